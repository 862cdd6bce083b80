cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r3j; mkdir -p $O
( timeout -k 10 500 python tests/tools/fuzz_many.py 10000 1500 miz > $O/fuzz_miz.log 2>&1; tail -3 $O/fuzz_miz.log ) 
( timeout -k 10 300 python tests/tools/fuzz_many.py 10000 1000 imex > $O/fuzz_imex.log 2>&1; tail -3 $O/fuzz_imex.log )
( timeout -k 10 200 python tests/tools/fuzz_many.py 10000 1000 classic > $O/fuzz_classic.log 2>&1; tail -2 $O/fuzz_classic.log )
