cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r3f; mkdir -p $O
timeout -k 10 600 python tests/tools/ab_two_handles.py 200 5 > $O/two_handles.log 2>&1; echo rc=$?; cat $O/two_handles.log | grep handle
timeout -k 10 300 python -m pytest tests/test_gpu_zonal.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3
