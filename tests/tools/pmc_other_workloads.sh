# HBM traffic (FETCH_SIZE, WRITE_SIZE: one --pmc pass each, --kernel-trace only) of the step kernel on the
# ensemble shape, state-only and with savesol! of ten variables inside the step.  Run via gpurun;
# output: gpurun_out/pmc_<workload>/p{1,2} and a summary on stdout.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for W in miz_1024x512x32 miz_1024x512x32_integrate; do
  i=0
  for C in FETCH_SIZE WRITE_SIZE; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d gpurun_out/pmc_$W/p$i -- python3 bench.py --workload $W --steps 3 --warmup 0 --spinup 300 --cpu-budget 0 --preroll 0 --repeats 1 > gpurun_out/pmc_$W.p$i.log 2>&1 || echo "pass $W $i failed"
  done
  echo "== $W"
  python3 tests/tools/pmc_summary.py gpurun_out/pmc_$W miz_step_kernel 3
done
