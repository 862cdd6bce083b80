cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3a
timeout -k 10 900 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r3a/pytest.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -40 gpurun_out/r3a/pytest.log
if [ $rc -ne 124 ] && [ $rc -ne 137 ]; then
  python bench.py --steps 50 --warmup 10 > gpurun_out/r3a/bench_head.json 2> gpurun_out/r3a/bench_head.err && \
  python bench.py --workload miz_4096x2048_step --steps 50 --cpu-budget 0 > gpurun_out/r3a/bench_step.json 2> gpurun_out/r3a/bench_step.err && \
  python bench.py --workload miz_1024x512x32_integrate --steps 20 --cpu-budget 0 > gpurun_out/r3a/bench_integ.json 2> gpurun_out/r3a/bench_integ.err
  echo "bench rc=$?"
fi
