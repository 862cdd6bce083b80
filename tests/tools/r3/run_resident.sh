#!/bin/bash
# GPU box: the fused-K tests (register kernel, LDS-resident kernel, the extension's), then the large-shape fused bench lines
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3p
export EBM_TEST_NO_CHILDREN=1
timeout -k 10 500 python -m pytest tests/test_gpu_fused.py tests/test_gpu_imex.py -m gpu -x -q > gpurun_out/r3p/tests.log 2>&1
rc=$?
tail -15 gpurun_out/r3p/tests.log
[ $rc -eq 0 ] || exit $rc
for k in 1 16 64; do
  timeout -k 10 200 python bench.py --workload miz_4096x2048 --steps-per-launch $k --cpu-budget 0 > gpurun_out/r3p/bench_K$k.json 2> gpurun_out/r3p/bench_K$k.err || { tail -5 gpurun_out/r3p/bench_K$k.err; exit 1; }
  cut -c1-400 gpurun_out/r3p/bench_K$k.json
done
