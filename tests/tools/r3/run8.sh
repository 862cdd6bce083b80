cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r3h; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_zonal8 -- python3 tests/tools/zonal_profile.py 32 6 > $O/z8.log 2>&1
find $O/prof_zonal8 -name "*kernel_stats.csv" | head -1 | xargs -r grep zonal_sweep
EBM_LIB=$GRAFT_REPO_ROOT/build/libebm_zunr16.so rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_zonal16 -- python3 tests/tools/zonal_profile.py 32 6 > $O/z16.log 2>&1
find $O/prof_zonal16 -name "*kernel_stats.csv" | head -1 | xargs -r grep zonal_sweep
python bench.py --steps 50 --warmup 10 --cpu-budget 0 > $O/bench_head.json 2> $O/bench_head.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r3h/bench_head.json"))
print(d["ms_per_step"], d["host_transfer"]["download_GBps"], d["host_transfer"]["upload_GBps"])
PY
timeout -k 10 300 python -m pytest tests/test_gpu_zonal.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3
