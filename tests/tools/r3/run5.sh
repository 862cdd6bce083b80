cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r3e; mkdir -p $O
rm -f gpurun_out/measured_errors.jsonl
timeout -k 10 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -12 $O/pytest.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
grep -h "zonal\|explained" gpurun_out/measured_errors.jsonl | cut -c1-380
python bench.py --steps 50 --warmup 10 --cpu-budget 0 > $O/bench_head.json 2> $O/bench_head.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r3e/bench_head.json"))
print(d["ms_per_step"], d["roofline"]["frac"], d["host_transfer"]["download_GBps"], d["host_transfer"]["upload_GBps"])
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_zonal -- python3 tests/tools/zonal_profile.py 32 6 > $O/prof_zonal.log 2>&1
cat $O/prof_zonal.log | tail -2
find $O/prof_zonal -name "*kernel_stats.csv" | head -1 | xargs -r head -8
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmc_zonal_$C -- python3 tests/tools/zonal_profile.py 32 3 > $O/pmc_zonal_$C.log 2>&1 || echo "pass $C failed"
done
python tests/tools/pmc_summary.py $O "zonal_sweep_kernel" 2
