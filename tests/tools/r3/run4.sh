cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r3d; mkdir -p $O
rm -f gpurun_out/measured_errors.jsonl
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_validity.py -m gpu -q -p no:cacheprovider -k "full_size or criterion or slab or pinned or bench" > $O/pytest.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -5 $O/pytest.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
grep -h "sensitivity\|dT0/dphi" gpurun_out/measured_errors.jsonl | cut -c1-420
python bench.py --steps 50 --warmup 10 --cpu-budget 0 > $O/bench_head.json 2> $O/bench_head.err && \
python bench.py --workload miz_1024x512x32_integrate --steps 20 --cpu-budget 0 > $O/bench_integ.json 2> $O/bench_integ.err
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[1], d["ms_per_step"], d["roofline"]["frac"], d["host_transfer"]["download_GBps"], d["host_transfer"]["upload_GBps"], d.get("year_end_ms"))
PY
done
