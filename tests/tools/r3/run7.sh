cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r3g; mkdir -p $O
rm -f gpurun_out/measured_errors.jsonl
timeout -k 10 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -8 $O/pytest.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
python bench.py --steps 100 --warmup 10 --cpu-budget 0 > $O/bench_head.json 2> $O/bench_head.err && \
python bench.py --steps 100 --warmup 10 --cpu-budget 0 --launch-chains 2 > $O/bench_head_2chains.json 2> $O/bench_head_2chains.err && \
python bench.py --workload miz_4096x2048_step --steps 100 --cpu-budget 0 --launch-chains 2 > $O/bench_step_2chains.json 2> $O/bench_step_2chains.err && \
python bench.py --workload miz_imex_4096x2048 --steps 100 --cpu-budget 0 > $O/bench_imex.json 2> $O/bench_imex.err && \
python bench.py --workload miz_imex_4096x2048 --steps 100 --cpu-budget 0 --launch-chains 2 > $O/bench_imex_2chains.json 2> $O/bench_imex_2chains.err
echo "bench rc=$?"
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[1], d["metric"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["config"]["steps_per_launch"])
PY
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_2chains -- python3 bench.py --cpu-budget 0 --launch-chains 2 > $O/prof_2chains.json 2> $O/prof_2chains.err
find $O/prof_2chains -name "*kernel_stats.csv" | head -1 | xargs -r head -4
