cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r3b; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -15 $O/pytest.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
python bench.py --steps 50 --warmup 10 > $O/bench_head.json 2> $O/bench_head.err && \
python bench.py --workload miz_4096x2048_step --steps 50 --cpu-budget 0 > $O/bench_step.json 2> $O/bench_step.err && \
python bench.py --workload miz_1024x512x32_integrate --steps 20 --cpu-budget 0 > $O/bench_integ.json 2> $O/bench_integ.err && \
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_step -- python3 bench.py --workload miz_4096x2048_step --steps 100 --cpu-budget 0 > $O/prof_step.json 2> $O/prof_step.err
echo "bench rc=$?"
B="python3 bench.py --workload miz_4096x2048_step --steps 3 --warmup 0 --spinup 300 --cpu-budget 0 --preroll 0 --repeats 1"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmc_step_$C -- $B > $O/pmc_step_$C.log 2>&1 || echo "pass $C failed"
done
python tests/tools/pmc_summary.py $O "miz_step_kernel<4, 1, 1, 1024, false>" 3 > $O/pmc_step_summary.txt 2>&1
cat $O/pmc_step_summary.txt
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[1], d["ms_per_step"], d["roofline"]["frac"], d.get("host_transfer"), d.get("year_end_ms"))
PY
done
find $O/prof_step -name "*kernel_stats.csv" | head -1 | xargs -r head -5
