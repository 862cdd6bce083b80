# The round's final evidence batch (GPU box): rocprofv3 kernel stats of the headline, integrate, fused-K and
# extension workloads, PMC passes of the headline kernel, bench lines of the other workloads.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2f
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_headline -- python3 bench.py --cpu-budget 0 > $O/prof_headline.json 2> $O/prof_headline.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_integrate -- python3 bench.py --workload miz_1024x512x32_integrate --steps 200 --repeats 2 --cpu-budget 0 > $O/prof_integrate.json 2> $O/prof_integrate.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fused180 -- python3 bench.py --workload miz_180x1 --steps 2048 --steps-per-launch 64 --cpu-budget 0 > $O/prof_fused180.json 2> $O/prof_fused180.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fused1440 -- python3 bench.py --workload miz_1440x1 --steps 2048 --steps-per-launch 64 --cpu-budget 0 > $O/prof_fused1440.json 2> $O/prof_fused1440.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_imex -- python3 bench.py --workload miz_imex_4096x2048 --cpu-budget 0 --repeats 2 > $O/prof_imex.json 2> $O/prof_imex.err
bash tests/tools/pmc_passes.sh $O/pmc
: > $O/other_workloads.jsonl
for W in "miz_180x1 --steps 2000" "miz_180x1 --steps 2048 --steps-per-launch 64" "miz_180x1 --steps 2048 --steps-per-launch 1024" \
         "miz_1440x1 --steps 2000" "miz_1440x1 --steps 2048 --steps-per-launch 64" "miz_1440x1 --steps 2048 --steps-per-launch 1024" \
         "miz_180x8192 --steps 512" "miz_180x8192 --steps 512 --steps-per-launch 64" \
         "miz_1024x512x32" "miz_1024x512x32_integrate --steps 100 --repeats 3" "miz_2048x4096" \
         "classic_1024x512 --steps 2000" "classic_1024x512 --steps 2048 --steps-per-launch 64" "miz_imex_4096x2048"; do
  python bench.py --workload $W --cpu-budget 0 >> $O/other_workloads.jsonl 2>> $O/other_workloads.err
done
python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -c 300 $O/bench_default.json
