# The round's final evidence batch (GPU box): rocprofv3 kernel stats of the headline, integrate, fused-K and
# extension workloads, PMC passes of the headline kernel, the timing-only no-stores build.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2f
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2f/prof_headline -- python3 bench.py --cpu-budget 0 > gpurun_out/r2f/prof_headline.json 2> gpurun_out/r2f/prof_headline.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2f/prof_integrate -- python3 bench.py --workload miz_1024x512x32_integrate --steps 200 --repeats 2 --cpu-budget 0 > gpurun_out/r2f/prof_integrate.json 2> gpurun_out/r2f/prof_integrate.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2f/prof_fused180 -- python3 bench.py --workload miz_180x1 --steps 2048 --steps-per-launch 64 --cpu-budget 0 > gpurun_out/r2f/prof_fused180.json 2> gpurun_out/r2f/prof_fused180.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2f/prof_imex -- python3 bench.py --workload miz_imex_4096x2048 --cpu-budget 0 --repeats 2 > gpurun_out/r2f/prof_imex.json 2> gpurun_out/r2f/prof_imex.err
bash tests/tools/pmc_passes.sh gpurun_out/r2f/pmc
python bench.py > gpurun_out/r2f/bench_default.json 2> gpurun_out/r2f/bench_default.err
tail -c 400 gpurun_out/r2f/bench_default.json
