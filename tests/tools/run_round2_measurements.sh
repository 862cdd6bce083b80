set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
python tests/tools/error_budget.py > gpurun_out/r2/error_budget.txt 2> gpurun_out/r2/error_budget.err
tail -3 gpurun_out/r2/error_budget.txt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2/prof_headline -- python3 bench.py --cpu-budget 0 > gpurun_out/r2/prof_headline.json 2> gpurun_out/r2/prof_headline.err
python bench.py --workload miz_180x1 --steps 2000 --cpu-budget 2 > gpurun_out/r2/b_180_k1.json 2> gpurun_out/r2/b_180_k1.err
python bench.py --workload miz_180x1 --steps 2048 --steps-per-launch 64 --cpu-budget 0 > gpurun_out/r2/b_180_k64.json 2> gpurun_out/r2/b_180_k64.err
python bench.py --workload miz_180x1 --steps 2048 --steps-per-launch 1024 --cpu-budget 0 > gpurun_out/r2/b_180_k1024.json 2> gpurun_out/r2/b_180_k1024.err
python bench.py --workload miz_1440x1 --steps 2000 --cpu-budget 2 > gpurun_out/r2/b_1440_k1.json 2> gpurun_out/r2/b_1440_k1.err
python bench.py --workload miz_1440x1 --steps 2048 --steps-per-launch 64 --cpu-budget 0 > gpurun_out/r2/b_1440_k64.json 2> gpurun_out/r2/b_1440_k64.err
python bench.py --workload miz_1440x1 --steps 2048 --steps-per-launch 1024 --cpu-budget 0 > gpurun_out/r2/b_1440_k1024.json 2> gpurun_out/r2/b_1440_k1024.err
python bench.py --workload miz_1024x512x32 --cpu-budget 0 > gpurun_out/r2/b_ens.json 2> gpurun_out/r2/b_ens.err
python bench.py --workload miz_1024x512x32_integrate --steps 100 --repeats 3 --cpu-budget 0 > gpurun_out/r2/b_integrate.json 2> gpurun_out/r2/b_integrate.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2/prof_integrate -- python3 bench.py --workload miz_1024x512x32_integrate --steps 100 --repeats 2 --cpu-budget 0 > gpurun_out/r2/prof_integrate.json 2> gpurun_out/r2/prof_integrate.err
bash tests/tools/pmc_passes.sh gpurun_out/r2/pmc
for f in gpurun_out/r2/b_*.json; do echo $f; python -c "import json,sys; d=json.load(open('$f')); print(d['metric'], d['ms_per_step'], d['value'], d['roofline']['frac'])"; done
