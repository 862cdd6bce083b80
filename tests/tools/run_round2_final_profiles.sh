set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
EBM_LIB=build/libebm_nostores.so python bench.py --cpu-budget 0 > gpurun_out/r2/b_nostores.json 2> gpurun_out/r2/b_nostores.err
python -c "import json; d=json.load(open('gpurun_out/r2/b_nostores.json')); print('NO STORES', d['ms_per_step'])"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2/prof_headline2 -- python3 bench.py --cpu-budget 0 > gpurun_out/r2/prof_headline2.json 2> gpurun_out/r2/prof_headline2.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2/prof_integrate2 -- python3 bench.py --workload miz_1024x512x32_integrate --steps 200 --repeats 2 --cpu-budget 0 > gpurun_out/r2/prof_integrate2.json 2> gpurun_out/r2/prof_integrate2.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2/prof_fused180 -- python3 bench.py --workload miz_180x1 --steps 2048 --steps-per-launch 64 --cpu-budget 0 > gpurun_out/r2/prof_fused180.json 2> gpurun_out/r2/prof_fused180.err
python tests/tools/soak_year.py > gpurun_out/r2/soak_year.log 2>&1
tail -3 gpurun_out/r2/soak_year.log
