# GPU box: which fused-K kernel is faster where the register kernel exists (<= 512 threads, four cells per thread)?
# shipped (registers, 231 ... 256 VGPRs, two waves per SIMD) against build/libebm_resall.so (-DEBM_RESIDENT_ALWAYS: state in LDS,
# 128 VGPRs, up to four waves per SIMD) on the many-column shapes and on one short meridian.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3p
L=gpurun_out/r3p/ab_resident_always.log; : > $L
EBM_TEST_NO_CHILDREN=1 EBM_LIB=$GRAFT_REPO_ROOT/build/libebm_resall.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused.py -q -x -m gpu -p no:cacheprovider -k "every_workgroup_size or fused_run_equals" 2>&1 | tail -1 | sed "s/^/resall parity check: /" | tee -a $L
run() { # name lib workload-args
  EBM_LIB=$2 python bench.py --cpu-budget 0 --repeats 3 ${@:3} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['config']['workload'].split(':')[0], '|', round(d['ms_per_step'],5), 'ms/step', d['roofline']['kernel'])" | tee -a $L; }
S=$GRAFT_REPO_ROOT/energybalancemodel.jl_amd/libebm_hip.so; R=$GRAFT_REPO_ROOT/build/libebm_resall.so
for W in "miz_1024x512x32 --steps 512 --steps-per-launch 64" "miz_180x8192 --steps 512 --steps-per-launch 64" "miz_2048x4096 --steps 512 --steps-per-launch 64"; do
  EBM_CELLS_PER_THREAD=4 run shipped $S --workload $W
  EBM_CELLS_PER_THREAD=4 run resall $R --workload $W
  EBM_CELLS_PER_THREAD=4 run shipped $S --workload $W
done
EBM_CELLS_PER_THREAD=4 run shipped $S --workload miz_180x1 --steps 2048 --steps-per-launch 64
EBM_CELLS_PER_THREAD=4 run resall $R --workload miz_180x1 --steps 2048 --steps-per-launch 64
