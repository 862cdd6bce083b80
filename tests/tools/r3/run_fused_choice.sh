# GPU box: tests of the fused-K kernel choice, then the fused bench lines of the many-column shapes either way
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3p
EBM_TEST_NO_CHILDREN=1 timeout -k 10 600 python -m pytest tests/test_gpu_fused.py tests/test_gpu_validity.py tests/test_gpu_parity.py -q -x -m gpu -p no:cacheprovider -k "every_workgroup_size or fused or options" 2>&1 | tail -3
[ ${PIPESTATUS[0]} -eq 0 ] || exit 1
L=gpurun_out/r3p/fused_choice.jsonl; : > $L
for W in "miz_1024x512x32 --steps 512" "miz_180x8192 --steps 512" "miz_2048x4096 --steps 512" "miz_180x1 --steps 2048"; do
  for S in registers lds auto; do
    python bench.py --cpu-budget 0 --repeats 3 --workload $W --steps-per-launch 64 --fused-state $S 2>/dev/null | tee -a $L | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$S', d['config']['workload'].split(':')[0], '|', round(d['ms_per_step'],5), 'ms/step', d['roofline']['kernel'])"
  done
done
