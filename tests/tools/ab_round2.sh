cd $GRAFT_REPO_ROOT
run() { python bench.py --cpu-budget 0 --repeats 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['blocks_ms_per_step'])"; }
run default
EBM_LIB=build/libebm_r2.so run R2
EBM_LIB=build/libebm_r8.so run R8
EBM_PREFETCH_COLS=0 run prefetch0
EBM_PREFETCH_COLS=128 run prefetch128
EBM_PREFETCH_COLS=512 run prefetch512
EBM_PREFETCH_COLS=264 run prefetch264
run default_again
