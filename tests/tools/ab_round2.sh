# A/B runs of this round's kernel experiments: bash tests/tools/ab_round2.sh (on the GPU box); variants are
# alternative builds of the library under build/ selected with EBM_LIB
cd $GRAFT_REPO_ROOT
run() { python bench.py --cpu-budget 0 --repeats 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['blocks_ms_per_step'])"; }
run default
EBM_LIB=build/libebm_ntEiD.so run nontemporal_loads_of_Ei_D
run default_again
EBM_LIB=build/libebm_ntEiD.so run nontemporal_again
