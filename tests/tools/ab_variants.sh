# A/B timing of alternative builds of the library on the GPU box: bash tests/tools/ab_variants.sh name1 name2 ...
# runs the headline bench (3 blocks) with the shipped library, then with build/libebm_<name>.so (built here with
# `make -C energybalancemodel.jl_amd/csrc -j4 OUT=$PWD/build/libebm_<name>.so BUILD=/tmp/<name> EXTRA=-D...`), then with
# the shipped one again.  Lines go to stdout and gpurun_out/ab_variants.log.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { python bench.py --cpu-budget 0 --repeats 3 $EBM_AB_ARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['blocks_ms_per_step'], d['config'].get('mean_tridiagonal_solves_per_column_step'))" | tee -a gpurun_out/ab_variants.log; }
run shipped
for v in "$@"; do EBM_LIB=build/libebm_$v.so run $v; done
run shipped_again
