# A/B timing of alternative builds of the library on the GPU box: bash tests/tools/ab_variants.sh name1 name2 ...
# runs the headline bench (3 blocks) with the shipped library, then with build/libebm_<name>.so (built here with
# `make -C energybalancemodel.jl_amd/csrc -j4 OUT=$PWD/build/libebm_<name>.so BUILD=/tmp/<name> EXTRA=-D...`), then with
# the shipped one again.  Lines go to stdout and gpurun_out/ab_variants.log.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { python bench.py --cpu-budget 0 --repeats 3 $EBM_AB_ARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['blocks_ms_per_step'], d['config'].get('mean_tridiagonal_solves_per_column_step'))" | tee -a gpurun_out/ab_variants.log; }
run shipped
for v in "$@"; do
  case $v in nostores|CHEAP_DIV|NO_TRANS) ;;   # timing-only builds: results are garbage by construction
  *) # a variant that is meant to compute the same thing is CHECKED before its time means anything (a mid-round variant
     # once "won" 9 % by overrunning an LDS buffer into NaNs)
     EBM_LIB=build/libebm_$v.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -p no:cacheprovider -k "every_workgroup_size or sizes_vs_oracle" 2>&1 | tail -1 | sed "s/^/$v parity check: /" | tee -a gpurun_out/ab_variants.log;;
  esac
  EBM_LIB=build/libebm_$v.so run $v
done
run shipped_again
