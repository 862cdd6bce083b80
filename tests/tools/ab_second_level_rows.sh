cd $GRAFT_REPO_ROOT
one() {
for W in "miz_180x1 --steps 2048 --steps-per-launch 64" "miz_1440x1 --steps 2048 --steps-per-launch 64" "miz_180x8192 --steps 512" "miz_1024x512x32" "classic_1024x512 --steps 2000"; do
  python bench.py --workload $W --cpu-budget 0 --repeats 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', '$W', round(d['ms_per_step']*1e3,3))"
done
}
one shipped
EBM_LIB=build/libebm_R2.so one R2
EBM_LIB=build/libebm_R8.so one R8
one shipped_again
