# Same-box A/B over the small and mid shapes: bash tests/tools/small_shapes_bench.sh [variant]  (shipped, build/libebm_<variant>.so, shipped)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
one() {
for W in "miz_180x1 --steps 2000" "miz_180x1 --steps 2048 --steps-per-launch 64" "miz_180x8192 --steps 512" "miz_180x8192 --steps 512 --steps-per-launch 64" "miz_1024x512x32" "classic_1024x512 --steps 2000" "classic_1024x512 --steps 2048 --steps-per-launch 64"; do
  python bench.py --workload $W --cpu-budget 0 --repeats 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', '$W', round(d['ms_per_step']*1e3,3))"
done
}
one shipped
[ -n "$1" ] && EBM_LIB=build/libebm_$1.so one $1
one shipped_again
