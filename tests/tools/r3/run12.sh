cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for L in base park base park; do
  if [ $L = park ]; then export EBM_LIB=$GRAFT_REPO_ROOT/build/libebm_park.so; else unset EBM_LIB; fi
  python bench.py --workload miz_imex_4096x2048 --cpu-budget 0 --repeats 5 --preroll 0 --warmup 0 --spinup 7000 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$L', round(d['ms_per_step'],5), d['config']['mean_tridiagonal_solves_per_column_step'])"
done
