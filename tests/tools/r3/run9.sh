cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r3i; mkdir -p $O
for L in base light base light; do
  if [ $L = light ]; then export EBM_LIB=$GRAFT_REPO_ROOT/build/libebm_light.so; else unset EBM_LIB; fi
  for W in "miz_4096x2048" "miz_imex_4096x2048" "miz_180x8192 --steps 512" "miz_1024x512x32" "miz_4096x2048_step"; do
    python bench.py --workload $W --cpu-budget 0 --repeats 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$L', d['metric'][22:60].ljust(40), round(d['ms_per_step'],5), d['config']['mean_tridiagonal_solves_per_column_step'])"
  done
done
export EBM_LIB=$GRAFT_REPO_ROOT/build/libebm_light.so
EBM_TEST_NO_CHILDREN=1 timeout -k 10 600 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -4
