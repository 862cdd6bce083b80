cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for C in 1 2; do
    python bench.py --cpu-budget 0 --repeats 5 --launch-chains $C 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('chains $C', round(d['ms_per_step'],5), [round(b,4) for b in d['blocks_ms_per_step']])"
  done
done
python tests/tools/ab_two_handles.py 200 5 2>&1 | grep handle | cut -c1-120
