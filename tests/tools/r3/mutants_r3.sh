# GPU box: the -m gpu suite (stop at the first failure) against the round-3 mutant builds (tests/tools/mutants.py, sixth batch).
# EBM_TEST_NO_CHILDREN=1: the sessions start no child processes (a session stopped by -x would leave them on the GPU while the
# next starts its own; the box allows six processes on the card) — the tests that read the children's output skip.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3mut; : > gpurun_out/r3mut/mutants_b.log
for name in "$@"; do
  lib=build/libebm_mut_$name.so
  EBM_TEST_NO_CHILDREN=1 EBM_LIB=$GRAFT_REPO_ROOT/$lib timeout -k 10 240 python -m pytest tests -q -m gpu -x -p no:cacheprovider > gpurun_out/r3mut/mut_$name.txt 2>&1
  rc=$?
  first=$(grep -m1 "^FAILED\|^ERROR" gpurun_out/r3mut/mut_$name.txt | cut -c1-150)
  if [ $rc -eq 0 ]; then echo "$name: whole suite: SURVIVED" | tee -a gpurun_out/r3mut/mutants_b.log; else echo "$name: whole suite: KILLED (rc $rc), first: $first" | tee -a gpurun_out/r3mut/mutants_b.log; fi
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "stopping: a run was killed at its limit"; break; fi
done
