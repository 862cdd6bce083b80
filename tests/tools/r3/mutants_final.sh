# GPU box: the -m gpu suite of the END of round 3 (final code: fused paths included; stop at the first failure) against mutant builds named on the command line
# (tests/tools/mutants.py).  No child processes (EBM_TEST_NO_CHILDREN=1).  Appends to gpurun_out/r3mut/mutants_final.log
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3mut
for name in "$@"; do
  lib=build/libebm_mut_$name.so
  if [ ! -f $lib ]; then echo "$name: not built" | tee -a gpurun_out/r3mut/mutants_final.log; continue; fi
  EBM_TEST_NO_CHILDREN=1 EBM_LIB=$GRAFT_REPO_ROOT/$lib timeout -k 10 240 python -m pytest tests -q -m gpu -x -p no:cacheprovider > gpurun_out/r3mut/final_$name.txt 2>&1
  rc=$?
  first=$(grep -m1 "^FAILED\|^ERROR" gpurun_out/r3mut/final_$name.txt | cut -c1-150)
  if [ $rc -eq 0 ]; then echo "$name: whole suite: SURVIVED" | tee -a gpurun_out/r3mut/mutants_final.log; else echo "$name: whole suite: KILLED (rc $rc), first: $first" | tee -a gpurun_out/r3mut/mutants_final.log; fi
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "stopping: a run was killed at its limit" | tee -a gpurun_out/r3mut/mutants_final.log; break; fi
done
