# GPU box: the seventh mutant batch (tests/tools/mutants.py: the LDS-resident fused-K kernel, its compact solve, halo and vote)
# against the two test files that exercise ebm_run_fused on long meridians and on the extension; stop at the first failure.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3mut; : > gpurun_out/r3mut/mutants_7.log
for name in "$@"; do
  lib=build/libebm_mut_$name.so
  [ -f $lib ] || { echo "$name: not built" | tee -a gpurun_out/r3mut/mutants_7.log; continue; }
  EBM_TEST_NO_CHILDREN=1 EBM_LIB=$GRAFT_REPO_ROOT/$lib timeout -k 10 200 python -m pytest tests/test_gpu_fused.py tests/test_gpu_imex.py -q -m gpu -x -p no:cacheprovider > gpurun_out/r3mut/mut_$name.txt 2>&1
  rc=$?
  first=$(grep -m1 "^FAILED\|^ERROR" gpurun_out/r3mut/mut_$name.txt | cut -c1-150)
  if [ $rc -eq 0 ]; then echo "$name: fused + extension tests: SURVIVED" | tee -a gpurun_out/r3mut/mutants_7.log; else echo "$name: fused + extension tests: KILLED (rc $rc), first: $first" | tee -a gpurun_out/r3mut/mutants_7.log; fi
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "stopping: a run was killed at its limit"; break; fi
done
