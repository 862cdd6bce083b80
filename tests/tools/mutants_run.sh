# GPU box: the -m gpu suite (stop at the first failure) against every mutant build of tests/tools/mutants.py, then
# ONLY the tests that involve no oracle (analytic solutions, closed forms, scalar recurrences) against each.
# One line per mutant and pass: KILLED by <first failing test> / the anchors that fail, or SURVIVED.
# Output: gpurun_out/mutants.log
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; : > gpurun_out/mutants.log
ANCHORS="analytic or closed_form or scalar_recurrence or mode_recurrence or manufactured or step1_closed or melt_through or cellwise_recurrence or couples_ice or lateral_melt or docstring_forcing or integrate_saves_from_registers or hemispheric or fused_run or with_insolation"
for lib in build/libebm_mut_*.so; do
  name=$(basename $lib .so); name=${name#libebm_mut_}
  EBM_TEST_NO_CHILDREN=1 EBM_LIB=$lib timeout -k 10 300 python -m pytest tests -q -m gpu -x -p no:cacheprovider > gpurun_out/mut_$name.txt 2>&1
  rc=$?
  first=$(grep -m1 "^FAILED\|^ERROR" gpurun_out/mut_$name.txt | cut -c1-140)
  if [ $rc -eq 0 ]; then echo "$name: whole suite: SURVIVED" | tee -a gpurun_out/mutants.log; else echo "$name: whole suite: KILLED, first: $first" | tee -a gpurun_out/mutants.log; fi
  EBM_TEST_NO_CHILDREN=1 EBM_LIB=$lib timeout -k 10 300 python -m pytest tests -q -m gpu -p no:cacheprovider -k "$ANCHORS" > gpurun_out/mut_anchors_$name.txt 2>&1
  n=$(grep -c "^FAILED" gpurun_out/mut_anchors_$name.txt)
  which=$(grep "^FAILED" gpurun_out/mut_anchors_$name.txt | sed 's/^FAILED tests\///; s/ - .*//' | cut -d: -f3 | cut -d[ -f1 | sort -u | tr '\n' ' ')
  echo "$name: oracle-free anchors: $n failing ($which)" | tee -a gpurun_out/mutants.log
done
