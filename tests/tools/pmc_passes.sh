# Counter passes for the bench workload (run via gpurun; one rocprofv3 --pmc invocation per pass,
# with --kernel-trace only, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE do not
# fit one pass).  Output: gpurun_out/pmc/<pass>/...counter_collection.csv
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python3 bench.py --steps 3 --warmup 0 --spinup 300 --cpu-budget 0 --preroll 0 --repeats 1 ${EBM_PMC_BENCH_ARGS:-}"
OUT=${1:-gpurun_out/pmc}
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" \
         "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
         "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/p$i -- $B > $OUT.p$i.log 2>&1 || echo "pass $i failed"
done
