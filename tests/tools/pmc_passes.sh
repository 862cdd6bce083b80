cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python3 bench.py --steps 3 --warmup 0 --spinup 300 --cpu-budget 0"
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
         "SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_INSTS_FLAT SQ_ACTIVE_INST_MISC SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS_F64" \
         "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d gpurun_out/pmc1/p$i -- $B > gpurun_out/pmc1_p$i.log 2>&1 || echo "pass $i failed"
done
find gpurun_out/pmc1 -name "*counter_collection.csv" | head -20
