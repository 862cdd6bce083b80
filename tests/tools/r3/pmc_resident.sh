# GPU box: SQ counter passes of the LDS-resident fused-K kernel (64 steps per launch) — what the "VALU-issue bound" claim rests on
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3p/pmc_resident
mkdir -p $O
B="python3 bench.py --steps 128 --warmup 0 --spinup 300 --cpu-budget 0 --preroll 0 --repeats 1 --steps-per-launch 64"
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/p$i -- $B > $O.p$i.log 2>&1 || echo "pass $i failed"
done
python tests/tools/pmc_summary.py $O "miz_resident_kernel" 2 | tee gpurun_out/r3p/pmc_resident_summary.txt
