# PMC passes over one NerfMLP-shaped dense layer for the ring variants (MIP360_GEMM_RING = 1 lock-step, 0 pipelined)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_gemm_one.txt; rm -f $OUT
for ring in ${RINGS:-1 0}; do
  echo "# MIP360_GEMM_RING=$ring" >> $OUT
  i=0
  for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
    i=$((i+1))
    MIP360_GEMM_RING=$ring timeout 200 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pm/r${ring}g$i -- python $R/tools/probes/mip360_gemm_one.py > /tmp/pm_g$i.log 2>&1
    python $R/tools/rocpd_pmc.py $(ls /tmp/pm/r${ring}g$i/*/*.db | head -1) linear_bf16 >> $OUT 2>&1
  done
done
cat $OUT
