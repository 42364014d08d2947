# PMC passes over the MipNeRF-360 forward (ring GEMM): wave-cycle split, MFMA busy, LDS, clock
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rm -f $R/gpurun_out/pmc_mip360.txt
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pm/g$i -- python $R/tools/mip360_bench.py --rays 4096 --steps 2 --warmup 1 --forward_only > /tmp/pm_g$i.log 2>&1
  python $R/tools/rocpd_pmc.py $(ls /tmp/pm/g$i/*/*.db | head -1) ring_kernel >> $R/gpurun_out/pmc_mip360.txt 2>&1
done
