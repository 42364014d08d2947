# PMC passes over the split-bf16 inference forward (kbench level-1 launch): where do the wave cycles go?
#   bash tools/probes/pmc_split_infer.sh <out dir under gpurun_out> [lib variant name]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O
if [ -n "$2" ]; then export NERFPP_HIP_LIB=$R/outdoor_nerf_depth_amd/csrc/build/variants/$2.so; fi
cd /tmp; export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_INST_LEVEL_VMEM" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $grp -d $O/g$i -- python $R/tools/kbench.py --prec 2 --only infer --iters 4 > $O/g$i.log 2>&1
  python $R/tools/rocpd_pmc.py $(ls $O/g$i/*/*.db | head -1) mlp_fwd >> $O/summary.txt 2>&1
  rm -rf $O/g$i
done
cat $O/summary.txt
