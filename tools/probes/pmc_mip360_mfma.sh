# MFMA occupancy x clock of the MipNeRF-360 step's GEMM kernels (rocprofv3 --pmc passes, counters only): is linear_fm at the
# power budget the NeRF++ kernels sit at (profiles/r06_split_stamps.md)?   bash tools/probes/pmc_mip360_mfma.sh <out dir>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_ANY" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d $O/g$i -- python $R/tools/mip360_bench.py --rays 4096 --steps 4 --warmup 2 > $O/g$i.log 2>&1
  python $R/tools/rocpd_pmc.py $(ls $O/g$i/*/*.db | head -1) linear_fm >> $O/summary.txt 2>&1
  rm -rf $O/g$i
done
timeout 300 rocprofv3 --kernel-trace --stats -d $O/t -- python $R/tools/mip360_bench.py --rays 4096 --steps 4 --warmup 2 > $O/t.log 2>&1
python $R/tools/rocpd_stats.py $(ls $O/t/*/*.db | head -1) | grep -i "linear_fm\|Name\|---" | head -12 >> $O/summary.txt
rm -rf $O/t
cat $O/summary.txt
