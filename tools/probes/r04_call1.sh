#!/bin/bash
# Round 4, GPU call 1: suite + default bench + the CU-mask partition probe + component-removal timings and PMC passes of the
# training forward (stores / MFMAs removed).  Output: gpurun_out/r04a/
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r04a; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
V=$R/outdoor_nerf_depth_amd/csrc/build/variants
( cd $R && timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log )
timeout 600 python $R/bench.py > $O/bench.json 2> $O/bench.err
NERFPP_HIP_LIB=$V/libnerfpp_hip_probes.so NERFPP_DEFER_DW=1 timeout 400 python $R/tools/probes/cu_mask_probe.py > $O/cu_mask.json 2> $O/cu_mask.err
for v in stock dbg1 nomfma nomfma_dbg1; do
  if [ $v = stock ]; then unset NERFPP_HIP_LIB; else export NERFPP_HIP_LIB=$V/$v.so; fi
  for n in 1024 8192; do
    timeout 120 python $R/tools/kbench.py --only train --n_rays $n --iters 10 2>/dev/null | sed "s/^/$v /" >> $O/kbench.txt
  done
done
G1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
G2="SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
G3="TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_TAG_STALL_sum"
for v in stock dbg1 nomfma; do
  if [ $v = stock ]; then unset NERFPP_HIP_LIB; else export NERFPP_HIP_LIB=$V/$v.so; fi
  i=0
  for grp in "$G1" "$G2" "$G3"; do
    i=$((i+1))
    timeout 200 rocprofv3 --kernel-trace --pmc $grp -d $O/pmc_${v}_g$i -- python $R/tools/kbench.py --only train --iters 4 > $O/pmc_${v}_g$i.log 2>&1
    echo "### $v group $i" >> $O/pmc_summary.txt
    python $R/tools/rocpd_pmc.py $(ls $O/pmc_${v}_g$i/*/*.db | head -1) mlp_fwd >> $O/pmc_summary.txt 2>&1
    rm -rf $O/pmc_${v}_g$i
  done
done
unset NERFPP_HIP_LIB
ls -la $O
