#!/bin/bash
# Round 4, GPU call 2: new tests (trajectory, schedule bit-identity), bench with the cli_loop leg, power / clock sampling
# during a long bench run, per-block cycle stamps of the training forward / backward, remaining PMC groups.
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r04b; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
V=$R/outdoor_nerf_depth_amd/csrc/build/variants
( cd $R && timeout 900 python -m pytest tests/test_gpu_round4.py -m gpu -x -q -s > $O/pytest_round4.log 2>&1; echo "pytest rc=$?" >> $O/pytest_round4.log )
cp $R/gpurun_out/trajectory_*.json $O/ 2>/dev/null
timeout 600 python $R/bench.py --no_cpu_baseline > $O/bench.json 2> $O/bench.err
# power + clocks while the bf16 step runs for ~8 s
( for i in $(seq 1 24); do rocm-smi --showpower --showclocks --showtemp --json 2>/dev/null | head -c 3000; echo; sleep 0.4; done > $O/smi_during_bench.txt ) &
timeout 300 python $R/bench.py --precision bf16 --no_cpu_baseline --large_batch 0 --mip360_rays 0 --render_frames 0 --cli_steps 0 --steps 3000 --warmup 10 > $O/bench_long.json 2> $O/bench_long.err
wait
rocm-smi --showpower --showclocks --showtemp --showmaxpower > $O/smi_idle.txt 2>&1
NERFPP_HIP_LIB=$V/stamps4.so timeout 200 python $R/tools/probes/stamps_probe.py --what fwd --out $O/stamps_fwd > $O/stamps_fwd.txt 2>&1
NERFPP_HIP_LIB=$V/stamps8.so timeout 200 python $R/tools/probes/stamps_probe.py --what bwd --out $O/stamps_bwd > $O/stamps_bwd.txt 2>&1
G2="SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT"
G4="TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
for v in stock dbg1 nomfma; do
  if [ $v = stock ]; then unset NERFPP_HIP_LIB; else export NERFPP_HIP_LIB=$V/$v.so; fi
  for i in 2 4; do
    if [ $i = 2 ]; then grp=$G2; else grp=$G4; fi
    timeout 200 rocprofv3 --kernel-trace --pmc $grp -d $O/pmc_${v}_g$i -- python $R/tools/kbench.py --only train --iters 4 > $O/pmc_${v}_g$i.log 2>&1
    echo "### $v group $i" >> $O/pmc_summary.txt
    python $R/tools/rocpd_pmc.py $(ls $O/pmc_${v}_g$i/*/*.db | head -1) mlp_fwd >> $O/pmc_summary.txt 2>&1
    rm -rf $O/pmc_${v}_g$i
  done
done
unset NERFPP_HIP_LIB
ls -la $O
