#!/bin/bash
# Round 4, GPU call 3: the grouped weight-DMA issue + split-bf16 training kernels on the roles pipe: full suite, bench
# (headline, parity modes, cli_loop), cycle stamps of the new build, weight-gradient launch stamps, power / clocks.
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r04c; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
V=$R/outdoor_nerf_depth_amd/csrc/build/variants
( cd $R && timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log )
cp $R/gpurun_out/trajectory_*.json $O/ 2>/dev/null
timeout 600 python $R/bench.py --no_cpu_baseline > $O/bench.json 2> $O/bench.err
( for i in $(seq 1 20); do rocm-smi --showpower --showclocks --json 2>/dev/null | head -c 4000; echo; sleep 0.4; done > $O/smi_during_bench.txt ) &
timeout 300 python $R/bench.py --precision bf16 --no_cpu_baseline --large_batch 0 --mip360_rays 0 --render_frames 0 --cli_steps 0 --steps 3000 --warmup 10 > $O/bench_long.json 2> $O/bench_long.err
wait
NERFPP_HIP_LIB=$V/stamps4.so timeout 200 python $R/tools/probes/stamps_probe.py --what fwd --out $O/stamps_fwd > $O/stamps_fwd.txt 2>&1
NERFPP_HIP_LIB=$V/stamps8.so timeout 200 python $R/tools/probes/stamps_probe.py --what bwd --out $O/stamps_bwd > $O/stamps_bwd.txt 2>&1
NERFPP_HIP_LIB=$V/libnerfpp_hip_probes.so timeout 200 python $R/tools/probes/dw_stamps_probe.py --out $O/dw_stamps > $O/dw_stamps.txt 2>&1
for n in 1024 8192; do for p in 1 2; do timeout 120 python $R/tools/kbench.py --n_rays $n --prec $p --iters 10 2>/dev/null >> $O/kbench.txt; done; done
ls -la $O
