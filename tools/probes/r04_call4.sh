#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r04d; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
V=$R/outdoor_nerf_depth_amd/csrc/build/variants
( cd $R && timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log )
cp $R/gpurun_out/trajectory_*.json $O/ 2>/dev/null
timeout 600 python $R/bench.py --no_cpu_baseline > $O/bench.json 2> $O/bench.err
NERFPP_HIP_LIB=$V/libnerfpp_hip_probes.so timeout 200 python $R/tools/probes/dw_stamps_probe.py --out $O/dw_stamps > $O/dw_stamps.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace1 -- python $R/bench.py --steps 6 --warmup 2 --no_cpu_baseline --large_batch 0 --mip360_rays 0 --render_frames 0 --cli_steps 0 --precision bf16 > $O/trace1.log 2>&1
python $R/tools/rocpd_timeline.py $(ls $O/trace1/*/*.db | head -1) > $O/timeline_bf16.md 2>&1
python $R/tools/rocpd_stats.py $(ls $O/trace1/*/*.db | head -1) > $O/kernel_stats_bf16.md 2>&1
rm -rf $O/trace1
ls -la $O
