#!/bin/bash
# Round 4, GPU call 5: pair launches (fg + bg MLP kernels, full + narrow weight-gradient kernels in one launch each):
# suite, A/B against the two-launch form inside one process image (probes library, env switches), bench, timeline.
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r04e; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
V=$R/outdoor_nerf_depth_amd/csrc/build/variants
( cd $R && timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log )
B="--precision bf16 --no_cpu_baseline --large_batch 0 --mip360_rays 0 --render_frames 0 --cli_steps 0 --steps 100 --warmup 5"
export NERFPP_HIP_LIB=$V/libnerfpp_hip_probes.so
for rep in 1 2 3; do
  for arm in pair split_mlp split_dw split_both; do
    unset NERFPP_MLP_SPLIT NERFPP_DW_SPLIT
    [ $arm = split_mlp ] && export NERFPP_MLP_SPLIT=1
    [ $arm = split_dw ] && export NERFPP_DW_SPLIT=1
    [ $arm = split_both ] && export NERFPP_MLP_SPLIT=1 NERFPP_DW_SPLIT=1
    timeout 200 python $R/bench.py $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$arm', round(d['value']), round(d['ms_per_step'],4), d['roofline']['share_ms_per_step'])" >> $O/ab_pair.txt
  done
done
unset NERFPP_MLP_SPLIT NERFPP_DW_SPLIT NERFPP_HIP_LIB
timeout 600 python $R/bench.py --no_cpu_baseline > $O/bench.json 2> $O/bench.err
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace1 -- python $R/bench.py --steps 6 --warmup 2 --no_cpu_baseline --large_batch 0 --mip360_rays 0 --render_frames 0 --cli_steps 0 --precision bf16 > $O/trace1.log 2>&1
python $R/tools/rocpd_timeline.py $(ls $O/trace1/*/*.db | head -1) > $O/timeline_bf16.md 2>&1
python $R/tools/rocpd_stats.py $(ls $O/trace1/*/*.db | head -1) > $O/kernel_stats_bf16.md 2>&1
rm -rf $O/trace1
ls -la $O
