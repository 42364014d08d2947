#!/bin/bash
# Round 4, GPU call 11: merged weight-gradient jobs against the unmerged build (commit 9e3f382) on ONE box: alternating bench
# runs, then rocprofv3 kernel stats + one step's timeline of each.
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
TAG=${1:-r04n}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
V=$R/outdoor_nerf_depth_amd/csrc/build/variants
B="--no_cpu_baseline --large_batch 0 --mip360_rays 0 --render_frames 0 --cli_steps 0 --precision bf16"
for rep in 1 2 3; do
  for v in merged unmerged; do
    if [ $v = merged ]; then unset NERFPP_HIP_LIB; else export NERFPP_HIP_LIB=$V/libnerfpp_hip_unmerged.so; fi
    timeout 300 python $R/bench.py $B --steps 100 --warmup 10 > $O/bench_${v}_$rep.json 2> $O/bench_${v}_$rep.err
    python - <<PY
import json
d=json.load(open('$O/bench_${v}_$rep.json'))
print('$v rep$rep', round(d['ms_per_step'],4), {k: v['ms'] for k,v in d['roofline']['all_kernels'].items()})
PY
  done
done | tee $O/ab.txt
for v in merged unmerged; do
  if [ $v = merged ]; then unset NERFPP_HIP_LIB; else export NERFPP_HIP_LIB=$V/libnerfpp_hip_unmerged.so; fi
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_$v -- python $R/bench.py --steps 6 --warmup 2 $B > $O/trace_$v.log 2>&1
  python $R/tools/rocpd_stats.py $(ls $O/trace_$v/*/*.db | head -1) > $O/kernel_stats_$v.md
  python $R/tools/rocpd_timeline.py $(ls $O/trace_$v/*/*.db | head -1) > $O/timeline_$v.md
  rm -rf $O/trace_$v
done
unset NERFPP_HIP_LIB
( cd $R && timeout 900 python -m pytest tests/test_gpu_round4.py -m gpu -q -k trajectory > $O/pytest_traj.log 2>&1; echo "pytest rc=$?" >> $O/pytest_traj.log )
grep "dw_kernel\|unpack" $O/kernel_stats_*.md $O/timeline_*.md
