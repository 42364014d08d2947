#!/bin/bash
# Round 4, GPU call 16: what the slab sum (unpack_grads_kernel) and the remap fix-up cost the step each (probes build,
# NERFPP_REDUCE_SKIP = 0 none / 1 no slab sum / 2 no fix-up / 3 neither), alternating bench runs.
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/${1:-r04ab}; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
export NERFPP_HIP_LIB=$R/outdoor_nerf_depth_amd/csrc/build/variants/libnerfpp_hip_probes.so
B="--no_cpu_baseline --large_batch 0 --mip360_rays 0 --cli_steps 0 --render_frames 0 --precision bf16"
for rep in 1 2 3; do
  for v in 0 1 2 3; do
    NERFPP_REDUCE_SKIP=$v timeout 300 python $R/bench.py $B --steps 100 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('skip=$v rep$rep', round(d['ms_per_step'],4))"
  done
done | tee $O/reduce_skip.txt
