#!/bin/bash
# Round 4, GPU call 18: store policy of the ReLU sign words (0 default, 1 sc1, 2 sc0 sc1, 4 sc1 nt, 5 sc0), alternating bench runs.
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/${1:-r04ai}; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
V=$R/outdoor_nerf_depth_amd/csrc/build/variants
B="--no_cpu_baseline --large_batch 0 --mip360_rays 0 --cli_steps 0 --render_frames 0 --precision bf16"
for rep in 1 2 3; do
  for v in mk0 mk1 mk2 mk4 mk5; do
    export NERFPP_HIP_LIB=$V/$v.so
    timeout 300 python $R/bench.py $B --steps 100 --warmup 10 > $O/bench_${v}_$rep.json 2> $O/bench_${v}_$rep.err
    python - <<PY
import json
d=json.load(open('$O/bench_${v}_$rep.json'))
print('$v rep$rep', round(d['ms_per_step'],4), {k: v['ms'] for k,v in d['roofline']['all_kernels'].items()})
PY
  done
done | tee $O/ab.txt
