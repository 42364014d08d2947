#!/bin/bash
# Round 4, GPU call 17: cache policy of the ReLU sign words: A = non-temporal stores (forward) + default DMA (backward),
# B = default stores + non-temporal DMA, C = both default, D = both non-temporal; prev = the build before the slab / Adam / table
# policies.  Alternating bench runs on one box.
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/${1:-r04ae}; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
V=$R/outdoor_nerf_depth_amd/csrc/build/variants
B="--no_cpu_baseline --large_batch 0 --mip360_rays 0 --cli_steps 0 --render_frames 0 --precision bf16"
for rep in 1 2 3; do
  for v in A B C D cpb4; do
    export NERFPP_HIP_LIB=$V/libnerfpp_hip_$v.so
    timeout 300 python $R/bench.py $B --steps 100 --warmup 10 > $O/bench_${v}_$rep.json 2> $O/bench_${v}_$rep.err
    python - <<PY
import json
d=json.load(open('$O/bench_${v}_$rep.json'))
print('$v rep$rep', round(d['ms_per_step'],4), {k: v['ms'] for k,v in d['roofline']['all_kernels'].items()})
PY
  done
done | tee $O/ab.txt
