#!/bin/bash
# alternate several library builds (names under csrc/build/variants, NEW = the in-tree one) 3x on one box
for rep in 1 2 3; do
for v in "$@"; do
  if [ $v = NEW ]; then unset NERFPP_HIP_LIB; else export NERFPP_HIP_LIB=$PWD/outdoor_nerf_depth_amd/csrc/build/variants/$v.so; fi
  python bench.py --precision bf16 --no_cpu_baseline --large_batch 0 --mip360_rays 0 --render_frames 0 --cli_steps 0 --steps 60 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', round(d['value']), round(d['ms_per_step'],4), d['roofline']['share_ms_per_step'])"
done; done
