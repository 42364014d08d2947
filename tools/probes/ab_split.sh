#!/bin/bash
# A/B of a variant library (csrc/build/variants/<name>.so) against the in-tree one in the split-bf16 and split_fwd modes
V=${1:-baseline}
for rep in 1 2 3; do
for v in $V NEW; do
  if [ $v = NEW ]; then unset NERFPP_HIP_LIB; else export NERFPP_HIP_LIB=$PWD/outdoor_nerf_depth_amd/csrc/build/variants/$v.so; fi
  python bench.py --precision both --no_cpu_baseline --large_batch 0 --mip360_rays 0 --render_frames 0 --cli_steps 0 --steps 40 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v bf16', round(d['value']), 'split', round(d['parity_mode']['value']), 'split_fwd', round(d['parity_forward_mode']['value']), 'fp16_fwd', round(d['fp16_forward_mode']['value']))"
done; done
