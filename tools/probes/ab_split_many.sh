#!/bin/bash
# alternate several library builds (names under csrc/build/variants) in the split-bf16 modes + one split-bf16 rendered frame, 2 rounds
for rep in 1 2; do
for v in "$@"; do
  if [ $v = NEW ]; then unset NERFPP_HIP_LIB; else export NERFPP_HIP_LIB=$PWD/outdoor_nerf_depth_amd/csrc/build/variants/$v.so; fi
  t=$(python bench.py --precision both --no_cpu_baseline --large_batch 0 --mip360_rays 0 --render_frames 0 --cli_steps 0 --steps 40 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('split', round(d['parity_mode']['value']), 'split_fwd', round(d['parity_forward_mode']['value']), 'fp16_fwd', round(d['fp16_forward_mode']['value']))")
  r=$(python tools/render_bench.py --frames 1 --precision split 2>/dev/null | tail -1 | python -c "import json,sys; print(round(json.loads(sys.stdin.read())['s_per_frame'],4))")
  echo "$v $t render_split $r"
done; done
