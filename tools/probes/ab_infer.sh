#!/bin/bash
# alternate library builds (names under csrc/build/variants, NEW = the in-tree one) on the inference forward: kbench level-1
# launch + one rendered frame, 3 rounds on one box:  tools/probes/ab_infer.sh stock skew1
for rep in 1 2 3; do
for v in "$@"; do
  if [ $v = NEW ]; then unset NERFPP_HIP_LIB; else export NERFPP_HIP_LIB=$PWD/outdoor_nerf_depth_amd/csrc/build/variants/$v.so; fi
  k=$(python tools/kbench.py --n_rays 8192 --S 192 --prec 1 --iters 10 --only infer 2>/dev/null | tail -1)
  r=$(python tools/render_bench.py --frames 1 --precision bf16 2>/dev/null | tail -1 | cut -c1-120)
  echo "$v | $k | $r"
done; done
