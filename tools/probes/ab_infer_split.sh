#!/bin/bash
# alternate library builds (names under csrc/build/variants, NEW = the in-tree one) on the split-bf16 inference forward: kbench
# level-1 launch at the training batch shape (1024 x 192), 3 rounds on one box:  tools/probes/ab_infer_split.sh NEW p1_noepi
for rep in 1 2 3; do
for v in "$@"; do
  if [ $v = NEW ]; then unset NERFPP_HIP_LIB; else export NERFPP_HIP_LIB=$PWD/outdoor_nerf_depth_amd/csrc/build/variants/$v.so; fi
  k=$(python tools/kbench.py --n_rays 1024 --S 192 --prec 2 --iters 30 --only ${ONLY:-infer} 2>/dev/null | tail -1)
  echo "$v | $k"
done; done
