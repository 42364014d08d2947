#!/bin/bash
# Round 4, GPU call 13: what would halving the LDS weight-fragment reads per MFMA buy (64-row waves)?  kbench of the stock
# build, the default-schedule build (LDS_PREFETCH=0) and one read per 2 / 4 MFMAs, alternating.
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/${1:-r04t}; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
V=$R/outdoor_nerf_depth_amd/csrc/build/variants
for rep in 1 2; do
  for v in stock pf0 reuse2 reuse4; do
    if [ $v = stock ]; then unset NERFPP_HIP_LIB; else export NERFPP_HIP_LIB=$V/$v.so; fi
    timeout 120 python $R/tools/kbench.py --iters 30 2>/dev/null | sed "s/^/$v rep$rep: /"
  done
done | tee $O/lds_reuse_kbench.txt
