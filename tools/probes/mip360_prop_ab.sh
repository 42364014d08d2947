#!/bin/bash
# kernel-level A/B of libmip360 variants of the fused PropMLP forward (NEW = in-tree), 2 rounds: tools/probes/mip360_prop_ab.sh NEW nb4 ...
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
for v in "$@"; do
  if [ $v = NEW ]; then unset MIP360_HIP_LIB; else export MIP360_HIP_LIB=$PWD/outdoor_nerf_depth_amd/csrc/build/variants/mip360_$v.so; fi
  echo "$v: $(timeout 200 python tools/probes/mip360_prop_bench.py 2>/dev/null | awk '{print $1, $4}' | tr '\n' ' ')"
done; done
