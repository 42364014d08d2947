#!/bin/bash
# average duration of the two weight-gradient launches (rocprofv3 kernel stats) for the in-tree library and for variants
# usage: tools/probes/dw_kernel_times.sh [variant ...]   (NEW = in-tree)
R=$PWD; cd /tmp; export TMPDIR=/tmp
for v in "$@"; do
  if [ $v = NEW ]; then unset NERFPP_HIP_LIB; else export NERFPP_HIP_LIB=$R/outdoor_nerf_depth_amd/csrc/build/variants/$v.so; fi
  rm -rf /tmp/dwk_$v
  rocprofv3 --kernel-trace --stats -d /tmp/dwk_$v -- python $R/bench.py --precision bf16 --no_cpu_baseline --large_batch 0 --mip360_rays 0 --render_frames 0 --cli_steps 0 --steps 30 --warmup 5 > /dev/null 2>&1
  echo "== $v"; python $R/tools/rocpd_stats.py $(ls /tmp/dwk_$v/*/*.db | head -1) | grep -E "dw_kernel|mlp_fwd_pair|mlp_bwd_pair" | cut -c1-110
done
