#!/bin/bash
# rocprofv3 kernel stats + one-step timeline of the MipNeRF-360 training step: tools/probes/mip360_profile.sh <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/m360 -- python $R/tools/mip360_bench.py --rays 4096 --steps 6 --warmup 2 > $O/mip360_bench.json 2> $O/mip360.err
python $R/tools/rocpd_stats.py $(ls $O/m360/*/*.db | head -1) > $O/mip360_kernel_stats.md
python $R/tools/rocpd_timeline.py $(ls $O/m360/*/*.db | head -1) resample_kernel 3 > $O/mip360_timeline.md
rm -rf $O/m360
tail -1 $O/mip360_bench.json | cut -c150-260
