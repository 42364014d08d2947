#!/bin/bash
# HBM traffic of one MipNeRF-360 training step per kernel (separate --pmc passes FETCH_SIZE, WRITE_SIZE; KB; FETCH_SIZE on gfx950
# counts 64 B per 128-B request for wide streams): tools/probes/pmc_mip360_hbm.sh <tag> -> gpurun_out/<tag>/mip360_hbm.txt
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/${1:-m360hbm}; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rm -f $O/mip360_hbm.txt
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$c -- python $R/tools/mip360_bench.py --rays 4096 --steps 3 --warmup 1 > $O/pmc_$c.log 2>&1
  python $R/tools/rocpd_pmc.py $(ls $O/pmc_$c/*/*.db | head -1) _kernel >> $O/mip360_hbm.txt 2>&1
done
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
