#!/bin/bash
# Round profile: bench JSON, rocprofv3 kernel stats + per-step timeline, HBM traffic PMC passes, the rendering leg and
# the MipNeRF-360 step; assembled into gpurun_out/<tag>/<tag>_*.md, ready to copy into profiles/.
# usage (on the GPU box, from the repo root): bash tools/probes/profile_round.sh <tag> [full]
#   full: also the default `python bench.py` run (incl. the ~2.5 min CPU baseline) and the MipNeRF-360 GEMM probes
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
TAG=${1:-rXX}; FULL=$2; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
B="--no_cpu_baseline --large_batch 0 --mip360_rays 0 --render_frames 0 --cli_steps 0"
if [ -n "$FULL" ]; then timeout 1200 python $R/bench.py --cli_steps 1100 > $O/bench.json 2> $O/bench.err; fi
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -- python $R/bench.py --steps 10 --warmup 2 $B > $O/trace.log 2>&1
python $R/tools/rocpd_stats.py $(ls $O/trace/*/*.db | head -1) > $O/kernel_stats.md
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace1 -- python $R/bench.py --steps 6 --warmup 2 $B --precision bf16 > $O/trace1.log 2>&1
python $R/tools/rocpd_timeline.py $(ls $O/trace1/*/*.db | head -1) > $O/timeline_bf16.md
rm -f $O/hbm_traffic.txt
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$c -- python $R/bench.py --steps 4 --warmup 1 $B --precision bf16 > $O/pmc_$c.log 2>&1
  python $R/tools/rocpd_pmc.py $(ls $O/pmc_$c/*/*.db | head -1) _kernel >> $O/hbm_traffic.txt 2>&1
done
# rendering leg: kernel stats of render_single_image, bf16 and split-bf16
for prec in bf16 split; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/render_$prec -- python $R/tools/render_bench.py --frames 2 --precision $prec > $O/render_$prec.json 2> $O/render_$prec.err
  python $R/tools/rocpd_stats.py $(ls $O/render_$prec/*/*.db | head -1) > $O/render_${prec}_kernel_stats.md
done
rm -rf $O/trace $O/trace1 $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/render_bf16 $O/render_split
# MipNeRF-360 step (config 5): kernel stats + one step's timeline
timeout 300 rocprofv3 --kernel-trace --stats -d $O/m360 -- python $R/tools/mip360_bench.py --rays 4096 --steps 6 --warmup 2 > $O/mip360_bench.json 2> $O/mip360.err
python $R/tools/rocpd_stats.py $(ls $O/m360/*/*.db | head -1) > $O/mip360_kernel_stats.md
python $R/tools/rocpd_timeline.py $(ls $O/m360/*/*.db | head -1) resample_kernel 3 > $O/mip360_timeline.md
rm -rf $O/m360
python $R/tools/probes/assemble_profile.py $O $TAG
if [ -n "$FULL" ]; then
  timeout 200 python $R/tools/probes/mip360_gemm_bench.py --check > $O/mip360_gemm_bench.txt 2>&1
  timeout 200 python $R/tools/ab_step.py > $O/ab_step.json 2> $O/ab_step.err
fi
