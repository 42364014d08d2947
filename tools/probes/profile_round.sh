#!/bin/bash
# Round profile: bench JSON, rocprofv3 kernel stats + per-step timeline, HBM traffic PMC passes.
# usage (on the GPU box, from the repo root): bash tools/probes/profile_round.sh <tag>
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
TAG=${1:-rXX}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 400 python $R/bench.py > $O/bench.json 2> $O/bench.err
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -- python $R/bench.py --steps 10 --warmup 2 --no_cpu_baseline --large_batch 0 --mip360_rays 0 > $O/trace.log 2>&1
DB=$(ls $O/trace/*/*.db | head -1)
python $R/tools/rocpd_stats.py $DB > $O/kernel_stats.md
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace1 -- python $R/bench.py --steps 6 --warmup 2 --no_cpu_baseline --large_batch 0 --mip360_rays 0 --precision bf16 > $O/trace1.log 2>&1
python $R/tools/rocpd_timeline.py $(ls $O/trace1/*/*.db | head -1) > $O/timeline_bf16.md
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$c -- python $R/bench.py --steps 4 --warmup 1 --no_cpu_baseline --large_batch 0 --mip360_rays 0 --precision bf16 > $O/pmc_$c.log 2>&1
  python $R/tools/rocpd_pmc.py $(ls $O/pmc_$c/*/*.db | head -1) _kernel >> $O/hbm_traffic.txt 2>&1
done
rm -rf $O/trace $O/trace1 $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
# MipNeRF-360 step (config 5): kernel stats + one step's timeline
timeout 300 rocprofv3 --kernel-trace --stats -d $O/m360 -- python $R/tools/mip360_bench.py --rays 4096 --steps 6 --warmup 2 > $O/mip360_bench.json 2> $O/mip360.err
python $R/tools/rocpd_stats.py $(ls $O/m360/*/*.db | head -1) > $O/mip360_kernel_stats.md
python $R/tools/rocpd_timeline.py $(ls $O/m360/*/*.db | head -1) resample_kernel 3 > $O/mip360_timeline.md
rm -rf $O/m360
# MipNeRF-360 dense-layer probes: GEMM micro-benchmarks (next to torch / hipBLASLt), time against K and tile count, DMA
# address-pattern rate, PMC passes over one NerfMLP-shaped layer
timeout 200 python $R/tools/probes/mip360_gemm_bench.py --check > $O/mip360_gemm_bench.txt 2>&1
timeout 200 python $R/tools/probes/mip360_gemm_shapes.py > $O/mip360_gemm_shapes.txt 2>&1
timeout 200 python $R/tools/probes/mip360_dw_bench.py > $O/mip360_dw_bench.txt 2>&1
[ -x $R/tools/probes/dma_pattern_probe ] && timeout 100 $R/tools/probes/dma_pattern_probe > $O/dma_pattern_probe.txt 2>&1
RINGS="1 0" bash $R/tools/probes/pmc_gemm_one.sh > /dev/null 2>&1; cp $R/gpurun_out/pmc_gemm_one.txt $O/pmc_gemm_one.txt
