#!/bin/bash
# Round 4, GPU call 12: the remap layer folded into the colour head (forward and dX chain) against the build before it, on ONE
# box: GPU suite, alternating bench runs, rocprofv3 kernel stats + one step's timeline.
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
TAG=${1:-r04q}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
V=$R/outdoor_nerf_depth_amd/csrc/build/variants
( cd $R && timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log )
tail -8 $O/pytest.log
B="--no_cpu_baseline --large_batch 0 --mip360_rays 0 --cli_steps 0"
for rep in 1 2 3; do
  for v in fold nofold; do
    if [ $v = fold ]; then unset NERFPP_HIP_LIB; else export NERFPP_HIP_LIB=$V/libnerfpp_hip_nofold.so; fi
    timeout 300 python $R/bench.py $B --steps 100 --warmup 10 > $O/bench_${v}_$rep.json 2> $O/bench_${v}_$rep.err
    python - <<PY
import json
d=json.load(open('$O/bench_${v}_$rep.json'))
print('$v rep$rep', round(d['ms_per_step'],4), {k: v['ms'] for k,v in d['roofline']['all_kernels'].items()}, 'split', round(d['parity_mode']['ms_per_step'],3), 'split_fwd', round(d['parity_forward_mode']['ms_per_step'],3), 'render', round(d['render']['bf16']['s_per_frame'],4), d['render']['bf16']['mlp_kernels']['frac_of_bf16_mfma_peak'])
PY
  done
done | tee $O/ab.txt
unset NERFPP_HIP_LIB
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -- python $R/bench.py --steps 6 --warmup 2 $B --render_frames 0 --precision bf16 > $O/trace.log 2>&1
python $R/tools/rocpd_stats.py $(ls $O/trace/*/*.db | head -1) > $O/kernel_stats.md
python $R/tools/rocpd_timeline.py $(ls $O/trace/*/*.db | head -1) > $O/timeline.md
rm -rf $O/trace
grep "mlp_\|dw_kernel" $O/kernel_stats.md $O/timeline.md
