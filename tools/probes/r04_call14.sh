#!/bin/bash
# Round 4, GPU call 14: h7 saved under the sigma + colour-head stages (3 chunks per block) against under the colour head alone
# (4 per block): parity tests, alternating bench runs on one box.
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/${1:-r04v}; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
V=$R/outdoor_nerf_depth_amd/csrc/build/variants
( cd $R && timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round3.py tests/test_gpu_round4.py tests/test_gpu_train_cli.py -m gpu -q -x -k "not seeds" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log )
tail -3 $O/pytest.log
B="--no_cpu_baseline --large_batch 0 --mip360_rays 0 --cli_steps 0 --render_frames 0"
for rep in 1 2 3; do
  for v in new cpb4; do
    if [ $v = new ]; then unset NERFPP_HIP_LIB; else export NERFPP_HIP_LIB=$V/libnerfpp_hip_cpb4.so; fi
    timeout 300 python $R/bench.py $B --steps 100 --warmup 10 > $O/bench_${v}_$rep.json 2> $O/bench_${v}_$rep.err
    python - <<PY
import json
d=json.load(open('$O/bench_${v}_$rep.json'))
print('$v rep$rep', round(d['ms_per_step'],4), {k: v['ms'] for k,v in d['roofline']['all_kernels'].items()}, 'split', round(d['parity_mode']['ms_per_step'],3), 'split_fwd', round(d['parity_forward_mode']['ms_per_step'],3))
PY
  done
done | tee $O/ab.txt
