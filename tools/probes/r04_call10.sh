#!/bin/bash
# Round 4, GPU call 10: merged weight-gradient jobs (stamps, parity tests, bench) + the multi-seed trajectory comparison
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
TAG=${1:-r04m}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
V=$R/outdoor_nerf_depth_amd/csrc/build/variants
NERFPP_HIP_LIB=$V/libnerfpp_hip_probes.so timeout 200 python $R/tools/probes/dw_stamps_probe.py --out $O/dw_stamps > $O/dw_stamps.txt 2>&1
( cd $R && timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log )
timeout 900 python $R/bench.py --no_cpu_baseline --mip360_rays 0 --render_frames 0 --cli_steps 0 > $O/bench.json 2> $O/bench.err
timeout 1500 python $R/tools/probes/traj_seeds.py --modes l1 kl mse --seeds 5 --steps 200 1000 --out $O/traj_seeds.json > $O/traj_seeds.txt 2>&1
grep "job" $O/dw_stamps.txt | tail -8; tail -5 $O/pytest.log; grep "==" $O/traj_seeds.txt; python - <<PY
import json
d=json.load(open('$O/bench.json'))
print(d['value'], d['ms_per_step'], d['roofline']['all_kernels'], d.get('parity_mode',{}).get('value'), d.get('parity_forward_mode',{}).get('value'))
PY
