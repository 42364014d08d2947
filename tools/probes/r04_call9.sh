#!/bin/bash
# Round 4, GPU call 9: the narrow weight-gradient kernel per job shape (wave = long block axis, hoisted LDS reads, two tiles
# per ring slot for the 160-column jobs): per-workgroup stamps, GPU suite, bench.
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
TAG=${1:-r04j}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
V=$R/outdoor_nerf_depth_amd/csrc/build/variants
NERFPP_HIP_LIB=$V/libnerfpp_hip_probes.so timeout 200 python $R/tools/probes/dw_stamps_probe.py --out $O/dw_stamps > $O/dw_stamps.txt 2>&1
( cd $R && timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log )
timeout 900 python $R/bench.py --no_cpu_baseline --mip360_rays 0 --render_frames 0 --cli_steps 0 > $O/bench.json 2> $O/bench.err
cat $O/dw_stamps.txt; tail -5 $O/pytest.log; python - <<PY
import json
d=json.load(open('$O/bench.json'))
print(d['value'], d['ms_per_step'], d['roofline']['all_kernels'], d.get('parity_mode',{}).get('value'), d.get('parity_forward_mode',{}).get('value'))
PY
