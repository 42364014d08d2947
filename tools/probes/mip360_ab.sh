#!/bin/bash
# A/B of libmip360 variants on one box: tools/probes/mip360_ab.sh <variant> ... (alternating with the in-tree build, 2 rounds)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
for v in base "$@"; do
  if [ $v = base ]; then unset MIP360_HIP_LIB; else export MIP360_HIP_LIB=$PWD/outdoor_nerf_depth_amd/csrc/build/variants/mip360_$v.so; fi
  ok=$(timeout 200 python tools/probes/mip360_fm_bench.py --check-only 2>&1 | grep -c "^check")
  t=$(timeout 200 python tools/mip360_bench.py 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), round(d['value']))")
  echo "$v checks_ok=$ok step: $t"
done; done
