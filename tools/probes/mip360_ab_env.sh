#!/bin/bash
# A/B of an environment switch of outdoor_nerf_depth_amd/mip360.py on one box, alternating, 3 rounds:
#   tools/probes/mip360_ab_env.sh MIP360_NO_FUSED_PROP
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2 3; do
for on in 0 1; do
  if [ $on = 1 ]; then export $1=1; else unset $1; fi
  t=$(timeout 200 python tools/mip360_bench.py 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), round(d['value']))")
  echo "$1=$on step: $t"
done; done
