#!/bin/bash
# VERDICT r04 item 1, bounding build: libnerfpp_hip.so variants whose bf16 training forward leaves some H_l unsaved
# (-DNERFPP_SKIP_H=mask) next to the probes build of the weight-gradient kernel, which can emulate the one-layer recompute
# (NERFPP_DW_DEBUG bit 4; slices via NERFPP_DW_RC_K / NERFPP_DW_RC_K0).  Garbage gradients: timing only.
#   tools/probes/build_recompute_probe.sh -1   -> csrc/build/variants/skiph_-1.so: the mask is read per launch from NERFPP_SKIP_H_RT
#   (tools/probes/recompute_probe.py toggles it, NERFPP_DW_DEBUG and the slice plan between blocks of one process)
set -e
cd "$(dirname "$0")/../.."
tools/probes/build_probes.sh > /dev/null
C=outdoor_nerf_depth_amd/csrc
V=$C/build/variants
CC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fhip-fp32-correctly-rounded-divide-sqrt -DNERFPP_PROBES"
$CC -c $C/nerfpp_api.hip -o $V/q_nerfpp_api.o &
$CC -c $C/nerfpp_dw.hip -o $V/q_nerfpp_dw.o &
wait
for m in "$@"; do
  $CC -DNERFPP_SKIP_H=$m -DNERFPP_MLP_PART=2 -c $C/nerfpp_mlp.hip -o $V/q_mlp2_$m.o
  objs="$V/q_nerfpp_api.o $V/q_nerfpp_dw.o $C/build/nerfpp_tables.o $C/build/nerfpp_render.o $C/build/nerfpp_optim.o $C/build/nerfpp_comm.o $V/q_mlp2_$m.o"
  for k in 0 1 3 4 5 6 7 8; do objs="$objs $C/build/nerfpp_mlp_$k.o"; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $V/skiph_$m.so $objs
done
rm -f $V/q_*.o
ls -la $V/*.so
