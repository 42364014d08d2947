#!/bin/bash
# Diagnostic builds of libnerfpp_hip.so: recompile the MLP kernels (nerfpp_mlp.hip, all 7 parts) with extra -D flags and
# link them with the stock objects of the other sources.
#   tools/probes/variant.sh <name> "<extra flags>"   -> outdoor_nerf_depth_amd/csrc/build/variants/<name>.so
# (-DNERFPP_PROBES is always added: the experiment switches live in csrc/nerfpp_mlp_probes.h, which only it includes)
# Run with NERFPP_HIP_LIB=$PWD/outdoor_nerf_depth_amd/csrc/build/variants/<name>.so (the directory travels with gpurun).
set -e
cd "$(dirname "$0")/../.."
C=outdoor_nerf_depth_amd/csrc
V=$C/build/variants
mkdir -p $V
name=$1; flags=$2
objs=""
for k in 0 1 2 3 4 5 6 7 8; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fhip-fp32-correctly-rounded-divide-sqrt \
    -DNERFPP_PROBES $flags -DNERFPP_MLP_PART=$k -c $C/nerfpp_mlp.hip -o $V/${name}_$k.o &
  objs="$objs $V/${name}_$k.o"
done
wait
for s in nerfpp_tables nerfpp_render nerfpp_dw nerfpp_optim nerfpp_api nerfpp_comm; do objs="$objs $C/build/$s.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $V/$name.so $objs
rm -f $V/${name}_*.o
echo $V/$name.so
