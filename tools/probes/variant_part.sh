#!/bin/bash
# Diagnostic build that recompiles ONE instantiation of the MLP kernels (nerfpp_mlp.hip, -DNERFPP_MLP_PART=<part>) with extra
# flags and links it with the stock objects of everything else (tools/probes/variant.sh rebuilds all nine parts).
#   tools/probes/variant_part.sh <name> <part> "<extra flags>"   -> outdoor_nerf_depth_amd/csrc/build/variants/<name>.so
set -e
cd "$(dirname "$0")/../.."
C=outdoor_nerf_depth_amd/csrc
V=$C/build/variants
mkdir -p $V
name=$1; part=$2; flags=$3
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fhip-fp32-correctly-rounded-divide-sqrt \
  -DNERFPP_PROBES $flags -DNERFPP_MLP_PART=$part -c $C/nerfpp_mlp.hip -o $V/${name}_$part.o
objs="$V/${name}_$part.o"
for k in 0 1 2 3 4 5 6 7 8; do [ $k = $part ] || objs="$objs $C/build/nerfpp_mlp_$k.o"; done
for s in nerfpp_tables nerfpp_render nerfpp_dw nerfpp_optim nerfpp_api nerfpp_comm; do objs="$objs $C/build/$s.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $V/$name.so $objs
rm -f $V/${name}_$part.o
echo $V/$name.so
