#!/bin/bash
# Build a variant of libmip360_hip.so with extra -D flags for one source: tools/probes/mip360_variant.sh <name> <source.hip> <flags...>
# -> outdoor_nerf_depth_amd/csrc/build/variants/mip360_<name>.so (select with MIP360_HIP_LIB)
set -e
cd "$(dirname "$0")/../.."
name=$1; src=$2; shift 2
C=outdoor_nerf_depth_amd/csrc
mkdir -p $C/build/variants
python $C/build.py > /dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -fhip-fp32-correctly-rounded-divide-sqrt -DNERFPP_PROBES "$@" -c $C/$src -o $C/build/variants/mip360_${name}_$(basename $src .hip).o
objs=""
for f in mip360_kernels mip360_gemm mip360_fm mip360_train mip360_api; do
  if [ "$f.hip" = "$src" ]; then objs="$objs $C/build/variants/mip360_${name}_$f.o"; else objs="$objs $C/build/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $C/build/variants/mip360_$name.so $objs
echo $C/build/variants/mip360_$name.so
