#!/bin/bash
# Diagnostic builds with the experiment switches compiled in (-DNERFPP_PROBES, csrc/probe_env.h): the sources that read
# the environment are recompiled, the MLP kernel objects are the stock ones.
#   tools/probes/build_probes.sh  -> outdoor_nerf_depth_amd/csrc/build/variants/lib{nerfpp,mip360}_hip_probes.so
# Run with NERFPP_HIP_LIB=... / MIP360_HIP_LIB=... pointing at them (the directory travels with gpurun).
set -e
cd "$(dirname "$0")/../.."
python outdoor_nerf_depth_amd/csrc/build.py > /dev/null
C=outdoor_nerf_depth_amd/csrc
V=$C/build/variants
mkdir -p $V
CC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fhip-fp32-correctly-rounded-divide-sqrt -DNERFPP_PROBES"
$CC -c $C/nerfpp_api.hip -o $V/p_nerfpp_api.o &
$CC -c $C/nerfpp_dw.hip -o $V/p_nerfpp_dw.o &
$CC -c $C/mip360_gemm.hip -o $V/p_mip360_gemm.o &
$CC -c $C/mip360_train.hip -o $V/p_mip360_train.o &
wait
objs="$V/p_nerfpp_api.o $V/p_nerfpp_dw.o $C/build/nerfpp_tables.o $C/build/nerfpp_render.o $C/build/nerfpp_optim.o $C/build/nerfpp_comm.o"
for k in 0 1 2 3 4 5 6 7 8; do objs="$objs $C/build/nerfpp_mlp_$k.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $V/libnerfpp_hip_probes.so $objs
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $V/libmip360_hip_probes.so $V/p_mip360_gemm.o $V/p_mip360_train.o \
  $C/build/mip360_kernels.o $C/build/mip360_api.o $C/build/mip360_fm.o
rm -f $V/p_*.o
ls -la $V/*_probes.so
