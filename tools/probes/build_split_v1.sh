#!/bin/bash
# The round-5 split-bf16 bodies (stage at a time, roles pipe in training) as a reference build for tools/probes/split_dump.py and the
# A/B scripts: parts 1 / 3 / 5 of nerfpp_mlp.hip with -DNERFPP_SPLIT_V2=0 -DNERFPP_TRICKLE=0, everything else the stock objects.
#   tools/probes/build_split_v1.sh  -> outdoor_nerf_depth_amd/csrc/build/variants/split_v1.so
set -e
cd "$(dirname "$0")/../.."
python outdoor_nerf_depth_amd/csrc/build.py > /dev/null
C=outdoor_nerf_depth_amd/csrc; V=$C/build/variants; mkdir -p $V
for k in 1 3 5; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fhip-fp32-correctly-rounded-divide-sqrt \
    -DNERFPP_PROBES -DNERFPP_SPLIT_V2=0 -DNERFPP_TRICKLE=0 -DNERFPP_MLP_PART=$k -c $C/nerfpp_mlp.hip -o $V/v1_$k.o &
done
wait
objs="$V/v1_1.o $V/v1_3.o $V/v1_5.o"
for k in 0 2 4 6 7 8; do objs="$objs $C/build/nerfpp_mlp_$k.o"; done
for s in nerfpp_tables nerfpp_render nerfpp_dw nerfpp_optim nerfpp_api nerfpp_comm; do objs="$objs $C/build/$s.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $V/split_v1.so $objs
rm -f $V/v1_1.o $V/v1_3.o $V/v1_5.o
echo $V/split_v1.so
