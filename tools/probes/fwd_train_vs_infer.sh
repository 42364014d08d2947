#!/bin/bash
# the bf16 training forward decomposed against the inference forward -- one stamps build per component
# removed (cumulative), each run in its own process: launch time (HIP events), tile cycles (s_memtime stamps), clock.
#   tools/probes/variant_part.sh st0_infer 0 "-DNERFPP_STAMPS=0"; st2_train 2 "-DNERFPP_STAMPS=2"; st2_nostore ... -DNERFPP_DBG=1;
#   st2_nosave -DNERFPP_DBG=2; st2_nohandoff -DNERFPP_DBG=18; st2_nosign -DNERFPP_DBG=146; st2_ring -DNERFPP_DBG=146 -DNERFPP_EXP=8
O=gpurun_out/${1:-fwd_train_vs_infer}; mkdir -p $O
V=$PWD/outdoor_nerf_depth_amd/csrc/build/variants
for rep in 1 2; do
for v in st0_infer st2_train st2_nostore st2_nosave st2_nohandoff st2_nosign st2_ring; do
  extra=""; [ $v = st0_infer ] && extra="--infer"
  NERFPP_HIP_LIB=$V/$v.so python tools/probes/stamps_probe.py --what fwd --precision 1 $extra --out $O/${v}_$rep 2>&1 | grep -E "^launch|^tile|^summary" | tr '\n' ' ' | sed "s/^/$v /"; echo
done; done
