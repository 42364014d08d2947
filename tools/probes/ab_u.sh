# A/B of library variants (csrc/build/variants/<name>.so): launch time of the training forward (or forward + backward with ARGS=--bwd)
V=$PWD/outdoor_nerf_depth_amd/csrc/build/variants
for rep in 1 2 3; do for v in ${VARIANTS:-u0 u4}; do NERFPP_HIP_LIB=$V/$v.so python tools/probes/time_split_fwd.py $ARGS 2>&1 | tail -n 1; done; done
