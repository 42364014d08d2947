#!/bin/bash
# Round 4, GPU call 8: cycles against wall time INSIDE one process: per-block stamps + HIP-event launch times of the training
# forward for the stock stream, without stores, without MFMAs and with idle cycles added to the loader wave; then suite + bench.
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r04h; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
V=$R/outdoor_nerf_depth_amd/csrc/build/variants
for rep in 1 2; do
  for v in st_base st_dbg1 st_nomfma st_sleep4 st_sleep8 st_sleep16; do
    NERFPP_HIP_LIB=$V/$v.so timeout 200 python $R/tools/probes/stamps_probe.py --what fwd --out $O/${v}_$rep 2>/dev/null | grep -E "launch|tile:" | sed "s/^/$v rep$rep: /" >> $O/cycles_vs_time.txt
  done
done
( cd $R && timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log )
timeout 900 python $R/bench.py --no_cpu_baseline > $O/bench.json 2> $O/bench.err
ls -la $O
