# L2 behaviour of the MipNeRF-360 GEMMs: hit / miss, fabric read bytes (double FETCH_SIZE on gfx950, MI355X_MICROARCH.md)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rm -f $R/gpurun_out/pmc_mip360_l2.txt
i=0
for grp in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pl/g$i -- python $R/tools/mip360_bench.py --rays 4096 --steps 2 --warmup 1 > /tmp/pl_g$i.log 2>&1
  python $R/tools/rocpd_pmc.py $(ls /tmp/pl/g$i/*/*.db | head -1) _kernel >> $R/gpurun_out/pmc_mip360_l2.txt 2>&1
done
