for rep in 1 2 3; do
for v in pre_p2pack NEW; do
  if [ $v = NEW ]; then unset NERFPP_HIP_LIB; else export NERFPP_HIP_LIB=$PWD/outdoor_nerf_depth_amd/csrc/build/variants/$v.so; fi
  python bench.py --precision split --no_cpu_baseline --large_batch 0 --mip360_rays 0 --render_frames 0 --cli_steps 0 --steps 40 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', round(d['value']), round(d['ms_per_step'],4), d['roofline']['share_ms_per_step'])"
  python tools/kbench.py --prec 2 --only infer 2>&1 | tail -1
done; done
