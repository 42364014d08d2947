#!/usr/bin/env python
"""Socket power and shader clock while the MLP kernels loop -- direct evidence for (or against) the "power budget" reading of the
round-6 stamps (profiles/r06_split_stamps.md: matrix-pipe occupancy x clock stays constant when waiting cycles are removed).

A sampler thread reads the amdgpu hwmon files (power1_average / power1_input in uW, power1_cap, freq1_input in Hz; rocm-smi --json
as a fallback) every 50 ms while one kernel at a time is launched back to back for a few seconds:

    python tools/probes/power_trace.py --out gpurun_out/x/power_trace.json
"""
import argparse
import glob
import json
import os
import subprocess
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from outdoor_nerf_depth_amd import ops, _lib as L            # noqa: E402
from outdoor_nerf_depth_amd.model import init_level_params   # noqa: E402
from outdoor_nerf_depth_amd.synthetic import SyntheticKitti  # noqa: E402


def pci_bus_id():
    """PCI address of HIP device 0 (the box shows every GPU of the node in sysfs; this process sees one of them)"""
    import ctypes as C
    hip = C.CDLL('libamdhip64.so')
    buf = C.create_string_buffer(64)
    if hip.hipDeviceGetPCIBusId(buf, 64, 0) != 0:
        return None
    return buf.value.decode().lower()


def hwmon_files():
    out = {}
    bdf = pci_bus_id()
    dirs = sorted(glob.glob('/sys/bus/pci/devices/%s/hwmon/hwmon*' % bdf)) if bdf else []
    out['pci'] = bdf
    for d in dirs or sorted(glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*')):
        for key, names in (('power_uw', ('power1_average', 'power1_input')), ('cap_uw', ('power1_cap',)), ('sclk_hz', ('freq1_input',)),
                           ('mclk_hz', ('freq2_input',)), ('temp_mc', ('temp1_input',))):
            for n in names:
                p = os.path.join(d, n)
                if key not in out and os.path.exists(p):
                    try:
                        int(open(p).read())
                        out[key] = p
                    except Exception:
                        pass
        if 'power_uw' in out:
            break
    return out


def read_all(files):
    r = {}
    for k, p in files.items():
        if k == 'pci':
            continue
        try:
            r[k] = int(open(p).read())
        except Exception:
            r[k] = None
    return r


def smi_sample():
    try:
        j = json.loads(subprocess.run(['rocm-smi', '--showpower', '--showclocks', '--json'], capture_output=True, text=True, timeout=5).stdout)
        return j.get('card0', j)
    except Exception as e:
        return {'error': str(e)}


class Sampler(threading.Thread):
    def __init__(self, files, period=0.05):
        super().__init__(daemon=True)
        self.files, self.period, self.rows, self.on = files, period, [], True

    def run(self):
        while self.on:
            self.rows.append((time.time(), read_all(self.files)))
            time.sleep(self.period)


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--seconds', type=float, default=4.0)
    p.add_argument('--out', default='power_trace.json')
    a = p.parse_args()
    dev = torch.device('cuda:0')
    files = hwmon_files()
    print('hwmon files:', files, flush=True)
    rep = {'hwmon': files, 'smi_idle': smi_sample(), 'legs': []}
    T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    b = SyntheticKitti().random_batch(1024, np.random.RandomState(0))
    ray_o, ray_d = T(b['ray_o']), T(b['ray_d'])
    far, fg_z, bg_z = ops.sample_coarse(ray_o, ray_d, T(b['min_depth']), 192)
    engines = {n: ops.LevelEngine(init_level_params(1)[0].to(dev), precision=pr) for n, pr in (('bf16', L.PREC_BF16), ('split', L.PREC_SPLIT_BF16))}

    def fwd(eng, training):
        return lambda: eng.forward(ray_o, ray_d, far, fg_z, bg_z, training=training)

    def fwd_bwd(eng):
        def f():
            ret = eng.forward(ray_o, ray_d, far, fg_z, bg_z, training=True)
            eng.backward(torch.full_like(ret['rgb'], 1e-3), torch.full_like(ret['depth'], 1e-3), None)
        return f

    x = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
    big = torch.empty(1 << 30, device=dev, dtype=torch.uint8)
    legs = [('idle', None),
            ('bf16 inference forward (196 608 samples)', fwd(engines['bf16'], False)),
            ('bf16 training forward', fwd(engines['bf16'], True)),
            ('bf16 forward + backward + weight gradients', fwd_bwd(engines['bf16'])),
            ('split-bf16 inference forward', fwd(engines['split'], False)),
            ('split-bf16 training forward', fwd(engines['split'], True)),
            ('split-bf16 forward + backward + weight gradients', fwd_bwd(engines['split'])),
            ('torch bf16 GEMM 8192^3 (hipBLASLt)', lambda: torch.mm(x, x)),
            ('1 GiB memset (HBM writes)', lambda: big.zero_())]
    from outdoor_nerf_depth_amd import mip360
    once = {'MipNeRF-360 training step, 4096 rays (config 5; 300 steps, joined + deferred updates)':
            lambda: mip360.benchmark_step(dev, 4096, steps=300, warmup=3)}
    for name, f in legs + [(k, None) for k in once]:
        s = Sampler(files)
        torch.cuda.synchronize()
        t0 = time.time()
        s.start()
        n = 0
        if name in once:
            res = once[name]()
            n = 600
        elif f is None:
            time.sleep(a.seconds)
        else:
            while time.time() - t0 < a.seconds:
                for _ in range(20):
                    f()
                torch.cuda.synchronize()
                n += 20
        t1 = time.time()
        s.on = False
        s.join()
        smi = smi_sample() if name == 'idle' else None
        rows = [r for t, r in s.rows if t - t0 > 1.0]                # after a second of warm-up
        if name in once:                                             # (set-up at both ends: the middle 60 % of the call)
            rows = [r for t, r in s.rows if t0 + 0.2 * (t1 - t0) <= t <= t0 + 0.8 * (t1 - t0)]
        leg = {'leg': name, 'launches': n, 'ms_per_launch': (res.get('ms_per_step') if name in once else (t1 - t0) * 1e3 / n) if n else None, 'samples': len(rows)}
        for k in files:
            if k == 'pci':
                continue
            v = [r[k] for r in rows if r.get(k) is not None]
            if v:
                leg[k] = {'mean': float(np.mean(v)), 'min': int(np.min(v)), 'max': int(np.max(v))}
        if smi:
            leg['smi'] = smi
        rep['legs'].append(leg)
        pw = leg.get('power_uw', {}).get('mean')
        print('%-52s %s ms/launch  power %s W (cap %s W)  sclk %s MHz' % (
            name, '%.3f' % leg['ms_per_launch'] if n else '  -  ', '%.0f' % (pw / 1e6) if pw else '?',
            '%.0f' % (leg['cap_uw']['mean'] / 1e6) if 'cap_uw' in leg else '?',
            '%.0f' % (leg['sclk_hz']['mean'] / 1e6) if 'sclk_hz' in leg else '?'), flush=True)
    rep['smi_after'] = smi_sample()
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    json.dump(rep, open(a.out, 'w'), indent=1)


if __name__ == '__main__':
    main()
