"""GPU: bench.py honours the driver's contract -- one JSON line with the required keys, the roofline
and cpu_baseline objects, whole-job throughput consistent with ms_per_step."""
import json
import os
import subprocess
import sys

import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_json_contract():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '4', '--warmup', '1',
                          '--n_rand', '128', '--large_batch', '256', '--cpu_rays', '8', '--mip360_rays', '256'], capture_output=True, text=True,
                         timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith('{')]
    assert len(lines) == 1
    r = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
              'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in r, k
    assert r['n_gpus'] == 1 and r['steps'] == 4 and r['warmup'] == 1 and r['vs_baseline'] is None
    assert r['unit'] == 'rays/s' and r['higher_is_better'] is True and r['scaling'] == 'weak' and r['data'] == 'synthetic'
    assert 'workload' in r['config'] and 'model' not in r['config']
    assert abs(r['value'] - 128 / (r['ms_per_step'] * 1e-3)) <= 1e-6 * r['value']
    rf = r['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert k in rf, k
    assert rf['bound'] in ('hbm', 'mfma') and abs(rf['frac'] - rf['achieved'] / rf['peak']) < 1e-9
    cb = r['cpu_baseline']
    assert cb['kind'] == 'port' and cb['unit'] == 'rays/s' and cb['value'] > 0 and cb['cores'] >= 1 and cb['sample']
    assert r['parity_mode']['value'] > 0 and r['large_batch']['n_rand_per_gpu'] == 256
    assert r['config5_mip360']['value'] > 0 and r['config5_mip360']['unit'] == 'rays/s'
