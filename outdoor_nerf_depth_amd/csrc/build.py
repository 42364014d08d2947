"""Build libnerfpp_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python outdoor_nerf_depth_amd/csrc/build.py [--force]
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
OUT = os.path.join(PKG, 'libnerfpp_hip.so')
OBJ = os.path.join(HERE, 'build')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
COMMON = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function',
          '-fhip-fp32-correctly-rounded-divide-sqrt']
SOURCES = {
    'nerfpp_tables.hip': [],
    'nerfpp_render.hip': ['-ffp-contract=off'],     # bit-exact sample bins: no implicit FMA
    # the fully unrolled MLP kernels: one translation unit per instantiation (nerfpp_mlp.hip, NERFPP_MLP_PART)
    **{('nerfpp_mlp.hip', k): ['-DNERFPP_MLP_PART=%d' % k] for k in range(9)},
    'nerfpp_dw.hip': [],
    'nerfpp_optim.hip': ['-ffp-contract=off'],      # Adam rounds like torch
    'nerfpp_api.hip': [],
    'nerfpp_comm.hip': [],                         # RCCL entry points (librccl.so.1 bound with dlopen at first use)
}
HEADERS = ['nerfpp_common.h', 'nerfpp_kernels.h', 'probe_env.h', 'nerfpp_mlp_probes.h', 'nerfpp_mlp_split.h', os.path.join('..', '..', 'include', 'nerfpp_hip.h')]
# SURVEY 8 f-4 (MipNeRF-360 path): its own shared object and C ABI (include/mip360_hip.h)
OUT_MIP360 = os.path.join(PKG, 'libmip360_hip.so')
SOURCES_MIP360 = {
    'mip360_kernels.hip': ['-ffp-contract=off'],    # arithmetic order of the oracle
    'mip360_gemm.hip': [],
    'mip360_fm.hip': [],
    'mip360_prop.hip': [],                          # the PropMLP forward / dX chain as one launch each
    'mip360_view.hip': [],                          # the NerfMLP's view branch forward as one launch
    'mip360_train.hip': [],
    'mip360_api.hip': [],
}
HEADERS_MIP360 = ['probe_env.h', 'mip360_gemm_probes.h', 'mip360_fm_probes.h', os.path.join('..', '..', 'include', 'mip360_hip.h')]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src, flags, headers=None):
    part = ''
    if isinstance(src, tuple):
        src, part = src[0], '_%d' % src[1]
    obj = os.path.join(OBJ, src.replace('.hip', part + '.o'))
    deps = [os.path.join(HERE, src)] + [os.path.join(HERE, h) for h in (headers or HEADERS)] + [__file__]
    if _stale(obj, deps):
        cmd = [HIPCC] + COMMON + flags + ['-c', os.path.join(HERE, src), '-o', obj]
        subprocess.check_call(cmd)
    return obj


def build(force=False):
    os.makedirs(OBJ, exist_ok=True)
    if force:
        for f in os.listdir(OBJ):
            if os.path.isfile(os.path.join(OBJ, f)):
                os.remove(os.path.join(OBJ, f))
    with ThreadPoolExecutor(max_workers=int(os.environ.get('NERFPP_BUILD_JOBS', '8'))) as ex:
        objs = list(ex.map(lambda kv: _compile(*kv), SOURCES.items()))
        objs2 = list(ex.map(lambda kv: _compile(kv[0], kv[1], HEADERS_MIP360), SOURCES_MIP360.items()))
    if force or _stale(OUT, objs):
        subprocess.check_call([HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', OUT] + objs)
    if force or _stale(OUT_MIP360, objs2):
        subprocess.check_call([HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', OUT_MIP360] + objs2)
    return OUT


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
