#!/usr/bin/env python
"""Golden vectors for the MipNeRF-360 oracle.  Runs ONLY in the build container: it imports the
one module of nerf-methods/mipnerf360 that is importable here (internal/geopoly.py is pure numpy; everything else
needs jax / flax / gin) and stores its OUTPUT -- the positional-encoding basis the MLPs are built on
(models.py:387-389: generate_basis('icosahedron', 2)) -- as tests/golden/mip360_basis.npz."""
import importlib.util
import os
import sys

import numpy as np

sys.dont_write_bytecode = True
REF = '/root/reference/nerf-methods/mipnerf360/internal/geopoly.py'
HERE = os.path.dirname(os.path.abspath(__file__))

spec = importlib.util.spec_from_file_location('ref_geopoly', REF)
geopoly = importlib.util.module_from_spec(spec)
spec.loader.exec_module(geopoly)
out = {}
for tess in (1, 2, 3):
    out['icosahedron_%d' % tess] = np.asarray(geopoly.generate_basis('icosahedron', tess), np.float64)
np.savez(os.path.join(HERE, 'mip360_basis.npz'), **out)
print({k: v.shape for k, v in out.items()})
