import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
from outdoor_nerf_depth_amd import mip360 as M
sys.path.insert(0, 'tools')
import mip360_bench as B
dev = torch.device('cuda:0')
rs = np.random.RandomState(0)
n = int(sys.argv[1])
prop, nerf = B.he_uniform(B.shapes(M.PROP_CFG), rs), B.he_uniform(B.shapes(M.NERF_CFG), rs)
d = rs.randn(n, 3).astype(np.float32); d /= np.linalg.norm(d, axis=-1, keepdims=True)
T = lambda x: torch.from_numpy(x).to(dev)
rays = dict(origins=T((rs.randn(n, 3) * 0.3).astype(np.float32)), directions=T(d), viewdirs=T(d.copy()), radii=T(np.full((n, 1), 2e-3, np.float32)), near=T(np.full((n, 1), 0.2, np.float32)), far=T(np.full((n, 1), 1e6, np.float32)))
tr = M.Mip360Trainer(prop, nerf, dev)
torch.cuda.synchronize(); print('init ok', flush=True)
sdist = torch.tensor([[0., 1.]], device=dev).repeat(n, 1); w = torch.ones(n, 1, device=dev)
sd, td = M.resample(sdist, w, 0.0, 0.5, 64, rays['near'], rays['far'], None); torch.cuda.synchronize(); print('resample ok', flush=True)
tm = tr.prop; rows = n * 64
enc_buf = torch.empty(rows, tm.W + 512, dtype=torch.bfloat16, device=dev)
M.cast_encode(td, rays['origins'], rays['directions'], rays['radii'], tr.basis_t, out=enc_buf[:, tm.W:], ld=tm.W + 512); torch.cuda.synchronize(); print('cast ok', flush=True)
out = torch.empty(rows, tm.W, dtype=torch.bfloat16, device=dev)
M.linear(enc_buf[:, tm.W:], tm.w[0], tm.b[0], act=1, out_bf16=out, m=rows, n=tm.W, k=512); torch.cuda.synchronize(); print('linear0 ok', flush=True)
dens, rgb, saved = M.mlp_forward_train(tm, enc_buf, rows, rays['viewdirs'], n, 64); torch.cuda.synchronize(); print('mlp ok', flush=True)
r = M.render_level(dens.reshape(n, 64), None, td, rays['directions']); torch.cuda.synchronize(); print('render ok', flush=True)
sd2, td2 = M.resample(sd, r['weights'], 0.5025, 0.5, 64, rays['near'], rays['far'], None); torch.cuda.synchronize(); print('resample2 ok', flush=True)
