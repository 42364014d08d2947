"""Where a tile's time goes in the persistent ping-pong GEMM: s_memrealtime stamps per workgroup and tile (needs the library built
with -DMIP360_EXP_STAMPS, passed as MIP360_HIP_LIB).  Stamps: 0 tile start, 1 stage 0 landed, 2 K loop done, 3 accumulators staged
(before the wait), 4 staging barrier passed, 5 output stores issued, 6 final barrier passed."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from outdoor_nerf_depth_amd import mip360 as M                                  # noqa: E402
dev = torch.device('cuda:0')
m, n, k = 131072, 1024, int(sys.argv[1]) if len(sys.argv) > 1 else 1024
torch.manual_seed(0)
a = torch.randn(m, k, device=dev).to(torch.bfloat16)
w = (torch.randn(n, k, device=dev) / k ** 0.5).to(torch.bfloat16)
out = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
stamps = torch.zeros(256 * 16 * 2 * 8, dtype=torch.int64, device=dev)
for _ in range(3):
    M.linear(a, w, None, act=0, out_bf16=out, aux=stamps.view(torch.bfloat16))
torch.cuda.synchronize()
s = stamps.cpu().numpy().reshape(256, 16, 2, 8)[:, :8].astype(np.float64) * 0.01      # 100 MHz ticks -> us
t0 = s[:, 0, 0, 0].min()
names = ['tile start', 'stage 0 landed', 'K loop done', 'staged', 'staging barrier', 'stores issued', 'final barrier']
print('K = %d; mean over 256 workgroups, per tile: stamp time since kernel start (us), group 0 | group 1' % k)
for t in range(8):
    print('tile %d: ' % t + '  '.join('%s %.1f|%.1f' % (names[i][:14], (s[:, t, 0, i] - t0).mean(), (s[:, t, 1, i] - t0).mean()) for i in range(7)))
d = s[:, :, :, 1:7] - s[:, :, :, 0:6]
print('mean segment lengths (us), tiles 1..7:')
for i in range(6):
    print('  %-16s -> %-16s group 0 %.2f  group 1 %.2f' % (names[i], names[i + 1], d[:, 1:, 0, i].mean(), d[:, 1:, 1, i].mean()))
print('  tile period %.2f us' % (s[:, 1:, 0, 0] - s[:, :-1, 0, 0]).mean())
