"""One dense layer of the NerfMLP shape (131072 x 1024 x 1024, ReLU + bit mask), a few launches: a PMC target."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from outdoor_nerf_depth_amd import mip360 as M                                  # noqa: E402

dev = torch.device('cuda:0')
m, n, k = 131072, 1024, int(sys.argv[1]) if len(sys.argv) > 1 else 1024
torch.manual_seed(0)
a = torch.randn(m, k, device=dev).to(torch.bfloat16)
w = (torch.randn(n, k, device=dev) / k ** 0.5).to(torch.bfloat16)
b = torch.randn(n, device=dev)
out = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
mask, ld = M.relu_mask_buffer(m, n, dev)
for _ in range(6):
    M.linear_relu_mask(a, w, b, out, mask, ld)
torch.cuda.synchronize()
