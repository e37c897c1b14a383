"""The CLAHE blend kernel (k_clahe_interp_vm, with the select's counting and window) alone on ONE large synthetic image -- 8192 x 8192 px,
a limb-darkened disk with noise, 16 disks' worth of pixels -- so that its ablations need no scan to go on.  Run under rocprofv3
--kernel-trace --stats and read the kernel's average with tools/kstats.py; SHG_SELECT_WINDOW=1 SHG_INTERP_SHAPE=290 give the launch a
Doppler stack gets (eight pixels a lane, 4 lanes x 4 waves across, the percentile window)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from solex_ser_recon_en_amd import hostmath, ops  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
g = torch.Generator(device='cuda').manual_seed(5)
yy, xx = torch.meshgrid(torch.arange(n, device='cuda', dtype=torch.float32), torch.arange(n, device='cuda', dtype=torch.float32), indexing='ij')
r = torch.sqrt((yy - n / 2) ** 2 + (xx - n / 2) ** 2) / (0.44 * n)
mu = torch.sqrt(torch.clamp(1 - r * r, min=0))
img = torch.where(r < 1, 0.8 * 65535 * (0.4 + 0.6 * mu), torch.full_like(r, 0.02 * 65535)) + 262 * torch.randn((n, n), device='cuda', generator=g)
frame = ops.pitched_u16(n, n, img.device)
frame.copy_(img.clamp(0, 65535).to(torch.int32).to(torch.uint16))
out5 = torch.zeros(5, dtype=torch.float64, device='cuda')
npx = n * n
rf = hostmath.percentile_plan(npx, 99.9999)[:2] if hasattr(hostmath, 'percentile_plan') else (npx - 8, npx - 7)
rc = (npx // 10, npx // 10 + 1, npx - 1)
for _ in range(6):
    cl1 = ops.contrast_stats_u16(frame, rf, rc, out5)
torch.cuda.synchronize()
print('out5', out5.cpu().numpy())
