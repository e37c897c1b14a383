import sys, os
sys.path.insert(0, '.')
import numpy as np, torch
from solex_ser_recon_en_amd import ops
from tools.bench_kernels import timeit
h, w = 2000, 2100
yy, xx = np.mgrid[0:h, 0:w]
r2 = ((xx - w/2)/900.0)**2 + ((yy - h/2)/900.0)**2
img = np.where(r2 < 1, 0.35 + 0.65*np.sqrt(np.clip(1-r2, 0, 1)), 0.02) * 0.8 * 65535
img = np.clip(img + 260*np.random.default_rng(0).standard_normal(img.shape), 0, 65535).astype(np.uint16)
t = ops.pitched_u16(h, w, 'cuda'); t[:, :] = torch.from_numpy(img).cuda()
n = h*w
for ranks in ([int(0.999999*(n-1))], [n//10, n-1]):
    med, best = timeit(lambda: ops.select_u16(t, ranks), iters=30)
    print(os.environ.get('SHG_SEL_BLOCKS'), ranks, 'select_u16 %.1f us median %.1f best' % (med*1e3, best*1e3))
