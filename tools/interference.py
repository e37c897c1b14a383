"""What does each part of a scan's chain cost pass A when it runs beside it?  Pass A (k_accumulate_vec, the HBM-bound kernel the
rate of a batch is bound by) loops on one stream; ONE workload of the chain loops on another; printed: pass A's duration beside
it, the workload's duration alone and beside pass A, and the duty cycle of the workload's stream.  (The chains move a tenth of
pass A's bytes and cost it a third of its speed: this says which kernels do that.)
    python3 tools/interference.py [seconds per workload]"""
import contextlib
import io
import os
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from solex_ser_recon_en_amd import SHG_MAIN, Solex_recon, _lib, ops, solex_util, stages, synth  # noqa: E402
from solex_ser_recon_en_amd.video_reader import array_reader  # noqa: E402

dur = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
dev = torch.device('cuda', 0)
stack = synth.synth_frames_torch(2000, 2000, 200, 16, seed=0, padded=True)
stack2 = synth.synth_frames_torch(2000, 2000, 200, 16, seed=1, padded=True)
opts = SHG_MAIN.default_options()
opts['_nolog'] = True
with contextlib.redirect_stdout(io.StringIO()):
    disk_list, bounds, hdr = Solex_recon.solex_read(array_reader(stack2), opts)
mean, mx = ops.accumulate_mean_max(stack2)
fit_res = stages.mean_fit(stack2, 2000)
fit = fit_res['fit'] if isinstance(fit_res, dict) else fit_res[1]
disk = disk_list[0].t if hasattr(disk_list[0], 't') else disk_list[0]
geo = stages.limb_fit(disk)
frame = ops.warp_rows_u16(disk, geo['h00'], geo['h01'], geo['h02'], geo['out_h'], geo['out_w'])
h, w = frame.shape
cl1 = ops.clahe(frame)
window = 301
trans = dict(circle=geo['circle'], borders=geo['borders'], taps=solex_util.savgol_taps(window), window=window)
disc = (int(geo['circle'][0]), int(geo['circle'][1]), int(geo['circle'][2]))
n_px = h * w
ranks = [int(0.1 * (n_px - 1)), int(0.1 * (n_px - 1)) + 1, n_px - 1]
out5 = torch.zeros(5, dtype=torch.float64, device=dev)
y1, y2 = int(geo['circle'][1] - geo['circle'][2]) + 2, int(geo['circle'][1] + geo['circle'][2]) - 2
xa = np.zeros(y2 - y1, dtype=np.int32) + int(geo['circle'][0] - 0.5 * geo['circle'][2])
xb = np.zeros(y2 - y1, dtype=np.int32) + int(geo['circle'][0] + 0.5 * geo['circle'][2])
shifts = opts['shift']
torch.cuda.synchronize()

workloads = {
    'nothing': None,
    'blur+argmin (mean image)': lambda: ops.blur_argmin_u16(mean, 25, 17, 12, 187),
    'extract (pass B, S=2)': lambda: stages.extract(stack2, fit, shifts),
    'limb fit (9 launches + host)': lambda: stages.limb_fit(disk),
    'warp': lambda: ops.warp_rows_u16(disk, geo['h00'], geo['h01'], geo['h02'], geo['out_h'], geo['out_w']),
    'rowpair stats': lambda: ops.rowpair_logratio_stats(frame, y1, y2, xa, xb),
    'process_frames (transv. + CLAHE + products)': lambda: stages.process_frames([frame], trans, None, disc),
    'process_frames without transversalium': lambda: stages.process_frames([frame], None, None, disc),
    'clahe (hist + reduce + lut + blend)': lambda: ops.clahe(frame),
    'select on cl1 (2 passes)': lambda: ops.select_u16(cl1, ranks),
    'products': lambda: ops.contrast_products_u16(frame, cl1, [0, 60000, 0, 50000, 1000, 60000], disc),
}
# plain streaming kernels of the same sizes (PyTorch's own): is it what the chain's kernels do, or that anything runs at all?
_a16 = torch.zeros(8 << 20, dtype=torch.int16, device=dev)
_b16 = torch.ones(8 << 20, dtype=torch.int16, device=dev)
_a1g = torch.zeros(256 << 20, dtype=torch.int16, device=dev)
_b1g = torch.ones(256 << 20, dtype=torch.int16, device=dev)
_m64 = torch.rand(1024, 1024, dtype=torch.float64, device=dev)
_m64o = torch.empty_like(_m64)
_m32 = torch.rand(2048, 2048, dtype=torch.float32, device=dev)
_m32o = torch.empty_like(_m32)
_v32 = torch.rand(1 << 20, dtype=torch.float32, device=dev)
_v32o = torch.empty_like(_v32)
workloads.update({
    'torch: copy 16 MB -> 16 MB': lambda: _a16.copy_(_b16),
    'torch: fill 16 MB': lambda: _a16.fill_(3),
    'torch: sum of 16 MB': lambda: _b16.sum(),
    'torch: copy 512 MB -> 512 MB': lambda: _a1g.copy_(_b1g),
    'torch: fill 512 MB': lambda: _a1g.fill_(3),
    'torch: sum of 512 MB': lambda: _b1g.sum(),
    # compute-bound co-runners whose operands stay in L2: is it memory traffic at all?
    'torch: fp64 matmul 1024^3': lambda: torch.mm(_m64, _m64, out=_m64o),
    'torch: fp32 matmul 2048^3': lambda: torch.mm(_m32, _m32, out=_m32o),
    'torch: sin of 1 M floats': lambda: torch.sin(_v32, out=_v32o),
})
if os.environ.get('SHG_INTERFERENCE_ONLY'):
    keep = os.environ['SHG_INTERFERENCE_ONLY'].split(',')
    workloads = {k: v for k, v in workloads.items() if v is None or any(t in k for t in keep)}

ws = torch.empty(_lib.lib.shg_accumulate_workspace_bytes(2000, 200, 2000, 2), dtype=torch.uint8, device=dev)
lane = torch.cuda.Stream(device=dev, priority=-1)
side = torch.cuda.Stream(device=dev)


def pass_a_loop(seconds):
    """-> mean duration (ms) of pass A launched back to back for `seconds` (the library's own events around the kernel)"""
    _lib.profile_reset()
    _lib.profile_enable(True, only=('accumulate',))
    with torch.cuda.stream(lane):
        t_end = time.perf_counter() + seconds
        k = 0
        while time.perf_counter() < t_end:
            ops.accumulate_sum_max(stack, ws)
            k += 1
            if k % 8 == 0:
                lane.synchronize()
        lane.synchronize()
    _lib.profile_enable(False)
    ms, n = _lib.profile_get('accumulate')
    _lib.profile_reset()
    return ms / max(n, 1), n


def side_loop(fn, stop, out):
    torch.cuda.set_device(dev)
    n = 0
    t0 = time.perf_counter()
    with torch.cuda.stream(side):
        while not stop.is_set():
            fn()
            n += 1
            if n % 8 == 0:
                side.synchronize()
        side.synchronize()
    out.append((n, time.perf_counter() - t0))


def alone(fn, seconds):
    n = 0
    with torch.cuda.stream(side):
        side.synchronize()
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            fn()
            n += 1
            if n % 8 == 0:
                side.synchronize()
        side.synchronize()
        return (time.perf_counter() - t0) / n * 1e3


base = None
print('%-46s %10s %10s %12s %12s' % ('workload beside pass A', 'pass A ms', 'x alone', 'workload ms', 'alone ms'))
for name, fn in workloads.items():
    if fn is None:
        a_ms, n = pass_a_loop(dur)
        base = a_ms
        print('%-46s %10.3f %10s %12s %12s' % (name, a_ms, '1.00', '-', '-'))
        continue
    fn()
    torch.cuda.synchronize()
    solo = alone(fn, dur / 2)
    stop, out = threading.Event(), []
    t = threading.Thread(target=side_loop, args=(fn, stop, out))
    t.start()
    time.sleep(0.05)
    a_ms, n = pass_a_loop(dur)
    stop.set()
    t.join()
    k, wall = out[0]
    print('%-46s %10.3f %10.2f %12.3f %12.3f' % (name, a_ms, a_ms / base, wall / k * 1e3, solo))
