"""How the frame-walking read rate depends on the frame pitch (probe mode 2, XOR only)."""
import sys
sys.path.insert(0, '.')
from solex_ser_recon_en_amd import ops, synth
stack = synth.synth_frames_torch(2100, 2000, 200, 16, seed=0)
res = []
for kib4 in range(3125, 3330, 3):          # pitch in units of 256 B: 781.25 KiB .. 832 KiB
    vecs = kib4 * 16
    rate, shape = ops.stream_read_ceiling(stack, mode=2, vecs_per_frame=vecs, shapes=((2, 4), (1, 8)), reps=4)
    res.append((rate, vecs * 16))
    print('pitch %8d B = %8.2f KiB  (mod 4K=%4d, mod 64K=%6d)  %.0f GB/s %s' % (vecs * 16, vecs * 16 / 1024, (vecs * 16) % 4096, (vecs * 16) % 65536, rate, shape))
print('best', max(res), 'worst', min(res))
