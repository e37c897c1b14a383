import sys
sys.path.insert(0, '.')
from solex_ser_recon_en_amd import ops, synth
stack = synth.synth_frames_torch(2100, 2000, 200, 16, seed=0)
for rep in range(2):
    for pitch in [800000, 802816, 806912, 811008, 815104, 819200, 823296, 827392, 835584, 843776, 851968, 860160, 868352, 876544, 884736]:
        vecs = pitch // 16
        rate, shape = ops.stream_read_ceiling(stack[:1900], mode=2, vecs_per_frame=vecs, shapes=((2, 4), (1, 8)), reps=5)
        print('pitch %8d B = %7.2f KiB  /8K=%.2f  %.0f GB/s %s' % (pitch, pitch / 1024, pitch / 8192, rate, shape))
