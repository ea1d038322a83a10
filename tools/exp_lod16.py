"""16x16 tiles, 16384^2: detector-only launch vs full step, and all-clone / all-small factor extremes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from __graft_entry__ import load_product
P = load_product()
h = P.Handle(0)
frames = h.synth_frames_device(1, 16384, 16384, 4, 0, 0)
def t(fn):
    for _ in range(30): fn()
    torch.cuda.synchronize(); h.enable_timing(True)
    for _ in range(30): fn()
    ms = h.last_kernel_ms(); h.enable_timing(False); return ms
for bs in (16, 32):
    print(bs, "lod only %.3f" % t(lambda: h.lod_frames_device(frames, bs, bs, 1, 16.0)))
    for factor in (16.0, 0.01, 1e6):
        out = h.shrink_frames_device(frames, bs, bs, 1, 4, factor)
        ms = t(lambda: h.shrink_frames_device(frames, bs, bs, 1, 4, factor, out=out))
        import collections
        hist = collections.Counter(zip(out[1].flatten().tolist()[:20000], out[2].flatten().tolist()[:20000])).most_common(3)
        print(bs, "factor %g: %.3f ms  %s" % (factor, ms, hist), flush=True)
        del out
