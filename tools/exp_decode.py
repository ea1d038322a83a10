"""Decode-side kernel times on the bench batch (8 x 8K RGBA frames, 32x32 tiles): device reader and expand."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from __graft_entry__ import load_product
P = load_product()
h = P.Handle(0)
frames = h.synth_frames_device(8, 4320, 7680, 4, 0, 0)
shape = tuple(frames.shape)
vals, ow, oh, slots = h.shrink_frames_device(frames, 32, 32, 1, 4, 16.0)
offs, buf = h.encode_frames_device(shape, 32, 32, vals, ow, oh, slots)
torch.cuda.synchronize()
def timeit(fn, n=30, warm=30):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
dout = h.decode_frames_device(buf, offs, shape, 32, 32)
dec = timeit(lambda: h.decode_frames_device(buf, offs, shape, 32, 32, out=dout), n=20, warm=10)
out = h.expand_frames_device(shape, 32, 32, 4, ow, oh, slots)
res = {"decode_ms": round(dec, 3), "file_bytes": int(offs[-1])}
for filt, name in ((0, "nearest"), (1, "bilinear"), (2, "catmullrom"), (3, "gaussian"), (4, "lanczos3")):
    res["expand_%s_ms" % name] = round(timeit(lambda: h.expand_frames_device(shape, 32, 32, filt, ow, oh, slots, out=out)), 3)
res["frame_bytes_written"] = int(out.numel())
print(res)
