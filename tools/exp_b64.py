"""64x64 tiles on the 16384^2 frame: detector-only and full step, size histogram."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from __graft_entry__ import load_product
P = load_product()
h = P.Handle(0)
bs = int(os.environ.get("BS", "64"))
frames = h.synth_frames_device(1, 16384, 16384, 4, 0, int(os.environ.get("DIST", "0")))
def timeit(fn, n=60):
    for _ in range(40): fn()
    torch.cuda.synchronize()
    h.enable_timing(True)
    for _ in range(n): fn()
    ms = h.last_kernel_ms(); h.enable_timing(False)
    return ms
lod = timeit(lambda: h.lod_frames_device(frames, bs, bs, 1, 16.0))
out = h.shrink_frames_device(frames, bs, bs, 1, 4, 16.0)
full = timeit(lambda: h.shrink_frames_device(frames, bs, bs, 1, 4, 16.0, out=out))
ow, oh = out[1], out[2]
key = (ow.long() * 1000 + oh.long()).flatten()
u, c = torch.unique(key, return_counts=True)
print({k: v for k, v in os.environ.items() if k.startswith("PXZ_")}, "lod %.4f ms  step %.4f ms" % (lod, full))
print({f"{int(k) // 1000}x{int(k) % 1000}": int(n) for k, n in zip(u.tolist(), c.tolist())})
