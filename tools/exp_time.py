"""Ablation timings on the GPU box (not part of the product): kernel time per variant."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from __graft_entry__ import load_product
P = load_product()
h = P.Handle(0)
NF = int(os.environ.get("NF", "8"))
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    h.enable_timing(True)
    for _ in range(n): fn()
    ms = h.last_kernel_ms(); h.enable_timing(False)
    return ms
res = {}
for dist, dname in ((0, "opaque"), (2, "flat"), (3, "noise")):
    frames = h.synth_frames_device(NF, 4320, 7680, 4, 0, dist)
    gb = frames.numel() / 1e9
    for mode, mname, factor in ((1, "dir", 16.0), (0, "oklab", 1.0)):
        ms_lod = timeit(lambda: h.lod_frames_device(frames, 32, 32, mode, factor))
        out = h.shrink_frames_device(frames, 32, 32, mode, 4, factor)
        ms_full = timeit(lambda: h.shrink_frames_device(frames, 32, 32, mode, 4, factor, out=out))
        ms_near = timeit(lambda: h.shrink_frames_device(frames, 32, 32, mode, 0, factor, out=out))
        res[f"{dname}/{mname}"] = dict(lod_ms=ms_lod, full_ms=ms_full, nearest_ms=ms_near, read_GBps_full=gb / ms_full * 1e3, read_GBps_lod=gb / ms_lod * 1e3)
        del out
    del frames
for bs in (16, 64):
    frames = h.synth_frames_device(4, 4096, 4096, 4, 0, 0)
    out = h.shrink_frames_device(frames, bs, bs, 1, 4, 16.0)
    ms = timeit(lambda: h.shrink_frames_device(frames, bs, bs, 1, 4, 16.0, out=out))
    res[f"block{bs}/dir"] = dict(full_ms=ms, read_GBps=frames.numel() / 1e9 / ms * 1e3)
    del out, frames
print(json.dumps(res, indent=1))
