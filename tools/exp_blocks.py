"""BASELINE config 3: 16384x16384 RGBA8, tile sweep 16/32/64 (directional, Lanczos3, opaque): kernel time, MP/s, GB/s."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from __graft_entry__ import load_product
P = load_product()
h = P.Handle(0)
frames = h.synth_frames_device(1, 16384, 16384, 4, 0, 0)
res = {}
for bs in (16, 32, 64):
    out = h.shrink_frames_device(frames, bs, bs, 1, 4, 16.0)
    for _ in range(40): h.shrink_frames_device(frames, bs, bs, 1, 4, 16.0, out=out)  # let the clocks settle
    torch.cuda.synchronize()
    h.enable_timing(True)
    for _ in range(40): h.shrink_frames_device(frames, bs, bs, 1, 4, 16.0, out=out)
    ms = h.last_kernel_ms(); h.enable_timing(False)
    ow, oh = out[1], out[2]
    wbytes = int((ow.long() * oh.long()).sum().item()) * 4 + 12 * ow.numel()
    algo = frames.numel() + wbytes
    res[bs] = dict(kernel_ms=round(ms, 3), mp_per_s=round(frames.numel() / 4 / 1e6 / ms * 1e3), achieved_gbps=round(algo / ms / 1e6), tiles=ow.numel(),
                   kernel={16: "shrink16_kernel", 32: "shrink32_kernel", 64: "shrink64_kernel"}[bs])
    del out
print(json.dumps(res))
