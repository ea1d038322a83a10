"""BASELINE config 5 on one GPU: a batch of 64 8K RGBA frames (8.5 GB in, 8.5 GB of slots out) in one call.
Checks batch independence on the last frame (offsets beyond 4 GB) and prints the step times."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from __graft_entry__ import load_product
P = load_product()
h = P.Handle(0)
N = 64
frames = h.synth_frames_device(N, 4320, 7680, 4, 0, 0)
for mode, factor, name in ((1, 16.0, "directional"), (0, 1.0, "shrink_by")):
    out = h.shrink_frames_device(frames, 32, 32, mode, 4, factor)
    torch.cuda.synchronize()
    h.enable_timing(True)
    for _ in range(5): h.shrink_frames_device(frames, 32, 32, mode, 4, factor, out=out)
    ms = h.last_kernel_ms(); h.enable_timing(False)
    alone = h.shrink_frames_device(frames[N - 1:N], 32, 32, mode, 4, factor)
    ok = all(bool((a[N - 1:N] == b).all()) for a, b in zip(out[:3], alone[:3]))
    valid = (alone[1].long() * alone[2].long() * 4)[0]
    idx = torch.arange(4096, device="cuda")[None, :] < valid[:, None]
    ok = ok and bool(((out[3][N - 1] == alone[3][0]) | ~idx).all())
    print("%s: %.3f ms per %d frames = %.0f MP/s; last frame equals its own single-frame call: %s" % (
        name, ms, N, N * 33.1776 / ms * 1e3, ok), flush=True)
    del out, alone
