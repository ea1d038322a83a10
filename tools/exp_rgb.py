"""RGB (3-channel) frames: step times per mode and tile size (8 x 8K frames)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from __graft_entry__ import load_product
P = load_product()
h = P.Handle(0)
c = int(os.environ.get("C", "3"))
HINT = os.environ.get("HINT") == "1"
frames = h.synth_frames_device(8, 4320, 7680, c, 0, int(os.environ.get("DIST", "0")))
for mode, factor, name in ((1, 16.0, "directional"), (0, 1.0, "shrink_by")):
    for bs in (32, 64):
        out = h.shrink_frames_device(frames, bs, bs, mode, 4, factor, transparency_hint=HINT)
        for _ in range(3): h.shrink_frames_device(frames, bs, bs, mode, 4, factor, out=out, transparency_hint=HINT)
        for _ in range(30): h.shrink_frames_device(frames, bs, bs, mode, 4, factor, out=out, transparency_hint=HINT)
        torch.cuda.synchronize()
        import time
        t0 = time.perf_counter()
        for _ in range(30): h.shrink_frames_device(frames, bs, bs, mode, 4, factor, out=out, transparency_hint=HINT)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 30 * 1e3  # wall clock: includes the RGB widening / slot narrowing kernels
        print("C=%d %s %dx%d: %.3f ms per 8 frames" % (c, name, bs, bs, ms), flush=True)
        del out
