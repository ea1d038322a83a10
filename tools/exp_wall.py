"""Wall clock per step with and without the timing events (8 x 8K frames, 32x32, directional)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from __graft_entry__ import load_product
P = load_product()
h = P.Handle(0)
frames = h.synth_frames_device(8, 4320, 7680, 4, 0, 0)
out = h.shrink_frames_device(frames, 32, 32, 1, 4, 16.0)
step = lambda: h.shrink_frames_device(frames, 32, 32, 1, 4, 16.0, out=out)
for _ in range(200): step()
torch.cuda.synchronize()
for timing in (0, 1, 8, 0, 1, 8):
    h.enable_timing(timing > 0, every=max(timing, 1))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(500): step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 500 * 1e3
    extra = ""
    if timing:
        extra = " first kernel %.4f, all kernels %.4f" % (h.last_first_kernel_ms(), h.last_kernel_ms())
    print("events %s: wall %.4f ms/step%s" % (timing, ms, extra), flush=True)
h.enable_timing(False)
