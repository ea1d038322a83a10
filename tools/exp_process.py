"""Legacy process() on the device (Oklab detector with the identity closure -> shrink -> resize back): 8 x 8K frames."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, time
from __graft_entry__ import load_product
P = load_product()
h = P.Handle(0)
for c in (4, 3):
    frames = h.synth_frames_device(8, 4320, 7680, c, 0, 0)
    for block in (32, 64):
        out = h.process_frames_device(frames, block, block, 4, 0)
        for _ in range(10): h.process_frames_device(frames, block, block, 4, 0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): h.process_frames_device(frames, block, block, 4, 0)
        torch.cuda.synchronize()
        print("process c%d block %d (Lanczos3 down, Nearest up): %.3f ms per 8 frames" % (c, block, (time.perf_counter() - t0) / 20 * 1e3), flush=True)
