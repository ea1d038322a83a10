"""Per size class cost of the 32x32 directional step: noise frames (DIST 3), where the factor picks one class for
every tile.  Prints the dominant class and the step time (events) per factor."""
import os, sys, collections, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from __graft_entry__ import load_product
P = load_product()
h = P.Handle(0)
frames = h.synth_frames_device(8, 4320, 7680, 4, 0, 3)
for factor in [float(x) for x in (sys.argv[1:] or "0.25 0.5 1 2 4 8 16 32 64 128 256".split())]:
    out = h.shrink_frames_device(frames, 32, 32, 1, 4, factor)
    w = out[1].cpu().numpy().ravel().astype(int); hh = out[2].cpu().numpy().ravel().astype(int)
    c = collections.Counter(zip(w.tolist(), hh.tolist())).most_common(2)
    for _ in range(100): h.shrink_frames_device(frames, 32, 32, 1, 4, factor, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100): h.shrink_frames_device(frames, 32, 32, 1, 4, factor, out=out)
    torch.cuda.synchronize()
    print("factor %-6g %-40s %.4f ms" % (factor, str(c), (time.perf_counter() - t0) * 10), flush=True)
