"""Time of the device bitstream (QOI tiles + container) on the bench workload."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from __graft_entry__ import load_product
P = load_product()
h = P.Handle(0)
frames = h.synth_frames_device(8, 4320, 7680, 4, 0, int(os.environ.get("DIST", "0")))
for mode, factor in ((1, 16.0), (0, 1.0)):
    vals, ow, oh, slots = h.shrink_frames_device(frames, 32, 32, mode, 4, factor)
    out = h.encode_frames_device(tuple(frames.shape), 32, 32, vals, ow, oh, slots)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): h.encode_frames_device(tuple(frames.shape), 32, 32, vals, ow, oh, slots, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    total = int(out[0][-1].item())
    px = int((ow.long() * oh.long()).sum().item())
    print(f"mode {mode}: encode {ms:.3f} ms per 8 frames, files {total/1e6:.1f} MB from {px*4/1e6:.1f} MB of shrunk pixels ({8*7680*4320/1e6/ms*1e3:.0f} source MP/s)")
    del out, slots
