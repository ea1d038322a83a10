"""Decode side: expand of 8 x 8K frames (32x32 tiles, shrunk at factor 16; CH=3|4 channels), per filter: step time (wall clock)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from __graft_entry__ import load_product
P = load_product()
h = P.Handle(0)
bs = int(os.environ.get('BS', '32'))
ch = int(os.environ.get('CH', '4'))
frames = h.synth_frames_device(8, 4320, 7680, ch, 0, 0)
vals, ow, oh, slots = h.shrink_frames_device(frames, bs, bs, 1, 4, 16.0)
for filt in (0, 2, 4):
    out = h.expand_frames_device(tuple(frames.shape), bs, bs, filt, ow, oh, slots)
    for _ in range(60): h.expand_frames_device(tuple(frames.shape), bs, bs, filt, ow, oh, slots, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100): h.expand_frames_device(tuple(frames.shape), bs, bs, filt, ow, oh, slots, out=out)
    torch.cuda.synchronize()
    print("expand %dx%d channels %d filter %d: %.4f ms" % (bs, bs, ch, filt, (time.perf_counter() - t0) * 10), flush=True)
