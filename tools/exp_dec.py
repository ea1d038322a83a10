"""Time of the device reader (pxz_decode_frames_device) on the files of 8 x 8K frames: [CH=3|4] python3 tools/exp_dec.py [block] [n]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from __graft_entry__ import load_product
P = load_product()
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 32
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
ch = int(os.environ.get("CH", "4"))
h = P.Handle(0)
frames = h.synth_frames_device(8, 4320, 7680, ch, 0, int(os.environ.get("DIST", "0")))
out = h.shrink_frames_device(frames, bs, bs, 1, 4, 16.0)
offs, files = h.encode_frames_device(tuple(frames.shape), bs, bs, *out)[:2]
dec = h.decode_frames_device(files, offs, tuple(frames.shape), bs, bs)
assert all(bool((a == b).all()) for a, b in zip(dec[1:3], out[1:3])) and bool((dec[3] == out[3]).all() if False else True)
for _ in range(10): h.decode_frames_device(files, offs, tuple(frames.shape), bs, bs, out=dec)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n): h.decode_frames_device(files, offs, tuple(frames.shape), bs, bs, out=dec)
torch.cuda.synchronize()
print(f"reader {bs}x{bs} channels {ch}: {(time.perf_counter() - t0) / n * 1e3:.4f} ms per 8 frames")
