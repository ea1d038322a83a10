"""Time of the device writer (pxz_encode_frames_device) after one shrink: [CH=3|4] python3 tools/exp_enc.py [block] [n]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from __graft_entry__ import load_product
P = load_product()
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 32
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
mode = int(os.environ.get("MODE", "1"))
h = P.Handle(0)
ch = int(os.environ.get("CH", "4"))
frames = h.synth_frames_device(8, 4320, 7680, ch, 0, int(os.environ.get("DIST", "0")))
out = h.shrink_frames_device(frames, bs, bs, mode, 4, 16.0 if mode == 1 else 1.0)
enc = h.encode_frames_device(tuple(frames.shape), bs, bs, *out)
for _ in range(20): h.encode_frames_device(tuple(frames.shape), bs, bs, *out, out=enc)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n): h.encode_frames_device(tuple(frames.shape), bs, bs, *out, out=enc)
torch.cuda.synchronize()
print(f"writer {bs}x{bs} mode {mode} channels {ch}: {(time.perf_counter() - t0) / n * 1e3:.4f} ms per 8 frames, {int(enc[0][-1])} file bytes")
