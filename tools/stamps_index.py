"""Diagnostic: phase shares of pixlzr_index_kernel from the -DPXZ_STAMPS build
   (make -C pixlzr-rust_amd/csrc libpixlzr_hip_stamps.so; PXZ_LIB=.../libpixlzr_hip_stamps.so python3 tools/stamps_index.py)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from __graft_entry__ import load_product
P = load_product()
h = P.Handle(0)
frames = h.synth_frames_device(8, 4320, 7680, 4, 0, 0)
vals, ow, oh, slots = h.shrink_frames_device(frames, 32, 32, 1, 4, 16.0)
enc = h.encode_frames_device(tuple(frames.shape), 32, 32, vals, ow, oh, slots)
torch.cuda.synchronize()
L = P.load_library()
L.pxz_debug_read_status.restype = C.c_int
L.pxz_debug_read_status.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t]
def read():
    buf = (C.c_uint64 * 8)()
    assert L.pxz_debug_read_status(h._h, buf, 8, 64) == 0
    return list(buf)
offs, files = enc[0], enc[1]
dec = h.decode_frames_device(files, offs, tuple(frames.shape), 32, 32)
torch.cuda.synchronize()
before = read()
n = 5
for _ in range(n): h.decode_frames_device(files, offs, tuple(frames.shape), 32, 32, out=dec)
torch.cuda.synchronize()
d = [a - b for a, b in zip(read(), before)]
names = ["file header, line table", "chunk staged", "walk", "checks, publishing, closing sync"]
rows = 8 * 135
tot = sum(d[:4])
for nm, v in zip(names, d[:4]): print(f"{nm:36s} {v / n / rows / 100:9.2f} us/row  {100 * v / tot:5.1f} %")
print("batches per row %.1f, records per row %.1f, total %.1f us per row" % (d[4] / n / rows, d[5] / n / rows, tot / n / rows / 100))
