"""Diagnostic: phase shares of expand_kernel from the -DPXZ_STAMPS build
   (make -C pixlzr-rust_amd/csrc libpixlzr_hip_stamps.so; PXZ_LIB=.../libpixlzr_hip_stamps.so python3 tools/stamps_expand.py [filter])."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from __graft_entry__ import load_product
P = load_product()
h = P.Handle(0)
filt = int(sys.argv[1]) if len(sys.argv) > 1 else 4
frames = h.synth_frames_device(8, 4320, 7680, 4, 0, 0)
vals, ow, oh, slots = h.shrink_frames_device(frames, 32, 32, 1, 4, 16.0)
out = h.expand_frames_device(tuple(frames.shape), 32, 32, filt, ow, oh, slots)
torch.cuda.synchronize()
L = P.load_library()
L.pxz_debug_read_status.restype = C.c_int
L.pxz_debug_read_status.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t]
def read():
    buf = (C.c_uint64 * 8)()
    assert L.pxz_debug_read_status(h._h, buf, 8, 64) == 0
    return list(buf)
before = read()
n = 5
for _ in range(n): h.expand_frames_device(tuple(frames.shape), 32, 32, filt, ow, oh, slots, out=out)
torch.cuda.synchronize()
d = [a - b for a, b in zip(read(), before)]
tot = sum(d)
names = ["ticket", "wait for the prefetched sizes / pixels", "other paths (clone, nearest, general)", "mfma: planes staged", "mfma: horizontal",
         "mfma: one-row replicate", "mfma: vertical + stores", "closing sync"]
tiles = ow.numel()
for nm, v in zip(names, d): print(f"{nm:42s} {v / n / tiles:9.1f} clocks/tile  {100 * v / tot:5.1f} %")
print("total per tile (s_memtime clocks, 100 MHz):", tot / n / tiles)
