"""Diagnostic: phase shares of shrink32_kernel from the -DPXZ_STAMPS build (PXZ_LIB=...stamps.so)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from __graft_entry__ import load_product
P = load_product()
h = P.Handle(0)
dist = int(os.environ.get("DIST", "0"))
frames = h.synth_frames_device(8, 4320, 7680, 4, 0, dist)
LOD = os.environ.get("LOD") == "1"
run = (lambda o=None: h.lod_frames_device(frames, 32, 32, 1, 16.0)) if LOD else (lambda o=None: h.shrink_frames_device(frames, 32, 32, 1, 4, 16.0, out=o))
out = run()
torch.cuda.synchronize()
# the stamps live behind the worklist inside the handle's private buffer: read via a second launch's side effect
# -> expose through hipMemcpy using torch: find the buffer by re-running and copying from the known layout
L = P.load_library()
L.pxz_debug_read_work.restype = C.c_int
L.pxz_debug_read_work.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t]
n_tiles = out[1].numel() if not LOD else out[0].numel()
def read():
    buf = (C.c_uint64 * 8)()
    off = ((2 * n_tiles + 132 + 2) & ~1) * 4
    assert L.pxz_debug_read_work(h._h, buf, off, 64) == 0
    return list(buf)
before = read()
for _ in range(5): run(None if LOD else out)
torch.cuda.synchronize()
after = read()
d = [a - b for a, b in zip(after, before)]
tot = sum(d[:4])
names = ["wait prefetch + stage", "issue prefetch", "detector+decision+meta", "clone/resample"]
for n, v in zip(names, d): print(f"{n:28s} {v/5/n_tiles:10.1f} cycles/tile  {100*v/tot:5.1f}%")
print("total per tile (wave cycles):", tot / 5 / n_tiles)

# per-wave run times of the last launch (100 MHz wall clock) and per-block start times (slot 15)
import numpy as np
buf = (C.c_uint64 * (256 * 17))()
off = ((2 * n_tiles + 132 + 2) & ~1) * 4 + 64
assert L.pxz_debug_read_work(h._h, buf, off, 256 * 17 * 8) == 0
raw = np.array(list(buf), dtype=np.uint64).reshape(256, 17)  # 16 waves per block + the block's start time
start = raw[:, 16].astype(np.float64) / 100
dur = (raw[:, :16] & np.uint64(0xffffffffffff)).astype(np.float64) / 100
cnt = (raw[:, :16] >> np.uint64(48)).astype(np.int64)
print("tiles per wave: min %d median %d max %d; per block sum min %d max %d" % (cnt.min(), np.median(cnt), cnt.max(), cnt.sum(axis=1).min(), cnt.sum(axis=1).max()))
print("corr(run time, tiles) = %.3f" % np.corrcoef(dur.ravel(), cnt.ravel())[0, 1])
print("block start spread (us): %.1f" % (start.max() - start.min()))
print("per-wave run time percentiles (us):", ["%.1f" % np.percentile(dur, q) for q in (0, 1, 5, 25, 50, 75, 95, 99, 100)])
end = start[:, None] + dur - start.min()
print("per-wave finish percentiles (us):", ["%.1f" % np.percentile(end, q) for q in (0, 1, 5, 25, 50, 75, 95, 99, 100)])
spread = end.max(axis=1) - end.min(axis=1)
print("within-block finish spread (us): min %.1f median %.1f max %.1f" % (spread.min(), np.median(spread), spread.max()))
eb = end.max(axis=1)
print("block finish (us): min %.1f median %.1f max %.1f" % (eb.min(), np.median(eb), eb.max()))
print("mean block finish per XCD (block % 8):", ["%.1f" % eb[x::8].mean() for x in range(8)])
print("block 0 waves: run time us", ["%.1f" % x for x in dur[0]], "tiles", cnt[0].tolist())
print("block 100 waves: run time us", ["%.1f" % x for x in dur[100]], "tiles", cnt[100].tolist())
