"""Runs a few launches of ONE flow of the library (for rocprofv3 --pmc / --kernel-trace) and, with OUTJSON set, writes
the flow's algorithmic bytes per step beside the counters (source once + shrunk pixels + 12 B per tile; the
writer: valid slot bytes + value/w/h read, file bytes written).

  python3 tools/pmc_run.py <variant>       variant = dir | by | enc | dec | exp | dirlod | bylod | sqdir | sqby (one 16384^2 frame)     (dir_full = dir, kept for old scripts)
  env: BLOCK (tile side, 32), NF (frames, 8), N (steps, 3), DIST (0 opaque .. 3 noise), FACTOR, FILTER (4), OUTJSON
"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from __graft_entry__ import load_product
P = load_product()
variant = sys.argv[1] if len(sys.argv) > 1 else "dir"
nf = int(os.environ.get("NF", "8"))
n = int(os.environ.get("N", "3"))
dist = int(os.environ.get("DIST", "0"))
bs = int(os.environ.get("BLOCK", "32"))  # tile size (square)
flt = int(os.environ.get("FILTER", "4"))
h = P.Handle(0)
W, H = 7680, 4320
if variant.startswith("sq"):  # BASELINE configs[3]: ONE 16384 x 16384 frame (sqdir / sqby)
    variant, nf, W, H = variant[2:], 1, 16384, 16384
frames = h.synth_frames_device(nf, H, W, 4, 0, dist)
mode, factor = (1, 16.0) if variant.startswith("dir") else (0, 1.0)
if os.environ.get("FACTOR"):
    factor = float(os.environ["FACTOR"])  # with DIST=3 (noise) the factor picks ONE size class for every tile
info = {"variant": variant, "block": bs, "frames": nf, "steps": n, "dist": dist, "factor": factor, "filter": flt}
if variant == "enc":  # shrink once, then the device writer n times
    out = h.shrink_frames_device(frames, bs, bs, 1, flt, 16.0)
    enc = h.encode_frames_device(tuple(frames.shape), bs, bs, *out)
    for _ in range(n - 1): h.encode_frames_device(tuple(frames.shape), bs, bs, *out, out=enc)
    vals, ow, oh, slots = out
    valid = int((ow.long() * oh.long()).sum().item()) * 4
    info.update(step_kernels="qoi_,pack_", setup_kernels="shrink,oklab",
                algo_bytes=valid + 12 * ow.numel() + int(enc[0][-1].item()), file_bytes=int(enc[0][-1].item()))
elif variant == "dec":  # shrink + write once, then the device reader n times
    out = h.shrink_frames_device(frames, bs, bs, 1, flt, 16.0)
    offs, buf = h.encode_frames_device(tuple(frames.shape), bs, bs, *out)
    for _ in range(n): d = h.decode_frames_device(buf, offs, tuple(frames.shape), bs, bs)
    info.update(step_kernels="pixlzr_index,qoi_decode,qoi_bin", setup_kernels="shrink,oklab,qoi_tiles,qoi_splice,pack_",
                algo_bytes=int(offs[-1].item()) + int((out[1].long() * out[2].long()).sum().item()) * 4 + 12 * out[1].numel())
elif variant == "exp":  # shrink once (directional, factor 16, Lanczos3), then expand n times with FILTER
    vals, ow, oh, slots = h.shrink_frames_device(frames, bs, bs, 1, 4, 16.0)
    out = h.expand_frames_device(tuple(frames.shape), bs, bs, flt, ow, oh, slots)
    for _ in range(n - 1): h.expand_frames_device(tuple(frames.shape), bs, bs, flt, ow, oh, slots, out=out)
    info.update(step_kernels="expand", setup_kernels="shrink,oklab",
                algo_bytes=int((ow.long() * oh.long()).sum().item()) * 4 + 8 * ow.numel() + frames.numel())
elif variant.endswith("lod"):
    for _ in range(n): h.lod_frames_device(frames, bs, bs, mode, factor)
    info.update(step_kernels="", algo_bytes=frames.numel() + 8 * (frames.shape[0] * ((H + bs - 1) // bs) * ((W + bs - 1) // bs)))
else:
    out = h.shrink_frames_device(frames, bs, bs, mode, flt, factor)
    for _ in range(n - 1): h.shrink_frames_device(frames, bs, bs, mode, flt, factor, out=out)
    vals, ow, oh, slots = out
    info.update(step_kernels="", algo_bytes=frames.numel() + int((ow.long() * oh.long()).sum().item()) * 4 + 12 * ow.numel())
torch.cuda.synchronize()
if os.environ.get("OUTJSON"):
    with open(os.environ["OUTJSON"], "w") as f:
        json.dump(info, f)
print("done", variant)
