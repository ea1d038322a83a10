"""Randomized parity of the decode side against the oracle: expand (every filter, RGB/RGBA, ragged grids, random
tile sizes) and process.  Prints the first mismatch, or the number of cases that passed.  Not part of the suite."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from __graft_entry__ import load_product
from oracle import binding as oracle
oracle.build()
P = load_product()
gpu = P.Handle(0)
rng = np.random.default_rng(int(os.environ.get("SEED", "5")))
n_ok = 0
for it in range(int(os.environ.get("N", "120"))):
    bw = int(rng.choice([4, 8, 12, 16, 20, 24, 32, 48, 64])); bh = bw if rng.random() < 0.6 else int(rng.choice([4, 8, 16, 24, 32, 48, 64]))
    w = int(rng.integers(bw, 5 * bw + 40)); h = int(rng.integers(bh, 4 * bh + 40))
    c = int(rng.choice([3, 4])); filt = int(rng.integers(0, 5)); mode = int(rng.integers(0, 2))
    if mode == 1 and min(w % bw or bw, h % bh or bh, bw, bh) <= 1: continue
    factor = float(rng.choice([0.5, 4.0, 16.0, 64.0])) if mode == 1 else float(rng.choice([0.1, 0.5, 1.0, 3.0]))
    img = oracle.synth_frame(w, h, c, it, 1 if (c == 4 and it % 2) else 0)
    vals, ow, oh, slots = oracle.shrink_image(img, bw, bh, mode, int(rng.integers(0, 5)), factor)
    exp = oracle.expand_image(w, h, bw, bh, c, filt, ow, oh, slots)
    got = gpu.expand_image(w, h, c, bw, bh, filt, ow, oh, slots)
    if (got != exp).any():
        print("MISMATCH expand", (w, h, bw, bh, c, filt, mode, factor), int((got != exp).any(axis=2).sum())); sys.exit(1)
    n_ok += 1
print("expand: %d passed" % n_ok, flush=True)
n_ok = 0
for it in range(int(os.environ.get("NP", "40"))):
    block = int(rng.choice([8, 16, 20, 32, 48, 64])); c = int(rng.choice([3, 4])); dist = int(rng.integers(0, 2)) if c == 4 else 0
    w = int(rng.integers(block, 5 * block + 40)); h = int(rng.integers(block, 4 * block + 40))
    down, up = int(rng.integers(0, 5)), int(rng.integers(0, 5))
    frames = gpu.synth_frames_device(2, h, w, c, first_frame=it, dist=dist)
    out = gpu.process_frames_device(frames, block, block, down, up).cpu().numpy()
    f = frames.cpu().numpy()
    for k in range(2):
        exp = oracle.process_image(f[k], block, block, down, up)
        if (out[k] != exp).any():
            print("MISMATCH process", (w, h, block, c, dist, down, up), int((out[k] != exp).any(axis=2).sum())); sys.exit(1)
    n_ok += 1
print("process: %d passed" % n_ok)
