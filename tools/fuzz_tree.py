"""Randomized parity of tree::process on the device (per-level grids and rectangle lists, pxz_tree.hip) against the oracle's
recursion: frame sizes, block sizes 5..128 (square and not), minimum blocks, thresholds of either sign, every filter pair,
RGB / RGBA opaque / transparent.  Prints the first mismatch, or the number of cases that passed.  Not part of the suite."""
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
ok = 0
for it in range(int(os.environ.get("N", "150"))):
    bw = int(rng.integers(5, 129)); bh = bw if rng.random() < 0.6 else int(rng.integers(5, 129))
    w = int(rng.integers(max(8, bw // 2), 3 * bw + 40)); h = int(rng.integers(max(8, bh // 2), 3 * bh + 40))
    c = int(rng.choice([3, 4])); dist = int(rng.choice([0, 1])) if c == 4 else 0
    down = int(rng.integers(0, 5)); up = int(rng.integers(0, 5))
    mw = int(rng.choice([1, 4, 6, 11, 16])); mh = int(rng.choice([1, 4, 6, 11, 16]))
    thr = float(rng.choice([0.0, 0.006, 0.012, 0.03, 0.08, 0.3, -0.02, -0.1]))
    frames = gpu.synth_frames_device(2, h, w, c, first_frame=int(rng.integers(0, 200)), dist=dist)
    f = frames.cpu().numpy()
    out = gpu.tree_process_frames_device(frames, bw, bh, thr, mw, mh, down, up).cpu().numpy()
    for n in range(2):
        exp = oracle.tree_process_image(f[n], bw, bh, thr, mw, mh, down, up)
        if (out[n] != exp).any():
            bad = np.argwhere((out[n] != exp).any(axis=2))
            print("MISMATCH", dict(w=w, h=h, bw=bw, bh=bh, c=c, dist=dist, down=down, up=up, mw=mw, mh=mh, thr=thr, frame=n), len(bad), bad[0]); sys.exit(1)
    ok += 1
print("tree: %d passed" % ok)
