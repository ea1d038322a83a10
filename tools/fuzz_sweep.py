"""A longer run of tests/test_gpu_parity.py's seeded sweep (other seeds, more cases), plus batches of device frames with
ragged grids in both modes: prints the first mismatch, or the number of cases that passed.  Not part of the suite."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import pytest
from __graft_entry__ import load_product
import test_gpu_parity as T
from oracle import binding as oracle
oracle.build()
P = load_product()
gpu = P.Handle(0)
seed = int(os.environ.get("SEED", "777"))
n = int(os.environ.get("N", "400"))
ok = skipped = 0
for case in T._sweep_cases(n, seed):
    try:
        T.test_seeded_sweep_of_geometries(gpu, oracle, *case)
        ok += 1
    except pytest.skip.Exception:
        skipped += 1
    except Exception as e:  # noqa: BLE001
        print("MISMATCH", case, repr(e)[:400]); sys.exit(1)
print("sweep: %d passed, %d skipped" % (ok, skipped), flush=True)
# device batches: aligned widths so that the fast paths and the Oklab regions are exercised
rng = np.random.default_rng(seed)
cases = 0
for _ in range(int(os.environ.get("NB", "60"))):
    bw = int(rng.choice([8, 12, 16, 20, 24, 32, 48, 64, 96, 100, 128])); bh = bw if rng.random() < 0.6 else int(rng.choice([8, 16, 24, 32, 48, 64, 96]))
    if bw * bh > 12000: continue
    w = int(rng.integers(bw, 5 * bw + 64)); h = int(rng.integers(bh, 4 * bh + 50))
    if rng.random() < 0.5: w = (w & ~3) or 4  # half of the batches with 16-byte-multiple rows, half re-pitched
    c = int(rng.choice([3, 4])); mode = int(rng.integers(0, 2)); filt = int(rng.integers(0, 5))
    if mode == 0 and bw * bh > 7168 and bw % 4: continue  # tiles of whole quads go through oklab_kernel (RGB widened)
    if mode == 1 and min(w % bw or bw, h % bh or bh) == 1: continue
    dist = int(rng.choice([0, 1])) if c == 4 else 0
    factor = float(rng.choice([0.5, 4.0, 16.0, 64.0])) if mode == 1 else float(rng.choice([0.1, 0.5, 1.0, 3.0]))
    frames = gpu.synth_frames_device(2, h, w, c, first_frame=int(rng.integers(0, 100)), dist=dist)
    f = frames.cpu().numpy()
    vals, ow, oh, slots = gpu.shrink_frames_device(frames, bw, bh, mode, filt, factor, transparency_hint=bool(rng.integers(0, 2)))
    for k in range(2):
        exp = oracle.shrink_image(f[k], bw, bh, mode, filt, factor, nthreads=8)
        got = (vals[k].cpu().numpy(), ow[k].cpu().numpy().astype(np.uint32), oh[k].cpu().numpy().astype(np.uint32), slots[k].cpu().numpy())
        try:
            T.assert_same_tiles(got, exp, c, f"{w}x{h} b{bw}x{bh} c{c} mode{mode} f{filt} k={factor} dist{dist}")
        except AssertionError as e:
            print("MISMATCH", repr(e)[:400]); sys.exit(1)
    cases += 1
print("device batches: %d passed" % cases)
