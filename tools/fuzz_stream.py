"""Randomized parity of the device .pixlzr writer and reader against the oracle on adversarial tile content: few colours
(INDEX hits), long and short runs (incl. runs of one and tiles that open with opaque black), gradients (DIFF / LUMA),
alpha flicker (RGBA ops), noise (RGB ops).  Tiles stay unshrunk (factor picks 'clone') or shrink, both sizes of files
are compared byte for byte, then the files are read back on the device.  Not part of the suite."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from __graft_entry__ import load_product
from oracle import binding as oracle
oracle.build()
P = load_product()
gpu = P.Handle(0)
rng = np.random.default_rng(int(os.environ.get("SEED", "11")))
def make(w, h, c, kind):
    img = np.zeros((h, w, c), np.uint8)
    if c == 4: img[..., 3] = 255
    if kind == 0:    # few colours, long runs
        pal = rng.integers(0, 256, size=(int(rng.integers(2, 6)), c), dtype=np.uint8)
        if c == 4 and rng.random() < 0.5: pal[:, 3] = 255
        pal[0, :3] = 0
        runs = rng.integers(1, 70, size=w * h)
        idx = np.repeat(rng.integers(0, len(pal), size=w * h), runs)[: w * h]
        img = pal[idx].reshape(h, w, c)
    elif kind == 1:  # gradients: small steps
        steps = rng.integers(-3, 4, size=(h, w, 3)).cumsum(axis=1).astype(np.int64)
        img[..., :3] = (steps + rng.integers(0, 256)) & 255
    elif kind == 2:  # luma-sized steps
        g = rng.integers(-20, 21, size=(h, w, 1)).cumsum(axis=1)
        img[..., :3] = (g + rng.integers(-6, 7, size=(h, w, 3)).cumsum(axis=1) + 128) & 255
    elif kind == 3:  # alpha flicker / noise
        img[..., :3] = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        if c == 4: img[..., 3] = rng.choice([0, 128, 255], size=(h, w)).astype(np.uint8)
    else:            # runs of one between repeats
        a = rng.integers(0, 256, size=c, dtype=np.uint8); b = rng.integers(0, 256, size=c, dtype=np.uint8)
        pat = np.array([a, a, b, a, a, a, b, b], np.uint8)
        img = pat[(np.arange(w * h) + rng.integers(0, 8)) % 8].reshape(h, w, c)
    return np.ascontiguousarray(img)
n_ok = 0
for it in range(int(os.environ.get("N", "80"))):
    bs = int(rng.choice([8, 16, 32, 64])); c = int(rng.choice([3, 4]))
    w = int(rng.integers(bs, 4 * bs + 30)); h = int(rng.integers(bs, 3 * bs + 30))
    frames = [make(w, h, c, int(rng.integers(0, 5))) for _ in range(2)]
    dev = torch.from_numpy(np.stack(frames)).cuda()
    mode, factor = (1, float(rng.choice([1e6, 16.0]))) if min(w % bs or bs, h % bs or bs) > 1 else (0, float(rng.choice([100.0, 1.0])))
    vals, ow, oh, slots = gpu.shrink_frames_device(dev, bs, bs, mode, 4, factor)
    offs, buf = gpu.encode_frames_device(tuple(dev.shape), bs, bs, vals, ow, oh, slots)
    torch.cuda.synchronize()
    o = offs.cpu().numpy(); data = buf[: o[-1]].cpu().numpy().tobytes()
    for f in range(2):
        v, tw, th, s = oracle.shrink_image(frames[f], bs, bs, mode, 4, factor)
        ref = oracle.encode_container(w, h, bs, bs, c, 0, v, None, tw, th, s)
        if data[o[f]:o[f + 1]] != ref:
            print("MISMATCH writer", (w, h, bs, c, mode, factor, f)); sys.exit(1)
    d = gpu.decode_frames_device(buf, offs, tuple(dev.shape), bs, bs)
    if gpu.decode_status() != 0:
        print("MISMATCH reader status", (w, h, bs, c)); sys.exit(1)
    dv, dw, dh, ds = d
    valid = (ow.long() * oh.long() * c)
    idx = torch.arange(ds.shape[-1], device="cuda")[None, None, :] < valid[:, :, None]
    if not (bool((dw == ow).all()) and bool((dh == oh).all()) and bool((dv.view(torch.int32) == vals.view(torch.int32)).all()) and bool(((ds == slots) | ~idx).all())):
        print("MISMATCH reader", (w, h, bs, c, mode, factor)); sys.exit(1)
    n_ok += 1
print("writer + reader: %d passed" % n_ok)
