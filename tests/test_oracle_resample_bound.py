"""A BOUND, not a pin, on the oracle's restatement of fast_image_resize 4.2.1 (oracle/pixlzr_oracle.c, the
`orc_resize` convolution the GPU path reproduces bit for bit): compared with an independent implementation of the
same lineage -- Pillow's `Image.resize` (Pillow-SIMD is what the crate's u8 convolution was derived from: same
window geometry, normalisation in f64, fixed-point coefficients, u8 intermediate between the two passes) -- on every
(input size, reduced size) pair the encoder can produce for 16/24/32/56/64-px tile axes (reduce_image_section,
operations.rs:140-156: ceil(max(size * 2^-k, 1))).

What the bound says: window placement, kernels and the two-pass u8 intermediate are structurally right; the remaining
risk is coefficient precision / rounding, at most one least-significant bit per output byte.  What it does NOT say:
that the crate's bytes equal the oracle's -- DESIGN.md section 3 keeps "parity unpinned" for them (no fixture produced
by the crate's fir path exists in the reference).

Filter correspondence (data_types/mod.rs:65-107 down-scaling arm -> Pillow): Lanczos3 -> LANCZOS,
CatmullRom -> BICUBIC (a = -0.5), Triangle -> Hamming -> HAMMING.  Gaussian has no Pillow counterpart.
Opaque RGB only: Pillow does not premultiply alpha, fir does (block.rs:295-299, U8x4)."""
import numpy as np
import pytest
from PIL import Image

PIL_FILTER = {4: Image.LANCZOS, 2: Image.BICUBIC, 1: Image.HAMMING}
AXES = (16, 24, 32, 56, 64)


def reduced_sizes(n):
    out = []
    for k in range(1, 8):
        m = int(np.ceil(max(n * 2.0 ** -k, 1)))
        if m < n and m not in out:
            out.append(m)
    return out


def tiles(rng, w, h):
    yield rng.integers(0, 256, (h, w, 3), dtype=np.uint8)                                   # noise: every weight matters
    yield (np.linspace(0, 255, w)[None, :, None] + rng.integers(-20, 20, (h, w, 3))).clip(0, 255).astype(np.uint8)
    yield np.where(rng.random((h, w, 1)) < 0.5, 0, 255).astype(np.uint8).repeat(3, 2)        # black/white: overshoot, clipping


@pytest.mark.parametrize("filt", [4, 2, 1])
def test_oracle_convolution_within_one_lsb_of_pillow(oracle, filt):
    rng = np.random.default_rng(100 + filt)
    worst, equal, count, worst_pair = 0, 0, 0, 1.0
    for w in AXES:
        for h in AXES:
            for nw in reduced_sizes(w) + [w]:
                for nh in reduced_sizes(h) + [h]:
                    if nw == w and nh == h:
                        continue  # clone: no resample (block.rs:279-281)
                    eq = n = 0
                    for t in tiles(rng, w, h):
                        mine = oracle.resize(t, nw, nh, filt)
                        pil = np.asarray(Image.fromarray(t).resize((nw, nh), PIL_FILTER[filt]))
                        d = np.abs(mine.astype(np.int32) - pil.astype(np.int32))
                        worst = max(worst, int(d.max()))
                        eq += int((d == 0).sum())
                        n += d.size
                    equal += eq
                    count += n
                    if n >= 300:  # a 1x1 output of three tiles is nine bytes: no ratio to speak of
                        worst_pair = min(worst_pair, eq / n)
    assert worst <= 1, f"filter {filt}: max |oracle - Pillow| = {worst}"
    assert equal / count >= 0.99, f"filter {filt}: only {equal / count:.4f} of the bytes equal"
    assert worst_pair >= 0.95, f"filter {filt}: a size pair with only {worst_pair:.3f} equal bytes"


def test_constant_tiles_stay_constant(oracle):
    """block.rs:400-435 (the reference's only resample test), extended over the sizes above and every filter."""
    for filt in (1, 2, 3, 4):
        for value in (0, 1, 127, 254, 255):
            for w, h in ((100, 100), (32, 32), (56, 17), (64, 24)):
                t = np.full((h, w, 3), value, np.uint8)
                for nw in reduced_sizes(w)[:3]:
                    out = oracle.resize(t, nw, max(1, h // 2), filt)
                    assert (out == value).all(), (filt, value, w, h, nw)
