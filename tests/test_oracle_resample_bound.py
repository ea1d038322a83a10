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


# ---------------------------------------------------------------------------------------------------------------
# Round 3: an independent f64 model for ALL FOUR convolution filters (Gaussian included: data_types/mod.rs:99-101), and the
# alpha-premultiplied U8x4 path (block.rs:295-299, fir's default ResizeOptions) against Pillow's premultiplied mode.
# ---------------------------------------------------------------------------------------------------------------

def _kernel(filt):
    """The four down-scaling kernels (data_types/mod.rs:65-107) as plain numpy f64 functions + their support."""
    if filt == 1:  # Triangle -> Convolution(Hamming)
        def f(x):
            x = np.abs(x)
            with np.errstate(divide="ignore", invalid="ignore"):
                v = np.sin(np.pi * x) / (np.pi * x) * (0.54 + 0.46 * np.cos(np.pi * x))
            return np.where(x == 0, 1.0, np.where(x < 1, v, 0.0))
        return f, 1.0
    if filt == 2:  # CatmullRom (a = -0.5)
        def f(x):
            x = np.abs(x)
            a = -0.5
            return np.where(x < 1, ((a + 2) * x - (a + 3)) * x * x + 1, np.where(x < 2, (((x - 5) * x + 8) * x - 4) * a, 0.0))
        return f, 2.0
    if filt == 3:  # Gaussian, sigma 0.5, support 3
        def f(x):
            return np.where(np.abs(x) < 3, np.exp(-(x * x) / 0.5) / np.sqrt(2 * np.pi * 0.25), 0.0)
        return f, 3.0

    def f(x):  # Lanczos3
        with np.errstate(divide="ignore", invalid="ignore"):
            s = np.where(x == 0, 1.0, np.sin(np.pi * x) / (np.pi * x)) * np.where(x == 0, 1.0, np.sin(np.pi * x / 3) / (np.pi * x / 3))
        return np.where((x >= -3) & (x < 3), s, 0.0)
    return f, 3.0


def _f64_axis_matrix(n_in, n_out, filt):
    """[n_out, n_in] f64 weights, normalised per output sample: the convolution as real arithmetic (no fixed point)."""
    f, support = _kernel(filt)
    scale = n_in / n_out
    fs = max(scale, 1.0)
    m = np.zeros((n_out, n_in))
    for o in range(n_out):
        center = (o + 0.5) * scale
        lo, hi = max(int(np.floor(center - support * fs)), 0), min(int(np.ceil(center + support * fs)), n_in)
        x = np.arange(lo, hi)
        w = f((x + 0.5 - center) / fs)
        m[o, lo:hi] = w / w.sum()
    return m


def _f64_two_pass(t, nw, nh, filt):
    """horizontal pass -> round half up, clip to u8 -> vertical pass -> round, clip: every weight and sum in f64."""
    h, w, _ = t.shape
    cur = t.astype(np.float64)
    if nw != w:
        cur = np.einsum("ox,yxc->yoc", _f64_axis_matrix(w, nw, filt), cur)
        cur = np.clip(np.floor(cur + 0.5), 0, 255)
    if nh != h:
        cur = np.einsum("oy,yxc->oxc", _f64_axis_matrix(h, nh, filt), cur)
        cur = np.clip(np.floor(cur + 0.5), 0, 255)
    return cur.astype(np.uint8)


@pytest.mark.parametrize("filt", [1, 2, 3, 4])
def test_oracle_convolution_within_one_lsb_of_an_f64_model(oracle, filt):
    """Every filter, Gaussian included, against real-number convolution with the same window geometry: the oracle's i16
    coefficients, i32 sums, rounding constant and shift cost at most one LSB per byte (and that only where the real sum
    sits near a rounding boundary of either pass)."""
    rng = np.random.default_rng(300 + filt)
    worst, equal, count = 0, 0, 0
    for w in AXES:
        for h in AXES:
            for nw in reduced_sizes(w) + [w]:
                for nh in reduced_sizes(h) + [h]:
                    if nw == w and nh == h:
                        continue
                    for t in tiles(rng, w, h):
                        d = np.abs(oracle.resize(t, nw, nh, filt).astype(np.int32) - _f64_two_pass(t, nw, nh, filt).astype(np.int32))
                        worst = max(worst, int(d.max()))
                        equal += int((d == 0).sum())
                        count += d.size
    assert worst <= 1, f"filter {filt}: max |oracle - f64 model| = {worst}"
    assert equal / count >= 0.97, f"filter {filt}: only {equal / count:.4f} of the bytes equal"


def _rgba_tiles(rng, w, h):
    rgb = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    yield np.dstack([rgb, rng.integers(1, 256, (h, w, 1), dtype=np.uint8)])                 # alpha anywhere in 1..255
    yield np.dstack([rgb, rng.integers(200, 256, (h, w, 1), dtype=np.uint8)])               # nearly opaque
    ramp = (np.linspace(0, 255, w)[None, :, None] + rng.integers(-20, 20, (h, w, 3))).clip(0, 255).astype(np.uint8)
    yield np.dstack([ramp, np.where(rng.random((h, w, 1)) < 0.3, rng.integers(1, 40, (h, w, 1)), 255).astype(np.uint8)])  # holes


# What can differ between two correct implementations of this pipeline: one LSB of the convolution in each premultiplied
# colour AND in alpha, each of which the division by alpha turns into up to 255/alpha of the result, plus the division's
# own rounding (fir: reciprocal table, rounded; Pillow: 255*c/a, truncated): |difference| <= 2 * 255/alpha + 1.5.
# MEASURED over the cases below, in premultiplied units (|difference| * alpha / 255; recorded so that a change of either
# side shows): Lanczos3 2.13, CatmullRom 1.96, Hamming 1.90; alpha itself 1 LSB; colour bytes at alpha >= 128: 3.
MEASURED_PREMUL = {4: 2.13, 2: 1.96, 1: 1.90}


@pytest.mark.parametrize("filt", [4, 2, 1])
def test_oracle_premultiplied_rgba_against_pillow(oracle, filt):
    """The U8x4 path: premultiply (mul_div_255) -> convolve the four channels -> un-premultiply.  Pillow's "RGBa" mode is
    the same pipeline from the same lineage (its premultiply IS mul_div_255; its un-premultiply truncates where fir
    rounds).  Compared where the comparison means something: alpha to one LSB, colours in the premultiplied domain
    (|c_oracle * a - c_pillow * a| / 255 <= 1 LSB + the two divisions' rounding), and as plain bytes for the pixels whose
    alpha is large enough that a premultiplied LSB stays an LSB."""
    rng = np.random.default_rng(500 + filt)
    worst_alpha, worst_premul, worst_opaqueish, n_px, worst_bound = 0, 0.0, 0, 0, -1e9
    for w, h in ((16, 16), (32, 32), (64, 64), (32, 24), (56, 17)):
        for nw in reduced_sizes(w) + [w]:
            for nh in reduced_sizes(h) + [h]:
                if nw == w and nh == h:
                    continue
                for t in _rgba_tiles(rng, w, h):
                    mine = oracle.resize(t, nw, nh, filt).astype(np.int32)
                    pil = np.asarray(Image.fromarray(t, "RGBA").convert("RGBa").resize((nw, nh), PIL_FILTER[filt]).convert("RGBA")).astype(np.int32)
                    worst_alpha = max(worst_alpha, int(np.abs(mine[..., 3] - pil[..., 3]).max()))
                    a = np.minimum(mine[..., 3], pil[..., 3])[..., None].astype(np.float64)
                    dc = np.abs(mine[..., :3] - pil[..., :3])
                    live = a[..., 0] > 0
                    if live.any():
                        worst_premul = max(worst_premul, float((dc * a / 255.0)[live].max()))
                        worst_bound = max(worst_bound, float((dc - (2.0 * 255.0 / np.maximum(a, 1.0) + 1.5))[live].max()))
                    big = a[..., 0] >= 128
                    if big.any():
                        worst_opaqueish = max(worst_opaqueish, int(dc[big].max()))
                    n_px += int(live.sum())
    assert n_px > 20000
    assert worst_alpha <= 1, f"filter {filt}: alpha differs by {worst_alpha}"
    assert worst_premul <= MEASURED_PREMUL[filt] + 0.02, f"filter {filt}: premultiplied-domain difference {worst_premul:.3f}"
    assert worst_bound <= 0.0, f"filter {filt}: a colour byte {worst_bound:.2f} beyond 2 * 255/alpha + 1.5"
    assert worst_opaqueish <= 3, f"filter {filt}: colour bytes at alpha >= 128 differ by {worst_opaqueish}"


@pytest.mark.parametrize("filt", [1, 2, 3, 4])
def test_oracle_premultiplied_rgba_against_an_f64_model(oracle, filt):
    """The same path against real arithmetic for all four filters (Gaussian has no Pillow counterpart): exact
    premultiplication c*a/255 rounded to u8, the f64 two-pass convolution of test ..._f64_model, exact division
    255*c/a rounded.  Same units as above."""
    rng = np.random.default_rng(700 + filt)
    worst_alpha, worst_premul = 0, 0.0
    for w, h in ((16, 16), (32, 32), (64, 64), (32, 24)):
        for nw in reduced_sizes(w)[:4]:
            for nh in reduced_sizes(h)[:4]:
                for t in _rgba_tiles(rng, w, h):
                    mine = oracle.resize(t, nw, nh, filt).astype(np.int32)
                    a_in = t[..., 3:4].astype(np.float64)
                    pre = np.dstack([np.floor(t[..., :3] * a_in / 255.0 + 0.5), a_in]).astype(np.uint8)
                    conv = _f64_two_pass(pre, nw, nh, filt).astype(np.float64)
                    a_out = conv[..., 3:4]
                    with np.errstate(divide="ignore", invalid="ignore"):
                        col = np.where(a_out > 0, np.clip(np.floor(conv[..., :3] * 255.0 / a_out + 0.5), 0, 255), 0)
                    worst_alpha = max(worst_alpha, int(np.abs(mine[..., 3] - a_out[..., 0]).max()))
                    a = np.minimum(mine[..., 3:4], a_out)
                    live = a[..., 0] > 0
                    if live.any():
                        worst_premul = max(worst_premul, float((np.abs(mine[..., :3] - col) * a / 255.0)[live].max()))
    assert worst_alpha <= 1
    assert worst_premul <= 2.0, f"filter {filt}: premultiplied-domain difference {worst_premul:.3f}"
