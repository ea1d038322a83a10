"""GPU parity: the HIP path (through the C ABI) against the oracle on the same inputs.
Bit-exact everywhere: block values are compared as u32 bit patterns, pixels as bytes."""
import os

import numpy as np
import pytest
from PIL import Image

pytestmark = pytest.mark.gpu

FILTERS = {"nearest": 0, "triangle": 1, "catmullrom": 2, "gaussian": 3, "lanczos3": 4}


@pytest.fixture(scope="module")
def gpu(product):
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    h = product.Handle(0)
    yield h
    h.close()


def assert_same_tiles(got, exp, channels, what=""):
    gv, gw, gh, gs = got
    ev, ew, eh, es = exp
    assert (gw == ew).all() and (gh == eh).all(), f"{what}: reduced dims differ on {int(((gw != ew) | (gh != eh)).sum())} tiles"
    gb, eb = gv.view(np.uint32), ev.view(np.uint32)
    assert (gb == eb).all(), f"{what}: block values differ on {int((gb != eb).sum())} tiles, e.g. {gv[gb != eb][:3]} vs {ev[gb != eb][:3]}"
    if es is not None:
        valid = (ew.astype(np.int64) * eh * channels)
        idx = np.arange(es.shape[1])[None, :] < valid[:, None]
        diff = (gs != es) & idx
        assert not diff.any(), f"{what}: {int(diff.sum())} payload bytes differ in {int(diff.any(axis=1).sum())} tiles"


def histogram(ow, oh):
    out = {}
    for w, h in zip(ow.tolist(), oh.tolist()):
        out[(w, h)] = out.get((w, h), 0) + 1
    return out


def test_synth_generator_matches_oracle(gpu, oracle):
    for channels in (4, 3):
        for dist in range(4):
            d = gpu.synth_frames_device(2, 70, 100, channels, first_frame=5, dist=dist).cpu().numpy()
            for f in range(2):
                assert (d[f] == oracle.synth_frame(100, 70, channels, 5 + f, dist)).all(), (channels, dist, f)


@pytest.mark.parametrize("name,block", [("Big-Ruscher.png", 32), ("base.png", 64), ("base.png", 32), ("image.png", 8)])
def test_directional_detector_on_reference_images(gpu, oracle, golden_dir, name, block):
    """get_block_variance_directionally over every tile of the reference's own images (RGB and RGBA,
    edge tiles 24-high / 56x17 / 2-wide) — includes the SURVEY §8(c) KAT tiles."""
    import torch
    img = np.ascontiguousarray(np.asarray(Image.open(os.path.join(golden_dir, name))))
    H, W, C = img.shape
    l0, l1 = gpu.lod_frames_device(torch.from_numpy(img)[None].cuda(), block, block, 1)
    l0, l1 = l0.cpu().numpy()[0], l1.cpu().numpy()[0]
    cols, rows = oracle.grid(W, H, block, block)
    for t in range(cols * rows):
        x, y, w, h = oracle.tile_rect(W, H, block, block, t)
        hz, vr, _, _ = oracle.lod_directional(img[y:y + h, x:x + w])
        assert np.array([hz, vr], np.float32).tobytes() == np.array([l0[t], l1[t]], np.float32).tobytes(), (t, hz, l0[t])


def test_oklab_detector_on_big_ruscher(gpu, oracle, golden_dir):
    """get_block_variance (+shrink_by closures, factor 0.125) on the image behind Big-Ruscher.pix: values
    bit-equal to the oracle and reduced sizes equal to the 2040 decisions stored in the reference's file."""
    img = np.ascontiguousarray(np.asarray(Image.open(os.path.join(golden_dir, "Big-Ruscher.png"))))
    got = gpu.shrink_image(img, 32, 32, 0, 4, 0.125)
    exp = oracle.shrink_image(img, 32, 32, 0, 4, 0.125)
    assert_same_tiles(got, exp, 3, "Big-Ruscher shrink_by")
    d = oracle.decode_container(open(os.path.join(golden_dir, "Big-Ruscher.pix"), "rb").read())
    assert (got[1] == d["tw"]).all() and (got[2] == d["th"]).all()


CASES_1080P = [
    # (dist, mode, filter, factor)
    ("opaque", 1, "lanczos3", 16.0), ("opaque", 1, "lanczos3", 1.0), ("opaque", 0, "lanczos3", 1.0),
    ("opaque", 0, "lanczos3", 0.125), ("alpha", 1, "lanczos3", 16.0), ("alpha", 0, "lanczos3", 1.0),
    ("opaque", 1, "nearest", 16.0), ("alpha", 1, "catmullrom", 16.0), ("alpha", 0, "triangle", 0.5),
    ("opaque", 1, "gaussian", 8.0), ("alpha", 1, "nearest", 4.0), ("opaque", 1, "lanczos3", -0.5),
]


@pytest.mark.parametrize("dist,mode,filt,factor", CASES_1080P)
def test_shrink_1080p_rgba_32(gpu, oracle, dist, mode, filt, factor):
    """BASELINE config: 1920x1080 RGBA8 synthetic, 32x32 tiles (last tile row 24 px high)."""
    img = oracle.synth_frame(1920, 1080, 4, 0, {"opaque": 0, "alpha": 1}[dist])
    got = gpu.shrink_image(img, 32, 32, mode, FILTERS[filt], factor)
    exp = oracle.shrink_image(img, 32, 32, mode, FILTERS[filt], factor, nthreads=8)
    assert_same_tiles(got, exp, 4, f"{dist} mode{mode} {filt} k={factor}")
    if factor in (16.0, 1.0) and filt == "lanczos3" and dist == "opaque":
        assert len(histogram(got[1], got[2])) >= 4  # the workload spans the levels


@pytest.mark.parametrize("w,h,bw,bh,c", [
    (64, 64, 64, 64, 4), (65, 33, 32, 32, 4), (100, 37, 48, 20, 4), (1920, 1170, 8, 8, 3), (257, 130, 16, 16, 4),
    (160, 160, 80, 80, 4), (130, 70, 128, 64, 3), (34, 34, 32, 32, 4), (35, 35, 32, 32, 3), (31, 9, 64, 64, 4),
    (3, 3, 32, 32, 4), (2, 2, 2, 2, 4), (640, 360, 64, 64, 4), (96, 96, 24, 24, 3)])
@pytest.mark.parametrize("mode", [0, 1])
def test_shrink_ragged_grids_and_block_sizes(gpu, oracle, w, h, bw, bh, c, mode):
    """Edge tiles (clamped, split.rs:18-19), non-power-of-two tiles, multi-wave tiles (64x64 .. 128x64),
    2-px tiles (directional 0/0 -> 1x1), RGB and RGBA."""
    rng = np.random.default_rng(w * 131 + h)
    img = oracle.synth_frame(w, h, c, 3, 1 if c == 4 else 0)
    if mode == 1 and (w % bw == 1 or h % bh == 1):
        # a 1-px edge tile: the reference underflows `width - 2` and panics (operations.rs:220-221)
        with pytest.raises(Exception) as e:
            gpu.shrink_image(img, bw, bh, mode, 4, 1.0)
        assert getattr(e.value, "code", None) == -4
        with pytest.raises(RuntimeError):
            oracle.shrink_image(img, bw, bh, mode, 4, 1.0)
        return
    img[: h // 2] = (img[: h // 2].astype(np.int32) + rng.integers(-40, 40, size=img[: h // 2].shape)).clip(0, 255).astype(np.uint8)
    for filt, factor in ((4, 8.0 if mode == 1 else 0.7), (2, 2.0 if mode == 1 else 0.2), (0, 4.0 if mode == 1 else 0.4)):
        got = gpu.shrink_image(img, bw, bh, mode, filt, factor)
        exp = oracle.shrink_image(img, bw, bh, mode, filt, factor)
        assert_same_tiles(got, exp, c, f"{w}x{h} b{bw}x{bh} c{c} mode{mode} f{filt}")


def test_pitch_and_unaligned_rows(gpu, oracle):
    """Pitch larger than a row and rows that are not 16-byte aligned (scalar staging path)."""
    base = oracle.synth_frame(203, 77, 4, 1, 1)
    padded = np.zeros((77, 203 * 4 + 12), np.uint8)
    padded[:, : 203 * 4] = base.reshape(77, -1)
    view = padded[:, : 203 * 4].reshape(77, 203, 4)
    assert view.strides[0] == 203 * 4 + 12
    for mode, factor in ((1, 8.0), (0, 0.5)):
        got = gpu.shrink_image(view, 32, 32, mode, 4, factor)
        exp = oracle.shrink_image(base, 32, 32, mode, 4, factor)
        assert_same_tiles(got, exp, 4, "pitched")


def test_error_behaviour(gpu, product, oracle):
    img = oracle.synth_frame(33, 40, 4, 0, 0)  # 33 = 32 + 1: a 1-px-wide edge tile
    with pytest.raises(product.PxzError) as e:
        gpu.shrink_image(img, 32, 32, 1, 4, 1.0)   # reference: usize underflow panic (operations.rs:221)
    assert e.value.code == -4
    gpu.shrink_image(img, 32, 32, 0, 4, 1.0)        # shrink_by has no such restriction
    with pytest.raises(product.PxzError) as e:
        gpu.shrink_image(img, 32, 32, 1, 9, 1.0)
    assert e.value.code == -1
    with pytest.raises(product.PxzError) as e:
        gpu.shrink_image(img, 0, 32, 1, 4, 1.0)
    assert e.value.code == -1
    with pytest.raises(product.PxzError) as e:
        gpu.shrink_image(img, 32, 32, 1, 4, float("nan"))
    assert e.value.code == -1
    with pytest.raises(product.PxzError) as e:
        # (tiles beyond LDS residency run since round 4 -- test_tiles_beyond_lds_residency -- up to 2^20 - 1 pixels)
        gpu.shrink_image(oracle.synth_frame(1100, 1030, 4, 0, 0), 1024, 1024, 1, 4, 1.0)
    assert e.value.code == -5


@pytest.mark.parametrize("bw,bh", [(192, 192), (256, 256), (300, 200)])
@pytest.mark.parametrize("c,dist", [(4, 0), (4, 1), (3, 0)])
def test_tiles_beyond_lds_residency(gpu, oracle, bw, bh, c, dist):
    """Block sizes whose tile image does not fit the 160 KB of LDS (any -b the reference's CLI accepts, src/bin/main.rs:19-24):
    the generic kernel and the expand kernel keep the image in HBM.  Both callers, Lanczos3 / Nearest / CatmullRom, a frame with
    ragged edge tiles; shrink against the oracle (values as bits, sizes, pixels), then the way back (expand) against it too."""
    W, H = 700, 500
    img = oracle.synth_frame(W, H, c, 7, dist)
    for mode, factor, filt in ((1, 16.0, 4), (1, 64.0, 0), (0, 1.0, 4), (0, 0.25, 2), (1, 4.0, 2)):
        got = gpu.shrink_image(img, bw, bh, mode, filt, factor)
        exp = oracle.shrink_image(img, bw, bh, mode, filt, factor, nthreads=8)
        assert_same_tiles(got, exp, c, f"{bw}x{bh} c{c} dist{dist} mode{mode} filter {filt} k={factor}")
        back = gpu.expand_image(W, H, c, bw, bh, filt, exp[1], exp[2], exp[3])
        ref = oracle.expand_image(W, H, bw, bh, c, filt, exp[1], exp[2], exp[3])
        bad = (back != ref).any(axis=2)
        assert not bad.any(), f"expand {bw}x{bh} c{c} filter {filt}: {int(bad.sum())} pixels differ"


@pytest.mark.parametrize("mode,factor", [(1, 16.0), (0, 1.0)])
def test_full_file_bit_exact(gpu, product, oracle, mode, factor):
    """image -> shrink on the GPU -> product's container writer  ==  oracle shrink -> oracle writer."""
    for c in (4, 3):
        img = oracle.synth_frame(640, 360, c, 2, 1 if c == 4 else 0)
        gv, gw, gh, gs = gpu.shrink_image(img, 32, 32, mode, 4, factor)
        ev, ew, eh, es = oracle.shrink_image(img, 32, 32, mode, 4, factor)
        mine = product.encode_container(640, 360, 32, 32, c, 0, gv, None, gw, gh, gs)
        ref = oracle.encode_container(640, 360, 32, 32, c, 0, ev, None, ew, eh, es)
        assert mine == ref
        d = oracle.decode_container(mine)
        assert (d["tw"] == gw).all() and (d["values"].view(np.uint32) == gv.view(np.uint32)).all()


def test_8k_batch_against_oracle_and_properties(gpu, oracle):
    """BASELINE headline size: 7680x4320 RGBA8, 32x32 tiles, device-resident batch.
    Frame 0 is compared with the oracle in full; the rest through size-independent properties."""
    import torch
    frames = gpu.synth_frames_device(3, 4320, 7680, 4, first_frame=0, dist=0)
    for mode, factor in ((1, 16.0), (0, 1.0)):
        vals, ow, oh, slots = gpu.shrink_frames_device(frames, 32, 32, mode, 4, factor)
        torch.cuda.synchronize()
        img0 = frames[0].cpu().numpy()
        assert (img0 == oracle.synth_frame(7680, 4320, 4, 0, 0)).all()
        exp = oracle.shrink_image(img0, 32, 32, mode, 4, factor, nthreads=16)
        got = (vals[0].cpu().numpy(), ow[0].cpu().numpy().astype(np.uint32), oh[0].cpu().numpy().astype(np.uint32),
               slots[0].cpu().numpy())
        assert_same_tiles(got, exp, 4, f"8K mode{mode}")
        # determinism / idempotence: same launch again gives the same bytes
        vals2, ow2, oh2, slots2 = gpu.shrink_frames_device(frames, 32, 32, mode, 4, factor)
        assert torch.equal(vals.view(torch.int32), vals2.view(torch.int32)) and torch.equal(ow, ow2) and torch.equal(oh, oh2)
        # batch independence: frame 2 alone == frame 2 inside the batch
        v1, w1, h1, s1 = gpu.shrink_frames_device(frames[2:3], 32, 32, mode, 4, factor)
        assert torch.equal(v1[0].view(torch.int32), vals[2].view(torch.int32)) and torch.equal(w1[0], ow[2])
        valid = (w1[0].long() * h1[0].long() * 4)[:, None] > torch.arange(4096, device=s1.device)[None, :]
        assert torch.equal(s1[0][valid], slots[2][valid])
    # "noise": nothing shrinks and the slots are the source tiles (clone, block.rs:279-281)
    noise = gpu.synth_frames_device(1, 4320, 7680, 4, first_frame=9, dist=3)
    vals, ow, oh, slots = gpu.shrink_frames_device(noise, 32, 32, 1, 4, 16.0)
    assert int((ow != 32).sum()) == 0 and int((oh != 32).sum()) == 0
    tiles = noise[0].view(135, 32, 240, 32, 4).permute(0, 2, 1, 3, 4).reshape(135 * 240, 4096)
    assert torch.equal(tiles, slots[0])
    # "flat": every tile collapses to 1x1
    flat = gpu.synth_frames_device(1, 4320, 7680, 4, first_frame=9, dist=2)
    vals, ow, oh, slots = gpu.shrink_frames_device(flat, 32, 32, 1, 4, 1.0)
    assert int((ow != 1).sum()) == 0 and int((oh != 1).sum()) == 0


def test_pack_tiles_matches_numpy(gpu, oracle):
    """pxz_pack_tiles_device: exclusive scan of tile byte sizes + compaction of the slots (RGBA and RGB,
    more tiles than one scan chunk of 4096)."""
    import torch
    for c, (w, h) in ((4, (3072, 1568)), (3, (200, 96))):
        img = oracle.synth_frame(w, h, c, 4, 1 if c == 4 else 0)
        frames = torch.from_numpy(img)[None].cuda()
        vals, ow, oh, slots = gpu.shrink_frames_device(frames, 32, 32, 1, 4, 12.0)
        offsets, packed = gpu.pack_tiles_device(ow, oh, slots, c)
        torch.cuda.synchronize()
        sizes = (ow.long() * oh.long() * c).reshape(-1).cpu().numpy()
        exp_off = np.concatenate([[0], np.cumsum(sizes)])
        assert (offsets.cpu().numpy() == exp_off).all()
        assert ow.numel() > 4096 or c == 3
        sl = slots.reshape(-1, slots.shape[-1]).cpu().numpy()
        exp = np.concatenate([sl[t, : sizes[t]] for t in range(len(sizes))])
        assert (packed[: exp_off[-1]].cpu().numpy() == exp).all()


@pytest.mark.parametrize("axis", [0, 1])
def test_single_pass_resamples(gpu, oracle, axis):
    """Tiles that keep one axis at full size (width level from hz, height level from vr): strong variation
    along one axis plus a weak one along the other gives horizontal-only / vertical-only resamples at
    several levels (32x16, 32x8, ... or 16x32, 8x32, ...)."""
    rng = np.random.default_rng(11 + axis)
    h, w = 256, 512
    strong = rng.integers(0, 256, size=(h if axis == 0 else w, 3)).astype(np.int32)
    weak = rng.integers(0, 12, size=(w if axis == 0 else h, 3)).astype(np.int32)
    img = np.empty((h, w, 4), np.uint8)
    img[..., 3] = 255
    if axis == 0:
        img[..., :3] = np.clip(strong[:, None, :] + weak[None, :, :] - 6, 0, 255)
    else:
        img[..., :3] = np.clip(strong[None, :, :] + weak[:, None, :] - 6, 0, 255)
    seen = set()
    for factor in (256.0, 128.0, 64.0, 32.0, 16.0, 8.0):
        for filt in (4, 3, 2, 1, 0):
            got = gpu.shrink_image(img, 32, 32, 1, filt, factor)
            exp = oracle.shrink_image(img, 32, 32, 1, filt, factor)
            assert_same_tiles(got, exp, 4, f"axis{axis} k={factor} f{filt}")
            seen |= set(histogram(got[1], got[2]))
    one_pass = {k for k in seen if (k[0] == 32) != (k[1] == 32)}
    assert len(one_pass) >= 3, seen


@pytest.mark.parametrize("filt", [1, 2, 3, 4])
def test_block64_fast_kernel_every_class(gpu, oracle, filt):
    """The reference CLI's default 64x64 tiles through shrink64_kernel (four waves per tile, matrix-core
    resample): every reduced size from 64x64 (clone) down to 1x1, the one-pass classes (worklist), a ragged
    edge, and a batch of frames -- against the oracle."""
    import torch
    seen = set()
    img = oracle.synth_frame(1088, 600, 4, 5, 0)  # 17 x 10 tiles, last row 24 px high
    for factor in (64.0, 16.0, 4.0, 1.0, 0.25):
        got = gpu.shrink_image(img, 64, 64, 1, filt, factor)
        exp = oracle.shrink_image(img, 64, 64, 1, filt, factor, nthreads=8)
        assert_same_tiles(got, exp, 4, f"64x64 filter {filt} k={factor}")
        seen |= set(histogram(got[1], got[2]))
    assert {(64, 64), (32, 32), (16, 16), (8, 8), (4, 4), (2, 2), (1, 1)} <= seen, seen
    frames = gpu.synth_frames_device(3, 256, 512, 4, first_frame=2, dist=0)
    vals, ow, oh, slots = gpu.shrink_frames_device(frames, 64, 64, 1, filt, 16.0)
    f = frames.cpu().numpy()
    for n in range(3):
        exp = oracle.shrink_image(f[n], 64, 64, 1, filt, 16.0)
        assert_same_tiles((vals[n].cpu().numpy(), ow[n].cpu().numpy().astype(np.uint32), oh[n].cpu().numpy().astype(np.uint32),
                           slots[n].cpu().numpy()), exp, 4, f"batch frame {n}")


@pytest.mark.parametrize("filt", [0, 1, 2, 3, 4])
def test_block16_group_kernel_every_class(gpu, oracle, filt):
    """16x16 tiles through shrink16_kernel (2x2 groups of tiles per 32x32 LDS image, per-tile sums by a
    segmented reduction): every reduced size from 16x16 (clone) to 1x1, mixed classes inside a group, the
    one-pass classes and ragged / odd group counts (worklist), transparency, a batch of frames."""
    seen = set()
    img = oracle.synth_frame(1000, 328, 4, 9, 0)  # 63 x 21 tiles: odd counts, last column 8 px wide, last row 8 px high
    for factor in (64.0, 16.0, 4.0, 1.0, 0.25):
        got = gpu.shrink_image(img, 16, 16, 1, filt, factor)
        exp = oracle.shrink_image(img, 16, 16, 1, filt, factor, nthreads=8)
        assert_same_tiles(got, exp, 4, f"16x16 filter {filt} k={factor}")
        seen |= set(histogram(got[1], got[2]))
    assert {(16, 16), (8, 8), (4, 4), (2, 2), (1, 1)} <= seen, seen
    frames = gpu.synth_frames_device(3, 160, 256, 4, first_frame=1, dist=1)  # transparency: whole groups are deferred
    vals, ow, oh, slots = gpu.shrink_frames_device(frames, 16, 16, 1, filt, 16.0)
    f = frames.cpu().numpy()
    for n in range(3):
        exp = oracle.shrink_image(f[n], 16, 16, 1, filt, 16.0)
        assert_same_tiles((vals[n].cpu().numpy(), ow[n].cpu().numpy().astype(np.uint32), oh[n].cpu().numpy().astype(np.uint32),
                           slots[n].cpu().numpy()), exp, 4, f"batch frame {n}")


@pytest.mark.parametrize("block", [16, 64])
@pytest.mark.parametrize("dist", [0, 1])
def test_shrink_by_blocks_16_and_64(gpu, oracle, block, dist):
    """shrink_by (Oklab detector) with the reference CLI's default 64x64 tiles and with 16x16 tiles: the
    block-cooperative detector (oklab_kernel<16|64>; the 64x64 form parks a tile's colours in HBM between its
    two passes) + shrink64_kernel<0> / the generic kernel, opaque and transparent RGBA, ragged last row,
    a batch of frames -- bit-exact block values included."""
    frames = gpu.synth_frames_device(2, 280, 448, 4, first_frame=6, dist=dist)
    f = frames.cpu().numpy()
    seen = set()
    for factor, filt in ((1.0, 4), (0.25, 2), (4.0, 4)):
        vals, ow, oh, slots = gpu.shrink_frames_device(frames, block, block, 0, filt, factor)
        for n in range(2):
            exp = oracle.shrink_image(f[n], block, block, 0, filt, factor, nthreads=8)
            got = (vals[n].cpu().numpy(), ow[n].cpu().numpy().astype(np.uint32), oh[n].cpu().numpy().astype(np.uint32),
                   slots[n].cpu().numpy())
            assert_same_tiles(got, exp, 4, f"shrink_by {block}x{block} dist {dist} k={factor} frame {n}")
            seen |= set(histogram(got[1], got[2]))
    assert len(seen) >= 3, seen


@pytest.mark.parametrize("channels", [4, 3])
@pytest.mark.parametrize("block", [16, 32, 64])
def test_shrink_by_copies_ahead_of_the_value(product, oracle, block, channels):
    """shrink_by on the square tile sizes, RGBA and RGB (round 4): the detector copies every tile into its slot while it converts it, and
    the shrink kernel leaves the tiles that are stored at full size alone (32x32: skipped in the kernel; 64x64: finished by
    clone_split64_kernel, the rest listed; 16x16: a group of four such tiles is not read, single ones are not cloned again).
    A sequence on ONE handle and ONE set of output buffers: noise (every tile stored whole), flat frames (every tile small: the
    copies are overwritten), the benchmark mix with transparency in some tiles, a ragged last tile row whose height is a whole
    number of the detector's bands and one that is not, a single frame -- values, sizes and every valid payload byte equal the
    oracle after each launch (pixlzr.rs:155-185, block.rs:279-281)."""
    import torch
    h = product.Handle(0)
    hh, ww = 8 * block + block // 2, 12 * block  # a ragged last row of half a tile: whole bands at every size
    seq = [(3, 1.0, 2, hh, ww), (0, 0.001, 2, hh, ww), (1, 1.0, 2, hh, ww), (0, 1.0, 2, hh, ww), (2, 4.0, 2, hh, ww),
           (0, 1.0, 2, 8 * block + 3, 12 * block + 5), (0, 1.0, 1, 6 * block, 7 * block)]
    out, shape, seen = None, None, set()
    for k, (dist, factor, nf, fh, fw) in enumerate(seq):
        frames = h.synth_frames_device(nf, fh, fw, channels, first_frame=11 + k, dist=dist)
        if shape != (nf, fh, fw):
            out, shape = None, (nf, fh, fw)
        out = h.shrink_frames_device(frames, block, block, 0, 4, factor, out=out)
        torch.cuda.synchronize()
        f = frames.cpu().numpy()
        for n in range(nf):
            exp = oracle.shrink_image(f[n], block, block, 0, 4, factor, nthreads=8)
            got = (out[0][n].cpu().numpy(), out[1][n].cpu().numpy().astype(np.uint32), out[2][n].cpu().numpy().astype(np.uint32), out[3][n].cpu().numpy())
            assert_same_tiles(got, exp, channels, f"launch {k} (dist {dist}, k={factor}, {fw}x{fh}) frame {n}")
            seen |= set(histogram(got[1], got[2]))
    assert (block, block) in seen and len(seen) >= 4, seen
    # the detector only copies the tiles it GUESSES to be stored whole (from their first band); the guess must never show: tiles whose
    # first rows are flat and the rest noise (stored whole, not copied), the other way round (copied, not stored whole), plain noise
    rng = np.random.default_rng(block + channels)
    tiles_y, tiles_x = 6, 9
    img = np.empty((2, tiles_y * block, tiles_x * block, channels), np.uint8)
    img[...] = 90
    if channels == 4:
        img[..., 3] = 255
    top = block // 2 if block == 16 else block // 4  # at least the first 128-px band of a tile
    for n in range(2):
        for ty in range(tiles_y):
            for tx in range(tiles_x):
                kind = (tx + 2 * ty + n) % 3
                y0, x0 = ty * block, tx * block
                noise = rng.integers(0, 256, size=(block, block, 3), dtype=np.uint8)
                if kind == 0:
                    img[n, y0:y0 + block, x0:x0 + block, :3] = noise
                elif kind == 1:
                    img[n, y0 + top:y0 + block, x0:x0 + block, :3] = noise[top:]
                else:
                    img[n, y0:y0 + 1, x0:x0 + block, :3] = noise[:1]  # (one busy row: copied, hardly stored whole)
    frames = torch.from_numpy(img).cuda()
    kinds = set()
    for factor in (1.0, 0.2):
        got_all = h.shrink_frames_device(frames, block, block, 0, 4, factor)
        torch.cuda.synchronize()
        for n in range(2):
            exp = oracle.shrink_image(img[n], block, block, 0, 4, factor, nthreads=8)
            got = (got_all[0][n].cpu().numpy(), got_all[1][n].cpu().numpy().astype(np.uint32), got_all[2][n].cpu().numpy().astype(np.uint32), got_all[3][n].cpu().numpy())
            assert_same_tiles(got, exp, channels, f"guessed copies, k={factor} frame {n}")
            kinds |= set(histogram(got[1], got[2]))
    assert (block, block) in kinds and len(kinds) >= 2, kinds


def test_block64_one_pass_classes_and_transparency(gpu, oracle):
    """64 x n / n x 64 outputs (one matrix-core pass inside shrink64_kernel) and tiles with transparency
    (handed to the generic kernel through the worklist): results must not depend on who processed a tile."""
    rng = np.random.default_rng(5)
    h, w = 256, 512
    strong = rng.integers(0, 256, size=(h, 3)).astype(np.int32)
    weak = rng.integers(0, 12, size=(w, 3)).astype(np.int32)
    img = np.empty((h, w, 4), np.uint8)
    img[..., 3] = 255
    img[..., :3] = np.clip(strong[:, None, :] + weak[None, :, :] - 6, 0, 255)
    img[100:140, 300:340, 3] = 77  # transparency in a few tiles
    seen = set()
    for factor in (256.0, 64.0, 16.0):
        got = gpu.shrink_image(img, 64, 64, 1, 4, factor)
        exp = oracle.shrink_image(img, 64, 64, 1, 4, factor)
        assert_same_tiles(got, exp, 4, f"k={factor}")
        seen |= set(histogram(got[1], got[2]))
    assert any((k[0] == 64) != (k[1] == 64) for k in seen), seen


@pytest.mark.parametrize("axis", [0, 1])
def test_wide_flat_and_tall_thin_classes(gpu, oracle, axis):
    """Reduced sizes that are large on one axis and tiny on the other (16x2, 16x1, 8x1, 1x16, ...): the fast
    kernel sends 16 x (2|1) to the worklist and runs n x (2|1), (2|1) x n through the dot2 form; all of them
    must equal the oracle."""
    rng = np.random.default_rng(23 + axis)
    h, w = 256, 512
    strong = rng.integers(96, 160, size=(h if axis == 0 else w, 3)).astype(np.int32)
    weak = rng.integers(0, 3, size=(w if axis == 0 else h, 3)).astype(np.int32)
    img = np.empty((h, w, 4), np.uint8)
    img[..., 3] = 255
    if axis == 0:
        img[..., :3] = np.clip(strong[:, None, :] + weak[None, :, :], 0, 255)
    else:
        img[..., :3] = np.clip(strong[None, :, :] + weak[:, None, :], 0, 255)
    seen = set()
    for factor in (64.0, 32.0, 16.0, 8.0, 4.0, 2.0, 1.0):
        for filt in (4, 2):
            got = gpu.shrink_image(img, 32, 32, 1, filt, factor)
            exp = oracle.shrink_image(img, 32, 32, 1, filt, factor)
            assert_same_tiles(got, exp, 4, f"axis{axis} k={factor} f{filt}")
            seen |= set(histogram(got[1], got[2]))
    mixed = {k for k in seen if max(k) in (4, 8, 16) and min(k) <= 2}
    assert len(mixed) >= 3 and any(max(k) == 16 for k in mixed), seen


@pytest.mark.parametrize("filt", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("mode,factors", [(1, (64.0, 16.0, 4.0, 1.0)), (0, (2.0, 0.5))])
def test_every_filter_on_opaque_fast_path(gpu, oracle, filt, mode, factors):
    """All five FilterTypes (narrow Hamming windows .. wide Lanczos3) through the 32x32 fast kernels,
    every level from 32x32 down to 1x1, opaque RGBA."""
    img = oracle.synth_frame(1024, 512, 4, 7, 0)
    seen = set()
    for factor in factors:
        got = gpu.shrink_image(img, 32, 32, mode, filt, factor)
        exp = oracle.shrink_image(img, 32, 32, mode, filt, factor, nthreads=8)
        assert_same_tiles(got, exp, 4, f"filter {filt} mode {mode} k={factor}")
        seen |= set(histogram(got[1], got[2]))
    assert len(seen) >= (5 if mode == 1 else 3), seen


def test_device_bitstream_reproduces_base_pixlzr(gpu, oracle, golden_dir):
    """GPU QOI + container on the un-shrunk tiles of benches/base.png (64x64, edge tiles 56x64/64x17/56x17):
    must be the reference's own file byte for byte (bench-00.rs:55,66)."""
    import torch
    img = np.ascontiguousarray(np.asarray(Image.open(os.path.join(golden_dir, "base.png"))))
    gold = open(os.path.join(golden_dir, "base.pixlzr"), "rb").read()
    H, W, C = img.shape
    cols, rows = oracle.grid(W, H, 64, 64)
    n = cols * rows
    tw = np.zeros(n, np.int32)
    th = np.zeros(n, np.int32)
    slots = np.zeros((n, 64 * 64 * C), np.uint8)
    for t in range(n):
        x, y, w, h = oracle.tile_rect(W, H, 64, 64, t)
        tw[t], th[t] = w, h
        slots[t, : w * h * C] = img[y:y + h, x:x + w].reshape(-1)
    offs, buf = gpu.encode_frames_device((1, H, W, C), 64, 64, torch.zeros(n, device="cuda"),
                                         torch.from_numpy(tw).cuda(), torch.from_numpy(th).cuda(), torch.from_numpy(slots).cuda())
    torch.cuda.synchronize()
    offs = offs.cpu().numpy()
    assert offs[0] == 0 and offs[1] == len(gold)
    assert buf[: offs[1]].cpu().numpy().tobytes() == gold


@pytest.mark.parametrize("c,mode,factor", [(4, 1, 16.0), (4, 0, 1.0), (3, 1, 8.0), (4, 1, 1.0)])
def test_device_bitstream_equals_oracle_writer(gpu, oracle, c, mode, factor):
    """shrink on the GPU -> GPU QOI + container == oracle shrink -> oracle writer, for a batch of frames
    (all tile sizes 32x32 .. 1x1, opaque and transparent RGBA, RGB)."""
    import torch
    frames = []
    for f in range(3):
        frames.append(oracle.synth_frame(672, 416, c, 20 + f, (f % 2) if c == 4 else 0))
    dev = torch.from_numpy(np.stack(frames)).cuda()
    vals, ow, oh, slots = gpu.shrink_frames_device(dev, 32, 32, mode, 4, factor)
    offs, buf = gpu.encode_frames_device(tuple(dev.shape), 32, 32, vals, ow, oh, slots)
    torch.cuda.synchronize()
    offs = offs.cpu().numpy()
    data = buf[: offs[-1]].cpu().numpy().tobytes()
    for f in range(3):
        v, w, h, s = oracle.shrink_image(frames[f], 32, 32, mode, 4, factor)
        ref = oracle.encode_container(672, 416, 32, 32, c, 0, v, None, w, h, s)
        assert data[offs[f]:offs[f + 1]] == ref, f"frame {f}"


@pytest.mark.parametrize("c,bw,bh", [(4, 128, 128), (3, 128, 128), (4, 96, 40), (4, 20, 12), (3, 64, 64)])
def test_device_writer_on_tiles_of_every_class(gpu, oracle, c, bw, bh):
    """The writer's units (one wave = 64 segments of one class; rows of 512 bytes in the scratch; the splice that walks them):
    tiles of every size from 1x1 up to the slot -- 128x128 slots reach the classes whose segments are longer than 128 pixels --
    with content that makes short and long pieces (constant, few colours, noise, alpha flicker), stored sizes at random.
    GPU QOI + container == the oracle's writer (encoding/mod.rs:40-89, 168-200 with the qoi crate's encoder), byte for byte."""
    import torch
    rng = np.random.default_rng(1000 * c + bw + bh)
    cols, rows = 7, 5
    n = cols * rows
    tw = np.zeros((2, n), np.uint32); th = np.zeros((2, n), np.uint32)
    slots = np.zeros((2, n, bw * bh * c), np.uint8)
    vals = rng.random((2, n)).astype(np.float32)
    for f in range(2):
        for t in range(n):
            w = int(rng.choice([1, 2, 3, bw // 4, bw // 2, bw - 1, bw, int(rng.integers(1, bw + 1))]))
            h = int(rng.choice([1, 2, bh // 4 or 1, bh // 2, bh, int(rng.integers(1, bh + 1))]))
            w, h = max(w, 1), max(h, 1)
            kind = int(rng.integers(0, 4))
            if kind == 0:
                px = np.tile(rng.integers(0, 256, c, dtype=np.uint8), (w * h, 1))
            elif kind == 1:
                pal = rng.integers(0, 256, (4, c), dtype=np.uint8)
                px = pal[np.repeat(rng.integers(0, 4, w * h), rng.integers(1, 40, w * h))[: w * h]]
            elif kind == 2:
                px = rng.integers(0, 256, (w * h, c), dtype=np.uint8)
            else:
                px = (np.cumsum(rng.integers(-2, 3, (w * h, c)), axis=0) + 128).astype(np.uint8)
                if c == 4: px[:, 3] = rng.choice(np.array([0, 255], np.uint8), w * h)
            tw[f, t], th[f, t] = w, h
            slots[f, t, : w * h * c] = px.reshape(-1)
    W, H = cols * bw, rows * bh
    offs, buf = gpu.encode_frames_device((2, H, W, c), bw, bh, torch.from_numpy(vals).cuda(), torch.from_numpy(tw.astype(np.int32)).cuda(),
                                         torch.from_numpy(th.astype(np.int32)).cuda(), torch.from_numpy(slots).cuda())
    torch.cuda.synchronize()
    offs = offs.cpu().numpy()
    data = buf[: offs[-1]].cpu().numpy().tobytes()
    for f in range(2):
        ref = oracle.encode_container(W, H, bw, bh, c, 0, vals[f], None, tw[f], th[f], slots[f])
        got = data[offs[f]:offs[f + 1]]
        assert len(got) == len(ref), f"frame {f}: {len(got)} bytes, oracle {len(ref)}"
        if got != ref:
            i = next(k for k in range(len(ref)) if got[k] != ref[k])
            raise AssertionError(f"frame {f}: first difference at byte {i} of {len(ref)}")


# ---- decode side: Pixlzr::expand + to_image --------------------------------------------------

def test_expand_reproduces_big_ruscher_pix_png(gpu, oracle, golden_dir):
    """The reference's own decode fixture: Big-Ruscher.pix expanded with Nearest is Big-Ruscher.pix.png."""
    d = oracle.decode_container(open(os.path.join(golden_dir, "Big-Ruscher.pix"), "rb").read())
    ref = np.asarray(Image.open(os.path.join(golden_dir, "Big-Ruscher.pix.png")))[..., :3]
    slots = np.ascontiguousarray(d["slots"][:, : d["bw"] * d["bh"] * 3])
    img = gpu.expand_image(d["width"], d["height"], 3, d["bw"], d["bh"], 0, d["tw"], d["th"], slots)
    assert (img == ref).all()


@pytest.mark.parametrize("c", [4, 3])
@pytest.mark.parametrize("filt", [0, 1, 2, 3, 4])
def test_expand_matches_oracle(gpu, oracle, c, filt):
    """Shrink (so that every reduced size from 1x1 to the full tile occurs, ragged edge tiles included), then
    expand with every FilterType: clone, nearest, one-pass and two-pass convolutions, RGB and RGBA with
    transparency (premultiplied convolution)."""
    img = oracle.synth_frame(500, 300, c, 3, 1 if c == 4 else 0)
    for bw, bh, factor in ((32, 32, 16.0), (32, 32, 2.0), (48, 20, 8.0), (64, 64, 16.0), (16, 16, 16.0), (16, 16, 2.0)):
        vals, ow, oh, slots = oracle.shrink_image(img, bw, bh, 1, 4, factor)
        exp = oracle.expand_image(500, 300, bw, bh, c, filt, ow, oh, slots)
        got = gpu.expand_image(500, 300, c, bw, bh, filt, ow, oh, slots)
        bad = (got != exp).any(axis=2)
        assert not bad.any(), f"c{c} f{filt} {bw}x{bh} k{factor}: {int(bad.sum())} pixels differ"


@pytest.mark.parametrize("block", [32, 16, 64])
@pytest.mark.parametrize("filt", [0, 1, 2, 3, 4])
def test_expand_of_power_of_two_tiles(gpu, oracle, filt, block):
    """RGBA tiles of 32x32, 16x16 (expand16_kernel: 2x2 groups of tiles, mixed sizes inside a group, an odd column count and so a
    partial group per row) and 64x64 stored at every combination of 1, 2, 4, ... up to the full size (also the ones no shrink
    produces from a level pair, and a few odd sizes between them, which take the general forms): the matrix-core convolutions
    and the shift-indexed Nearest against the oracle's PixlzrBlock::resize (block.rs:273-334).  Random pixels; a third of the
    tiles opaque, a third with random alpha, a third with alpha in {0, 255} (premultiplied convolution, fir's U8x4 default)."""
    rng = np.random.default_rng(40 + filt + block)
    sizes = [s for s in (1, 2, 4, 8, 16, 32, 64) if s <= block]
    odd = [(3, 8), (8, 5), (block - 1, block // 2), (block // 2, block // 2 + 1), (12, 12)]
    pairs = [(a, b) for a in sizes for b in sizes] * 3 + odd
    pairs = [pairs[i] for i in rng.permutation(len(pairs))]  # every group of a 16x16 grid holds a mix
    cols, rows = 9, (len(pairs) + 8) // 9
    n = cols * rows
    tw = np.ones(n, np.uint32); th = np.ones(n, np.uint32)
    slots = np.zeros((n, block * block * 4), np.uint8)
    for t in range(n):
        w, h = pairs[t % len(pairs)]
        tw[t], th[t] = w, h
        px = rng.integers(0, 256, (h * w, 4), dtype=np.uint8)
        kind = t % 3
        if kind == 0: px[:, 3] = 255
        elif kind == 2: px[:, 3] = rng.choice(np.array([0, 255], np.uint8), h * w)
        if t % 7 == 0: px[:, :3] = rng.choice(np.array([0, 255], np.uint8), (h * w, 3))  # extremes: the clamps
        slots[t, : h * w * 4] = px.ravel()
    exp = oracle.expand_image(cols * block, rows * block, block, block, 4, filt, tw, th, slots)
    got = gpu.expand_image(cols * block, rows * block, 4, block, block, filt, tw, th, slots)
    bad = (got != exp).any(axis=2)
    if bad.any():
        t_bad = sorted({int((y // block) * cols + x // block) for y, x in zip(*np.nonzero(bad))})
        raise AssertionError(f"filter {filt} block {block}: {int(bad.sum())} pixels differ, tiles {[(t, int(tw[t]), int(th[t])) for t in t_bad[:8]]}")


@pytest.mark.parametrize("filt", [0, 1, 2, 3, 4])
def test_expand_of_power_of_two_rgb_tiles_64(gpu, oracle, filt):
    """RGB tiles of 64x64 (what image::open yields for most photographs, at the reference CLI's default block) stored at every
    combination of 1 .. 64 and a few odd sizes, into an RGB frame whose width is not a multiple of 4 pixels beyond the last full
    tile (rows at any byte alignment): expand64_kernel<3> -- three planes, no premultiplication, twelve-byte stores -- and the
    general form for the ragged column, against the oracle's PixlzrBlock::resize with U8x3 (block.rs:273-334)."""
    rng = np.random.default_rng(90 + filt)
    sizes = [1, 2, 4, 8, 16, 32, 64]
    pairs = [(a, b) for a in sizes for b in sizes] * 2 + [(3, 8), (63, 32), (32, 33), (12, 12)]
    pairs = [pairs[i] for i in rng.permutation(len(pairs))]
    cols, rows = 8, (len(pairs) + 6) // 7
    edge = 7  # the last column of tiles is 7 px wide
    W, H = (cols - 1) * 64 + edge, rows * 64
    n = cols * rows
    tw = np.ones(n, np.uint32); th = np.ones(n, np.uint32)
    slots = np.zeros((n, 64 * 64 * 3), np.uint8)
    k = 0
    for t in range(n):
        fw = 64 if t % cols < cols - 1 else edge
        w, h = pairs[k % len(pairs)]
        if fw == 64: k += 1
        w = min(w, fw)
        tw[t], th[t] = w, h
        px = rng.integers(0, 256, (h * w, 3), dtype=np.uint8)
        if t % 7 == 0: px[:] = rng.choice(np.array([0, 255], np.uint8), (h * w, 3))  # extremes: the clamps
        slots[t, : h * w * 3] = px.ravel()
    exp = oracle.expand_image(W, H, 64, 64, 3, filt, tw, th, slots)
    got = gpu.expand_image(W, H, 3, 64, 64, filt, tw, th, slots)
    bad = (got != exp).any(axis=2)
    if bad.any():
        t_bad = sorted({int((y // 64) * cols + x // 64) for y, x in zip(*np.nonzero(bad))})
        raise AssertionError(f"filter {filt}: {int(bad.sum())} pixels differ, tiles {[(t, int(tw[t]), int(th[t])) for t in t_bad[:8]]}")


@pytest.mark.parametrize("filt", [0, 4])
def test_expand_into_rows_at_any_alignment(gpu, oracle, filt):
    """A frame 289 px wide: its rows start 4 (mod 16) bytes apart, so the 16-byte moves of the 32x32 fast paths (full-size
    tiles copied from their slots, shift-indexed Nearest, the one-row replicate) land on addresses that are only 4-byte
    aligned; the ragged last column and row take the general forms beside them.  Equal to the oracle's expand."""
    rng = np.random.default_rng(77 + filt)
    width, height = 289, 3 * 32 + 7
    cols, rows = 10, 4
    n = cols * rows
    tw = np.ones(n, np.uint32); th = np.ones(n, np.uint32)
    slots = np.zeros((n, 32 * 32 * 4), np.uint8)
    sizes = [32, 32, 1, 2, 4, 8, 16]
    for t in range(n):
        fw = 32 if t % cols < cols - 1 else 1
        fh = 32 if t // cols < rows - 1 else 7
        w = min(sizes[int(rng.integers(0, len(sizes)))], fw); h = min(sizes[int(rng.integers(0, len(sizes)))], fh)
        if t % 5 == 0: w, h = fw, fh  # stored at full size
        if t % 11 == 3: w, h = 2, 1
        w, h = min(w, fw), min(h, fh)
        tw[t], th[t] = w, h
        px = rng.integers(0, 256, (h * w, 4), dtype=np.uint8)
        if t % 3 == 0: px[:, 3] = 255
        slots[t, : h * w * 4] = px.ravel()
    exp = oracle.expand_image(width, height, 32, 32, 4, filt, tw, th, slots)
    got = gpu.expand_image(width, height, 4, 32, 32, filt, tw, th, slots)
    bad = (got != exp).any(axis=2)
    assert not bad.any(), f"filter {filt}: {int(bad.sum())} pixels differ"


def test_expand_frames_device_round_trip_properties(gpu, oracle):
    """Device-resident batch: shrink -> expand.  Tiles kept at full size come back unchanged; the batch equals
    the per-frame oracle; an invalid stored size is flagged and leaves the tile untouched."""
    import torch
    frames = gpu.synth_frames_device(2, 256, 384, 4, first_frame=9, dist=0)
    vals, ow, oh, slots = gpu.shrink_frames_device(frames, 32, 32, 1, 4, 16.0)
    out = gpu.expand_frames_device(tuple(frames.shape), 32, 32, 4, ow, oh, slots)
    torch.cuda.synchronize()
    assert gpu.decode_status() == 0
    f, o = frames.cpu().numpy(), out.cpu().numpy()
    w, h = ow.cpu().numpy(), oh.cpu().numpy()
    cols = 384 // 32
    kept = 0
    for n in range(2):
        exp = oracle.expand_image(384, 256, 32, 32, 4, 4, w[n], h[n], slots[n].cpu().numpy())
        assert (o[n] == exp).all()
        for t in np.nonzero((w[n] == 32) & (h[n] == 32))[0]:
            ty, tx = divmod(int(t), cols)
            assert (o[n, ty * 32:ty * 32 + 32, tx * 32:tx * 32 + 32] == f[n, ty * 32:ty * 32 + 32, tx * 32:tx * 32 + 32]).all()
            kept += 1
    assert kept > 0
    bad_w = ow.clone()
    bad_w[0, 5] = 33
    out2 = torch.zeros_like(out)
    gpu.expand_frames_device(tuple(frames.shape), 32, 32, 4, bad_w, oh, slots, out=out2)
    assert gpu.decode_status() == 1
    ty, tx = divmod(5, cols)
    assert int(out2[0, ty * 32:ty * 32 + 32, tx * 32:tx * 32 + 32].max()) == 0


def test_decode_frames_device_reference_files(gpu, oracle, golden_dir):
    """The reference's own files through the device decoder: every tile's value, size and pixels equal the
    oracle's decode (which is pinned by re-encoding them byte for byte); then decode + expand(Nearest) of
    Big-Ruscher.pix is Big-Ruscher.pix.png."""
    import torch
    for name, c in (("Big-Ruscher.pix", 3), ("base.pixlzr", 4)):
        raw = open(os.path.join(golden_dir, name), "rb").read()
        d = oracle.decode_container(raw)
        files = torch.frombuffer(bytearray(raw), dtype=torch.uint8).cuda()
        offs = torch.tensor([0, len(raw)], dtype=torch.int64).cuda()
        shape = (1, d["height"], d["width"], c)
        vals, ow, oh, slots = gpu.decode_frames_device(files, offs, shape, d["bw"], d["bh"])
        torch.cuda.synchronize()
        assert gpu.decode_status() == 0
        assert (vals.cpu().numpy()[0].view(np.uint32) == d["values"].view(np.uint32)).all()
        assert (ow.cpu().numpy()[0] == d["tw"]).all() and (oh.cpu().numpy()[0] == d["th"]).all()
        got, exp = slots.cpu().numpy()[0], d["slots"]
        valid = d["tw"].astype(np.int64) * d["th"] * c
        idx = np.arange(got.shape[1])[None, :] < valid[:, None]
        assert not ((got != exp[:, : got.shape[1]]) & idx).any()
        if c == 3:
            img = gpu.expand_frames_device(shape, d["bw"], d["bh"], 0, ow, oh, slots)
            ref = np.asarray(Image.open(os.path.join(golden_dir, "Big-Ruscher.pix.png")))[..., :3]
            assert (img.cpu().numpy()[0] == ref).all()


@pytest.mark.parametrize("shift", [1, 5, 11])
def test_reader_on_files_at_any_address(gpu, oracle, golden_dir, shift):
    """The reader takes its files where they are: a buffer that starts `shift` bytes into an allocation (the index kernel
    stages aligned 16-byte granules around it, the decoder's windows are aligned 8-byte words) and whose last byte is the
    last byte of the tensor (the window requests stop at the last window of the files)."""
    import torch
    raw = open(os.path.join(golden_dir, "base.pixlzr"), "rb").read()
    d = oracle.decode_container(raw)
    whole = torch.zeros(shift + len(raw), dtype=torch.uint8, device="cuda")
    whole[shift:] = torch.frombuffer(bytearray(raw), dtype=torch.uint8).cuda()
    files = whole[shift:]
    offs = torch.tensor([0, len(raw)], dtype=torch.int64).cuda()
    vals, ow, oh, slots = gpu.decode_frames_device(files, offs, (1, d["height"], d["width"], 4), d["bw"], d["bh"])
    torch.cuda.synchronize()
    assert gpu.decode_status() == 0
    assert (ow.cpu().numpy()[0] == d["tw"]).all() and (oh.cpu().numpy()[0] == d["th"]).all()
    assert (vals.cpu().numpy()[0].view(np.uint32) == d["values"].view(np.uint32)).all()
    got, exp = slots.cpu().numpy()[0], d["slots"]
    valid = d["tw"].astype(np.int64) * d["th"] * 4
    idx = np.arange(got.shape[1])[None, :] < valid[:, None]
    assert not ((got != exp[:, : got.shape[1]]) & idx).any()


def test_writer_with_too_little_room(gpu, oracle):
    """pxz_encode_frames_device with a buffer shorter than the files: what fits is written where it belongs (records are
    left out whole, nothing is written past the capacity), the offsets still say how long the files are."""
    import torch
    frames = gpu.synth_frames_device(2, 160, 224, 4, first_frame=3, dist=0)
    out = gpu.shrink_frames_device(frames, 32, 32, 1, 4, 4.0)
    offs, buf = gpu.encode_frames_device(tuple(frames.shape), 32, 32, *out)
    torch.cuda.synchronize()
    total = int(offs[-1].item())
    full = buf[:total].cpu().numpy()
    cap = total // 2
    small = torch.full((cap + 4096,), 0xA5, dtype=torch.uint8, device="cuda")
    offs2 = torch.zeros_like(offs)
    gpu.encode_frames_device(tuple(frames.shape), 32, 32, *out, out=(offs2, small[:cap]))
    torch.cuda.synchronize()
    assert (offs2.cpu().numpy() == offs.cpu().numpy()).all()
    got = small.cpu().numpy()
    assert (got[cap:] == 0xA5).all(), "bytes behind the capacity were written"
    written = got[:cap] != 0xA5
    # (a byte that was written is the byte the full files have there; bytes of value 0xA5 cannot be told apart and are skipped)
    assert (got[:cap][written] == full[:cap][written]).all()
    assert written.mean() > 0.5
    # a capacity below file 0's length: file 1's header does not fit at all, and the offsets (here a reused array with
    # stale contents from another call) are still the full files'
    cap0 = int(offs[1].item()) // 2
    small.fill_(0xA5)
    offs3 = torch.full_like(offs, 12345)
    gpu.encode_frames_device(tuple(frames.shape), 32, 32, *out, out=(offs3, small[:cap0]))
    torch.cuda.synchronize()
    assert (offs3.cpu().numpy() == offs.cpu().numpy()).all()
    got = small.cpu().numpy()
    assert (got[cap0:] == 0xA5).all(), "bytes behind the capacity were written"
    written = got[:cap0] != 0xA5
    assert (got[:cap0][written] == full[:cap0][written]).all()


@pytest.mark.parametrize("c", [4, 3])
def test_encode_decode_round_trip_on_device(gpu, oracle, c):
    """shrink -> device writer -> device decoder gives back exactly the tiles that went in (values as bits,
    sizes, valid pixel bytes), for a batch of frames; a corrupted record is flagged and zero-sized."""
    import torch
    frames = gpu.synth_frames_device(3, 200, 328, c, first_frame=4, dist=1 if c == 4 else 0)
    vals, ow, oh, slots = gpu.shrink_frames_device(frames, 32, 32, 1, 4, 16.0)
    offs, buf = gpu.encode_frames_device(tuple(frames.shape), 32, 32, vals, ow, oh, slots)
    v2, w2, h2, s2 = gpu.decode_frames_device(buf, offs, tuple(frames.shape), 32, 32)
    torch.cuda.synchronize()
    assert gpu.decode_status() == 0
    assert (v2.view(torch.int32) == vals.view(torch.int32)).all()
    assert (w2 == ow).all() and (h2 == oh).all()
    valid = (ow.long() * oh.long() * c)[..., None]
    idx = torch.arange(slots.shape[-1], device=slots.device)[None, None, :] < valid
    assert not ((s2 != slots) & idx).any()
    bad = buf.clone()
    first_record = 26 + 4 * 7  # 200/32 -> 7 tile rows
    bad[int(offs[0]) + first_record] = ord("x")  # breaks the "block" magic of tile 0
    v3, w3, h3, s3 = gpu.decode_frames_device(bad, offs, tuple(frames.shape), 32, 32)
    assert gpu.decode_status() == 2
    assert int(w3[0, 0]) == 0 and int(h3[0, 0]) == 0
    assert (w3[1:] == ow[1:]).all()


# ---- legacy image -> image filter: process() / process_custom ----------------------------------

@pytest.mark.parametrize("c,dist", [(4, 0), (4, 1), (3, 0)])
@pytest.mark.parametrize("block,down,up", [(32, 4, 0), (32, 2, 4), (64, 4, 0), (20, 3, 1)])
def test_process_matches_oracle(gpu, oracle, c, dist, block, down, up):
    """process(image, n) = Lanczos3 down / Nearest up (process/mod.rs:107-121) and process_custom with other
    filter pairs: Oklab MAD with the identity closure, shrink, resize back, RGBA8 output; opaque RGBA through
    the 32x32 fast kernels, transparent RGBA, RGB (alpha 255 added), ragged grids."""
    import torch
    frames = gpu.synth_frames_device(2, 200, 328, c, first_frame=11, dist=dist)
    out = gpu.process_frames_device(frames, block, block, down, up).cpu().numpy()
    f = frames.cpu().numpy()
    for n in range(2):
        exp = oracle.process_image(f[n], block, block, down, up)
        bad = (out[n] != exp).any(axis=2)
        assert not bad.any(), f"frame {n}: {int(bad.sum())} pixels differ"
    assert gpu.decode_status() == 0


@pytest.mark.parametrize("c", [4, 3])
@pytest.mark.parametrize("block", [16, 32, 64])
def test_strided_batches_on_the_fast_paths(gpu, oracle, block, c):
    """Frames that are views into a larger allocation (row pitch and frame stride larger than the image,
    still 16-byte aligned) and their misaligned twins (odd column offset: every fast path must stand down),
    both detectors, RGBA and RGB (widened)."""
    import torch
    H, W = 192, 320
    big = torch.zeros((3, H + 7, W + 16, c), dtype=torch.uint8, device="cuda")
    src = gpu.synth_frames_device(3, H, W, c, first_frame=21, dist=0)
    for x0 in (0, 4 if c == 4 else 16, 1):  # 16-byte aligned at 0 and 16 bytes in; misaligned at 1 pixel
        big.zero_()
        view = big[:, 3:3 + H, x0:x0 + W]
        view.copy_(src)
        for mode, factor in ((1, 16.0), (0, 1.0)):
            vals, ow, oh, slots = gpu.shrink_frames_device(view, block, block, mode, 4, factor)
            f = src.cpu().numpy()
            for n in range(3):
                exp = oracle.shrink_image(f[n], block, block, mode, 4, factor, nthreads=8)
                got = (vals[n].cpu().numpy(), ow[n].cpu().numpy().astype(np.uint32), oh[n].cpu().numpy().astype(np.uint32),
                       slots[n].cpu().numpy())
                assert_same_tiles(got, exp, c, f"{block}x{block} c{c} x0={x0} mode {mode} frame {n}")


@pytest.mark.parametrize("mode,factor", [(1, 16.0), (1, 2.0), (0, 1.0)])
@pytest.mark.parametrize("filt", [0, 2, 4])
def test_transparent_tiles_through_the_alpha_kernel(gpu, oracle, mode, factor, filt):
    """Frames whose every tile carries alpha < 255, with PXZ_HINT_TRANSPARENCY: shrink32a_kernel (four LDS planes,
    premultiplied matrix-core convolution, un-premultiply) for the full tiles, the generic kernel for the ragged
    edge and the one-pass classes; and without the hint (generic kernel for all of them): same bits either way."""
    frames = gpu.synth_frames_device(2, 200, 328, 4, first_frame=14, dist=1)
    f = frames.cpu().numpy()
    seen = set()
    for hint in (True, False):
        vals, ow, oh, slots = gpu.shrink_frames_device(frames, 32, 32, mode, filt, factor, transparency_hint=hint)
        for n in range(2):
            exp = oracle.shrink_image(f[n], 32, 32, mode, filt, factor, nthreads=8)
            got = (vals[n].cpu().numpy(), ow[n].cpu().numpy().astype(np.uint32), oh[n].cpu().numpy().astype(np.uint32),
                   slots[n].cpu().numpy())
            assert_same_tiles(got, exp, 4, f"hint {hint} mode {mode} k={factor} f{filt} frame {n}")
            seen |= set(histogram(got[1], got[2]))
    assert len(seen) >= 3, seen


@pytest.mark.parametrize("mode,factors", [(1, (64.0, 16.0, 4.0, 1.0)), (0, (3.0, 1.0, 0.3))])
@pytest.mark.parametrize("filt", [1, 4])
def test_transparent_64x64_tiles_through_the_alpha_instance(gpu, oracle, mode, factors, filt):
    """64x64 tiles with alpha < 255 and PXZ_HINT_TRANSPARENCY: shrink64_kernel<MODE, true> (four planes, premultiply
    after the detector, four channels through both matrix-core passes, un-premultiply), including its one-pass
    classes and clones; un-hinted the generic kernel takes the same tiles from list A: same bits."""
    frames = gpu.synth_frames_device(2, 280, 448, 4, first_frame=5, dist=1)
    frames[1, :130, :130, 3] = 255     # some opaque tiles in between
    frames[0, 64:128, 64:256, 3] = 0   # and fully transparent ones
    f = frames.cpu().numpy()
    seen = set()
    for factor in factors:
        for hint in (True, False):
            vals, ow, oh, slots = gpu.shrink_frames_device(frames, 64, 64, mode, filt, factor, transparency_hint=hint)
            for n in range(2):
                exp = oracle.shrink_image(f[n], 64, 64, mode, filt, factor, nthreads=8)
                got = (vals[n].cpu().numpy(), ow[n].cpu().numpy().astype(np.uint32), oh[n].cpu().numpy().astype(np.uint32),
                       slots[n].cpu().numpy())
                assert_same_tiles(got, exp, 4, f"hint {hint} mode {mode} k={factor} f{filt} frame {n}")
                seen |= set(histogram(got[1], got[2]))
    assert len(seen) >= 4, seen


@pytest.mark.parametrize("w,h,bs,c,mode,factor", [(640, 360, 32, 4, 1, 8.0), (333, 217, 64, 3, 0, 0.5), (100, 75, 16, 4, 1, 2.0)])
def test_packed_host_boundary_equals_the_slots(gpu, oracle, w, h, bs, c, mode, factor):
    """pxz_shrink_image_packed + pxz_fetch_packed: same values and dimensions as pxz_shrink_image, and the stream is
    the tiles' valid bytes back to back in tile order (= the oracle's payloads)."""
    img = oracle.synth_frame(w, h, c, 9, 1 if c == 4 else 0)
    vals, ow, oh, slots = gpu.shrink_image(img, bs, bs, mode, 4, factor)
    pv, pw, ph, stream = gpu.shrink_image_packed(img, bs, bs, mode, 4, factor)
    assert (pv.view(np.uint32) == vals.view(np.uint32)).all() and (pw == ow).all() and (ph == oh).all()
    exp = oracle.shrink_image(img, bs, bs, mode, 4, factor)
    sizes = ow.astype(np.int64) * oh * c
    assert stream.size == int(sizes.sum())
    expect = np.concatenate([exp[3][t, :n] for t, n in enumerate(sizes.tolist())])
    assert (stream == expect).all()
    # capacity check of the second step
    with pytest.raises(Exception) as e:
        gpu._check(gpu._L.pxz_fetch_packed(gpu._h, stream.ctypes.data, stream.size - 1))
    assert getattr(e.value, "code", None) == -1


def test_oklab_conversion_of_every_colour(gpu, oracle):
    """All 2^24 RGB triples (alpha spread over 0..255) through the device function the Oklab detector kernels call,
    against the oracle's conversion: L, a, b and alpha bit for bit.  The detector's sums absorb single-ulp
    differences of one pixel, so this is the test that pins the conversion itself (incl. the glibc cbrtf steps)."""
    import torch
    idx = np.arange(1 << 24, dtype=np.uint32)
    rgba = np.empty((1 << 24, 4), np.uint8)
    rgba[:, 0] = idx & 255
    rgba[:, 1] = (idx >> 8) & 255
    rgba[:, 2] = (idx >> 16) & 255
    rgba[:, 3] = (idx * 7 + (idx >> 11)) & 255
    got = gpu.oklab_pixels_device(torch.from_numpy(rgba).cuda()).cpu().numpy()
    exp = oracle.oklab_pixels(rgba)
    diff = got.view(np.uint32) != exp.view(np.uint32)
    assert not diff.any(), f"{int(diff.any(axis=1).sum())} colours differ, first: {rgba[diff.any(axis=1)][:4]}"


@pytest.mark.parametrize("bw,bh", [(8, 8), (12, 8), (24, 24), (40, 12), (48, 48), (32, 16), (16, 32), (64, 32), (96, 64),
                                   (80, 80), (4, 16), (20, 36), (128, 48), (36, 28), (128, 64), (100, 100), (96, 112)])
def test_oklab_detector_with_run_time_geometry(gpu, oracle, bw, bh):
    """shrink_by on tiles that are not 16/32/64 squares: oklab_kernel<0, NBR> (tile width a multiple of 4, rows
    16-byte aligned) takes the full tiles -- 1..4 bands in registers or any number parked, a short last band padded
    with exact zeros -- and, region by region, the ragged edge.  Tiles beyond ~7000 pixels only work because of it:
    the generic kernel's LDS image is then sized without detector planes.  Two frames in one batch, bit for bit."""
    # ragged right column and bottom row; edge widths of whole quads (8) and not (6, 1, 3: rows walked padded, and the
    # batch's rows are no 16-byte multiples, so it is re-pitched first)
    for extra, (filt, factor) in zip((8, 6, 1, 3), ((4, 1.0), (2, 0.25), (4, 0.5), (1, 2.0))):
        w, h = 5 * bw + extra, 3 * bh + (bh // 2 or 1)
        frames = gpu.synth_frames_device(2, h, w, 4, first_frame=21 + extra, dist=1)
        f = frames.cpu().numpy()
        vals, ow, oh, slots = gpu.shrink_frames_device(frames, bw, bh, 0, filt, factor)
        for n in range(2):
            exp = oracle.shrink_image(f[n], bw, bh, 0, filt, factor, nthreads=8)
            got = (vals[n].cpu().numpy(), ow[n].cpu().numpy().astype(np.uint32), oh[n].cpu().numpy().astype(np.uint32),
                   slots[n].cpu().numpy())
            assert_same_tiles(got, exp, 4, f"{bw}x{bh} +{extra} f{filt} k={factor} frame {n}")


@pytest.mark.parametrize("bw,bh", [(24, 24), (48, 32), (8, 8), (80, 80), (100, 100), (128, 64)])
def test_rgb_frames_ride_the_run_time_geometry_detector(gpu, oracle, bw, bh):
    """RGB batches, shrink_by, tile sizes off the square fast paths: widened to RGBA for the run-time-geometry Oklab
    detector and the generic RGBA kernel, slots narrowed back -- the same bits as the oracle's RGB path."""
    w, h = 4 * bw + 12, 2 * bh + 5
    frames = gpu.synth_frames_device(2, h, w, 3, first_frame=31, dist=0)
    f = frames.cpu().numpy()
    for filt, factor in ((4, 1.0), (1, 0.3)):
        vals, ow, oh, slots = gpu.shrink_frames_device(frames, bw, bh, 0, filt, factor)
        for n in range(2):
            exp = oracle.shrink_image(f[n], bw, bh, 0, filt, factor, nthreads=8)
            got = (vals[n].cpu().numpy(), ow[n].cpu().numpy().astype(np.uint32), oh[n].cpu().numpy().astype(np.uint32),
                   slots[n].cpu().numpy())
            assert_same_tiles(got, exp, 3, f"rgb {bw}x{bh} f{filt} k={factor} frame {n}")


def _sweep_cases(n, seed):
    rng = np.random.default_rng(seed)
    sizes = [2, 3, 4, 6, 8, 12, 16, 20, 24, 32, 40, 48, 64, 80, 96]
    out = []
    for i in range(n):
        bw, bh = int(rng.choice(sizes)), int(rng.choice(sizes))
        if rng.random() < 0.5:
            bh = bw  # square tiles take the fast kernels for 16 / 32 / 64
        if rng.random() < 0.4:
            bw = bh = int(rng.choice([16, 32, 64]))
        w = int(rng.integers(bw, 6 * bw + 40))
        h = int(rng.integers(bh, 5 * bh + 40))
        out.append((i, w, h, bw, bh, int(rng.choice([3, 4])), int(rng.integers(0, 2)), int(rng.integers(0, 5))))
    return out


@pytest.mark.parametrize("i,w,h,bw,bh,c,mode,filt", _sweep_cases(48, 20260214))
def test_seeded_sweep_of_geometries(gpu, oracle, i, w, h, bw, bh, c, mode, filt):
    """Seeded sweep: frame and tile sizes, channels, detector, filter, factor and alpha layout drawn at random;
    frames mix flat, smooth and noisy regions (every level class shows up) with opaque, partly and fully
    transparent patches.  1-px edge tiles under the directional detector must fail as the reference does."""
    rng = np.random.default_rng(1000 + i)
    img = oracle.synth_frame(w, h, c, i, 1 if (c == 4 and i % 3 == 0) else 0).copy()
    for _ in range(4):  # noisy / flat patches
        x0, y0 = int(rng.integers(0, w)), int(rng.integers(0, h))
        x1, y1 = int(rng.integers(x0, w)) + 1, int(rng.integers(y0, h)) + 1
        if rng.random() < 0.5:
            amp = int(rng.choice([2, 8, 40, 120]))
            patch = img[y0:y1, x0:x1, :3].astype(np.int32) + rng.integers(-amp, amp + 1, size=(y1 - y0, x1 - x0, 3))
            img[y0:y1, x0:x1, :3] = patch.clip(0, 255).astype(np.uint8)
        else:
            img[y0:y1, x0:x1, :3] = rng.integers(0, 256, size=3, dtype=np.uint8)
    if c == 4:
        for _ in range(3):
            x0, y0 = int(rng.integers(0, w)), int(rng.integers(0, h))
            x1, y1 = int(rng.integers(x0, w)) + 1, int(rng.integers(y0, h)) + 1
            kind = rng.random()
            if kind < 0.3:
                img[y0:y1, x0:x1, 3] = 0
            elif kind < 0.6:
                img[y0:y1, x0:x1, 3] = rng.integers(0, 256, size=(y1 - y0, x1 - x0), dtype=np.uint8)
            else:
                img[y0:y1, x0:x1, 3] = 255
    factor = float(rng.choice([0.5, 2.0, 8.0, 16.0, 64.0])) if mode == 1 else float(rng.choice([0.05, 0.2, 0.5, 1.0, 3.0]))
    edge_w, edge_h = w % bw or bw, h % bh or bh
    if mode == 1 and (min(edge_w, edge_h, bw, bh) == 1):
        with pytest.raises(Exception) as e:
            gpu.shrink_image(img, bw, bh, mode, filt, factor)
        assert getattr(e.value, "code", None) == -4
        with pytest.raises(RuntimeError):
            oracle.shrink_image(img, bw, bh, mode, filt, factor)
        return
    got = gpu.shrink_image(img, bw, bh, mode, filt, factor)
    exp = oracle.shrink_image(img, bw, bh, mode, filt, factor)
    assert_same_tiles(got, exp, c, f"case {i}: {w}x{h} b{bw}x{bh} c{c} mode{mode} f{filt} k={factor}")


def _one_tile_file(w, h, c, ops):
    """A .pixlzr file of one w x h tile whose QOI op stream is `ops` (bytes), laid out as encode_block does."""
    import struct
    body = struct.pack(">IIBB", w, h, c, 0) + bytes(ops) + bytes([0, 0, 0, 0, 0, 0, 0, 1])
    rec = b"block" + struct.pack(">f", 0.5) + struct.pack(">I", len(body)) + body
    head = b"PIXLZR" + bytes([0, 0, 2, 0]) + struct.pack(">IIII", w, h, w, h) + struct.pack(">I", len(rec))
    return head + rec


def test_decoder_on_streams_the_crate_encoder_never_writes(gpu, oracle):
    """The qoi 0.4.1 decoder stores a pixel in its index after RGB / RGBA / DIFF / LUMA ops only: a stream that OPENS
    with a run of the implicit opaque black and later names that slot with an INDEX op gets the zero pixel there, not
    opaque black.  And a 3-channel stream has no RGBA op: at 0xff the crate's decoder stops consuming and repeats its pixel.
    Oracle and device decoder agree.
    UNVERIFIED AGAINST THE CRATE: both behaviours are restated from memory of qoi 0.4.1's decode_impl_slice (its source is not
    in this environment and no reference fixture holds such a stream -- the crate's own encoder never writes one), and
    oracle and device were changed together, so this test shows that the two agree, not that either equals the crate."""
    import torch
    # RUN of 3 (opaque black x3), INDEX 53 (= hash of opaque black: (255 * 11) % 64) -> zero pixel, then RGB
    ops = [0xc0 | 2, 53, 0xfe, 10, 20, 30, 0xc0 | 0]
    for c in (4, 3):
        raw = _one_tile_file(3, 2, c, ops)
        d = oracle.decode_container(raw)
        exp = d["slots"][0][: 6 * c].reshape(6, c)
        assert (exp[:3, :3] == 0).all() and (exp[3, :3] == 0).all()  # the INDEX op yields the zero pixel
        if c == 4:
            assert (exp[:3, 3] == 255).all() and exp[3, 3] == 0
        files = torch.frombuffer(bytearray(raw), dtype=torch.uint8).cuda()
        offs = torch.tensor([0, len(raw)], dtype=torch.int64).cuda()
        vals, ow, oh, slots = gpu.decode_frames_device(files, offs, (1, 2, 3, c), 3, 2)
        torch.cuda.synchronize()
        assert gpu.decode_status() == 0
        assert (slots.cpu().numpy()[0, 0, : 6 * c].reshape(6, c) == exp).all()
    # an RGBA op byte in a 3-channel stream: the crate's 3-channel decoder has no arm for it, consumes nothing and repeats
    # the pixel it holds for the rest of the tile (qoi 0.4.1's catch-all arm, restated from memory): RGB (9, 8, 7), then 0xff
    raw = _one_tile_file(3, 2, 3, [0xfe, 9, 8, 7, 0xff, 1, 2, 3, 4, 0xc0])
    d = oracle.decode_container(raw)
    exp = d["slots"][0][: 6 * 3].reshape(6, 3)
    assert (exp == np.array([9, 8, 7], np.uint8)).all()
    files = torch.frombuffer(bytearray(raw), dtype=torch.uint8).cuda()
    offs = torch.tensor([0, len(raw)], dtype=torch.int64).cuda()
    vals, ow, oh, slots = gpu.decode_frames_device(files, offs, (1, 2, 3, 3), 3, 2)
    torch.cuda.synchronize()
    assert gpu.decode_status() == 0 and int(ow[0, 0]) == 3 and int(oh[0, 0]) == 2
    assert (slots.cpu().numpy()[0, 0, : 6 * 3].reshape(6, 3) == exp).all()


# ---- legacy quad-tree filter: tree::process / tree::process_custom --------------------------------

@pytest.mark.parametrize("c,dist", [(4, 0), (4, 1), (3, 0)])
@pytest.mark.parametrize("bw,bh,down,up", [(64, 64, 4, 0), (32, 32, 2, 4), (48, 48, 4, 0), (64, 32, 3, 1), (40, 40, 4, 0)])
def test_tree_process_matches_oracle(gpu, oracle, c, dist, bw, bh, down, up):
    """tree::process(image, n, k) = Lanczos3 down / Nearest up, 4 x 4 minimum (tree.rs:89-109) and process_custom with other
    filters / non-square blocks: per level the tiles under the threshold are pixelised, the others split in four until
    the minimum block, where they keep their pixels.  Thresholds from "split everything" to "pixelise everything", and
    a negative one (the outermost level inverted, tree.rs:37-38).  RGBA opaque / transparent, RGB; ragged grids."""
    frames = gpu.synth_frames_device(2, 200, 328, c, first_frame=17, dist=dist)
    f = frames.cpu().numpy()
    split_some = False
    for thr in (0.0, 0.012, 0.04, 0.15, 0.4, 50.0, -0.04):
        out = gpu.tree_process_frames_device(frames, bw, bh, thr, 4, 4, down, up).cpu().numpy()
        for n in range(2):
            exp = oracle.tree_process_image(f[n], bw, bh, thr, 4, 4, down, up)
            bad = (out[n] != exp).any(axis=2)
            assert not bad.any(), f"thr {thr} frame {n}: {int(bad.sum())} pixels differ"
            if 0.0 < thr < 1.0:
                flat = oracle.process_image(f[n], bw, bh, down, up)
                src = np.concatenate([f[n], np.full(f[n].shape[:2] + (1,), 255, np.uint8)], axis=2) if c == 3 else f[n]
                split_some |= bool((exp != flat).any()) and bool((exp != src).any())
    assert split_some  # some threshold gave a real mixture of levels
    assert gpu.decode_status() == 0


def test_tree_process_edge_cases(gpu, product, oracle):
    """A block at or below the minimum hands the image back (tree.rs:34-36; as RGBA here); a larger minimum ends the
    recursion earlier; block sizes that do not halve evenly down to the last level are refused on the device."""
    frames = gpu.synth_frames_device(1, 96, 160, 3, first_frame=2, dist=0)
    f = frames.cpu().numpy()[0]
    out = gpu.tree_process_frames_device(frames, 4, 4, 0.05).cpu().numpy()[0]
    assert (out[..., :3] == f).all() and (out[..., 3] == 255).all()
    assert (out == oracle.tree_process_image(f, 4, 4, 0.05)).all()
    out = gpu.tree_process_frames_device(frames, 64, 64, 0.02, 16, 16).cpu().numpy()[0]
    assert (out == oracle.tree_process_image(f, 64, 64, 0.02, 16, 16)).all()
    with pytest.raises(product.PxzError) as e:
        gpu.tree_process_frames_device(frames, 160, 160, 0.05)  # blocks above 128 px are refused
    assert e.value.code == -5


@pytest.mark.parametrize("c,dist", [(4, 0), (4, 1), (3, 0)])
@pytest.mark.parametrize("bw,bh,down,up", [(128, 128, 4, 0), (50, 50, 4, 0), (50, 50, 2, 4), (100, 36, 1, 2), (128, 96, 3, 3), (37, 61, 4, 1)])
def test_tree_process_on_any_geometry(gpu, oracle, c, dist, bw, bh, down, up):
    """tree::process where the per-level grids do not exist (round 3, pxz_tree.hip): the 128-px blocks src/bin/tree.rs:6
    calls it with, and blocks whose halvings go odd (50 -> 25 -> 12 + 12 + 1: every tile is cut from its own corner,
    tree.rs:70-79), ragged frames, every filter pair, RGBA opaque / transparent and RGB -- against the oracle's recursion,
    pixel for pixel."""
    frames = gpu.synth_frames_device(2, 300, 428, c, first_frame=23, dist=dist)
    f = frames.cpu().numpy()
    mixtures = 0
    for thr in (0.0, 0.012, 0.04, 0.15, 50.0, -0.04):
        out = gpu.tree_process_frames_device(frames, bw, bh, thr, 4, 4, down, up).cpu().numpy()
        for n in range(2):
            exp = oracle.tree_process_image(f[n], bw, bh, thr, 4, 4, down, up)
            bad = (out[n] != exp).any(axis=2)
            assert not bad.any(), f"{bw}x{bh} thr {thr} frame {n}: {int(bad.sum())} pixels differ, first at {np.argwhere(bad)[0]}"
            src = np.concatenate([f[n], np.full(f[n].shape[:2] + (1,), 255, np.uint8)], axis=2) if c == 3 else f[n]
            mixtures += int(bool((exp != src).any()) and bool((exp == src).all(axis=2).any()))
    assert mixtures  # some threshold gave a real mixture of pixelised tiles and tiles that kept their pixels
    # a larger minimum ends the recursion earlier
    out = gpu.tree_process_frames_device(frames, bw, bh, 0.03, 20, 9, down, up).cpu().numpy()
    for n in range(2):
        assert (out[n] == oracle.tree_process_image(f[n], bw, bh, 0.03, 20, 9, down, up)).all()


def test_trim_releases_and_the_handle_goes_on(gpu, oracle):
    """pxz_trim between calls: the scratch buffers (and the rings of the list calls) are given back, results do not change."""
    imgs = [oracle.synth_frame(256, 160, 4, 70 + k, 0) for k in range(4)]
    before = gpu.shrink_images(imgs, 32, 32, 1, 4, 8.0)
    gpu.trim()
    after = gpu.shrink_images(imgs, 32, 32, 1, 4, 8.0)
    for k, img in enumerate(imgs):
        exp = oracle.shrink_image(img, 32, 32, 1, 4, 8.0)
        assert_same_tiles(before[k], exp, 4, f"before trim, image {k}")
        assert_same_tiles(after[k], exp, 4, f"after trim, image {k}")
    gpu.trim()
    got = gpu.shrink_image(imgs[0], 32, 32, 0, 4, 1.0)
    assert_same_tiles(got, oracle.shrink_image(imgs[0], 32, 32, 0, 4, 1.0), 4, "after trim")


@pytest.mark.parametrize("c,mode,factor,bs", [(4, 1, 8.0, 32), (3, 0, 0.5, 64), (4, 0, 1.0, 16)])
def test_pipelined_host_boundary_over_a_list_of_images(gpu, oracle, c, mode, factor, bs):
    """pxz_shrink_images / pxz_shrink_images_packed: seven images through the three-stage pipeline (more images than
    buffer sets, so every set is reused) give what pxz_shrink_image gives one by one -- and the oracle."""
    imgs = [oracle.synth_frame(333 if c == 3 else 640, 217 if c == 3 else 360, c, 40 + k, (k % 2) if c == 4 else 0) for k in range(7)]
    res = gpu.shrink_images(imgs, bs, bs, mode, 4, factor)
    pk = gpu.shrink_images(imgs, bs, bs, mode, 4, factor, packed=True)
    lod = gpu.shrink_images(imgs, bs, bs, mode, 4, factor, want_pixels=False)
    for k, img in enumerate(imgs):
        exp = oracle.shrink_image(img, bs, bs, mode, 4, factor)
        assert_same_tiles(res[k], exp, c, f"list image {k}")
        assert (lod[k][0].view(np.uint32) == exp[0].view(np.uint32)).all() and (lod[k][1] == exp[1]).all() and lod[k][3] is None
        sizes = exp[1].astype(np.int64) * exp[2] * c
        stream = np.concatenate([exp[3][t, :n] for t, n in enumerate(sizes.tolist())])
        assert (pk[k][0].view(np.uint32) == exp[0].view(np.uint32)).all() and (pk[k][1] == exp[1]).all() and (pk[k][2] == exp[2]).all()
        assert pk[k][3].size == stream.size and (pk[k][3] == stream).all()


@pytest.mark.parametrize("bs", [16, 32, 64])
@pytest.mark.parametrize("filt", [0, 2, 4])
@pytest.mark.parametrize("mode,factors", [(1, (64.0, 16.0, 4.0, 1.0)), (0, (3.0, 1.0, 0.25, 0.05))])
def test_rgb_frames_on_the_square_fast_paths(gpu, oracle, filt, mode, factors, bs):
    """RGB batches, 16x16, 32x32 and 64x64 tiles, both callers: shrink16/32/64_kernel<.., 3> read 12-byte pixel quads
    and write RGB slots themselves; oklab2_kernel<32, 3> cuts a lane's two pixels out of an aligned 8 bytes and
    oklab_kernel<64, 0, 3> a lane's four out of 12 (no widened copy): every reduced size from the clone to 1x1, one-pass
    classes, a ragged edge (generic kernel), views with a row pitch / column offset that keep 4-byte alignment and
    ones that do not (widened path)."""
    import torch
    seen = set()
    frames = gpu.synth_frames_device(2, 1080, 1920, 3, first_frame=3, dist=0)
    f = frames.cpu().numpy()
    for factor in factors:
        vals, ow, oh, slots = gpu.shrink_frames_device(frames, bs, bs, mode, filt, factor)
        for n in range(2):
            exp = oracle.shrink_image(f[n], bs, bs, mode, filt, factor, nthreads=8)
            got = (vals[n].cpu().numpy(), ow[n].cpu().numpy().astype(np.uint32), oh[n].cpu().numpy().astype(np.uint32), slots[n].cpu().numpy())
            assert_same_tiles(got, exp, 3, f"rgb{bs} mode{mode} f{filt} k={factor} frame {n}")
            seen |= set(histogram(got[1], got[2]))
    assert ({(bs, bs), (8, 8), (4, 4), (2, 2), (1, 1)} if mode == 1 else {(bs, bs), (8, 8), (4, 4)}) <= seen, seen
    big = torch.zeros((2, 200, 360, 3), dtype=torch.uint8, device="cuda")
    src = gpu.synth_frames_device(2, 192, 320, 3, first_frame=8, dist=0)
    for x0 in (0, 4, 1):  # 0 and 12 bytes in: 4-byte aligned rows; 3 bytes in: not
        view = big[:, 3:195, x0:x0 + 320]
        view.copy_(src)
        vals, ow, oh, slots = gpu.shrink_frames_device(view, bs, bs, mode, filt, factors[1])
        for n in range(2):
            exp = oracle.shrink_image(src[n].cpu().numpy(), bs, bs, mode, filt, factors[1])
            got = (vals[n].cpu().numpy(), ow[n].cpu().numpy().astype(np.uint32), oh[n].cpu().numpy().astype(np.uint32), slots[n].cpu().numpy())
            assert_same_tiles(got, exp, 3, f"rgb{bs} view x0={x0} frame {n}")


@pytest.mark.parametrize("bs", [32, 64])
@pytest.mark.parametrize("mode,factor", [(1, 16.0), (1, 2.0), (0, 1.0)])
def test_mostly_transparent_batches_go_to_the_four_plane_kernel_first(product, oracle, mode, factor, bs):
    """A handle that has seen a launch in which most full tiles had transparency sends the next launches to
    shrink32a_kernel / the four-plane instance of shrink64_kernel alone (every tile, no listing pass by the opaque kernel): the results must not depend on that -- for
    transparent frames, for the opaque frames that follow them (still in that mode: the statistic is one launch behind),
    with a ragged edge (those tiles go on to the generic kernel) and after the handle has swung back."""
    import torch
    h = product.Handle(0)
    seq = [(1, 2160, 3840), (1, 2160, 3840), (0, 2160, 3840), (0, 2160, 3840), (1, 2112, 3840), (1, 2112, 3840)]
    for k, (dist, hh, ww) in enumerate(seq):
        frames = h.synth_frames_device(2, hh, ww, 4, first_frame=5 + k, dist=dist)
        f = frames.cpu().numpy()
        vals, ow, oh, slots = h.shrink_frames_device(frames, bs, bs, mode, 4, factor)
        torch.cuda.synchronize()
        for n in range(2):
            exp = oracle.shrink_image(f[n], bs, bs, mode, 4, factor, nthreads=8)
            got = (vals[n].cpu().numpy(), ow[n].cpu().numpy().astype(np.uint32), oh[n].cpu().numpy().astype(np.uint32), slots[n].cpu().numpy())
            assert_same_tiles(got, exp, 4, f"launch {k} (dist {dist}, {ww}x{hh}) frame {n}")
