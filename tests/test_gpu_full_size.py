"""GPU parity at BASELINE.json's full sizes (configs[3] and configs[4]): the HIP path through the C ABI against the
oracle on whole frames -- block values as u32 bit patterns, reduced dimensions, payload bytes, .pixlzr bytes -- at the
sizes where 32-bit index paths would give way: 1 048 576 tiles per frame, 1 GiB of slots per frame, batches whose
last frame lies beyond 4 GiB."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu(product):
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    h = product.Handle(0)
    yield h
    h.close()


def assert_same_tiles_device(got, exp, channels, what):
    """got: device tensors of one frame (values, w, h, slots); exp: the oracle's numpy arrays.  The byte comparison
    runs on the device in chunks of tiles (a frame of 16384^2 has 1 GiB of slots)."""
    import torch
    gv, gw, gh, gs = got
    ev, ew, eh, es = exp
    dev = gs.device
    tw = torch.from_numpy(ew.astype(np.int32)).to(dev)
    th = torch.from_numpy(eh.astype(np.int32)).to(dev)
    nbad = int(((gw != tw) | (gh != th)).sum())
    assert nbad == 0, f"{what}: reduced dims differ on {nbad} tiles"
    eb = torch.from_numpy(ev.view(np.int32)).to(dev)
    nbad = int((gv.view(torch.int32) != eb).sum())
    assert nbad == 0, f"{what}: block values differ on {nbad} tiles"
    n, slot = gs.shape
    valid = tw.long() * th.long() * channels
    lane = torch.arange(slot, device=dev)[None, :]
    chunk = max(1, (256 << 20) // slot)
    bad_bytes = bad_tiles = 0
    for t0 in range(0, n, chunk):
        t1 = min(n, t0 + chunk)
        e = torch.from_numpy(es[t0:t1]).to(dev)
        d = (gs[t0:t1] != e) & (lane < valid[t0:t1, None])
        bad_bytes += int(d.sum())
        bad_tiles += int(d.any(dim=1).sum())
    assert bad_bytes == 0, f"{what}: {bad_bytes} payload bytes differ in {bad_tiles} tiles"


@pytest.mark.parametrize("mode,factor", [(1, 16.0), (0, 1.0)])
@pytest.mark.parametrize("block", [16, 32, 64])
def test_16384_square_frame_block_sweep(gpu, oracle, block, mode, factor):
    """BASELINE configs[3]: one 16384x16384 RGBA8 frame at 16 / 32 / 64-px tiles (1 048 576 / 262 144 / 65 536 tiles),
    both callers, against the oracle in full; then the stream offsets of pxz_pack_tiles_device and the files of
    pxz_encode_frames_device at that tile count (split.rs:37-61 grid, encoding/mod.rs:40-89 line table)."""
    import torch
    W = H = 16384
    frames = gpu.synth_frames_device(1, H, W, 4, first_frame=0, dist=0)
    img = frames[0].cpu().numpy()
    assert (img[:64] == oracle.synth_frame(W, 64, 4, 0, 0)).all()  # generator rows do not depend on the height
    vals, ow, oh, slots = gpu.shrink_frames_device(frames, block, block, mode, 4, factor)
    torch.cuda.synchronize()
    exp = oracle.shrink_image(img, block, block, mode, 4, factor, nthreads=16)
    assert exp[1].size == (W // block) * (H // block)
    assert_same_tiles_device((vals[0], ow[0], oh[0], slots[0]), exp, 4, f"16384^2 b{block} mode{mode}")
    hist = torch.unique(ow[0] * 256 + oh[0])
    assert hist.numel() >= 2, hist  # more than one size class (64-px tiles under shrink_by: 64x64 and 32x32 only)
    # compaction: offsets are the exclusive scan of the valid sizes, the stream is the valid bytes in tile order
    sizes = exp[1].astype(np.int64) * exp[2] * 4
    exp_off = np.concatenate([[0], np.cumsum(sizes)])
    offsets, packed = gpu.pack_tiles_device(ow, oh, slots, 4)
    torch.cuda.synchronize()
    assert (offsets.cpu().numpy() == exp_off).all()
    total = int(exp_off[-1])
    es = exp[3]
    mask = np.arange(es.shape[1])[None, :] < sizes[:, None] if es.shape[0] <= 65536 else None
    if mask is not None:
        assert (packed[:total].cpu().numpy() == es[mask]).all()
    else:  # 1 M tiles: check the stream in windows of tiles
        pk = packed[:total].cpu().numpy()
        for t0 in range(0, es.shape[0], 65536):
            t1 = t0 + 65536
            m = np.arange(es.shape[1])[None, :] < sizes[t0:t1, None]
            assert (pk[exp_off[t0]:exp_off[t1]] == es[t0:t1][m]).all(), (t0, t1)
    del packed, offsets
    # the device writer: the whole file equals the oracle's writer fed with the oracle's tiles
    offs, buf = gpu.encode_frames_device((1, H, W, 4), block, block, vals, ow, oh, slots)
    torch.cuda.synchronize()
    offs = offs.cpu().numpy()
    ref = oracle.encode_container(W, H, block, block, 4, 0, exp[0], None, exp[1], exp[2], exp[3])
    assert offs[0] == 0 and offs[1] == len(ref)
    mine = buf[: offs[1]].cpu().numpy()
    assert (mine == np.frombuffer(ref, np.uint8)).all()


@pytest.mark.parametrize("mode,factor", [(1, 16.0), (0, 1.0)])
def test_batch_of_64_8k_frames_in_one_call(gpu, oracle, mode, factor):
    """BASELINE configs[4] on one GPU: 64 x 7680x4320 RGBA8 (8.49 GB of source, 8.49 GB of slots) in ONE call.  The
    last frame (its pixels and slots start beyond 4 GiB) equals its own single-frame call and the oracle; the first
    frame equals the oracle; through pxz_encode_frames_device the last file equals the oracle's writer and every
    file length equals what its tiles add up to."""
    import torch
    N, H, W = 64, 4320, 7680
    frames = gpu.synth_frames_device(N, H, W, 4, first_frame=0, dist=0)
    assert frames.stride(0) * (N - 1) > (1 << 32)
    vals, ow, oh, slots = gpu.shrink_frames_device(frames, 32, 32, mode, 4, factor)
    torch.cuda.synchronize()
    T = ow.shape[1]
    for n in (N - 1, 0, 37):
        img = frames[n].cpu().numpy()
        if n == N - 1:
            assert (img == oracle.synth_frame(W, H, 4, n, 0)).all()
        exp = oracle.shrink_image(img, 32, 32, mode, 4, factor, nthreads=16)
        assert_same_tiles_device((vals[n], ow[n], oh[n], slots[n]), exp, 4, f"batch frame {n} mode{mode}")
        if n == N - 1:
            last = exp
    alone = gpu.shrink_frames_device(frames[N - 1:N], 32, 32, mode, 4, factor)
    assert torch.equal(alone[0][0].view(torch.int32), vals[N - 1].view(torch.int32))
    assert torch.equal(alone[1][0], ow[N - 1]) and torch.equal(alone[2][0], oh[N - 1])
    del alone
    # every frame differs from its neighbour (seeds differ) and all frames were written
    assert int((ow.min(dim=1).values < 1).sum()) == 0
    offs, buf = gpu.encode_frames_device((N, H, W, 4), 32, 32, vals, ow, oh, slots)
    torch.cuda.synchronize()
    offs = offs.cpu().numpy()
    assert offs[0] == 0 and (np.diff(offs) > 26 + 4 * 135 + 13 * T).all()
    ref = oracle.encode_container(W, H, 32, 32, 4, 0, last[0], None, last[1], last[2], last[3])
    assert offs[N] - offs[N - 1] == len(ref)
    assert (buf[offs[N - 1]:offs[N]].cpu().numpy() == np.frombuffer(ref, np.uint8)).all()
    # checksum of checksums: decoding all 64 files on the device gives back every tile that went in
    v2, w2, h2, s2 = gpu.decode_frames_device(buf, torch.from_numpy(offs).cuda(), (N, H, W, 4), 32, 32)
    torch.cuda.synchronize()
    assert gpu.decode_status() == 0
    assert torch.equal(v2.view(torch.int32), vals.view(torch.int32)) and torch.equal(w2, ow) and torch.equal(h2, oh)
    lane = torch.arange(4096, device=slots.device)[None, :]
    for n in range(N):
        valid = (ow[n].long() * oh[n].long() * 4)[:, None]
        assert not ((s2[n] != slots[n]) & (lane < valid)).any(), n


# ---------------------------------------------------------------------------------------------------------------
# The decode side and the legacy filter at full size: Pixlzr::expand + to_image (pixlzr.rs:77-122,
# pixlzr_image.rs:24-74) and process (process/mod.rs:71-121) against the oracle on whole 8K and 16384^2 frames, and a
# batch whose output passes 4 GiB.
# ---------------------------------------------------------------------------------------------------------------
def assert_same_image_device(got, exp, what):
    """got: device tensor [H, W, C]; exp: the oracle's numpy image.  Compared on the device in bands of rows."""
    import torch
    H = got.shape[0]
    band = max(1, (256 << 20) // (got.shape[1] * got.shape[2]))
    bad = 0
    for y0 in range(0, H, band):
        e = torch.from_numpy(exp[y0:y0 + band]).to(got.device)
        bad += int((got[y0:y0 + band] != e).sum())
    assert bad == 0, f"{what}: {bad} bytes differ"


@pytest.mark.parametrize("size,block,filt", [((4320, 7680), 32, 0), ((4320, 7680), 32, 4), ((4320, 7680), 64, 4),
                                              ((16384, 16384), 32, 0), ((16384, 16384), 32, 4), ((16384, 16384), 64, 4),
                                              ((16384, 16384), 16, 4)])
def test_expand_of_a_whole_frame(gpu, oracle, size, block, filt):
    """Pixlzr::expand + to_image of one whole frame (8K; 16384^2 = 1 GiB of output) at the reference's tile sizes against
    orc_expand_image, fed with the same stored tiles (what shrink_directionally(Lanczos3, 16) left: every size class)."""
    import torch
    H, W = size
    frames = gpu.synth_frames_device(1, H, W, 4, first_frame=2, dist=0)
    _, ow, oh, slots = gpu.shrink_frames_device(frames, block, block, 1, 4, 16.0)
    del frames
    back = gpu.expand_frames_device((1, H, W, 4), block, block, filt, ow, oh, slots)
    torch.cuda.synchronize()
    assert gpu.decode_status() == 0
    assert torch.unique(ow[0] * 256 + oh[0]).numel() >= 6
    exp = oracle.expand_image(W, H, block, block, 4, filt, ow[0].cpu().numpy(), oh[0].cpu().numpy(), slots[0].cpu().numpy())
    assert_same_image_device(back[0], exp, f"expand {W}x{H} b{block} filter {filt}")


def test_expand_of_a_batch_beyond_4_gib(gpu, oracle):
    """34 x 8K RGBA frames in ONE expand call: 4.51 GB of output, the last frame's rows lie beyond 4 GiB.  Frames 33, 0 and 17
    equal the oracle's expand of their own tiles; frame 33 equals its own single-frame call."""
    import torch
    N, H, W = 34, 4320, 7680
    frames = gpu.synth_frames_device(N, H, W, 4, first_frame=0, dist=0)
    _, ow, oh, slots = gpu.shrink_frames_device(frames, 32, 32, 1, 4, 16.0)
    del frames
    back = gpu.expand_frames_device((N, H, W, 4), 32, 32, 4, ow, oh, slots)
    torch.cuda.synchronize()
    assert back.stride(0) * (N - 1) > (1 << 32)
    assert gpu.decode_status() == 0
    for n in (N - 1, 0, 17):
        exp = oracle.expand_image(W, H, 32, 32, 4, 4, ow[n].cpu().numpy(), oh[n].cpu().numpy(), slots[n].cpu().numpy())
        assert_same_image_device(back[n], exp, f"expand batch frame {n}")
    alone = gpu.expand_frames_device((1, H, W, 4), 32, 32, 4, ow[N - 1:N].contiguous(), oh[N - 1:N].contiguous(), slots[N - 1:N].contiguous())
    assert torch.equal(alone[0], back[N - 1])


@pytest.mark.parametrize("block", [32, 64])
def test_process_of_a_whole_8k_frame(gpu, oracle, block):
    """process (process/mod.rs:107-121: Lanczos3 down, Nearest up, |x - avg|, identity) on one 8K frame against
    orc_process_image, pixel for pixel."""
    import torch
    H, W = 4320, 7680
    frames = gpu.synth_frames_device(1, H, W, 4, first_frame=5, dist=0)
    out = gpu.process_frames_device(frames, block, block)
    torch.cuda.synchronize()
    exp = oracle.process_image(frames[0].cpu().numpy(), block, block)
    assert_same_image_device(out[0], exp, f"process 8K b{block}")


def test_tree_process_of_a_whole_8k_frame(gpu, oracle):
    """tree::process (process/tree.rs:89-109) with the 128-px blocks of src/bin/tree.rs:6 on one 8K frame against
    orc_tree_process_image, pixel for pixel."""
    import torch
    H, W = 4320, 7680
    frames = gpu.synth_frames_device(1, H, W, 4, first_frame=6, dist=0)
    out = gpu.tree_process_frames_device(frames, 128, 128, 0.05)
    torch.cuda.synchronize()
    exp = oracle.tree_process_image(frames[0].cpu().numpy(), 128, 128, 0.05)
    assert_same_image_device(out[0], exp, "tree::process 8K b128")
