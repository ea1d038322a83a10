"""The C++ mirror of the reference's operator surface (include/pixlzr.hpp) driven like the CLI's
image_to_pix (src/bin/main.rs:142-175): raw image -> Pixlzr::from_image -> shrink_* -> save."""
import os
import subprocess

import numpy as np
import pytest
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "pixlzr-rust_amd", "csrc", "pxz_encode.bin")


def _run(tmp_path, img, bw, bh, mode, filt, factor):
    raw = tmp_path / "in.raw"
    out = tmp_path / "out.pixlzr"
    np.ascontiguousarray(img).tofile(raw)
    h, w, c = img.shape
    r = subprocess.run([TOOL, str(raw), str(w), str(h), str(c), str(bw), str(bh), mode, str(filt), repr(factor), str(out)],
                       capture_output=True, text=True)
    return r, (out.read_bytes() if out.exists() else None)


def test_from_image_and_encode_reproduce_base_pixlzr(tmp_path, product, golden_dir):
    """bench-00.rs:55,66 — from_image(64,64) + encode_to_vec, no shrink: the reference's own file."""
    product.build_library()
    img = np.asarray(Image.open(os.path.join(golden_dir, "base.png")))
    r, data = _run(tmp_path, img, 64, 64, "none", 4, 1.0)
    assert r.returncode == 0, r.stderr
    assert "442 blocks" in r.stdout
    assert data == open(os.path.join(golden_dir, "base.pixlzr"), "rb").read()


def test_shrink_without_gpu_fails_loudly(tmp_path, product, oracle):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r, data = _run(tmp_path, oracle.synth_frame(64, 64, 4, 0, 0), 32, 32, "dir", 4, 16.0)
    assert r.returncode == 1 and "no usable gfx950 device" in r.stderr and data is None


@pytest.mark.gpu
@pytest.mark.parametrize("name,block,mode,filt,factor", [
    ("Big-Ruscher.png", 32, "by", 4, 0.125), ("Big-Ruscher.png", 32, "dir", 4, 16.0),
    ("base.png", 64, "by", 2, 0.25), ("base.png", 64, "dir", 4, 8.0), ("image.png", 8, "by", 0, 0.5)])
def test_cli_flow_files_equal_oracle(tmp_path, oracle, golden_dir, name, block, mode, filt, factor):
    """image_to_pix with --force: whole .pixlzr file from the C++ mirror (GPU) == oracle (CPU)."""
    img = np.ascontiguousarray(np.asarray(Image.open(os.path.join(golden_dir, name))))
    r, data = _run(tmp_path, img, block, block, mode, filt, factor)
    assert r.returncode == 0, r.stderr
    h, w, c = img.shape
    pm = 0 if mode == "by" else 1
    v, ow, oh, slots = oracle.shrink_image(img, block, block, pm, filt, factor, nthreads=8)
    assert data == oracle.encode_container(w, h, block, block, c, 0, v, None, ow, oh, slots)
    if name == "Big-Ruscher.png" and mode == "by":
        # the same command that produced the reference's Big-Ruscher.pix: identical LOD decisions
        d = oracle.decode_container(open(os.path.join(golden_dir, "Big-Ruscher.pix"), "rb").read())
        mine = oracle.decode_container(data)
        assert (mine["tw"] == d["tw"]).all() and (mine["th"] == d["th"]).all()


DECODE_TOOL = os.path.join(ROOT, "pixlzr-rust_amd", "csrc", "pxz_decode.bin")


@pytest.mark.gpu
def test_pix_to_image_reproduces_big_ruscher_pix_png(tmp_path, product, golden_dir):
    """pix_to_image on the reference's own pair: Pixlzr::open(Big-Ruscher.pix).to_image(Nearest) is
    Big-Ruscher.pix.png (device reader + expand behind the C++ mirror)."""
    product.build_library()
    out = tmp_path / "out.raw"
    r = subprocess.run([DECODE_TOOL, os.path.join(golden_dir, "Big-Ruscher.pix"), "0", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout.split() == ["1920", "1080", "3", "2040"]
    got = np.fromfile(out, np.uint8).reshape(1080, 1920, 3)
    ref = np.asarray(Image.open(os.path.join(golden_dir, "Big-Ruscher.pix.png")))[..., :3]
    assert (got == ref).all()


@pytest.mark.gpu
@pytest.mark.parametrize("filt", ["file", "3", "4"])
def test_encode_then_decode_through_the_mirror(tmp_path, oracle, filt):
    """image_to_pix then pix_to_image: the decoded image equals the oracle's expand of the oracle's tiles
    ("file": the stored filter byte, which the writer leaves at Nearest)."""
    img = oracle.synth_frame(200, 136, 4, 2, 1)
    r, data = _run(tmp_path, img, 32, 32, "dir", 4, 16.0)
    assert r.returncode == 0, r.stderr
    pix = tmp_path / "out.pixlzr"
    raw = tmp_path / "back.raw"
    r = subprocess.run([DECODE_TOOL, str(pix), filt, str(raw)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = np.fromfile(raw, np.uint8).reshape(136, 200, 4)
    v, ow, oh, slots = oracle.shrink_image(img, 32, 32, 1, 4, 16.0)
    exp = oracle.expand_image(200, 136, 32, 32, 4, 0 if filt == "file" else int(filt), ow, oh, slots)
    assert (got == exp).all()
