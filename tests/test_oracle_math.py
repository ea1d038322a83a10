"""Pins the oracle's libm restatements against this image's glibc 2.35 (the libm the
reference's f32::cbrt / f32::hypot would call on this platform)."""
import numpy as np


def test_cbrtf_matches_platform_libm(oracle):
    L = oracle.lib()
    # strided sweep of every exponent in [0, 4.0] (dense sweep = 0 mismatches, 26 s)
    assert L.orc_selftest_cbrtf(0x00000000, 0x40800000, 13) == 0
    # dense windows: denormals, around 2^-k boundaries, the top of the [0,1] range Oklab uses
    for lo in (0x00000000, 0x3a000000, 0x3d7ff000, 0x3effff00, 0x3f7f0000):
        assert L.orc_selftest_cbrtf(lo, lo + (1 << 21), 1) == 0
    assert L.orc_cbrtf(0.0) == 0.0
    assert L.orc_cbrtf(1.0) == 1.0
    assert L.orc_cbrtf(0.125) == 0.5


def test_hypotf_matches_platform_libm(oracle):
    assert oracle.lib().orc_selftest_hypotf(5_000_000, 1234) == 0


def test_srgb_lut_is_correctly_rounded_eotf(oracle):
    L = oracle.lib()
    for v in range(256):
        x = v / 255.0
        lin = x / 12.92 if x <= 0.04045 else ((x + 0.055) / 1.055) ** 2.4
        assert np.float32(L.orc_srgb_u8_to_linear(v)) == np.float32(lin), v
    # first entries of fast-srgb8's published table
    assert np.float32(L.orc_srgb_u8_to_linear(1)) == np.float32(0.000303527)
    assert np.float32(L.orc_srgb_u8_to_linear(255)) == np.float32(1.0)


def test_reduce_dims_levels(oracle):
    # operations.rs:140-156: level = 2^min(round(log2 v),0); 32 -> {32,16,8,4,2,1}
    cases = [(1.0, 32), (5.0, 32), (0.75, 32), (0.70, 16), (0.5, 16), (0.36, 16), (0.35, 8), (0.25, 8),
             (0.125, 4), (0.0625, 2), (0.03125, 1), (1e-6, 1), (0.0, 1)]
    for v, exp in cases:
        nw, nh, st = oracle.reduce_dims(v, v, 32, 32)
        assert (nw, nh) == (exp, exp), (v, nw)
        assert st == np.float32(oracle.lib().orc_hypotf(v, v))
    # edge tile sizes use ceil: 24 -> {24,12,6,3,2,1}; 17 -> {17,9,5,3,2,1}
    assert [oracle.reduce_dims(2.0 ** -k, 2.0 ** -k, 24, 17)[:2] for k in range(6)] == \
        [(24, 17), (12, 9), (6, 5), (3, 3), (2, 2), (1, 1)]
    # parse_value (operations.rs:128-138): negative v -> max(1+v, 0)
    assert oracle.reduce_dims(-0.5, -0.75, 32, 32)[:2] == (16, 8)
    assert oracle.reduce_dims(-3.0, -1.0, 32, 32)[:2] == (1, 1)
    # anisotropic (shrink_directionally passes (hz*f, vr*f)): width from v0, height from v1
    assert oracle.reduce_dims(1.0, 0.25, 32, 32)[:2] == (32, 8)
