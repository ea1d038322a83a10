"""ctypes view of oracle/liboracle.so — the CPU restatement of the reference path.

TEST INFRASTRUCTURE.  Importable only from tests/, bench.py's cpu_baseline leg
and __graft_entry__.smoke(); the product package never imports this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")

NEAREST, TRIANGLE, CATMULLROM, GAUSSIAN, LANCZOS3 = range(5)
MODE_SHRINK_BY, MODE_SHRINK_DIRECTIONALLY = 0, 1
DIST_OPAQUE, DIST_ALPHA, DIST_FLAT, DIST_NOISE = range(4)

u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
i32p = C.POINTER(C.c_int32)
i16p = C.POINTER(C.c_int16)
f32p = C.POINTER(C.c_float)
u64p = C.POINTER(C.c_uint64)


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h", ".inc", "Makefile"))]
    if (not force and os.path.exists(_LIB)
            and all(os.path.getmtime(_LIB) >= os.path.getmtime(s) for s in srcs)):
        return _LIB
    subprocess.run(["make", "-C", _HERE, "liboracle.so"], check=True, capture_output=True)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        L = C.CDLL(_LIB)
        L.orc_cbrtf.restype = C.c_float
        L.orc_cbrtf.argtypes = [C.c_float]
        L.orc_hypotf.restype = C.c_float
        L.orc_hypotf.argtypes = [C.c_float, C.c_float]
        L.orc_srgb_u8_to_linear.restype = C.c_float
        L.orc_srgb_u8_to_linear.argtypes = [C.c_uint8]
        L.orc_selftest_cbrtf.restype = C.c_uint64
        L.orc_selftest_cbrtf.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
        L.orc_selftest_hypotf.restype = C.c_uint64
        L.orc_selftest_hypotf.argtypes = [C.c_uint64, C.c_uint32]
        L.orc_grid.restype = None
        L.orc_grid.argtypes = [C.c_uint32] * 4 + [u32p, u32p]
        L.orc_tile_rect.restype = None
        L.orc_tile_rect.argtypes = [C.c_uint32] * 5 + [u32p] * 4
        L.orc_lod_directional.restype = None
        L.orc_lod_directional.argtypes = [C.c_void_p] + [C.c_uint32] * 4 + [f32p, f32p, u64p, u64p]
        L.orc_lod_oklab.restype = C.c_float
        L.orc_lod_oklab.argtypes = [C.c_void_p] + [C.c_uint32] * 4 + [C.c_float]
        L.orc_oklab_pixel.restype = None
        L.orc_oklab_pixel.argtypes = [C.c_void_p, C.c_uint32, f32p]
        L.orc_oklab_pixels.restype = None
        L.orc_oklab_pixels.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p]
        L.orc_reduce_dims.restype = None
        L.orc_reduce_dims.argtypes = [C.c_float, C.c_float, C.c_uint32, C.c_uint32, u32p, u32p, f32p]
        L.orc_resize.restype = C.c_int
        L.orc_resize.argtypes = [C.c_void_p] + [C.c_uint32] * 4 + [C.c_void_p] + [C.c_uint32] * 3
        L.orc_resize_imagers.restype = C.c_int
        L.orc_resize_imagers.argtypes = L.orc_resize.argtypes
        L.orc_fir_coeffs.restype = C.c_int
        L.orc_fir_coeffs.argtypes = [C.c_uint32] * 3 + [C.c_void_p, C.c_void_p, C.c_void_p, i32p, i32p]
        L.orc_shrink_image.restype = C.c_int
        L.orc_shrink_image.argtypes = ([C.c_void_p] + [C.c_uint32] * 8 + [C.c_float]
                                       + [C.c_void_p] * 4 + [C.c_int])
        L.orc_qoi_bound.restype = C.c_size_t
        L.orc_qoi_bound.argtypes = [C.c_uint32] * 3
        L.orc_qoi_encode.restype = C.c_size_t
        L.orc_qoi_encode.argtypes = [C.c_void_p] + [C.c_uint32] * 3 + [C.c_void_p]
        L.orc_qoi_decode.restype = C.c_int
        L.orc_qoi_decode.argtypes = [C.c_void_p, C.c_size_t, u32p, u32p, u32p, C.c_void_p, C.c_size_t]
        L.orc_encode_container.restype = C.c_size_t
        L.orc_encode_container.argtypes = ([C.c_uint32] * 6 + [C.c_void_p] * 5 + [C.c_void_p, C.c_size_t])
        L.orc_decode_container.restype = C.c_int
        L.orc_decode_container.argtypes = ([C.c_void_p, C.c_size_t] + [u32p] * 5 + [C.c_void_p] * 5
                                           + [C.c_size_t, C.c_uint32])
        L.orc_synth_frame.restype = None
        L.orc_synth_frame.argtypes = [C.c_void_p] + [C.c_uint32] * 6
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def grid(iw, ih, bw, bh):
    c, r = C.c_uint32(), C.c_uint32()
    lib().orc_grid(iw, ih, bw, bh, C.byref(c), C.byref(r))
    return c.value, r.value


def tile_rect(iw, ih, bw, bh, t):
    v = [C.c_uint32() for _ in range(4)]
    lib().orc_tile_rect(iw, ih, bw, bh, t, *[C.byref(x) for x in v])
    return tuple(x.value for x in v)


def lod_directional(tile):
    """tile: (h, w, c) uint8 view (may be strided along rows). -> (hz, vr, sum_hz, sum_vr)"""
    h, w, c = tile.shape
    assert tile.strides[2] == 1 and tile.strides[1] == c
    hz, vr = C.c_float(), C.c_float()
    sh, sv = C.c_uint64(), C.c_uint64()
    lib().orc_lod_directional(C.c_void_p(tile.ctypes.data), w, h, c, tile.strides[0],
                              C.byref(hz), C.byref(vr), C.byref(sh), C.byref(sv))
    return np.float32(hz.value), np.float32(vr.value), sh.value, sv.value


def oklab_pixels(px):
    """px: uint8 [n, c] -> float32 [n, 4] = (L, a, b, alpha) of every pixel (operations.rs:56-59)."""
    px = np.ascontiguousarray(px, np.uint8)
    n, c = px.shape
    out = np.empty((n, 4), np.float32)
    lib().orc_oklab_pixels(C.c_void_p(px.ctypes.data), c, n, C.c_void_p(out.ctypes.data))
    return out


def lod_oklab(tile, factor):
    h, w, c = tile.shape
    assert tile.strides[2] == 1 and tile.strides[1] == c
    return np.float32(lib().orc_lod_oklab(C.c_void_p(tile.ctypes.data), w, h, c, tile.strides[0],
                                          C.c_float(factor)))


def reduce_dims(v0, v1, w, h):
    nw, nh, st = C.c_uint32(), C.c_uint32(), C.c_float()
    lib().orc_reduce_dims(C.c_float(v0), C.c_float(v1), w, h, C.byref(nw), C.byref(nh), C.byref(st))
    return nw.value, nh.value, np.float32(st.value)


def resize(tile, nw, nh, filt, imagers=False):
    h, w, c = tile.shape
    assert tile.strides[2] == 1 and tile.strides[1] == c
    out = np.empty((nh, nw, c), np.uint8)
    fn = lib().orc_resize_imagers if imagers else lib().orc_resize
    rc = fn(C.c_void_p(tile.ctypes.data), w, h, c, tile.strides[0], _ptr(out), nw, nh, filt)
    if rc != 0:
        raise RuntimeError(f"orc_resize rc={rc}")
    return out


def fir_coeffs(in_size, out_size, filt):
    window, prec = C.c_int32(), C.c_int32()
    rc = lib().orc_fir_coeffs(in_size, out_size, filt, None, None, None, C.byref(window), C.byref(prec))
    if rc != 0:
        raise RuntimeError("orc_fir_coeffs")
    starts = np.zeros(out_size, np.int32)
    sizes = np.zeros(out_size, np.int32)
    k = np.zeros((out_size, window.value), np.int16)
    lib().orc_fir_coeffs(in_size, out_size, filt, _ptr(starts), _ptr(sizes), _ptr(k),
                         C.byref(window), C.byref(prec))
    return starts, sizes, k, prec.value


def shrink_image(img, bw, bh, mode, filt, factor, want_pixels=True, nthreads=1):
    """img: (H, W, C) uint8, C-contiguous rows. Returns (values f32[t], w u32[t], h u32[t], slots u8[t, bw*bh*c] | None)"""
    H, W, Cc = img.shape
    assert img.strides[2] == 1 and img.strides[1] == Cc
    cols, rows = grid(W, H, bw, bh)
    n = cols * rows
    vals = np.zeros(n, np.float32)
    ow = np.zeros(n, np.uint32)
    oh = np.zeros(n, np.uint32)
    slots = np.zeros((n, bw * bh * Cc), np.uint8) if want_pixels else None
    rc = lib().orc_shrink_image(C.c_void_p(img.ctypes.data), W, H, Cc, img.strides[0], bw, bh, mode, filt,
                                C.c_float(factor), _ptr(vals), _ptr(ow), _ptr(oh),
                                _ptr(slots) if want_pixels else None, nthreads)
    if rc != 0:
        raise RuntimeError(f"orc_shrink_image rc={rc}")
    return vals, ow, oh, slots


def qoi_encode(tile):
    h, w, c = tile.shape
    tile = np.ascontiguousarray(tile)
    out = np.empty(lib().orc_qoi_bound(w, h, c), np.uint8)
    n = lib().orc_qoi_encode(_ptr(tile), w, h, c, _ptr(out))
    return out[:n].tobytes()


def qoi_decode(data):
    buf = np.frombuffer(data, np.uint8)
    w, h, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
    cap = int.from_bytes(data[4:8], "big") * int.from_bytes(data[8:12], "big") * 4
    out = np.empty(cap, np.uint8)
    rc = lib().orc_qoi_decode(_ptr(buf), len(data), C.byref(w), C.byref(h), C.byref(c), _ptr(out), cap)
    if rc != 0:
        raise RuntimeError(f"orc_qoi_decode rc={rc}")
    return out[: w.value * h.value * c.value].reshape(h.value, w.value, c.value).copy()


def encode_container(width, height, bw, bh, channels, filter_byte, vals, has_value, tw, th, slots):
    vals = np.ascontiguousarray(vals, np.float32)
    tw = np.ascontiguousarray(tw, np.uint32)
    th = np.ascontiguousarray(th, np.uint32)
    slots = np.ascontiguousarray(slots, np.uint8)
    hv = None if has_value is None else np.ascontiguousarray(has_value, np.uint8)
    args = [width, height, bw, bh, channels, filter_byte, _ptr(vals), None if hv is None else _ptr(hv),
            _ptr(tw), _ptr(th), _ptr(slots)]
    bound = lib().orc_encode_container(*args, None, 0)
    out = np.empty(bound, np.uint8)
    n = lib().orc_encode_container(*args, _ptr(out), bound)
    if n == 0:
        raise RuntimeError("orc_encode_container failed")
    return out[:n].tobytes()


def decode_container(data):
    buf = np.frombuffer(data, np.uint8)
    width = int.from_bytes(data[10:14], "big")
    height = int.from_bytes(data[14:18], "big")
    bw = int.from_bytes(data[18:22], "big")
    bh = int.from_bytes(data[22:26], "big")
    cols = -(-width // bw)
    rows = -(-height // bh)
    n = cols * rows
    hdr = [C.c_uint32() for _ in range(5)]
    vals = np.zeros(n, np.float32)
    tw = np.zeros(n, np.uint32)
    th = np.zeros(n, np.uint32)
    tc = np.zeros(n, np.uint32)
    stride = bw * bh * 4
    slots = np.zeros((n, stride), np.uint8)
    rc = lib().orc_decode_container(_ptr(buf), len(data), *[C.byref(x) for x in hdr], _ptr(vals), _ptr(tw),
                                    _ptr(th), _ptr(tc), _ptr(slots), stride, n)
    if rc != 0:
        raise RuntimeError(f"orc_decode_container rc={rc}")
    return dict(width=hdr[0].value, height=hdr[1].value, bw=hdr[2].value, bh=hdr[3].value, filter=hdr[4].value,
                values=vals, tw=tw, th=th, tc=tc, slots=slots, cols=cols, rows=rows)


def expand_image(width, height, bw, bh, channels, filt, tile_w, tile_h, slots):
    """Pixlzr::expand + to_image: tiles (slots[t] holds tile_w[t]*tile_h[t]*channels tightly packed bytes) ->
    (height, width, channels) image."""
    L = lib()
    L.orc_expand_image.restype = C.c_int
    L.orc_expand_image.argtypes = [C.c_uint32] * 6 + [C.c_void_p] * 3 + [C.c_size_t, C.c_void_p, C.c_uint32]
    tile_w = np.ascontiguousarray(tile_w, np.uint32)
    tile_h = np.ascontiguousarray(tile_h, np.uint32)
    slots = np.ascontiguousarray(slots, np.uint8)
    out = np.zeros((height, width, channels), np.uint8)
    rc = L.orc_expand_image(width, height, bw, bh, channels, filt, _ptr(tile_w), _ptr(tile_h), _ptr(slots),
                            slots.shape[1], _ptr(out), width * channels)
    if rc != 0:
        raise RuntimeError(f"orc_expand_image rc={rc}")
    return out


def process_image(img, bw, bh, filter_down=4, filter_up=0):
    """process_custom (process/mod.rs:71-102) with |x - avg| and the identity: (H, W, C) -> (H, W, 4)."""
    L = lib()
    L.orc_process_image.restype = C.c_int
    L.orc_process_image.argtypes = [C.c_void_p] + [C.c_uint32] * 8 + [C.c_void_p, C.c_uint32]
    img = np.ascontiguousarray(img, np.uint8)
    H, W, Cc = img.shape
    out = np.zeros((H, W, 4), np.uint8)
    rc = L.orc_process_image(_ptr(img), W, H, Cc, W * Cc, bw, bh, filter_down, filter_up, _ptr(out), W * 4)
    if rc != 0:
        raise RuntimeError(f"orc_process_image rc={rc}")
    return out


def tree_process_image(img, bw, bh, threshold, min_bw=4, min_bh=4, filter_down=4, filter_up=0):
    """tree::process_custom with the closures of tree::process (tree.rs:23-109): (H, W, C) -> RGBA (H, W, 4)."""
    H, W, Cc = img.shape
    L = lib()
    L.orc_tree_process_image.restype = C.c_int
    L.orc_tree_process_image.argtypes = [C.c_void_p] + [C.c_uint32] * 8 + [C.c_float] + [C.c_uint32] * 2 + [C.c_void_p, C.c_uint32]
    img = np.ascontiguousarray(img)
    out = np.zeros((H, W, 4), np.uint8)
    rc = L.orc_tree_process_image(_ptr(img), W, H, Cc, W * Cc, bw, bh, min_bw, min_bh, C.c_float(threshold), filter_down,
                                  filter_up, _ptr(out), W * 4)
    if rc != 0:
        raise RuntimeError(f"orc_tree_process_image rc={rc}")
    return out


def synth_frame(width, height, channels=4, frame_index=0, dist=DIST_OPAQUE):
    img = np.empty((height, width, channels), np.uint8)
    lib().orc_synth_frame(_ptr(img), width, height, channels, width * channels, frame_index, dist)
    return img
