"""ctypes binding of csrc/libpixlzr_hip.so (C ABI: include/pixlzr_hip.h)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
_LIB_PATH = os.environ.get("PXZ_LIB") or os.path.join(_CSRC, "libpixlzr_hip.so")  # PXZ_LIB: diagnostic builds

FILTER_NEAREST, FILTER_TRIANGLE, FILTER_CATMULLROM, FILTER_GAUSSIAN, FILTER_LANCZOS3 = range(5)
MODE_SHRINK_BY, MODE_SHRINK_DIRECTIONALLY = 0, 1
DIST_OPAQUE, DIST_ALPHA, DIST_FLAT, DIST_NOISE = range(4)

# every symbol include/pixlzr_hip.h declares
EXPORTED_SYMBOLS = [
    "pxz_version", "pxz_device_count", "pxz_create", "pxz_destroy", "pxz_last_error", "pxz_set_stream",
    "pxz_synchronize", "pxz_grid", "pxz_shrink_image", "pxz_shrink_image_packed", "pxz_fetch_packed", "pxz_shrink_images", "pxz_shrink_images_packed", "pxz_shrink_frames_device", "pxz_lod_frames_device", "pxz_oklab_pixels_device",
    "pxz_pack_tiles_device", "pxz_encode_frames_device", "pxz_encode_container", "pxz_qoi_encode", "pxz_qoi_bound", "pxz_synth_frames_device", "pxz_axis_table",
    "pxz_enable_timing", "pxz_last_kernel_ms", "pxz_last_first_kernel_ms", "pxz_handle_state",
    "pxz_debug_read_work", "pxz_expand_frames_device", "pxz_expand_image", "pxz_decode_frames_device", "pxz_decode_file", "pxz_decode_status", "pxz_process_frames_device", "pxz_tree_process_frames_device", "pxz_trim", "pxz_debug_read_status",
]

STATUS = {0: "PXZ_OK", -1: "PXZ_ERR_INVALID_ARG", -2: "PXZ_ERR_NO_DEVICE", -3: "PXZ_ERR_HIP",
          -4: "PXZ_ERR_TILE_TOO_SMALL", -5: "PXZ_ERR_UNSUPPORTED", -6: "PXZ_ERR_NOMEM",
          -7: "PXZ_ERR_BUFFER_TOO_SMALL", -8: "PXZ_ERR_INTERNAL"}


class PxzError(RuntimeError):
    def __init__(self, code, text=""):
        self.code = code
        super().__init__(f"{STATUS.get(code, code)}: {text}" if text else STATUS.get(code, str(code)))


class Frames(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("channels", C.c_uint32),
                ("pitch_bytes", C.c_uint32), ("n_frames", C.c_uint32), ("reserved", C.c_uint32),
                ("frame_stride_bytes", C.c_uint64)]


class Params(C.Structure):
    _fields_ = [("block_w", C.c_uint32), ("block_h", C.c_uint32), ("mode", C.c_uint32),
                ("filter", C.c_uint32), ("factor", C.c_float), ("reserved", C.c_uint32)]


def library_path():
    return _LIB_PATH


def build_library(force=False):
    """hipcc --offload-arch=gfx950 (cross-compiles without a GPU)."""
    srcs = [os.path.join(_CSRC, f) for f in os.listdir(_CSRC)
            if f.endswith((".hip", ".cpp", ".h", ".inc")) or f == "Makefile"]
    srcs.append(os.path.join(os.path.dirname(_HERE), "include", "pixlzr_hip.h"))
    srcs.append(os.path.join(os.path.dirname(_HERE), "include", "pixlzr.hpp"))
    if (not force and os.path.exists(_LIB_PATH)
            and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in srcs)):
        return _LIB_PATH
    r = subprocess.run(["make", "-j8", "-C", _CSRC, "all"], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("building libpixlzr_hip.so failed:\n" + r.stdout + r.stderr)
    return _LIB_PATH


_lib = None


def load_library():
    """Loads the HIP library; raises if it is missing (no fallback of any kind)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise ImportError(f"{_LIB_PATH} is missing: run __graft_entry__.build() (hipcc --offload-arch=gfx950)")
    try:  # share torch's HIP runtime when torch is in the process (same SONAME, must come first)
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(_LIB_PATH)
    vp, u32, f32 = C.c_void_p, C.c_uint32, C.c_float
    L.pxz_version.restype = C.c_char_p
    L.pxz_device_count.restype = C.c_int
    L.pxz_create.restype = C.c_int
    L.pxz_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.pxz_destroy.restype = None
    L.pxz_destroy.argtypes = [vp]
    L.pxz_last_error.restype = C.c_char_p
    L.pxz_last_error.argtypes = [vp]
    L.pxz_set_stream.restype = C.c_int
    L.pxz_set_stream.argtypes = [vp, vp]
    L.pxz_synchronize.restype = C.c_int
    L.pxz_synchronize.argtypes = [vp]
    L.pxz_trim.restype = C.c_int
    L.pxz_trim.argtypes = [vp]
    L.pxz_grid.restype = C.c_int
    L.pxz_grid.argtypes = [u32] * 4 + [C.POINTER(u32)] * 2
    L.pxz_shrink_image.restype = C.c_int
    L.pxz_shrink_image.argtypes = [vp, vp] + [u32] * 8 + [f32] + [vp] * 4
    L.pxz_shrink_image_packed.restype = C.c_int
    L.pxz_shrink_image_packed.argtypes = [vp, vp] + [u32] * 8 + [f32] + [vp] * 4
    L.pxz_shrink_images.restype = C.c_int
    L.pxz_shrink_images.argtypes = [vp, vp] + [u32] * 9 + [f32] + [vp] * 4
    L.pxz_shrink_images_packed.restype = C.c_int
    L.pxz_shrink_images_packed.argtypes = [vp, vp] + [u32] * 9 + [f32] + [vp] * 4 + [C.c_uint64, vp]
    L.pxz_fetch_packed.restype = C.c_int
    L.pxz_fetch_packed.argtypes = [vp, vp, C.c_uint64]
    L.pxz_shrink_frames_device.restype = C.c_int
    L.pxz_shrink_frames_device.argtypes = [vp, C.POINTER(Frames), C.POINTER(Params)] + [vp] * 5
    L.pxz_lod_frames_device.restype = C.c_int
    L.pxz_lod_frames_device.argtypes = [vp, C.POINTER(Frames), C.POINTER(Params)] + [vp] * 3
    L.pxz_oklab_pixels_device.restype = C.c_int
    L.pxz_oklab_pixels_device.argtypes = [vp, vp, u32, vp]
    L.pxz_pack_tiles_device.restype = C.c_int
    L.pxz_pack_tiles_device.argtypes = [vp, u32, u32, u32, vp, vp, vp, vp, vp, C.c_uint64]
    L.pxz_encode_frames_device.restype = C.c_int
    L.pxz_encode_frames_device.argtypes = [vp, C.POINTER(Frames), C.POINTER(Params), u32, vp, vp, vp, vp, vp, C.c_uint64, vp]
    L.pxz_encode_container.restype = C.c_int64
    L.pxz_encode_container.argtypes = [u32] * 6 + [vp] * 5 + [vp, C.c_size_t]
    L.pxz_qoi_encode.restype = C.c_int64
    L.pxz_qoi_encode.argtypes = [vp, u32, u32, u32, vp, C.c_size_t]
    L.pxz_qoi_bound.restype = C.c_size_t
    L.pxz_qoi_bound.argtypes = [u32] * 3
    L.pxz_expand_frames_device.restype = C.c_int
    L.pxz_expand_frames_device.argtypes = [vp, C.POINTER(Frames), C.POINTER(Params)] + [vp] * 4
    L.pxz_decode_file.restype = C.c_int
    L.pxz_decode_file.argtypes = [vp, vp, C.c_size_t] + [C.POINTER(u32)] * 6 + [vp] * 4
    L.pxz_process_frames_device.restype = C.c_int
    L.pxz_process_frames_device.argtypes = [vp, C.POINTER(Frames), C.POINTER(Params), u32, vp, vp, u32, C.c_uint64]
    L.pxz_tree_process_frames_device.restype = C.c_int
    L.pxz_tree_process_frames_device.argtypes = [vp, C.POINTER(Frames), C.POINTER(Params), u32, f32, u32, u32, vp, vp, u32, C.c_uint64]
    L.pxz_decode_status.restype = C.c_int
    L.pxz_decode_status.argtypes = [vp, C.POINTER(u32)]
    L.pxz_decode_frames_device.restype = C.c_int
    L.pxz_decode_frames_device.argtypes = [vp, C.POINTER(Frames), C.POINTER(Params)] + [vp] * 6
    L.pxz_expand_image.restype = C.c_int
    L.pxz_expand_image.argtypes = [vp] + [u32] * 7 + [vp] * 4
    L.pxz_synth_frames_device.restype = C.c_int
    L.pxz_synth_frames_device.argtypes = [vp, C.POINTER(Frames), vp, u32, u32]
    L.pxz_axis_table.restype = C.c_int
    L.pxz_axis_table.argtypes = [u32] * 3 + [vp] * 3 + [C.POINTER(C.c_int32)] * 2
    L.pxz_enable_timing.restype = C.c_int
    L.pxz_enable_timing.argtypes = [vp, C.c_int]
    L.pxz_last_first_kernel_ms.restype = C.c_int
    L.pxz_last_first_kernel_ms.argtypes = [vp, C.POINTER(f32)]
    L.pxz_last_kernel_ms.restype = C.c_int
    L.pxz_last_kernel_ms.argtypes = [vp, C.POINTER(f32)]
    if hasattr(L, "pxz_handle_state"):  # (an older diagnostic build named by PXZ_LIB may lack it)
        L.pxz_handle_state.restype = C.c_int
        L.pxz_handle_state.argtypes = [vp, C.POINTER(C.c_uint32)]
    _lib = L
    return L


def _p(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


def grid(width, height, bw, bh):
    c, r = C.c_uint32(), C.c_uint32()
    rc = load_library().pxz_grid(width, height, bw, bh, C.byref(c), C.byref(r))
    if rc != 0:
        raise PxzError(rc)
    return c.value, r.value


def qoi_encode(tile):
    tile = np.ascontiguousarray(tile, np.uint8)
    h, w, c = tile.shape
    L = load_library()
    out = np.empty(L.pxz_qoi_bound(w, h, c), np.uint8)
    n = L.pxz_qoi_encode(_p(tile), w, h, c, _p(out), out.size)
    if n < 0:
        raise PxzError(int(n))
    return out[:n].tobytes()


def encode_container(width, height, bw, bh, channels, filter_byte, values, has_value, tw, th, slots):
    """Pixlzr::encode_to_vec: tiles (slots + dims + values) -> .pixlzr bytes."""
    L = load_library()
    values = np.ascontiguousarray(values, np.float32)
    tw = np.ascontiguousarray(tw, np.uint32)
    th = np.ascontiguousarray(th, np.uint32)
    slots = np.ascontiguousarray(slots, np.uint8)
    hv = None if has_value is None else np.ascontiguousarray(has_value, np.uint8)
    head = [width, height, bw, bh, channels, filter_byte, _p(values), _p(hv), _p(tw), _p(th), _p(slots)]
    bound = L.pxz_encode_container(*head, None, 0)
    if bound < 0:
        raise PxzError(int(bound))
    out = np.empty(bound, np.uint8)
    n = L.pxz_encode_container(*head, _p(out), out.size)
    if n < 0:
        raise PxzError(int(n))
    return out[:n].tobytes()


def axis_table(in_size, out_size, filt):
    L = load_library()
    window, prec = C.c_int32(), C.c_int32()
    rc = L.pxz_axis_table(in_size, out_size, filt, None, None, None, C.byref(window), C.byref(prec))
    if rc != 0:
        raise PxzError(rc)
    starts = np.zeros(out_size, np.int32)
    sizes = np.zeros(out_size, np.int32)
    k = np.zeros((out_size, max(window.value, 1)), np.int16)
    L.pxz_axis_table(in_size, out_size, filt, _p(starts), _p(sizes), _p(k), C.byref(window), C.byref(prec))
    return starts, sizes, k, prec.value


class Handle:
    """One GPU.  Device entry points take/return torch CUDA tensors (plumbing only)."""

    def __init__(self, device_id=0):
        self._L = load_library()
        self._h = C.c_void_p()
        rc = self._L.pxz_create(device_id, C.byref(self._h))
        if rc != 0:
            raise PxzError(rc, "pxz_create: no usable gfx950 device (there is no CPU fallback)")
        self.device_id = device_id

    def close(self):
        if self._h:
            self._L.pxz_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise PxzError(rc, (self._L.pxz_last_error(self._h) or b"").decode())

    def set_stream(self, stream_ptr):
        self._check(self._L.pxz_set_stream(self._h, C.c_void_p(stream_ptr)))

    def use_torch_stream(self):
        import torch
        self.set_stream(torch.cuda.current_stream(self.device_id).cuda_stream)

    def trim(self):
        """pxz_trim: the handle's scratch buffers go back to the device (the next call grows them again)"""
        self._check(self._L.pxz_trim(self._h))

    def synchronize(self):
        self._check(self._L.pxz_synchronize(self._h))

    def enable_timing(self, on=True, every=1):
        """every = n > 1: only every n-th step is bracketed by events."""
        self._check(self._L.pxz_enable_timing(self._h, (max(int(every), 1) if on else 0)))

    def last_first_kernel_ms(self):
        ms = C.c_float()
        self._check(self._L.pxz_last_first_kernel_ms(self._h, C.byref(ms)))
        return ms.value

    def state(self):
        """which kernels the fast paths pick (pxz_handle_state): a timing is comparable only with one taken in the same state"""
        st = (C.c_uint32 * 4)()
        if not hasattr(self._L, "pxz_handle_state"):
            return None
        self._check(self._L.pxz_handle_state(self._h, st))
        return {"transparent_tiles_seen_by_last_finished_launch": int(st[0]),
                "tiles_listed_by_last_finished_launch": None if st[1] == 0xffffffff else int(st[1]),
                "alpha_kernel": bool(st[2]), "alpha_first": bool(st[3])}

    def last_kernel_ms(self):
        ms = C.c_float()
        self._check(self._L.pxz_last_kernel_ms(self._h, C.byref(ms)))
        return ms.value

    # ---- host-buffer entry point (Pixlzr::from_image + shrink_*) ----
    def shrink_image(self, img, bw, bh, mode, filt, factor, want_pixels=True):
        H, W, Cc = img.shape
        assert img.dtype == np.uint8 and img.strides[2] == 1 and img.strides[1] == Cc
        cols, rows = grid(W, H, bw, bh)
        n = cols * rows
        vals = np.zeros(n, np.float32)
        ow = np.zeros(n, np.uint32)
        oh = np.zeros(n, np.uint32)
        slots = np.zeros((n, bw * bh * Cc), np.uint8) if want_pixels else None
        self._check(self._L.pxz_shrink_image(self._h, C.c_void_p(img.ctypes.data), W, H, Cc, img.strides[0], bw, bh,
                                             mode, filt, C.c_float(factor), _p(vals), _p(ow), _p(oh), _p(slots)))
        return vals, ow, oh, slots

    def shrink_image_packed(self, img, bw, bh, mode, filt, factor):
        """pxz_shrink_image_packed + pxz_fetch_packed: (values, w, h, stream) with the tiles' pixels back to back."""
        H, W, Cc = img.shape
        assert img.dtype == np.uint8 and img.strides[2] == 1 and img.strides[1] == Cc
        cols, rows = grid(W, H, bw, bh)
        n = cols * rows
        vals = np.zeros(n, np.float32)
        ow = np.zeros(n, np.uint32)
        oh = np.zeros(n, np.uint32)
        total = C.c_uint64(0)
        self._check(self._L.pxz_shrink_image_packed(self._h, C.c_void_p(img.ctypes.data), W, H, Cc, img.strides[0], bw, bh,
                                                    mode, filt, C.c_float(factor), _p(vals), _p(ow), _p(oh), C.byref(total)))
        stream = np.empty(total.value, np.uint8)
        self._check(self._L.pxz_fetch_packed(self._h, _p(stream), total.value))
        return vals, ow, oh, stream

    def shrink_images(self, imgs, bw, bh, mode, filt, factor, want_pixels=True, packed=False):
        """pxz_shrink_images[_packed]: a list of equally sized (H, W, C) uint8 images through the pipelined host boundary.
        Returns a list of (values, w, h, slots | stream | None) per image."""
        n = len(imgs)
        H, W, Cc = imgs[0].shape
        for im in imgs:
            assert im.shape == (H, W, Cc) and im.dtype == np.uint8 and im.strides == imgs[0].strides and im.strides[2] == 1 and im.strides[1] == Cc
        cols, rows = grid(W, H, bw, bh)
        T = cols * rows
        vals = [np.zeros(T, np.float32) for _ in range(n)]
        ow = [np.zeros(T, np.uint32) for _ in range(n)]
        oh = [np.zeros(T, np.uint32) for _ in range(n)]
        ptrs = lambda arrs: (C.c_void_p * n)(*[a.ctypes.data for a in arrs])
        src = ptrs(imgs)
        if packed:
            cap = H * W * Cc
            px = [np.empty(cap, np.uint8) for _ in range(n)]
            lens = np.zeros(n, np.uint64)
            self._check(self._L.pxz_shrink_images_packed(self._h, src, n, W, H, Cc, imgs[0].strides[0], bw, bh, mode, filt, C.c_float(factor),
                                                         ptrs(vals), ptrs(ow), ptrs(oh), ptrs(px), cap, _p(lens)))
            return [(vals[k], ow[k], oh[k], px[k][: int(lens[k])]) for k in range(n)]
        px = [np.zeros((T, bw * bh * Cc), np.uint8) for _ in range(n)] if want_pixels else None
        self._check(self._L.pxz_shrink_images(self._h, src, n, W, H, Cc, imgs[0].strides[0], bw, bh, mode, filt, C.c_float(factor),
                                              ptrs(vals), ptrs(ow), ptrs(oh), ptrs(px) if px else None))
        return [(vals[k], ow[k], oh[k], px[k] if px else None) for k in range(n)]

    def oklab_pixels_device(self, rgba):
        """rgba: uint8 CUDA tensor [n, 4] -> float32 [n, 4] = (l, a, b, alpha) per pixel."""
        import torch
        assert rgba.is_cuda and rgba.dtype == torch.uint8 and rgba.dim() == 2 and rgba.shape[1] == 4 and rgba.is_contiguous()
        out = torch.empty((rgba.shape[0], 4), dtype=torch.float32, device=rgba.device)
        self.use_torch_stream()
        self._check(self._L.pxz_oklab_pixels_device(self._h, C.c_void_p(rgba.data_ptr()), rgba.shape[0], C.c_void_p(out.data_ptr())))
        return out

    # ---- device entry points ----
    @staticmethod
    def _frames_desc(frames):
        assert frames.is_cuda and frames.dim() == 4 and frames.dtype.is_floating_point is False
        N, H, W, Cc = frames.shape
        assert frames.stride(3) == 1 and frames.stride(2) == Cc
        return Frames(W, H, Cc, frames.stride(1), N, 0, frames.stride(0)), (N, H, W, Cc)

    def shrink_frames_device(self, frames, bw, bh, mode, filt, factor, want_pixels=True, out=None, transparency_hint=False):
        """frames: uint8 CUDA tensor [N,H,W,C].  Returns (values[N,T], w[N,T], h[N,T], slots[N,T,bw*bh*C]|None).
        transparency_hint: PXZ_HINT_TRANSPARENCY (many tiles with alpha < 255; a pure performance hint)."""
        import torch
        fd, (N, H, W, Cc) = self._frames_desc(frames)
        cols, rows = grid(W, H, bw, bh)
        T = cols * rows
        dev = frames.device
        if out is None:
            vals = torch.empty((N, T), dtype=torch.float32, device=dev)
            ow = torch.empty((N, T), dtype=torch.int32, device=dev)
            oh = torch.empty((N, T), dtype=torch.int32, device=dev)
            slots = torch.empty((N, T, bw * bh * Cc), dtype=torch.uint8, device=dev) if want_pixels else None
        else:
            vals, ow, oh, slots = out
        pd = Params(bw, bh, mode, filt, factor, 1 if transparency_hint else 0)
        self.use_torch_stream()
        self._check(self._L.pxz_shrink_frames_device(
            self._h, C.byref(fd), C.byref(pd), C.c_void_p(frames.data_ptr()), C.c_void_p(vals.data_ptr()),
            C.c_void_p(ow.data_ptr()), C.c_void_p(oh.data_ptr()),
            C.c_void_p(slots.data_ptr()) if slots is not None else None))
        return vals, ow, oh, slots

    # ---- decode side: Pixlzr::expand + to_image ----
    def expand_image(self, width, height, channels, bw, bh, filt, tile_w, tile_h, slots):
        """Host buffers: stored tiles (slots[t] holds tile_w[t]*tile_h[t]*channels tightly packed bytes in a slot of
        bw*bh*channels) -> (height, width, channels) image."""
        tile_w = np.ascontiguousarray(tile_w, np.uint32)
        tile_h = np.ascontiguousarray(tile_h, np.uint32)
        slots = np.ascontiguousarray(slots, np.uint8)
        assert slots.shape[1] == bw * bh * channels
        out = np.zeros((height, width, channels), np.uint8)
        self._check(self._L.pxz_expand_image(self._h, width, height, channels, width * channels, bw, bh, filt,
                                             _p(tile_w), _p(tile_h), _p(slots), _p(out)))
        return out

    def expand_frames_device(self, shape, bw, bh, filt, ow, oh, slots, out=None):
        """Device tensors as left by shrink_frames_device (w[N,T], h[N,T], slots[N,T,bw*bh*C]) -> frames [N,H,W,C]."""
        import torch
        N, H, W, Cc = shape
        if out is None:
            out = torch.empty((N, H, W, Cc), dtype=torch.uint8, device=slots.device)
        fd, _ = self._frames_desc(out)
        pd = Params(bw, bh, 0, filt, 0.0, 0)
        self.use_torch_stream()
        self._check(self._L.pxz_expand_frames_device(self._h, C.byref(fd), C.byref(pd), C.c_void_p(ow.data_ptr()),
                                                     C.c_void_p(oh.data_ptr()), C.c_void_p(slots.data_ptr()),
                                                     C.c_void_p(out.data_ptr())))
        return out

    def process_frames_device(self, frames, bw, bh, filter_down=4, filter_up=0):
        """process_custom with |x - avg| / identity (process/mod.rs:71-121): frames [N,H,W,C] -> RGBA [N,H,W,4]."""
        import torch
        fd, (N, H, W, Cc) = self._frames_desc(frames)
        out = torch.empty((N, H, W, 4), dtype=torch.uint8, device=frames.device)
        pd = Params(bw, bh, 0, filter_down, 1.0, 0)
        self.use_torch_stream()
        self._check(self._L.pxz_process_frames_device(self._h, C.byref(fd), C.byref(pd), filter_up,
                                                      C.c_void_p(frames.data_ptr()), C.c_void_p(out.data_ptr()),
                                                      W * 4, W * 4 * H))
        return out

    def tree_process_frames_device(self, frames, bw, bh, threshold, min_bw=4, min_bh=4, filter_down=4, filter_up=0):
        """tree::process_custom with |x - avg| / identity (process/tree.rs:23-109): frames [N,H,W,C] -> RGBA [N,H,W,4]."""
        import torch
        fd, (N, H, W, Cc) = self._frames_desc(frames)
        out = torch.empty((N, H, W, 4), dtype=torch.uint8, device=frames.device)
        pd = Params(bw, bh, 0, filter_down, 1.0, 0)
        self.use_torch_stream()
        self._check(self._L.pxz_tree_process_frames_device(self._h, C.byref(fd), C.byref(pd), filter_up, C.c_float(threshold),
                                                           min_bw, min_bh, C.c_void_p(frames.data_ptr()),
                                                           C.c_void_p(out.data_ptr()), W * 4, W * 4 * H))
        return out

    def decode_status(self):
        """bit 0: invalid stored tile size seen by expand; bit 1: malformed file/record seen by decode."""
        flags = C.c_uint32(0)
        self._check(self._L.pxz_decode_status(self._h, C.byref(flags)))
        return flags.value

    def decode_frames_device(self, files, file_offsets, shape, bw, bh, out=None):
        """files: uint8 CUDA tensor holding N .pixlzr files back to back, file_offsets int64[N+1] (CUDA).
        Returns (values[N,T], w[N,T], h[N,T], slots[N,T,bw*bh*C]) for frames of `shape` = (N,H,W,C)."""
        import torch
        N, H, W, Cc = shape
        cols, rows = grid(W, H, bw, bh)
        T = cols * rows
        dev = files.device
        if out is None:
            vals = torch.zeros((N, T), dtype=torch.float32, device=dev)
            ow = torch.zeros((N, T), dtype=torch.int32, device=dev)
            oh = torch.zeros((N, T), dtype=torch.int32, device=dev)
            slots = torch.zeros((N, T, bw * bh * Cc), dtype=torch.uint8, device=dev)
        else:
            vals, ow, oh, slots = out
        fd = Frames(W, H, Cc, W * Cc, N, 0, W * Cc * H)
        pd = Params(bw, bh, 0, 0, 0.0, 0)
        self.use_torch_stream()
        self._check(self._L.pxz_decode_frames_device(self._h, C.byref(fd), C.byref(pd), C.c_void_p(files.data_ptr()),
                                                     C.c_void_p(file_offsets.data_ptr()), C.c_void_p(vals.data_ptr()),
                                                     C.c_void_p(ow.data_ptr()), C.c_void_p(oh.data_ptr()),
                                                     C.c_void_p(slots.data_ptr())))
        return vals, ow, oh, slots

    def lod_frames_device(self, frames, bw, bh, mode, factor=1.0):
        import torch
        fd, (N, H, W, Cc) = self._frames_desc(frames)
        cols, rows = grid(W, H, bw, bh)
        T = cols * rows
        l0 = torch.empty((N, T), dtype=torch.float32, device=frames.device)
        l1 = torch.empty((N, T), dtype=torch.float32, device=frames.device)
        pd = Params(bw, bh, mode, FILTER_NEAREST, factor, 0)
        self.use_torch_stream()
        self._check(self._L.pxz_lod_frames_device(self._h, C.byref(fd), C.byref(pd), C.c_void_p(frames.data_ptr()),
                                                  C.c_void_p(l0.data_ptr()), C.c_void_p(l1.data_ptr())))
        return l0, l1

    def pack_tiles_device(self, ow, oh, slots, channels, out=None):
        """Compacts the valid bytes of the slots (any leading batch dims) into one stream.
        Returns (offsets int64[n_tiles+1], packed uint8[capacity]); offsets[-1] is the stream length."""
        import torch
        n = ow.numel()
        slot_bytes = slots.shape[-1]
        if out is None:
            offsets = torch.empty(n + 1, dtype=torch.int64, device=ow.device)
            packed = torch.empty(n * slot_bytes, dtype=torch.uint8, device=ow.device)
        else:
            offsets, packed = out
        self.use_torch_stream()
        self._check(self._L.pxz_pack_tiles_device(
            self._h, n, channels, slot_bytes, C.c_void_p(ow.data_ptr()), C.c_void_p(oh.data_ptr()),
            C.c_void_p(slots.data_ptr()), C.c_void_p(offsets.data_ptr()), C.c_void_p(packed.data_ptr()),
            packed.numel()))
        return offsets, packed

    def encode_frames_device(self, shape, bw, bh, vals, ow, oh, slots, filter_byte=0, out=None):
        """GPU bitstream: tiles of a batch -> the .pixlzr files, back to back.  shape = (N, H, W, C) of the frames.
        Returns (file_offsets int64[N+1], bytes uint8[capacity])."""
        import torch
        N, H, W, Cc = shape
        fd = Frames(W, H, Cc, W * Cc, N, 0, W * Cc * H)
        pd = Params(bw, bh, 0, 0, 1.0, 0)
        if out is None:
            cols, rows = grid(W, H, bw, bh)
            cap = N * (26 + rows * 4) + N * cols * rows * (13 + 10 + bw * bh * (Cc + 1) + 8)
            offs = torch.empty(N + 1, dtype=torch.int64, device=vals.device)
            buf = torch.empty(cap, dtype=torch.uint8, device=vals.device)
        else:
            offs, buf = out
        self.use_torch_stream()
        self._check(self._L.pxz_encode_frames_device(
            self._h, C.byref(fd), C.byref(pd), filter_byte, C.c_void_p(vals.data_ptr()), C.c_void_p(ow.data_ptr()),
            C.c_void_p(oh.data_ptr()), C.c_void_p(slots.data_ptr()), C.c_void_p(buf.data_ptr()), buf.numel(),
            C.c_void_p(offs.data_ptr())))
        return offs, buf

    def synth_frames_device(self, n_frames, height, width, channels=4, first_frame=0, dist=DIST_OPAQUE, out=None):
        import torch
        if out is None:
            out = torch.empty((n_frames, height, width, channels), dtype=torch.uint8,
                              device=torch.device("cuda", self.device_id))
        fd, _ = self._frames_desc(out)
        self.use_torch_stream()
        self._check(self._L.pxz_synth_frames_device(self._h, C.byref(fd), C.c_void_p(out.data_ptr()), first_frame, dist))
        return out
