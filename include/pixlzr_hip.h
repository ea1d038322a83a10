/*
 * pixlzr_hip.h — C ABI of libpixlzr_hip.so, the MI355X (gfx950) implementation of
 * the pixlzr encode hot path: per-tile level-of-detail detection + power-of-two
 * down-sampling over the regular tile grid, and the .pixlzr bitstream writer.
 *
 * The reference (guiga-zalu/pixlzr-rust 0.3.1) has no FFI of its own; these are
 * the entry points a Rust `extern "C"` block inside `Pixlzr::shrink_by` /
 * `Pixlzr::shrink_directionally` / `Pixlzr::encode_to_vec` would bind (see
 * INTEGRATION.md for that binding).  Each declaration cites the reference
 * interface it replaces (paths relative to the reference repo).
 *
 * Conventions
 *  - every function returns PXZ_OK (0) or a negative pxz_status; nothing aborts.
 *    (The reference panics on the same conditions: unwrap() inside the path.)
 *  - plain pointers and sizes only; "device" pointers are HIP device addresses
 *    on the handle's GPU.  *_device entry points are asynchronous on the
 *    handle's stream (pxz_set_stream); host-buffer entry points synchronise.
 *  - a handle is bound to one GPU and may be used by one thread at a time;
 *    distinct handles are independent (mirrors: concurrent calls on different
 *    `Pixlzr` objects are legal, src/data_types/pixlzr.rs:17-25).
 *  - tile order is the reference's: row-major, tile = ty*cols + tx
 *    (src/data_types/iter.rs:64-76); grid = ceil(w/bw) x ceil(h/bh).
 *  - there is no CPU fallback: without a gfx950 device pxz_create fails.
 */
#ifndef PIXLZR_HIP_H
#define PIXLZR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pxz_handle pxz_handle;

typedef enum pxz_status {
	PXZ_OK = 0,
	PXZ_ERR_INVALID_ARG = -1,   /* null pointer, channels not 3|4, zero sizes, non-finite factor */
	PXZ_ERR_NO_DEVICE = -2,     /* no HIP device / not gfx950 */
	PXZ_ERR_HIP = -3,           /* a HIP runtime call failed; see pxz_last_error */
	PXZ_ERR_TILE_TOO_SMALL = -4,/* directional mode with a tile narrower/lower than 2 px:
	                               the reference underflows usize and panics (operations.rs:220-221) */
	PXZ_ERR_UNSUPPORTED = -5,   /* a tile of 2^20 pixels or more (smaller tiles whose image exceeds the 160 KB of LDS run from an
	                             * HBM-resident image, slowly), an image side above 2^24, tree blocks above 128 px */
	PXZ_ERR_NOMEM = -6,
	PXZ_ERR_BUFFER_TOO_SMALL = -7,
	PXZ_ERR_INTERNAL = -8       /* a device-side consistency check failed (never expected; see pxz_last_error) */
} pxz_status;

/* FilterType, repr(u8): src/data_types/mod.rs:10-30 */
typedef enum pxz_filter {
	PXZ_FILTER_NEAREST = 0,
	PXZ_FILTER_TRIANGLE = 1,    /* down-scales with fir Hamming (data_types/mod.rs:93-95) */
	PXZ_FILTER_CATMULLROM = 2,
	PXZ_FILTER_GAUSSIAN = 3,
	PXZ_FILTER_LANCZOS3 = 4
} pxz_filter;

/* which caller of the hot path is replaced */
typedef enum pxz_mode {
	PXZ_MODE_SHRINK_BY = 0,            /* Pixlzr::shrink_by, pixlzr.rs:155-185 (Oklab MAD, isotropic) */
	PXZ_MODE_SHRINK_DIRECTIONALLY = 1  /* Pixlzr::shrink_directionally, pixlzr.rs:187-205 */
} pxz_mode;

/* ---- library / handle ------------------------------------------------- */
const char *pxz_version(void);
/* number of usable gfx950 devices (0 if none / no HIP runtime) */
int pxz_device_count(void);
int pxz_create(int device_id, pxz_handle **out);
void pxz_destroy(pxz_handle *h);
/* text of the last error on this handle ("" if none) */
const char *pxz_last_error(const pxz_handle *h);
/* run device entry points on the caller's hipStream_t (NULL = default stream) */
int pxz_set_stream(pxz_handle *h, void *hip_stream);
int pxz_synchronize(pxz_handle *h);

/* ---- geometry --------------------------------------------------------- */
/* ImageBlockIterator::new grid math, src/data_types/iter.rs:38-41 / src/split.rs:45-46 */
int pxz_grid(uint32_t width, uint32_t height, uint32_t block_w, uint32_t block_h,
             uint32_t *cols, uint32_t *rows);

/* A batch of equally sized, pitch-linear, interleaved 8-bit frames
 * (what `image::DynamicImage::ImageRgba8|ImageRgb8` holds, src/split.rs:10-27). */
typedef struct pxz_frames {
	uint32_t width, height;
	uint32_t channels;          /* 3 (RGB8) or 4 (RGBA8) */
	uint32_t pitch_bytes;       /* >= width*channels */
	uint32_t n_frames;          /* >= 1 */
	uint32_t reserved;          /* 0 */
	uint64_t frame_stride_bytes;/* distance between frames (ignored when n_frames==1) */
} pxz_frames;

typedef struct pxz_params {
	uint32_t block_w, block_h;  /* CLI -b / --block-height, src/bin/main.rs:19-24 */
	uint32_t mode;              /* pxz_mode */
	uint32_t filter;            /* pxz_filter (filter_downscale) */
	float factor;               /* shrinking factor, src/bin/main.rs:26-29 */
	uint32_t reserved;          /* 0, or PXZ_HINT_* bits */
} pxz_params;

/* Performance hints (pxz_params.reserved); results never depend on them.
 * PXZ_HINT_TRANSPARENCY: many tiles of these RGBA frames carry alpha < 255.  32x32 tiles with transparency are
 * then resampled by a dedicated kernel (four LDS planes, premultiplied matrix-core convolution) instead of the
 * generic one, at the price of one more launch per call.  pxz_shrink_image samples the image and sets it itself;
 * without the hint a handle switches to that kernel by itself once a finished launch has reported >= 2048 such
 * tiles (and back when a launch reports fewer). */
#define PXZ_HINT_TRANSPARENCY 1u

/* ---- the hot path ----------------------------------------------------- */

/* Pixlzr::from_image (pixlzr_image.rs:6-22) + shrink_by | shrink_directionally
 * (pixlzr.rs:155-205) for ALL tiles of one host-resident image:
 * get_block_variance[_directionally] (operations.rs:26-126,192-259) ->
 * reduce_image_section (operations.rs:140-156) -> PixlzrBlock::resize
 * (block.rs:273-334, fast_image_resize convolution / nearest).
 * Outputs (caller-allocated, tile order):
 *   block_value[t]  Some(value) of the shrunk tile = hypot(v0,v1) (operations.rs:154)
 *   out_w/out_h[t]  reduced tile dimensions
 *   out_pixels      fixed slots of block_w*block_h*channels bytes per tile, of
 *                   which out_w*out_h*channels are valid (tightly packed rows);
 *                   the rest of a slot is unspecified and may be written (shrink_by
 *                   leaves the tile's own pixels there on its way).
 *                   May be NULL: LOD + dimensions only. */
int pxz_shrink_image(pxz_handle *h, const uint8_t *pixels, uint32_t width, uint32_t height,
                     uint32_t channels, uint32_t pitch_bytes, uint32_t block_w, uint32_t block_h,
                     uint32_t mode, uint32_t filter, float factor,
                     float *block_value, uint32_t *out_w, uint32_t *out_h, uint8_t *out_pixels);

/* The same call with the pixels as ONE tightly packed stream (tile order, each tile out_w*out_h*channels
 * bytes) instead of fixed slots: what the reference keeps per block (PixlzrBlock's pixel Vec, block.rs) and
 * a fraction of the slot array over PCIe.  Two steps so that the caller allocates exactly what comes back:
 * pxz_shrink_image_packed runs the path, returns values and dimensions and *packed_len; the stream stays in
 * the handle until pxz_fetch_packed copies it out (capacity >= packed_len) or the next call on the handle.
 * Tile t starts at the sum of out_w*out_h*channels over the tiles before it. */
int pxz_shrink_image_packed(pxz_handle *h, const uint8_t *pixels, uint32_t width, uint32_t height,
                            uint32_t channels, uint32_t pitch_bytes, uint32_t block_w, uint32_t block_h,
                            uint32_t mode, uint32_t filter, float factor,
                            float *block_value, uint32_t *out_w, uint32_t *out_h, uint64_t *packed_len);
int pxz_fetch_packed(pxz_handle *h, uint8_t *dst, uint64_t capacity);

/* Releases every scratch buffer the handle has grown (inputs, tile slots, worklists, the rings of the pipelined list
 * calls, the writer's scratch ...) after finishing what is queued on its stream; tables stay.  The next call allocates
 * what it needs again.  Nothing in the reference corresponds to it (its Vecs are freed when a Pixlzr is dropped). */
int pxz_trim(pxz_handle *h);

/* pxz_shrink_image / pxz_shrink_image_packed over a LIST of equally sized host images (a folder of frames, what
 * src/bin/whole-folder.rs:69-117 loops over): outputs as for the single-image calls, one set of caller-allocated arrays
 * per image (out_pixels may be NULL, or hold NULLs, for detector + dimensions only; packed[k] needs packed_capacity
 * bytes -- width*height*channels always suffices -- and packed_len[k] receives the stream's length).  The images go
 * through a three-stage pipeline on the device -- upload of image k+1, kernels of image k, download of image k-1 at the
 * same time -- so that a list costs about one PCIe direction per image rather than the sum of both.  Synchronous.
 * On an error the arrays of the images that were finished before it are complete, those of later images are untouched
 * or partly written, and packed_len[k] is written for every image that was downloaded even when the call then returns
 * PXZ_ERR_BUFFER_TOO_SMALL (it names the capacity that would have sufficed).  The three sets of device buffers of
 * the pipeline stay with the handle for the next list (about 1.2 GB for 8K RGBA): pxz_trim releases them. */
int pxz_shrink_images(pxz_handle *h, const uint8_t *const *pixels, uint32_t n_images, uint32_t width, uint32_t height,
                      uint32_t channels, uint32_t pitch_bytes, uint32_t block_w, uint32_t block_h, uint32_t mode,
                      uint32_t filter, float factor, float *const *block_value, uint32_t *const *out_w,
                      uint32_t *const *out_h, uint8_t *const *out_pixels);
int pxz_shrink_images_packed(pxz_handle *h, const uint8_t *const *pixels, uint32_t n_images, uint32_t width, uint32_t height,
                             uint32_t channels, uint32_t pitch_bytes, uint32_t block_w, uint32_t block_h, uint32_t mode,
                             uint32_t filter, float factor, float *const *block_value, uint32_t *const *out_w,
                             uint32_t *const *out_h, uint8_t *const *packed, uint64_t packed_capacity, uint64_t *packed_len);

/* Same over a batch of device-resident frames: the measured path (frames come
 * from a GPU decoder / stay in HBM).  All pointers are device pointers; outputs
 * are frame-major: index = frame*tiles + tile; out_pixels slot stride as above.
 * Asynchronous on the handle's stream; one kernel launch for the whole batch. */
int pxz_shrink_frames_device(pxz_handle *h, const pxz_frames *frames, const pxz_params *params,
                             const uint8_t *d_pixels, float *d_block_value, uint32_t *d_out_w,
                             uint32_t *d_out_h, uint8_t *d_out_pixels);

/* The colour conversion inside get_block_variance, per pixel (operations.rs:56-59: Srgba<u8>::into_linear()
 * .into_color::<Oklaba<f32>>(), palette 0.7.6 + the platform's cbrtf): d_laba[4i..4i+3] = {l, a, b, alpha} of
 * RGBA pixel i.  The same device function the Oklab detector kernels call -- exposed so that its bits can be
 * checked for every one of the 2^24 colours.  Device pointers (pixels 4-byte, output 16-byte aligned); async. */
int pxz_oklab_pixels_device(pxz_handle *h, const uint8_t *d_rgba, uint32_t n_pixels, float *d_laba);

/* Detector only: get_block_variance_directionally (operations.rs:192-259) ->
 * lod0 = hz, lod1 = vr (raw, before `* factor`); get_block_variance with the
 * shrink_by closures (operations.rs:26-126, pixlzr.rs:160-162) -> lod0 = lod1 =
 * value (params->factor applied).  Device pointers, frame-major. */
int pxz_lod_frames_device(pxz_handle *h, const pxz_frames *frames, const pxz_params *params,
                          const uint8_t *d_pixels, float *d_lod0, float *d_lod1);

/* ---- decode side ---------------------------------------- */

/* Pixlzr::expand (pixlzr.rs:77-122) + Pixlzr::to_image (pixlzr_image.rs:24-74) on device-resident
 * tiles: every stored tile (tile_w x tile_h pixels, tightly packed, in a slot of block_w*block_h*channels
 * bytes -- the layout pxz_shrink_frames_device leaves and decode_block (encoding/mod.rs:202-242) yields)
 * is resized back to its full size by PixlzrBlock::resize (block.rs:273-334: clone | ResizeAlg::Nearest |
 * SuperSampling(filter, 2), which is a plain convolution when nothing shrinks; Triangle means Bilinear here,
 * mod.rs:72-90) and written to its place in the frame.  `frames` describes the OUTPUT; params->block_w,
 * block_h and filter are used.  Tiles whose stored size is zero or exceeds their place are skipped and
 * flagged (pxz_decode_status).  Asynchronous on the handle's stream. */
int pxz_expand_frames_device(pxz_handle *h, const pxz_frames *frames, const pxz_params *params,
                             const uint32_t *d_tile_w, const uint32_t *d_tile_h, const uint8_t *d_slots,
                             uint8_t *d_out_pixels);

/* Pixlzr::decode_from_vec (src/encoding/mod.rs:95-165) + decode_block (:202-242) + the `qoi` decoder it
 * calls, on the device: n_frames .pixlzr files, back to back in d_files (file f = [off[f], off[f+1])),
 * become the per-tile values, stored sizes and pixel slots (the layout pxz_expand_frames_device and
 * pxz_encode_frames_device use).  `frames` gives the geometry every file must carry (width, height,
 * channels; pitch fields unused), params block_w/block_h.  A malformed file or record is flagged
 * (pxz_decode_status) and its tiles get size 0x0.  Asynchronous on the handle's stream. */
int pxz_decode_frames_device(pxz_handle *h, const pxz_frames *frames, const pxz_params *params,
                             const uint8_t *d_files, const uint64_t *d_file_offsets, float *d_block_value,
                             uint32_t *d_tile_w, uint32_t *d_tile_h, uint8_t *d_slots);

/* The same for one host-resident file (Pixlzr::decode_from_vec, mod.rs:95-165).  The header fields are
 * always returned; with all four output pointers NULL the call stops there (size query: the grid is
 * pxz_grid(width, height, block_w, block_h), slots need block_w*block_h*channels bytes per tile).
 * PXZ_ERR_INVALID_ARG for a malformed file (the reference panics / returns the qoi error). */
int pxz_decode_file(pxz_handle *h, const uint8_t *file, size_t len, uint32_t *width, uint32_t *height,
                    uint32_t *block_w, uint32_t *block_h, uint32_t *channels, uint32_t *filter_byte,
                    float *block_value, uint32_t *tile_w, uint32_t *tile_h, uint8_t *slots);

/* Waits for the handle's stream; flags of the last decode-side call on this handle:
 * bit 0  pxz_expand_frames_device met a tile whose stored size is zero or exceeds its place,
 * bit 1  pxz_decode_frames_device met a malformed file or record. */
int pxz_decode_status(pxz_handle *h, uint32_t *flags);

/* The same for one host-resident image (copies in, expands, copies out; PXZ_ERR_INVALID_ARG on an
 * invalid stored size). */
int pxz_expand_image(pxz_handle *h, uint32_t width, uint32_t height, uint32_t channels, uint32_t pitch_bytes,
                     uint32_t block_w, uint32_t block_h, uint32_t filter, const uint32_t *tile_w,
                     const uint32_t *tile_h, const uint8_t *slots, uint8_t *out_pixels);

/* ---- legacy image -> image filter ------------------------ */

/* process_custom (src/process/mod.rs:71-102) with the closures of process() (:107-121: |x - avg| and the
 * identity): per tile get_block_variance -> reduce_image_section((v, v)) with params->filter (the
 * down-scaling filter) -> .resize(w0, h0, filter_upscale) -> copy_from into an RGBA8 image
 * (DynamicImage::new_rgba8, :80-81; RGB input gains alpha 255).  process(image, n) is block_w = block_h = n,
 * params->filter = PXZ_FILTER_LANCZOS3, filter_upscale = PXZ_FILTER_NEAREST.  params->mode and factor are
 * ignored.  Two launches' worth of device work (shrink into the handle's scratch, expand into d_out_rgba);
 * asynchronous on the handle's stream. */
int pxz_process_frames_device(pxz_handle *h, const pxz_frames *frames, const pxz_params *params,
                              uint32_t filter_upscale, const uint8_t *d_pixels, uint8_t *d_out_rgba,
                              uint32_t out_pitch_bytes, uint64_t out_frame_stride_bytes);

/* tree::process_custom (src/process/tree.rs:23-83) with the closures of tree::process (:89-109): per tile of the
 * block_w x block_h grid get_block_variance (|x - avg|, identity); a tile with (value >= |threshold|) ^ (threshold >= 0)
 * is pixelised as process() does it (reduce_image_section((v, v)) with params->filter, resized back with
 * filter_upscale); any other tile is handed to the same function with both block sizes halved (|threshold| from
 * there on, so only the outermost level can be inverted) until a block size is no larger than max(min_block_*, 4)
 * (tree.rs:32-36), where the tile keeps its pixels.  tree::process(image, n, k) is block_w = block_h = n, min 4 x 4,
 * PXZ_FILTER_LANCZOS3 down, PXZ_FILTER_NEAREST up, threshold k.  Output RGBA8 (RGB input gains alpha 255; also in the
 * degenerate case of a block size at or below the minimum, where the reference returns the image unchanged in its own
 * colour type).  Any block geometry up to 128 x 128 (what src/bin/tree.rs:6 calls it with): blocks of at most 64 px that
 * halve evenly down to the last level run one detector + shrink + expand pass per level over that level's regular
 * grid; everything else -- 128-px blocks, halvings that go odd (50 -> 25 -> 12 + 12 + 1: every tile is cut from its own
 * corner, tree.rs:70-79) -- goes level by level over lists of rectangles, one block of threads per open tile, with one
 * 8-byte read-back per level: the next level's tile count and a consistency flag (a tile whose axis tables the host
 * did not list -> PXZ_ERR_INTERNAL; so that form is synchronous).  Blocks above 128 px: PXZ_ERR_UNSUPPORTED.  params->mode and factor are ignored. */
int pxz_tree_process_frames_device(pxz_handle *h, const pxz_frames *frames, const pxz_params *params,
                                   uint32_t filter_upscale, float threshold, uint32_t min_block_w, uint32_t min_block_h,
                                   const uint8_t *d_pixels, uint8_t *d_out_rgba, uint32_t out_pitch_bytes,
                                   uint64_t out_frame_stride_bytes);

/* Block-stream compaction (device): the valid out_w*out_h*channels bytes of every slot, in tile
 * order, into one contiguous stream -- the payload `encode_block` (src/encoding/mod.rs:168-200)
 * consumes tile after tile, and what one rank ships to the writer rank over RCCL.
 * d_offsets gets n_tiles+1 byte offsets (u64; the last one is the stream length).  Tiles that
 * would exceed packed_capacity are skipped (offsets stay valid, so the caller can detect it).
 * Asynchronous on the handle's stream. */
int pxz_pack_tiles_device(pxz_handle *h, uint32_t n_tiles, uint32_t channels, uint32_t slot_bytes,
                          const uint32_t *d_tile_w, const uint32_t *d_tile_h, const uint8_t *d_slots,
                          uint64_t *d_offsets, uint8_t *d_packed, uint64_t packed_capacity);

/* Pixlzr::encode_to_vec (src/encoding/mod.rs:40-89) + encode_block (:168-200) + the `qoi` crate 0.4.1
 * encoder it calls (:181-189), entirely on the device: the tiles of a batch of frames (as left by
 * pxz_shrink_frames_device: values, dims, slots) become the complete .pixlzr files, back to back in
 * d_out.  d_file_offsets gets n_frames+1 byte offsets (file f = [off[f], off[f+1])).  filter_byte as in
 * pxz_encode_container.  If out_capacity is too small the files are truncated but the offsets are
 * still exact (retry with off[n_frames] bytes).  Asynchronous on the handle's stream. */
int pxz_encode_frames_device(pxz_handle *h, const pxz_frames *frames, const pxz_params *params, uint32_t filter_byte,
                             const float *d_block_value, const uint32_t *d_tile_w, const uint32_t *d_tile_h,
                             const uint8_t *d_slots, uint8_t *d_out, uint64_t out_capacity, uint64_t *d_file_offsets);

/* ---- bitstream: Pixlzr::encode_to_vec, src/encoding/mod.rs:40-89,168-200 ---- */
/* Tiles given as produced by pxz_shrink_image (slots + dims + values).
 * has_value may be NULL (all Some); has_value[t]==0 writes 0.0 (mod.rs:173-178).
 * filter_byte = `self.filter.unwrap_or_default() as u8` (mod.rs:53): 0 after from_image.
 * Returns the number of bytes written into `out`, or a negative pxz_status;
 * call with out==NULL to obtain an upper bound. */
int64_t pxz_encode_container(uint32_t width, uint32_t height, uint32_t block_w, uint32_t block_h,
                             uint32_t channels, uint32_t filter_byte, const float *block_value,
                             const uint8_t *has_value, const uint32_t *tile_w, const uint32_t *tile_h,
                             const uint8_t *slots, uint8_t *out, size_t out_capacity);

/* qoi::Encoder::new(data,w,h).encode_to_vec() as called at mod.rs:181-189
 * (qoi crate 0.4.1 semantics incl. its run-of-one INDEX substitution).
 * Writes the full stream incl. "qoif"; returns its length or a negative status. */
int64_t pxz_qoi_encode(const uint8_t *data, uint32_t w, uint32_t h, uint32_t channels,
                       uint8_t *out, size_t out_capacity);
size_t pxz_qoi_bound(uint32_t w, uint32_t h, uint32_t channels);

/* ---- utilities -------------------------------------------------------- */
/* Deterministic synthetic frames for benchmarks/tests (integer-only generator,
 * DESIGN.md "Synthetic frames"); dist: 0 opaque, 1 alpha, 2 flat, 3 noise.
 * frame f of the batch uses seed 0x5049584C + first_frame_index + f. */
int pxz_synth_frames_device(pxz_handle *h, const pxz_frames *frames, uint8_t *d_pixels,
                            uint32_t first_frame_index, uint32_t dist);

/* Down-scaling tables the kernels use for one axis (for cross-checks):
 * fast_image_resize coefficient windows in i16 fixed point.  coeffs has room for
 * out_size*window entries; any output pointer may be NULL. */
int pxz_axis_table(uint32_t in_size, uint32_t out_size, uint32_t filter,
                   int32_t *starts, int32_t *sizes, int16_t *coeffs, int32_t *window, int32_t *precision);

/* Average device time (milliseconds) of the kernels launched by the last
 * *_device call on this handle, measured with HIP events on the handle's
 * stream; blocks until that work is done.  Enabled by pxz_enable_timing(h, 1); pxz_enable_timing(h, n) with
 * n > 1 brackets every n-th step only (three event records cost a 0.27 ms step about 2 %). */
int pxz_enable_timing(pxz_handle *h, int on);
int pxz_last_kernel_ms(pxz_handle *h, float *ms);
/* The same average for the FIRST kernel of each step alone (shrink32/64/16_kernel; oklab_kernel in shrink_by
 * steps): the figure a per-kernel profile shows for it.  Call before pxz_last_kernel_ms, which resets the record. */
int pxz_last_first_kernel_ms(pxz_handle *h, float *ms);

/* Diagnostics: the handle's kernel-selection state.  Two dwords in pinned memory, written by the worklist kernel of the last
 * FINISHED fast-path launch and read without synchronisation when the next one is set up, pick which kernels run (never what
 * they compute): state[0] = full tiles with transparency that launch saw (>= 2048: the four-plane kernel is launched; >= half
 * the tiles: it goes first), state[1] = tiles it listed for the worklist kernel (sizes that kernel's grid), state[2] = 1 if the
 * LAST launch set up through this handle ran the four-plane kernel, state[3] = 1 if it ran it first.  A timing is only
 * comparable with another taken in the same state; bench.py prints it beside its numbers. */
int pxz_handle_state(pxz_handle *h, uint32_t state[4]);

/* Diagnostics: copies `bytes` bytes at byte `offset` of the handle's worklist buffer to `dst` after waiting for the
 * handle's stream (the in-kernel phase stamps of the -DPXZ_STAMPS build land there; tools/stamps_run.py).  Not part of
 * the path; PXZ_ERR_INVALID_ARG when the range lies outside the buffer. */
int pxz_debug_read_work(pxz_handle *h, void *dst, size_t offset, size_t bytes);
/* The same for the status buffer of the decode side (the -DPXZ_STAMPS build keeps expand_kernel's phase stamps behind the
 * status word; tools/stamps_expand.py). */
int pxz_debug_read_status(pxz_handle *h, void *dst, size_t offset, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* PIXLZR_HIP_H */
