/*
 * pixlzr_oracle.h — CPU restatement of the pixlzr encode hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, bench.py's
 * `cpu_baseline` leg and __graft_entry__.smoke() may load this library; the
 * product (pixlzr-rust_amd/) never links, loads or calls anything in oracle/.
 *
 * Every function restates one piece of the reference (guiga-zalu/pixlzr-rust
 * 0.3.1, Rust) and cites the file:line it follows.  The reference cannot be
 * compiled here (no rustc/cargo, crates not vendored), so parity is pinned by
 * the reference's own checked-in fixtures (tests/golden/, see
 * tests/test_oracle_golden.py):
 *   - tiling + QOI + container: benches/base.png -> benches/base.pixlzr, whole
 *     file byte-exact;
 *   - Oklab-MAD detector + level decision: Big-Ruscher.png -> Big-Ruscher.pix,
 *     2040/2040 reduced (w,h) exact, stored f32 values within 1e-4 relative;
 *   - libm pieces (cbrtf, hypotf): bit-identical to this image's glibc 2.35
 *     over the whole input domain (tests/test_oracle_math.py);
 *   - fast_image_resize 4.2.1 arithmetic: restated from its published
 *     algorithm (source not vendored): PARITY UNPINNED at bit level beyond the
 *     reference's constant-colour test (src/data_types/block.rs:400-435).
 */
#ifndef PIXLZR_ORACLE_H
#define PIXLZR_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* FilterType repr(u8), src/data_types/mod.rs:10-30 */
enum { ORC_NEAREST = 0, ORC_TRIANGLE = 1, ORC_CATMULLROM = 2, ORC_GAUSSIAN = 3, ORC_LANCZOS3 = 4 };
/* mode: which Pixlzr::shrink_* the call restates */
enum { ORC_MODE_SHRINK_BY = 0, ORC_MODE_SHRINK_DIRECTIONALLY = 1 };

/* ---- math primitives (platform libm restated; see header comment) ---- */
float orc_cbrtf(float x);
float orc_hypotf(float x, float y);
float orc_srgb_u8_to_linear(uint8_t v);
/* mismatch counters vs this box's libm, over float bit patterns [lo,hi] */
uint64_t orc_selftest_cbrtf(uint32_t lo_bits, uint32_t hi_bits, uint32_t step);
uint64_t orc_selftest_hypotf(uint64_t n, uint32_t seed);

/* ---- tiling: src/split.rs:10-61, src/data_types/iter.rs:28-87 ---- */
void orc_grid(uint32_t iw, uint32_t ih, uint32_t bw, uint32_t bh, uint32_t *cols, uint32_t *rows);
void orc_tile_rect(uint32_t iw, uint32_t ih, uint32_t bw, uint32_t bh, uint32_t tile,
                   uint32_t *x, uint32_t *y, uint32_t *w, uint32_t *h);

/* ---- LOD detectors on ONE tile (pitch in bytes, c = 3|4) ---- */
/* src/operations.rs:192-259 */
void orc_lod_directional(const uint8_t *tile, uint32_t w, uint32_t h, uint32_t c, uint32_t pitch,
                         float *hz, float *vr, uint64_t *sum_hz, uint64_t *sum_vr);
/* src/operations.rs:26-126 with the closures of Pixlzr::shrink_by (pixlzr.rs:160-162) */
float orc_lod_oklab(const uint8_t *tile, uint32_t w, uint32_t h, uint32_t c, uint32_t pitch, float factor);
/* one pixel -> Oklab (l,a,b) + alpha, palette 0.7.6 model */
void orc_oklab_pixel(const uint8_t *px, uint32_t c, float out_laba[4]);
void orc_oklab_pixels(const uint8_t *px, uint32_t c, uint64_t n, float *out_laba);

/* ---- level decision: src/operations.rs:128-156 ---- */
void orc_reduce_dims(float v0, float v1, uint32_t w, uint32_t h,
                     uint32_t *nw, uint32_t *nh, float *stored_value);

/* ---- resample: src/data_types/block.rs:273-334 + fast_image_resize 4.2.1 ---- */
/* returns 0 on success; dst is nw*nh*c tightly packed */
int orc_resize(const uint8_t *src, uint32_t w, uint32_t h, uint32_t c, uint32_t pitch,
               uint8_t *dst, uint32_t nw, uint32_t nh, uint32_t filter);
/* i16 coefficient table for one axis (for cross-checking the product's tables) */
int orc_fir_coeffs(uint32_t in_size, uint32_t out_size, uint32_t filter,
                   int32_t *starts, int32_t *sizes, int16_t *coeffs /* out_size*window */,
                   int32_t *window, int32_t *precision);
/* image-rs 0.25 imageops::resize (float intermediate) — ONLY used to check the
 * Big-Ruscher.pix payloads, which predate the fir default. */
int orc_resize_imagers(const uint8_t *src, uint32_t w, uint32_t h, uint32_t c, uint32_t pitch,
                       uint8_t *dst, uint32_t nw, uint32_t nh, uint32_t filter);

/* ---- whole image: Pixlzr::from_image + shrink_by|shrink_directionally ----
 * src/data_types/pixlzr_image.rs:6-22, src/data_types/pixlzr.rs:155-205.
 * Outputs are tile-indexed (row-major ty*cols+tx); out_pixels uses fixed
 * slots of bw*bh*c bytes, of which out_w*out_h*c are valid.
 * out_pixels may be NULL (LOD + dims only).  nthreads>=1 (tile rows split). */
int orc_shrink_image(const uint8_t *pixels, uint32_t width, uint32_t height, uint32_t channels,
                     uint32_t pitch, uint32_t bw, uint32_t bh, uint32_t mode, uint32_t filter,
                     float factor, float *block_value, uint32_t *out_w, uint32_t *out_h,
                     uint8_t *out_pixels, int nthreads);

/* ---- bitstream: qoi 0.4.1 + src/encoding/mod.rs:40-89,168-200 ---- */
/* worst-case bytes for a w*h*c tile including the 14-byte header and 8-byte tail */
/* Pixlzr::expand + to_image (pixlzr.rs:77-122, pixlzr_image.rs:24-74): tiles (tile_w x tile_h pixels at
 * slots + t*slot_stride, tightly packed) back to a width x height image; filter as FilterType repr(u8). */
int orc_expand_image(uint32_t width, uint32_t height, uint32_t bw, uint32_t bh, uint32_t channels, uint32_t filter,
                     const uint32_t *tile_w, const uint32_t *tile_h, const uint8_t *slots, size_t slot_stride,
                     uint8_t *out_pixels, uint32_t out_pitch);

/* process() / process_custom (process/mod.rs:71-121): per tile Oklab MAD with the identity closure ->
 * reduce_image_section((v, v)) with filter_down -> resize back with filter_up -> RGBA8 image. */
int orc_process_image(const uint8_t *pixels, uint32_t width, uint32_t height, uint32_t channels, uint32_t pitch,
                      uint32_t bw, uint32_t bh, uint32_t filter_down, uint32_t filter_up, uint8_t *out_rgba,
                      uint32_t out_pitch);
/* tree::process_custom (src/process/tree.rs:23-83) with the closures of tree::process (:89-109): recursive halving of
 * the block while (value >= threshold) ^ is_positive fails; out is RGBA8.  process(image, n, k) = bw = bh = n,
 * min 4 x 4, Lanczos3 down, Nearest up. */
int orc_tree_process_image(const uint8_t *pixels, uint32_t width, uint32_t height, uint32_t channels, uint32_t pitch,
                           uint32_t bw, uint32_t bh, uint32_t min_bw, uint32_t min_bh, float threshold,
                           uint32_t filter_down, uint32_t filter_up, uint8_t *out_rgba, uint32_t out_pitch);

size_t orc_qoi_bound(uint32_t w, uint32_t h, uint32_t c);
/* full QOI stream incl. "qoif" magic; returns length */
size_t orc_qoi_encode(const uint8_t *data, uint32_t w, uint32_t h, uint32_t c, uint8_t *out);
/* decode a full QOI stream; returns 0 on success */
int orc_qoi_decode(const uint8_t *in, size_t len, uint32_t *w, uint32_t *h, uint32_t *c,
                   uint8_t *out, size_t out_cap);
/* container writer. tiles given as slots (slot stride bw*bh*c) + dims + values
 * (has_value[i]==0 -> block_value None -> 0.0 written). returns length written
 * (call with out==NULL to get the bound). */
size_t orc_encode_container(uint32_t width, uint32_t height, uint32_t bw, uint32_t bh, uint32_t channels,
                            uint32_t filter_byte, const float *block_value, const uint8_t *has_value,
                            const uint32_t *tw, const uint32_t *th, const uint8_t *slots,
                            uint8_t *out, size_t out_cap);
/* container reader (src/encoding/mod.rs:95-165,202-242): fills header fields and per-tile
 * value/dims, decoding payloads into slots (stride bw*bh*4, channel count returned per tile). */
int orc_decode_container(const uint8_t *in, size_t len, uint32_t *width, uint32_t *height,
                         uint32_t *bw, uint32_t *bh, uint32_t *filter_byte,
                         float *block_value, uint32_t *tw, uint32_t *th, uint32_t *tc,
                         uint8_t *slots, size_t slot_stride, uint32_t max_tiles);

/* ---- synthetic frames; dist: 0 opaque, 1 alpha, 2 flat, 3 noise ---- */
void orc_synth_frame(uint8_t *pixels, uint32_t width, uint32_t height, uint32_t channels,
                     uint32_t pitch, uint32_t frame_index, uint32_t dist);

#ifdef __cplusplus
}
#endif
#endif
