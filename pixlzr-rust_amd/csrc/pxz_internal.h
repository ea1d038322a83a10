// pxz_internal.h — structures shared by the host runtime and the gfx950 kernels.
#pragma once
#include <stdint.h>

namespace pxz {

constexpr int kMaxLevel = 18;        // level exponents 0..17 (2^17 > any LDS-resident tile side)
constexpr int kNumThresholds = 32;   // float thresholds for round(log2f(v)) >= -k, k = 0..31
constexpr int kWave = 64;

// One down-scaling table along one axis: in_size -> out_size.
// Convolution (windows padded to whole quads of 4 source samples, zero weights in the padding):
//   bounds[bounds_off + 2*o]     first quad (source index / 4) of output o
//   bounds[bounds_off + 2*o + 1] number of quads
//   coeffs[coeff_off + o*wquads*2 + 2*q + {0,1}]  packed i16 pairs (k0 | k1<<16), (k2 | k3<<16)
//   ksums[ksum_off + o]          sum of the window's weights (constant-input shortcut)
// Nearest: bounds[bounds_off + o] = source index.
struct AxisTab {
	uint32_t bounds_off;
	uint32_t coeff_off;
	uint32_t ksum_off;
	uint32_t rows_off;       // unified rows (fast path): per output row_stride dwords =
	uint32_t row_stride;     //   {first quad, quads, weight sum, 0} + wquads*2 packed pairs, 16-B aligned
	uint16_t out_size;
	uint16_t wquads;
	uint16_t precision;
	uint16_t in_size;
	uint32_t mf_off;         // matrix-core operand table of a 32 -> 16|8 axis (dword offset in the rows blob, 0: none)
};

// Matrix-core operand table of one 32 -> out axis (out = 16 or 8), kMfDwords dwords, see resample_mfma32:
//   [0, 128)    low bytes  of the weights: lane l = (o = l & 15, g = l >> 4), 8 x i8 = K_lo[o][src(g, j)]
//   [128, 256)  high bytes of the weights, same arrangement (K = 256 * K_hi + K_lo, both signed 8-bit)
//   [256, 272)  bias[o]  = 128 * sum_i K[o][i] + 2^(precision - 1)
//   [272, 288)  ksum[o]  = sum_i K[o][i]
//   [288]       1 if clip8(2^(precision-1) + 255 * ksum[o]) == 255 for every o (opaque stays opaque)
// src(g, j) = 4g + j (j < 4), 16 + 4g + (j - 4) (j >= 4): the order in which the first product's
// accumulator registers hand their rows to the second product.
constexpr uint32_t kMfDwords = 292;
// Matrix-core operand table of one 16 -> out axis (out = 8, 4, 2 or 1; resample_group16_mfma): dwords [0..31] the low weight
// bytes, dword o * 4 + g = W[o][4g .. 4g+3] (rows o >= out are zero); [32..63] the high bytes; [64..71] bias = 128 * sum + half;
// [72..79] the weight sums; [80] 1 if a constant-255 alpha convolves to 255 at every output; [81] the precision
constexpr uint32_t kMf16Dwords = 84;

// Layout of the handle's worklist buffer (dwords): [0], [1] the two worklist counters (list B: tiles for the
// generic kernel) used alternately, [2 + 64*s, 2 + 64*s + 64) the tile-ticket counters of shrink64_kernel for
// counter set s, [kWorkA + s] the counters of list A, then list B and list A (n_tiles entries each).
constexpr uint32_t kTicketCounters = 64;
constexpr uint32_t kWorkA = 2 + 2 * kTicketCounters;  // two counters of list A (transparent tiles for shrink32a_kernel)
constexpr uint32_t kWorkList = kWorkA + 2;             // list B starts here (n_tiles entries), list A follows it

// q = n / d as (t + ((n - t) >> sh1)) >> sh2 with t = mulhi(n, mul)  (Granlund-Montgomery)
struct FastDiv {
	uint32_t mul, sh1, sh2;
};

struct ShrinkArgs {
	const uint8_t *src;      // frames, pitch-linear
	uint64_t frame_stride;   // bytes
	uint32_t pitch;          // bytes
	uint32_t width, height;
	uint32_t bw, bh;         // nominal tile size
	uint32_t cols, rows;
	uint32_t tiles_per_frame;
	uint32_t n_tiles;        // n_frames * tiles_per_frame
	FastDiv div_tpf, div_cols;
	uint32_t edge_w, edge_h; // size of the last column / row of tiles
	uint32_t mode, filter;
	float factor;
	// outputs (device); out_px may be null (no resample), out_w/out_h may be null
	uint32_t oklab_given;    // 1: full 32x32 RGBA tiles already carry their Oklab value in sums[] (oklab32_kernel)
	float scale2;            // Oklab mode: value = mean deviation * factor * scale2 (10 = BASE_FACTOR of shrink_by,
	                         //   pixlzr.rs:15,162; 1 with factor 1 = the identity closure of process(), process/mod.rs:107-121)
	FastDiv div_gpf, div_gcols;  // shrink16_kernel: divisors for its 2x2 tile groups (groups per frame, group columns)
	uint32_t n_frames_x_groups;  //   and the number of groups in the batch
	uint32_t ok_bands;       // Oklab detector with run-time geometry: bands (256 pixels) per tile, ceil(w * h / 256)
	uint32_t ok_region;      //   which tiles that launch takes: 0 the full ones, 1 the right column (edge_w x bh), 2 the bottom
	                         //   row (bw x edge_h), 3 the corner tile (edge_w x edge_h)
	uint32_t ok_count;       //   and how many items it enumerates (region 0: n_tiles)
	uint32_t ok_edges;       // consumers: edge regions whose values are in sums[] already (bit 0 right, 1 bottom, 2 corner)
	uint32_t clone_ahead;    // shrink_by, square RGBA tiles (round 4): oklab2_kernel copies every tile it converts into its slot as if it
	                         //   were stored at full size; the fast kernels behind it do not read such tiles again (PXZ_NO_CLONE_AHEAD=1: 0)
	uint32_t *clone_list;    //   64x64: room for the list of the tiles that are NOT stored at full size (8 + n_tiles dwords; clone_split64_kernel)
	uint32_t *ahead_spare;   //   2 KB nobody reads: where that kernel's copy of a band goes when there is no tile behind it
	float *ok_scratch;       // Oklab detector on 64x64 tiles: 16 floats per pixel quad between its passes (HBM)
	const uint32_t *mf64;    // 64x64 fast path: matrix-core operand tables (global memory), see Fast64Args
	uint32_t ok_rows;        // tile rows the block-cooperative Oklab detector takes: full_rows, plus the ragged last row
	                         //   when its height is a whole number of that detector's bands
	uint32_t full_cols, full_rows;  // 32x32 fast path: tile (tx, ty) is eligible iff tx < full_cols && ty < full_rows
	                                //   (full size, 16-byte aligned rows; 0/0 when the batch is not aligned)
	uint32_t alpha_kernel;   // frames with transparency announced (pxz_params.reserved bit 0) or seen by the last launch
	uint32_t alpha_first;    // most tiles of the last launch had transparency: the four-plane kernel takes every tile
	uint32_t finish_scan;    // worklist kernel: 1 = finish every tile the fast kernel completed (scan of all sums); 0 = the fast
	                         //   kernel did that itself, only the tiles of list A are left (when shrink32a_kernel took them)
	uint32_t list_a_too;     // worklist kernel: list A (full tiles with transparency) was not taken by shrink32a_kernel
	void *mid_event;         // host side only: hipEvent_t to record behind the first kernel of the step, or null
	uint32_t *stats;         // pinned host dwords (device address) <- [0] list-A tiles, [1] all listed tiles of this launch; may be null
	uint32_t expect_listed;  // what [1] said after the last finished launch (0xffffffff: unknown): sizes the worklist kernel's grid, nothing else
	uint32_t stats_sig;      // signature of this launch's configuration, written to stats[2] beside the counts: the host only trusts
	                         //   counts whose signature is its next launch's (round 4: a handle that went from 32x32 tiles, nothing
	                         //   listed, to 64x64 frames with a ragged bottom row ran its first launches with an 8-block worklist grid)
	uint32_t *work;          // worklist: [work_slot] = count, [2..] = tile ids (null: all tiles).  The two
	uint32_t work_slot;      //   counters alternate between launches; a launch zeroes the other one
	float *value;            // worklist mode only: the kernel finishes its tiles itself (finish_tile) and,
	float *lod0;             //   after the worklist, the tiles shrink32_kernel completed; each may be null
	float *lod1;
	uint32_t *sums;          // 2 per tile: gradient sums (directional) | f32 value bits (Oklab); -> finish_kernel
	uint32_t *out_w;
	uint32_t *out_h;
	uint8_t *out_px;
	uint32_t slot_bytes;     // bw*bh*channels
	// tables (device)
	AxisTab tabs[4 * kMaxLevel];  // [axis 0=x,1=y][cls 0=full,1=edge][kMaxLevel], in the kernarg segment:
	                              // scalar loads, never behind the vector-memory counter
	const uint16_t *bounds;
	const uint32_t *coeffs;
	const int32_t *ksums;
	const uint32_t *trows;   // unified table rows, padded by 32 dwords
	uint32_t tab_dw;         // dwords of trows to keep in LDS (multiple of 4), 0 = too large / unused
	// Level decision: m = #{j < kMaxLevel : key < breaks[cls][j]} (or key >= ... when breaks_asc[cls]),
	// cls = ycls*2 + xcls.  directional: key = integer gradient sum; Oklab: key = bits of the parsed value.
	uint32_t breaks[4][kMaxLevel];
	uint32_t breaks_asc[4];
	// LDS carve-up (dwords per tile): 4 planes of u16 pairs [y][x], 4 transposed planes [ox][y], Oklab scratch
	uint32_t rs;             // plane row stride: round_up(ceil(bw/2), 2), +2 of bank skew when a multiple of 16
	uint32_t plane_dw;       // rs * bh
	uint32_t hps;            // transposed row stride: round_up(ceil(bh/2), 2), +2 of bank skew likewise
	uint32_t tmp_dw;         // ceil(bw/2) * hps   (0 when no convolution)
	uint32_t lab_dw;         // Oklab mode: 3*bw*bh floats (aliases the transposed planes), else 0
	uint32_t tile_dw;        // total dwords per tile incl. over-read slack
	uint32_t *big_scratch;   // tiles whose image exceeds LDS (round 4): big_blocks images of tile_dw dwords in HBM, one per block
	uint32_t big_blocks;     //   of the generic kernel (0: the image is in LDS)
};

// Arguments of shrink32_kernel (full 32x32 RGBA tiles only): the subset of ShrinkArgs it needs
struct Fast32Args {
	const uint8_t *src;
	uint64_t frame_stride;
	uint32_t pitch, cols, rows, tiles_per_frame, n_tiles;
	FastDiv div_tpf, div_cols;
	uint32_t n_groups;       // shrink16_kernel: 2x2 groups of tiles, ceil(cols/2) * ceil(rows/2) per frame
	FastDiv div_gpf, div_gcols;
	uint32_t full_cols, full_rows, ok_rows, ok_edges;  // as in ShrinkArgs
	uint32_t filter;
	uint32_t *sums;
	uint32_t *out_w;
	uint32_t *out_h;
	uint8_t *out_px;
	uint32_t *work;
	uint32_t work_slot;
	uint32_t alpha_list;     // 1: full tiles with transparency go to list A (shrink32a_kernel), else to list B (generic kernel)
	const uint32_t *trows;
	uint32_t tab_dw, tile_dw;
	uint32_t chunk_lg;       // tickets deal runs of 2^chunk_lg adjacent tiles
	uint32_t finish_here;    // shrink32_kernel: every block finishes the tiles it completed (value / lod outputs) at its end
	uint32_t all_tiles;      // shrink32a_kernel: every tile of the batch (not list A): the launch that skips shrink32_kernel
	uint32_t narrow;         // shrink32_kernel: 4/2/1-px-wide outputs take resample_mfma32_narrow (0: PXZ_NO_NARROW=1, the round-1 forms)
	uint32_t group16;        // shrink16_kernel: the two-pass tiles of a group go through resample_group16_mfma (0: PXZ_NO_GROUP16=1)
	uint32_t clone_ahead;    // MODE 0: tiles stored at full size are in their slots already (ShrinkArgs::clone_ahead)
	float factor;            //   with these, as the worklist kernel's scan over all tiles would
	float *value, *lod0, *lod1;
	uint32_t breaks[kMaxLevel];
	uint32_t breaks_asc;
	AxisTab tabs[kMaxLevel];
};

// shrink64_kernel: full, aligned, opaque 64x64 RGBA tiles (the reference CLI's default block size), one
// tile per block of four waves.  Matrix-core operand tables live in global memory (mf64):
//   per level with out = 32|16|8|4|2|1:  nblk = max(1, out/16) output blocks, each
//     [lo: 64 lanes x 16 B][hi: 64 lanes x 16 B]   lane (o = l & 15, g = l >> 4): K[16*blk + o][16g .. 16g+15]
//   then bias[32], ksum[32], flag (as in kMfDwords)
struct Fast64Args {
	const uint8_t *src;
	uint64_t frame_stride;
	uint32_t pitch, cols, rows, tiles_per_frame, n_tiles;
	FastDiv div_tpf, div_cols;
	uint32_t full_cols, full_rows, ok_rows, ok_edges;
	uint32_t filter;
	uint32_t *sums;
	uint32_t *out_w;
	uint32_t *out_h;
	uint8_t *out_px;
	uint32_t *work;
	uint32_t work_slot;
	uint32_t clone_ahead;    // MODE 0: tiles stored at full size are in their slots already (ShrinkArgs::clone_ahead)
	uint32_t *clone_list;    //   and, when not null, finished: [0] = how many tiles are left, [8 ..] = which (clone_split64_kernel)
	uint32_t all_tiles;      // ALPHA instance: every tile of the batch (not list A): the launch that skips the opaque instance
	const uint32_t *mf64;
	uint32_t mf_off[kMaxLevel];      // dword offset of the level's table in mf64 (0: none; the blob starts with a pad)
	uint32_t precision[kMaxLevel];
	uint32_t breaks[kMaxLevel];
	uint32_t breaks_asc;
	uint32_t finish_here;    // every block finishes the tiles it completed (value / lod outputs) at its end, as shrink32_kernel does
	float factor;
	float *value, *lod0, *lod1;
};

// Diagnostic switches (PXZ_* environment variables), read ONCE per process -- never on the call path.  Every one of
// them only picks between kernels that produce the same bytes (the tests run both sides); none is part of the ABI.
struct Knobs {
	bool no_alpha_kernel;   // PXZ_NO_ALPHA_KERNEL: tiles with transparency stay on the generic kernel
	bool no_oklab_general;  // PXZ_NO_OKLAB_GENERAL: no run-time-geometry Oklab detector
	bool no_oklab32;        // PXZ_NO_OKLAB32: no block-cooperative Oklab detector at all
	bool no_oklab_edges;    // PXZ_NO_OKLAB_EDGES: ragged edge tiles keep their four-lane chains
	bool no_repitch;        // PXZ_NO_REPITCH: unaligned device batches are staged pixel by pixel
	bool no_widen;          // PXZ_NO_WIDEN: RGB batches never ride the RGBA kernels
	bool no_native_rgb;     // PXZ_NO_NATIVE_RGB: RGB batches are widened to RGBA even where a kernel reads RGB itself
	bool no_alpha_first;    // PXZ_NO_ALPHA_FIRST: transparent batches keep the two-kernel flow (shrink32_kernel lists, shrink32a_kernel takes the list)
	bool no_narrow;         // PXZ_NO_NARROW: 4/2/1-px-wide outputs of 32x32 tiles keep the round-1 resample forms
	bool no_big_tiles;      // PXZ_NO_BIG_TILES: tiles whose image exceeds LDS are refused (PXZ_ERR_UNSUPPORTED), as before round 4
	bool no_group16;        // PXZ_NO_GROUP16: the two-pass tiles of a 16x16 group keep their own dot2 resamples (no block-diagonal matrix-core products)
	bool no_clone_ahead;    // PXZ_NO_CLONE_AHEAD: shrink_by's detector does not copy tiles into their slots ahead of the value (round-3 flow: the shrink kernel reads every tile again)
	bool oklab_v1;          // PXZ_OKLAB_V1: round-1 detector (one chain wave, two barriers per band; 64-px tiles parked in HBM)
	bool no_expand_fast32;    // PXZ_NO_EXPAND_FAST32: expand_kernel keeps its general forms for 32x32 RGBA tiles (no matrix-core convolutions, no shift-indexed Nearest)
	bool tree_rects;        // PXZ_TREE_RECTS: tree::process always goes over rectangle lists (pxz_tree.hip), also where the per-level grids apply
	int wpb;                // PXZ_WPB: waves per block of the persistent kernels (0: default)
	int chunk_lg;           // PXZ_CHUNK_LG: log2 of the ticket run length of shrink32_kernel (-1: default)
};
const Knobs &knobs();

struct LaunchGeom {
	uint32_t blocks, threads, lds_bytes;
};

struct FinishArgs {
	const uint32_t *sums;
	float *value;            // may be null
	float *lod0;             // may be null
	float *lod1;             // may be null
	uint32_t n_tiles, tiles_per_frame, cols, rows, bw, bh, edge_w, edge_h, mode;
	float factor;
};

struct PackArgs {
	const uint32_t *w, *h;        // per tile
	const uint32_t *sizes;        // optional: explicit byte sizes per tile (else w*h*channels)
	const uint8_t *slots;
	unsigned long long *offsets;  // n_tiles + 1
	unsigned long long *chunk_totals;
	uint8_t *packed;
	unsigned long long capacity;
	uint32_t n_tiles, n_chunks, channels, slot_bytes;
};

struct QoiArgs {
	const uint8_t *slots;
	const uint32_t *w, *h;
	const float *value;
	uint32_t *perm;               // tiles ordered by pixel count (largest first)
	uint32_t *bins;               // 64: histogram + cursors
	uint8_t *scratch;             // the encoder's units (pxz_stream.hip: qoi_class_rows), classes in descending order
	uint32_t *rec_len;
	unsigned long long *offsets;  // n_tiles + 1
	unsigned long long *chunk_totals;
	uint8_t *out;
	unsigned long long *file_offsets;  // n_frames + 1
	unsigned long long capacity;
	uint32_t n_tiles, n_chunks, tiles_per_frame, cols, rows;
	uint32_t channels, slot_bytes, hdr_bytes;
	uint32_t width, height, bw, bh, filter_byte;
	uint32_t splice_unit_blocks;  // (set by launch_qoi) the blocks of qoi_splice_kernel below this one take units, the others headers
};

// Decode side (expand_kernel): one table per (axis, size class, source size) of an up-scale to the full
// tile size.  starts/sizes hold one entry per output sample, coeffs out_size * window i16 weights.
struct ExpandTab {
	uint32_t start_off;      // into starts[] / sizes[] (nearest: starts[] is the source index)
	uint32_t coeff_off;      // into coeffs[]
	uint16_t window;
	uint16_t precision;
};

// expand_kernel's matrix-core tables (32x32 tiles, the convolutions): one block of kXmfDw dwords per stored size 2^li,
// li = 0 .. kXmfLevels-1 (1, 2, 4, 8, 16 px -> 32 px; one table serves both axes).  Operand of v_mfma_i32_32x32x16_i8
// for output sample o = lane & 31 and k group kg = lane >> 5: k slot j stands for source sample xmf_src(kg, j).
//   [0, 128)    low bytes of the weights, lane l at dwords 2l, 2l+1      [128, 256)  high bytes (weight = 256 hi + lo, lo signed)
//   [256, 288)  bias per output sample (128 * weight sum + half)           [288, 320)  the same in accumulator order [g][reg]
//   [320]       precision        [321]  1: a one-sample source is copied (every window is the weight 2^precision alone)
constexpr uint32_t kXmfLevels = 5, kXmfDw = 324;
constexpr uint32_t xmf_src(uint32_t kg, uint32_t j) { return j < 4u ? 4u * kg + j : 8u + 4u * kg + (j - 4u); }
constexpr uint32_t xmf_row(uint32_t g, uint32_t reg) { return (reg & 3u) + 8u * (reg >> 2) + 4u * g; }  // accumulator reg -> row

// expand16_kernel's matrix-core tables (16x16 tiles in 2x2 groups, the convolutions): one block of kXmf16Dw dwords per stored size
// 2^li, li = 0 .. kXmf16Levels-1 (1, 2, 4, 8 px -> 16 px; one table serves both axes):
//   [0, 32)   low bytes of the weights: dword o * 2 + kg = K[o][4 kg .. 4 kg + 3] for output sample o < 16 (source samples >= the
//             stored size carry zero)                       [32, 64)  high bytes
//   [64, 80)  bias per output sample (128 * weight sum + half)
//   [80, 96)  the same in accumulator order: [g][r], r < 8, for output sample (r & 3) + 8 (r >> 2) + 4 g
//   [96]      precision
constexpr uint32_t kXmf16Levels = 4, kXmf16Dw = 100;

// expand64_kernel's matrix-core tables (64x64 tiles, the convolutions): one block of kXmf64Dw dwords per stored size 2^li,
// li = 0 .. kXmf64Levels-1 (1 .. 32 px -> 64 px; one table serves both axes).  The 64 outputs are two halves q of 32, the source
// samples steps s of 16 (one step up to 16 px, two at 32):
//   [((q * 2 + s) * 2 + piece) * 128 + 2 l, + 1]   lane l = kg * 32 + o: weight bytes (piece 0 low, 1 high) of output 32 q + o for
//                                                  the source samples 16 s + xmf_src(kg, j), j = 0 .. 7
//   [1024, 1088)  bias per output sample (128 * weight sum + half)
//   [1088, 1152)  the same in accumulator order: [q][g][reg] for output 32 q + xmf_row(g, reg)
//   [1152]        precision
constexpr uint32_t kXmf64Levels = 6, kXmf64Dw = 1156;

struct ExpandArgs {
	const uint32_t *tile_w, *tile_h;  // per tile: stored size
	const uint8_t *slots;             // per tile slot_bytes, tile_w*tile_h*channels valid, tightly packed
	uint8_t *dst;                     // frames, pitch-linear
	uint64_t frame_stride;
	uint32_t pitch, width, height, channels;
	uint32_t bw, bh, cols, rows, tiles_per_frame, n_tiles, edge_w, edge_h;
	uint32_t slot_bytes, filter;
	uint32_t out_channels;            // bytes per pixel in dst (4 with channels == 3: alpha 255 is added, process())
	// tabs[(axis*2 + cls) * dir_stride + in_size], in_size in [1, full]; dir_stride = max(bw, bh) + 1
	const ExpandTab *tabs;
	uint32_t dir_stride;
	const uint16_t *starts, *sizes;
	const int16_t *coeffs;
	uint32_t tile_dw;                 // LDS dwords per wave: source pixels + horizontal-pass result
	const uint32_t *xmf;              // kXmfLevels * kXmfDw dwords, or null (other tile sizes, Nearest, weights beyond 16 bits)
	uint32_t fast32;                  // 0: the general forms only (PXZ_NO_EXPAND_FAST32)
	uint32_t *status;                 // set to 1 when a tile's stored size is 0 or exceeds its full size
	uint32_t quiet_empty;             // 1: tiles of stored size 0 x 0 are simply not written (tree::process: not this level's)
	const uint32_t *xmf16;            // kXmf16Levels * kXmf16Dw dwords (16x16 tiles, a convolution filter), or null
	const uint32_t *xmf64;            // kXmf64Levels * kXmf64Dw dwords (64x64 tiles, a convolution filter), or null
	uint32_t *list;                   // 16x16 flow: tiles expand16_kernel left to expand_kernel (status[1] counts them), or null
	uint32_t list_mode;               // expand_kernel: 1 = take the tiles of `list` (status[1] of them) instead of every tile
	FastDiv div_gpf, div_gcols;       // expand16_kernel: divisors for its 2x2 tile groups (groups per frame, group columns)
	uint32_t *big_scratch;            // tiles whose image exceeds LDS (round 4): one image of tile_dw dwords per wave of the grid in HBM
	uint32_t big_waves;               //   (0: the images are in LDS)
};

// Decode side: .pixlzr files -> tile values, sizes and pixel slots (pixlzr_index_kernel, qoi_decode_kernel)
struct DecodeArgs {
	const uint8_t *files;                   // the files back to back
	const unsigned long long *file_offsets; // n_frames + 1 byte offsets
	float *value;
	uint32_t *tile_w, *tile_h;
	uint8_t *slots;
	unsigned long long *rec_off;            // scratch, per tile: offset of the record's QOI body in files[]
	uint32_t *rec_len;                      // scratch, per tile: length of that body (0: unusable)
	uint32_t *perm, *bins;                  // scratch: tiles ordered by pixel count (as in the encoder), 64 counters
	uint32_t *status;                       // bit 1: malformed file / record
	uint32_t width, height, bw, bh, cols, rows, tiles_per_frame, n_tiles, n_frames, channels, slot_bytes;
	uint32_t edge_w, edge_h;
};

// RGB frames ride the RGBA fast paths: widen to RGBA (alpha 255) on the way in, narrow the tile slots on the way out
struct WidenArgs {
	const uint8_t *src;
	uint8_t *dst;
	uint64_t src_frame_stride, dst_frame_stride;
	uint32_t src_pitch, dst_pitch, width, height, n_frames;
};
struct NarrowArgs {
	const uint8_t *slots4;
	uint8_t *slots3;
	const uint32_t *w, *h;
	uint32_t n_tiles, slot4_bytes, slot3_bytes;
};

// tree::process (src/process/tree.rs:23-83), one level of the quad tree = the regular grid of that level's block size
struct TreeArgs {
	const uint8_t *src;           // the source frames (for the tiles that are handed back unchanged)
	uint8_t *dst;                 // RGBA output frames
	uint64_t src_frame_stride, dst_frame_stride;
	uint32_t src_pitch, dst_pitch, channels;
	const float *value;           // this level's grid: get_block_variance of every tile
	uint32_t *tile_w, *tile_h;    //   reduced sizes; set to 0 for tiles that are not pixelised at this level
	const uint8_t *parent_open;   // previous level's grid: 1 = the tile went on to this level (null at level 0: all tiles)
	uint8_t *open;                // this level's grid: 1 = the tile goes on to the next level
	uint32_t bw, bh, cols, rows, tiles_per_frame, n_tiles, edge_w, edge_h;
	uint32_t parent_cols, parent_tiles_per_frame;
	float threshold;              // |threshold|
	uint32_t positive;            // tree.rs:37: threshold >= 0 (deeper levels: always)
	uint32_t last;                // no further level: a tile that fails the test keeps its pixels (tree.rs:34-36)
};

// tree::process on rectangle lists (pxz_tree.hip): the tiles of one level of the recursion, wherever they lie
struct TreeRect {
	uint32_t x, y;      // origin in the frame
	uint16_t w, h;
	uint32_t frame;
};
// one axis table of the rectangle kernel: windows of `out` samples over `in` samples; up = 1: the table of the way back
// (PixlzrBlock::resize with the upscale flag set, block.rs:301-306)
struct TreeAxisEntry {
	uint16_t in, out;
	uint16_t up, window;
	uint32_t precision;
	uint32_t starts_off;   // into starts[] / sizes[] (Nearest: starts[] is the source index, no sizes / coefficients)
	uint32_t coeff_off;    // into coeffs[]: out * window weights
};
struct TreeRectArgs {
	const uint8_t *src;
	uint8_t *dst;                  // RGBA frames
	uint64_t src_frame_stride, dst_frame_stride;
	uint32_t src_pitch, dst_pitch, channels;
	const TreeRect *rects;
	uint32_t n_rects;
	float threshold;               // |threshold|
	uint32_t positive;             // tree.rs:37 (deeper levels: always)
	uint32_t next_bw, next_bh;     // block of the next level (tree.rs:73)
	uint32_t next_is_leaf;         // ... which is not above the minimum: a tile that goes on keeps its pixels (tree.rs:34-36)
	TreeRect *next_rects;
	uint32_t *next_count;
	uint32_t next_capacity;
	uint32_t filter_down, filter_up;
	const TreeAxisEntry *dir;
	uint32_t n_dir;
	const int32_t *starts;
	const int32_t *sizes;
	const int16_t *coeffs;
	float thresholds[kNumThresholds];
};

struct SynthArgs {
	uint8_t *dst;
	uint64_t frame_stride;
	uint32_t pitch, width, height, channels, n_frames, first_frame, dist;
};

}  // namespace pxz
