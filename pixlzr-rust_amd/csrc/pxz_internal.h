// pxz_internal.h — structures shared by the host runtime and the gfx950 kernels.
#pragma once
#include <stdint.h>

namespace pxz {

constexpr int kMaxLevel = 18;        // level exponents 0..17 (2^17 > any LDS-resident tile side)
constexpr int kNumThresholds = 32;   // thresholds for round(log2f(v)) >= -k, k = 0..31
constexpr int kWave = 64;

// One down-scaling table along one axis: in_size -> out_size.
// Convolution: bounds[bounds_off + 2*o] = first source index, [+1] = tap count;
//              coeffs[coeff_off + o*window + i] = i16 fixed-point weight.
// Nearest:     bounds[bounds_off + o] = source index.
struct AxisTab {
	uint32_t bounds_off;
	uint32_t coeff_off;
	uint16_t out_size;
	uint16_t window;
	uint16_t precision;
	uint16_t in_size;
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
	uint32_t edge_w, edge_h; // size of the last column / row of tiles
	uint32_t mode, filter;
	float factor;
	// outputs (device); out_px may be null (no resample), lod0/lod1 may be null
	float *value;
	uint32_t *out_w;
	uint32_t *out_h;
	uint8_t *out_px;
	float *lod0;
	float *lod1;
	uint32_t slot_bytes;     // bw*bh*channels
	// tables (device)
	const AxisTab *tabs;     // [axis 0=x,1=y][cls 0=full,1=edge][kMaxLevel]
	const uint16_t *bounds;
	const int16_t *coeffs;
	float thresholds[kNumThresholds];
	// LDS carve-up (in dwords, per tile)
	uint32_t lds_src_dw;     // bw*bh
	uint32_t lds_tmp_dw;     // ceil(bw/2)*bh
	uint32_t lds_lab_dw;     // 3*bw*bh in Oklab mode, else 0
};

struct SynthArgs {
	uint8_t *dst;
	uint64_t frame_stride;
	uint32_t pitch, width, height, channels, n_frames, first_frame, dist;
};

}  // namespace pxz
