// pxz_kernels.hip — gfx950 (MI355X / CDNA4) kernels of the pixlzr encode hot path.
//
// One fused kernel per batch of frames: every tile of the regular grid
// (reference src/split.rs:10-61) is staged ONCE from HBM into LDS, its
// level-of-detail value is reduced (reference src/operations.rs:26-126 /
// :192-259), the power-of-two target size is decided (operations.rs:140-156)
// and the tile is resampled out of the same LDS copy (reference
// src/data_types/block.rs:273-334 -> fast_image_resize convolution) into its
// output slot.  HBM traffic = each source byte read once + shrunk pixels and
// 12 B of metadata per tile written once.  No MFMA: integer/byte work, HBM-bound.
//
// Compiled with -ffp-contract=off: the f32 results of the Oklab detector are
// written into the bitstream, and the reference (Rust) never fuses a*b+c.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "pxz_internal.h"

namespace pxz {

// ---------------------------------------------------------------------------
// constant tables
// ---------------------------------------------------------------------------

// sRGB u8 -> linear f32 (bits): the 256-entry table palette 0.7.6 / fast-srgb8
// 1.0.0 use for `Srgb<u8>::into_linear()` (reference operations.rs:56-59).
__constant__ uint32_t kSrgbToLinearBits[256] = {
#include "srgb_lut.inc"
};

// fast_image_resize un-premultiply: recip[a] = ((255 << 9) / a + 1) >> 1
struct RecipAlphaTable {
	uint32_t v[256];
	constexpr RecipAlphaTable() : v{}
	{
		for (uint32_t a = 1; a < 256; ++a) v[a] = ((255u * 512u) / a + 1u) >> 1;
	}
};
__constant__ RecipAlphaTable kRecipAlpha = RecipAlphaTable();

// ---------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------

// Walks i = first, first+step, ... while tracking (row, col) = (i / width, i % width)
// without a division per element.
// n / d for a host-prepared divisor (round-up magic number): stays on the scalar unit for
// wave-uniform n, where a plain `/` would expand to ~25 VALU instructions.
__device__ __forceinline__ uint32_t fastdiv(uint32_t n, const FastDiv &d)
{
	const uint32_t t = __umulhi(n, d.mul);
	return (t + ((n - t) >> d.sh1)) >> d.sh2;
}

// n / d for n < 2^20, d >= 1 in five instructions instead of the ~30 of a general 32-bit division:
// x = (n + 0.5) / d is at least 0.5/d away from every integer, and the float product with the 1-ulp
// reciprocal is within x * 2^-22 of it, which is below 0.5/d for every n < 2^21.
__device__ __forceinline__ uint32_t small_div(uint32_t n, uint32_t d)
{
	return (uint32_t)(((float)n + 0.5f) * __builtin_amdgcn_rcpf((float)d));
}

struct RowWalker {
	uint32_t row, col, drow, dcol, width;
	__device__ RowWalker(uint32_t first, uint32_t step, uint32_t width_) : width(width_)
	{
		// first and step are lane / thread counts (<= 1024 + a tile's pixel count < 2^20)
		row = small_div(first, width_);
		col = first - row * width_;
		drow = small_div(step, width_);
		dcol = step - drow * width_;
	}
	__device__ void next()
	{
		row += drow;
		col += dcol;
		if (col >= width) {
			col -= width;
			++row;
		}
	}
};

template <int NW>
__device__ __forceinline__ void tile_sync()
{
	if constexpr (NW == 1) {
		// one wave owns the tile: LDS operations of a wave execute in order,
		// only the compiler must not move them across this point
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
	} else {
		__syncthreads();
	}
}

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
#pragma unroll
	for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
	return v;
}
__device__ __forceinline__ uint32_t wave_and_u32(uint32_t v)
{
#pragma unroll
	for (int off = 32; off >= 1; off >>= 1) v &= __shfl_xor(v, off, 64);
	return v;
}

__device__ __forceinline__ uint32_t fbits(float f) { return __float_as_uint(f); }

// reference operations.rs:128-138
__device__ __forceinline__ float parse_value(float value)
{
	if ((fbits(value) >> 31) == 0) return value;
	float t = 1.0f + value;
	// f32::max(t, 0.0): NaN -> 0.0
	float v = (t != t) ? 0.0f : (t > 0.0f ? t : 0.0f);
	return v;  // never negative-signed here, so the `else 1f32` arm is unreachable
}

// n = min(round(log2f(v)), 0) as the exponent m = -n in [0, 32]; thresholds[k] is
// the smallest float with round(log2f(v)) >= -k (host-built with the platform log2f).
__device__ __forceinline__ uint32_t level_exponent(float v, const float *thresholds)
{
	uint32_t m = 0;
#pragma unroll
	for (int k = 0; k < kNumThresholds; ++k) m += (v < thresholds[k]) ? 1u : 0u;
	return m;
}

// ceil(max(size * 2^-m, 1)) (operations.rs:150-151)
__device__ __forceinline__ uint32_t reduced_size(uint32_t size, uint32_t m)
{
	if (m >= 31) return 1;
	uint32_t r = (uint32_t)(((uint64_t)size + ((1ull << m) - 1ull)) >> m);
	return r < 1 ? 1 : r;
}

// f32::hypot as glibc computes it (double sqrt of the exact squares' sum)
__device__ __forceinline__ float hypot_f32(float x, float y)
{
	double dx = (double)x, dy = (double)y;
	return (float)__dsqrt_rn(dx * dx + dy * dy);
}

// f32::cbrt = glibc 2.35 cbrtf: frexp, quadratic seed and one Halley step in double.
__device__ __forceinline__ float cbrt_f32(float x)
{
	if (x == 0.0f) return x + x;
	int xe;
	float xm = frexpf(fabsf(x), &xe);
	float u = (float)(0.492659620528969547 +
	                  (0.697570460207922770 - 0.191502161678719066 * (double)xm) * (double)xm);
	float t2 = u * u * u;
	int r = xe % 3;  // C semantics, sign follows xe
	double scale = r == 0 ? 1.0
	             : r == 1 ? 1.2599210498948731648
	             : r == 2 ? 1.5874010519681994748
	             : r == -1 ? 1.0 / 1.2599210498948731648
	                       : 1.0 / 1.5874010519681994748;
	float ym = (float)((double)u * ((double)t2 + 2.0 * (double)xm) / (2.0 * (double)t2 + (double)xm) * scale);
	return ldexpf(x > 0.0f ? ym : -ym, xe / 3);
}

// palette 0.7.6 LinSrgb<f32> -> Oklab<f32> (Ottosson's matrices, left-to-right f32 sums)
__device__ __forceinline__ void oklab_from_rgba(uint32_t px, float &L, float &A, float &B)
{
	float r = __uint_as_float(kSrgbToLinearBits[px & 255u]);
	float g = __uint_as_float(kSrgbToLinearBits[(px >> 8) & 255u]);
	float b = __uint_as_float(kSrgbToLinearBits[(px >> 16) & 255u]);
	float l = 0.4122214708f * r + 0.5363325363f * g + 0.0514459929f * b;
	float m = 0.2119034982f * r + 0.6806995451f * g + 0.1073969566f * b;
	float s = 0.0883024619f * r + 0.2817188376f * g + 0.6299787005f * b;
	float l_ = cbrt_f32(l), m_ = cbrt_f32(m), s_ = cbrt_f32(s);
	L = 0.2104542553f * l_ + 0.7936177850f * m_ - 0.0040720468f * s_;
	A = 1.9779984951f * l_ - 2.4285922050f * m_ + 0.4505937099f * s_;
	B = 0.0259040371f * l_ + 0.7827717662f * m_ - 0.8086757660f * s_;
}

// fast_image_resize alpha premultiply: mul_div_255
__device__ __forceinline__ uint32_t mul_div_255(uint32_t a, uint32_t b)
{
	uint32_t t = a * b + 128u;
	return ((t >> 8) + t) >> 8;
}
__device__ __forceinline__ uint32_t premultiply(uint32_t px)
{
	uint32_t al = px >> 24;
	return mul_div_255(px & 255u, al) | (mul_div_255((px >> 8) & 255u, al) << 8) |
	       (mul_div_255((px >> 16) & 255u, al) << 16) | (al << 24);
}
__device__ __forceinline__ uint32_t unpremultiply(uint32_t px)
{
	uint32_t al = px >> 24;
	uint32_t rc = kRecipAlpha.v[al];
	uint32_t r = ((px & 255u) * rc + 128u) >> 8;
	uint32_t g = (((px >> 8) & 255u) * rc + 128u) >> 8;
	uint32_t b = (((px >> 16) & 255u) * rc + 128u) >> 8;
	r = r > 255u ? 255u : r;
	g = g > 255u ? 255u : g;
	b = b > 255u ? 255u : b;
	return r | (g << 8) | (b << 16) | (al << 24);
}

__device__ __forceinline__ uint32_t clip8(int32_t acc, int precision)
{
	int32_t v = acc >> precision;
	return (uint32_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

template <int C>
__device__ __forceinline__ void store_pixel(uint8_t *dst, uint32_t index, uint32_t px)
{
	if constexpr (C == 4) {
		reinterpret_cast<uint32_t *>(dst)[index] = px;
	} else {
		uint8_t *p = dst + (size_t)index * 3;
		p[0] = (uint8_t)px;
		p[1] = (uint8_t)(px >> 8);
		p[2] = (uint8_t)(px >> 16);
	}
}

// ---------------------------------------------------------------------------
// packed-math helpers
// ---------------------------------------------------------------------------
typedef short short2v __attribute__((ext_vector_type(2)));
typedef unsigned short ushort2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ ushort2v us2(uint32_t v) { return __builtin_bit_cast(ushort2v, v); }
__device__ __forceinline__ uint32_t u32(ushort2v v) { return __builtin_bit_cast(uint32_t, v); }

// acc + a.lo*b.lo + a.hi*b.hi on i16 halves (v_dot2_i32_i16)
__device__ __forceinline__ int32_t dot2(uint32_t a, uint32_t b, int32_t acc)
{
	return __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, a), __builtin_bit_cast(short2v, b), acc, false);
}
// acc + |a.lo-b.lo| + |a.hi-b.hi| on u16 halves (v_sad_u16)
__device__ __forceinline__ uint32_t sad16(uint32_t a, uint32_t b, uint32_t acc)
{
	return __builtin_amdgcn_sad_u16(a, b, acc);
}
// horizontal 1-2-1 smoothing of the pixel pairs D0=(x,x+1), D1=(x+2,x+3): (x+2(x+1)+(x+2), (x+1)+2(x+2)+(x+3))
__device__ __forceinline__ uint32_t smooth121(uint32_t d0, uint32_t d1)
{
	const uint32_t mid = __builtin_amdgcn_alignbit(d1, d0, 16);  // (x+1, x+2)
	return u32(us2(mid) * (ushort2v)(2) + (us2(d0) + us2(d1)));
}

// v_pk_mad_u16: a*b+c on both u16 halves (the compiler prefers shift+add, one VALU op more)
__device__ __forceinline__ uint32_t pk_mad_u16(uint32_t a, uint32_t b, uint32_t c)
{
	uint32_t d;
	asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
	return d;
}

template <int CTRL>
__device__ __forceinline__ uint32_t dpp_mov(uint32_t v)
{
	return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}
// sum over the wave, result in an SGPR: butterfly inside each row of 16 lanes (DPP), then 4 readlanes
__device__ __forceinline__ uint32_t wave_sum_sgpr(uint32_t v)
{
	v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
	v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
	v += dpp_mov<0x124>(v);  // row_ror:4
	v += dpp_mov<0x128>(v);  // row_ror:8
	return (uint32_t)__builtin_amdgcn_readlane((int)v, 0) + (uint32_t)__builtin_amdgcn_readlane((int)v, 16) +
	       (uint32_t)__builtin_amdgcn_readlane((int)v, 32) + (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
}
__device__ __forceinline__ uint32_t wave_and_sgpr(uint32_t v)
{
	v &= dpp_mov<0xB1>(v);
	v &= dpp_mov<0x4E>(v);
	v &= dpp_mov<0x124>(v);
	v &= dpp_mov<0x128>(v);
	return (uint32_t)__builtin_amdgcn_readlane((int)v, 0) & (uint32_t)__builtin_amdgcn_readlane((int)v, 16) &
	       (uint32_t)__builtin_amdgcn_readlane((int)v, 32) & (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
}

__device__ __forceinline__ uint32_t level_count(uint32_t key, const uint32_t *breaks, uint32_t asc)
{
	key = __builtin_amdgcn_readfirstlane(key);  // tile-uniform: keep the compares on the scalar unit
	uint32_t m = 0;
#pragma unroll
	for (int j = 0; j < kMaxLevel; ++j) m += ((key < breaks[j]) != (asc != 0)) ? 1u : 0u;
	return m;
}

// ---------------------------------------------------------------------------
// resample fast path: full, opaque 32x32 RGBA tile, both passes needed.
// Compile-time geometry (plane row stride 18 dwords -- 16 + 2 of bank skew --, plane 576
// dwords, transposed plane [ox][18 dwords] = 288 dwords) turns every LDS address into
// base+immediate and keeps the row groups / output columns on distinct banks.
// Work items are spread over all 64 lanes: when there are fewer than 64 outputs the
// filter window itself is split over 2..8 lanes and summed with DPP.
// ---------------------------------------------------------------------------
constexpr uint32_t kRS32 = 18, kPD32 = 18 * 32, kHS32 = 18;
// transposed planes of the dot2 two-pass form: up to 8 output columns (16/8/4-px outputs on both axes go
// through the matrix cores; the rare 16 x (2|1) tile is handed to the worklist)
constexpr uint32_t kTD32 = 8 * 18;
constexpr uint32_t kOut32 = 512;  // dwords of the region after the planes: transposed planes / parked output pixels

template <int LPI>
__device__ __forceinline__ int32_t group_sum(int32_t v)
{
	if constexpr (LPI == 8) v += (int32_t)dpp_mov<0x141>((uint32_t)v);  // row_half_mirror: i <-> 7-i
	if constexpr (LPI >= 4) v += (int32_t)dpp_mov<0x4E>((uint32_t)v);   // quad_perm [2,3,0,1]
	if constexpr (LPI >= 2) v += (int32_t)dpp_mov<0xB1>((uint32_t)v);   // quad_perm [1,0,3,2]
	return v;
}

// A lane's table row: header {first quad, quads, weight sum} + up to 8 quads of packed weights,
// fetched as independent 16-byte loads (one memory latency instead of one per window step).
struct RowRegs {
	uint32_t fq, nq;
	int32_t ksum;
	uint32_t k[16];
};
__device__ __forceinline__ void load_row(const uint32_t *rowp, RowRegs &r)
{
	const uint4 h = *reinterpret_cast<const uint4 *>(rowp);
	const uint4 c0 = *reinterpret_cast<const uint4 *>(rowp + 4), c1 = *reinterpret_cast<const uint4 *>(rowp + 8);
	const uint4 c2 = *reinterpret_cast<const uint4 *>(rowp + 12), c3 = *reinterpret_cast<const uint4 *>(rowp + 16);
	r.fq = h.x;
	r.nq = h.y;
	r.ksum = (int32_t)h.z;
	r.k[0] = c0.x; r.k[1] = c0.y; r.k[2] = c0.z; r.k[3] = c0.w;
	r.k[4] = c1.x; r.k[5] = c1.y; r.k[6] = c1.z; r.k[7] = c1.w;
	r.k[8] = c2.x; r.k[9] = c2.y; r.k[10] = c2.z; r.k[11] = c2.w;
	r.k[12] = c3.x; r.k[13] = c3.y; r.k[14] = c3.z; r.k[15] = c3.w;
}

// horizontal pass, one source row per item: item = (ox, y), the window split over LPI lanes
// (each lane takes quads part, part+LPI, ...; weights beyond the window are zero in the table)
template <int LPI>
__device__ __forceinline__ void fast32_h_rows(const uint32_t *trows, const AxisTab &tx, const uint32_t *s_pl, uint32_t *s_tmp,
                                              uint32_t lane, uint32_t nw, uint32_t lgx, uint32_t in_rows = 32u)
{
	constexpr int QPL = 8 / LPI;  // quads per lane
	const uint32_t item = lane / LPI, part = lane % LPI;
	const bool live = item < nw * in_rows;
	const uint32_t ox = item & (nw - 1u), y = live ? item >> lgx : 0u;
	const uint32_t *rowp = trows + tx.rows_off + ox * tx.row_stride;
	const uint4 hdr = *reinterpret_cast<const uint4 *>(rowp);
	uint2 kk[QPL];  // table rows are zero-padded to 8 quads (get_tables): quads past the window carry no weight
#pragma unroll
	for (int j = 0; j < QPL; ++j) kk[j] = *reinterpret_cast<const uint2 *>(rowp + 4 + 2 * (part + j * LPI));
	const uint32_t *row = s_pl + y * kRS32 + hdr.x * 2u + part * 2u;
	int32_t a0 = 0, a1 = 0, a2 = 0;
#pragma unroll
	for (int j = 0; j < QPL; ++j) {
		if (LPI > 1 || (uint32_t)j < tx.wquads) {  // LPI == 1: wave-uniform trim of the zero-weight tail
			const uint2 d0 = *reinterpret_cast<const uint2 *>(row + j * LPI * 2);
			const uint2 d1 = *reinterpret_cast<const uint2 *>(row + kPD32 + j * LPI * 2);
			const uint2 d2 = *reinterpret_cast<const uint2 *>(row + 2 * kPD32 + j * LPI * 2);
			a0 = dot2(d0.y, kk[j].y, dot2(d0.x, kk[j].x, a0));
			a1 = dot2(d1.y, kk[j].y, dot2(d1.x, kk[j].x, a1));
			a2 = dot2(d2.y, kk[j].y, dot2(d2.x, kk[j].x, a2));
		}
	}
	a0 = group_sum<LPI>(a0);
	a1 = group_sum<LPI>(a1);
	a2 = group_sum<LPI>(a2);
	if (live && part == 0) {
		const int prec = tx.precision;
		const int32_t init = 1 << (prec - 1);
		uint16_t *t16 = reinterpret_cast<uint16_t *>(s_tmp) + ox * (2 * kHS32) + y;
		t16[0] = (uint16_t)clip8(a0 + init, prec);
		t16[2 * kTD32] = (uint16_t)clip8(a1 + init, prec);
		t16[4 * kTD32] = (uint16_t)clip8(a2 + init, prec);
	}
}

// vertical pass over the transposed planes, window split over LPI lanes; item = (oy fastest, ox),
// so that a lane keeps the same output row (= the same table row) across iterations
template <int LPI>
__device__ __forceinline__ void fast32_v(const uint32_t *trows, const AxisTab &tx, const AxisTab &ty, const uint32_t *s_tmp,
                                         uint32_t lane, uint32_t nw, uint32_t nh, uint32_t *out)
{
	constexpr int QPL = 8 / LPI;
	const uint32_t items = nw * nh;
	const uint32_t lgy = 31u - (uint32_t)__builtin_clz(nh);
	const uint32_t part = lane % LPI, item0 = lane / LPI;
	const uint32_t oy = item0 & (nh - 1u);  // invariant: the item step (64/LPI) is a multiple of nh
	const uint32_t *rowp = trows + ty.rows_off + oy * ty.row_stride;
	const uint4 hdr = *reinterpret_cast<const uint4 *>(rowp);
	const uint32_t wq = ty.wquads;
	uint2 kk[QPL];  // zero-padded rows: no guard
#pragma unroll
	for (int j = 0; j < QPL; ++j) kk[j] = *reinterpret_cast<const uint2 *>(rowp + 4 + 2 * (part + j * LPI));
	const int px_ = tx.precision, py = ty.precision;
	const int32_t ix = 1 << (px_ - 1), iy = 1 << (py - 1);
	for (uint32_t item = item0; item < ((items + 63u / LPI) & ~(64u / LPI - 1u)); item += 64u / LPI) {
		const bool live = item < items;
		const uint32_t ox = live ? item >> lgy : 0u;
		const uint32_t *colp = s_tmp + ox * kHS32 + hdr.x * 2u + part * 2u;
		int32_t a0 = 0, a1 = 0, a2 = 0;
#pragma unroll
		for (int j = 0; j < QPL; ++j) {
			if (LPI > 1 || (uint32_t)j < wq) {  // LPI == 1: wave-uniform trim of the zero-weight tail
				const uint2 d0 = *reinterpret_cast<const uint2 *>(colp + j * LPI * 2);
				const uint2 d1 = *reinterpret_cast<const uint2 *>(colp + kTD32 + j * LPI * 2);
				const uint2 d2 = *reinterpret_cast<const uint2 *>(colp + 2 * kTD32 + j * LPI * 2);
				a0 = dot2(d0.y, kk[j].y, dot2(d0.x, kk[j].x, a0));
				a1 = dot2(d1.y, kk[j].y, dot2(d1.x, kk[j].x, a1));
				a2 = dot2(d2.y, kk[j].y, dot2(d2.x, kk[j].x, a2));
			}
		}
		a0 = group_sum<LPI>(a0);
		a1 = group_sum<LPI>(a1);
		a2 = group_sum<LPI>(a2);
		if (live && part == 0) {
			// opaque tile: alpha is the convolution of the constant 255 = the windows' weight sums
			const int32_t ksx = (int32_t)trows[tx.rows_off + ox * tx.row_stride + 2];
			const int32_t ah = (int32_t)clip8(ix + 255 * ksx, px_);
			const uint32_t al = clip8(iy + ah * (int32_t)hdr.z, py);
			uint32_t px = clip8(a0 + iy, py) | (clip8(a1 + iy, py) << 8) | (clip8(a2 + iy, py) << 16) | (al << 24);
			if (al != 255u) px = unpremultiply(px);
			out[oy * nw + ox] = px;
		}
	}
}

// single-pass cases of the fast path (one axis keeps its 32 samples), opaque tile.
// Vertical only: item = (pair of columns, output row) straight on the [y][x] planes; the (row j,
// row j+1) sample pairs dot2 needs are built with two perms per column pair and row pair.
__device__ __forceinline__ void fast32_v_only(const uint32_t *trows, const AxisTab &ty, const uint32_t *s_pl, uint32_t lane,
                                              uint32_t nh, uint32_t *out)
{
	const uint32_t lgy = 31u - (uint32_t)__builtin_clz(nh);
	const uint32_t oy = lane & (nh - 1u);  // invariant per lane: 64 is a multiple of nh
	RowRegs r;
	load_row(trows + ty.rows_off + oy * ty.row_stride, r);
	const int prec = ty.precision;
	const int32_t init = 1 << (prec - 1);
	const uint32_t al = clip8(init + 255 * r.ksum, prec);  // constant-255 alpha through the same window
	const uint32_t wq = ty.wquads;
	for (uint32_t i = lane; i < 16u * nh; i += 64u) {
		const uint32_t qx = i >> lgy;  // column pair
		const uint32_t *colp = s_pl + (r.fq * 4u) * kRS32 + qx;
		int32_t acc[3][2];
#pragma unroll
		for (int c = 0; c < 3; ++c) acc[c][0] = acc[c][1] = init;
#pragma unroll
		for (int q = 0; q < 8; ++q) {
			if ((uint32_t)q < wq) {
#pragma unroll
				for (int c = 0; c < 3; ++c) {
					const uint32_t *p = colp + c * kPD32 + (q * 4) * (int)kRS32;
					const uint32_t r0 = p[0], r1 = p[kRS32], r2 = p[2 * kRS32], r3 = p[3 * kRS32];
					const uint32_t l01 = __builtin_amdgcn_perm(r1, r0, 0x05040100u), l23 = __builtin_amdgcn_perm(r3, r2, 0x05040100u);
					const uint32_t h01 = __builtin_amdgcn_perm(r1, r0, 0x07060302u), h23 = __builtin_amdgcn_perm(r3, r2, 0x07060302u);
					acc[c][0] = dot2(l23, r.k[2 * q + 1], dot2(l01, r.k[2 * q], acc[c][0]));
					acc[c][1] = dot2(h23, r.k[2 * q + 1], dot2(h01, r.k[2 * q], acc[c][1]));
				}
			}
		}
		uint2 o;
		o.x = clip8(acc[0][0], prec) | (clip8(acc[1][0], prec) << 8) | (clip8(acc[2][0], prec) << 16) | (al << 24);
		o.y = clip8(acc[0][1], prec) | (clip8(acc[1][1], prec) << 8) | (clip8(acc[2][1], prec) << 16) | (al << 24);
		if (al != 255u) {
			o.x = unpremultiply(o.x);
			o.y = unpremultiply(o.y);
		}
		*reinterpret_cast<uint2 *>(out + (oy * 16u + qx) * 2u) = o;  // pixels (2qx, 2qx+1) of output row oy, row length 32
	}
}

// Horizontal only: item = (output column, pair of rows), results go straight to the slot.
__device__ __forceinline__ void fast32_h_only(const uint32_t *trows, const AxisTab &tx, const uint32_t *s_pl, uint32_t lane,
                                              uint32_t nw, uint32_t *out)
{
	const uint32_t lgx = 31u - (uint32_t)__builtin_clz(nw);
	const uint32_t ox = lane & (nw - 1u);
	RowRegs r;
	load_row(trows + tx.rows_off + ox * tx.row_stride, r);
	const int prec = tx.precision;
	const int32_t init = 1 << (prec - 1);
	const uint32_t al = clip8(init + 255 * r.ksum, prec);
	const uint32_t wq = tx.wquads;
	for (uint32_t i = lane; i < nw * 16u; i += 64u) {
		const uint32_t yp = i >> lgx;
		const uint32_t *row = s_pl + yp * (2 * kRS32) + r.fq * 2u;
		int32_t acc[3][2];
#pragma unroll
		for (int c = 0; c < 3; ++c) acc[c][0] = acc[c][1] = init;
#pragma unroll
		for (int q = 0; q < 8; ++q) {
			if ((uint32_t)q < wq) {
#pragma unroll
				for (int c = 0; c < 3; ++c) {
					const uint2 d = *reinterpret_cast<const uint2 *>(row + c * kPD32 + q * 2);
					const uint2 e = *reinterpret_cast<const uint2 *>(row + c * kPD32 + kRS32 + q * 2);
					acc[c][0] = dot2(d.y, r.k[2 * q + 1], dot2(d.x, r.k[2 * q], acc[c][0]));
					acc[c][1] = dot2(e.y, r.k[2 * q + 1], dot2(e.x, r.k[2 * q], acc[c][1]));
				}
			}
		}
#pragma unroll
		for (uint32_t rr = 0; rr < 2; ++rr) {
			uint32_t px = clip8(acc[0][rr], prec) | (clip8(acc[1][rr], prec) << 8) | (clip8(acc[2][rr], prec) << 16) | (al << 24);
			if (al != 255u) px = unpremultiply(px);
			out[(2u * yp + rr) * nw + ox] = px;
		}
	}
}

// horizontal pass, item = (ox, pair of rows): nw*16 items, nw/4 per lane, the same ox (table row)
// every time; WQ = quads per window (weights past a lane's own window are zero in the table)
template <int WQ>
__device__ __forceinline__ void fast32_h_pairs(const uint32_t *trows, const AxisTab &tx, const uint32_t *s_pl, uint32_t *s_tmp,
                                               uint32_t lane, uint32_t nw, uint32_t lgx, uint32_t row_pairs = 16u)
{
	const uint32_t ox = lane & (nw - 1u);
	RowRegs r;
	load_row(trows + tx.rows_off + ox * tx.row_stride, r);
	const int prec = tx.precision;
	const int32_t init = 1 << (prec - 1);
	for (uint32_t i = lane; i < nw * row_pairs; i += 64u) {
		const uint32_t yp = i >> lgx;
		const uint32_t *row = s_pl + yp * (2 * kRS32) + r.fq * 2u;
		int32_t a0 = init, a1 = init, a2 = init, b0 = init, b1 = init, b2 = init;
#pragma unroll
		for (int q = 0; q < WQ; ++q) {
			const uint2 d0 = *reinterpret_cast<const uint2 *>(row + q * 2);
			const uint2 e0 = *reinterpret_cast<const uint2 *>(row + kRS32 + q * 2);
			const uint2 d1 = *reinterpret_cast<const uint2 *>(row + kPD32 + q * 2);
			const uint2 e1 = *reinterpret_cast<const uint2 *>(row + kPD32 + kRS32 + q * 2);
			const uint2 d2 = *reinterpret_cast<const uint2 *>(row + 2 * kPD32 + q * 2);
			const uint2 e2 = *reinterpret_cast<const uint2 *>(row + 2 * kPD32 + kRS32 + q * 2);
			const uint32_t k01 = r.k[2 * q], k23 = r.k[2 * q + 1];
			a0 = dot2(d0.y, k23, dot2(d0.x, k01, a0));
			b0 = dot2(e0.y, k23, dot2(e0.x, k01, b0));
			a1 = dot2(d1.y, k23, dot2(d1.x, k01, a1));
			b1 = dot2(e1.y, k23, dot2(e1.x, k01, b1));
			a2 = dot2(d2.y, k23, dot2(d2.x, k01, a2));
			b2 = dot2(e2.y, k23, dot2(e2.x, k01, b2));
		}
		uint32_t *t = s_tmp + ox * kHS32 + yp;
		t[0] = clip8(a0, prec) | (clip8(b0, prec) << 16);
		t[kTD32] = clip8(a1, prec) | (clip8(b1, prec) << 16);
		t[2 * kTD32] = clip8(a2, prec) | (clip8(b2, prec) << 16);
	}
}

__device__ __forceinline__ void resample_fast32_hv(const uint32_t *trows, const AxisTab &tx, const AxisTab &ty,
                                                   uint32_t *s_pl, uint32_t *s_tmp, uint32_t lane, uint32_t nw,
                                                   uint32_t nh, uint32_t *out)
{
	const uint32_t lgx = 31u - (uint32_t)__builtin_clz(nw);  // nw is a power of two <= 16
	if (nw >= 4) {
		switch (tx.wquads) {  // straight-line window code per size: no branches between LDS reads and dot2s
		case 1: fast32_h_pairs<1>(trows, tx, s_pl, s_tmp, lane, nw, lgx); break;
		case 2: fast32_h_pairs<2>(trows, tx, s_pl, s_tmp, lane, nw, lgx); break;
		case 3: fast32_h_pairs<3>(trows, tx, s_pl, s_tmp, lane, nw, lgx); break;
		case 4: fast32_h_pairs<4>(trows, tx, s_pl, s_tmp, lane, nw, lgx); break;
		case 5: fast32_h_pairs<5>(trows, tx, s_pl, s_tmp, lane, nw, lgx); break;
		case 6: fast32_h_pairs<6>(trows, tx, s_pl, s_tmp, lane, nw, lgx); break;
		case 7: fast32_h_pairs<7>(trows, tx, s_pl, s_tmp, lane, nw, lgx); break;
		default: fast32_h_pairs<8>(trows, tx, s_pl, s_tmp, lane, nw, lgx); break;
		}
	} else if (nw == 2) {
		fast32_h_rows<1>(trows, tx, s_pl, s_tmp, lane, nw, lgx);
	} else {
		fast32_h_rows<2>(trows, tx, s_pl, s_tmp, lane, nw, lgx);
	}
	tile_sync<1>();
	const uint32_t items = nw * nh;
	if (items >= 64u) fast32_v<1>(trows, tx, ty, s_tmp, lane, nw, nh, out);
	else if (items >= 32u) fast32_v<2>(trows, tx, ty, s_tmp, lane, nw, nh, out);
	else if (items >= 16u) fast32_v<4>(trows, tx, ty, s_tmp, lane, nw, nh, out);
	else fast32_v<8>(trows, tx, ty, s_tmp, lane, nw, nh, out);
}

// The same for a 16x16 tile that sits somewhere inside the 32x32 LDS image (s_pl points at its first pixel
// pair): windows of at most 4 quads, 16 source rows, nw and nh in {8, 4, 2, 1}.
__device__ __forceinline__ void resample_fast16_hv(const uint32_t *trows, const AxisTab &tx, const AxisTab &ty, const uint32_t *s_pl,
                                                   uint32_t *s_tmp, uint32_t lane, uint32_t nw, uint32_t nh, uint32_t *out)
{
	const uint32_t lgx = 31u - (uint32_t)__builtin_clz(nw);
	if (nw >= 4) {
		switch (tx.wquads) {
		case 1: fast32_h_pairs<1>(trows, tx, s_pl, s_tmp, lane, nw, lgx, 8u); break;
		case 2: fast32_h_pairs<2>(trows, tx, s_pl, s_tmp, lane, nw, lgx, 8u); break;
		case 3: fast32_h_pairs<3>(trows, tx, s_pl, s_tmp, lane, nw, lgx, 8u); break;
		default: fast32_h_pairs<4>(trows, tx, s_pl, s_tmp, lane, nw, lgx, 8u); break;
		}
	} else if (nw == 2) {
		fast32_h_rows<2>(trows, tx, s_pl, s_tmp, lane, nw, lgx, 16u);
	} else {
		fast32_h_rows<4>(trows, tx, s_pl, s_tmp, lane, nw, lgx, 16u);
	}
	tile_sync<1>();
	const uint32_t items = nw * nh;
	if (items >= 64u) fast32_v<1>(trows, tx, ty, s_tmp, lane, nw, nh, out);
	else if (items >= 32u) fast32_v<2>(trows, tx, ty, s_tmp, lane, nw, nh, out);
	else if (items >= 16u) fast32_v<4>(trows, tx, ty, s_tmp, lane, nw, nh, out);
	else fast32_v<8>(trows, tx, ty, s_tmp, lane, nw, nh, out);
	tile_sync<1>();  // the next tile of the group reuses the transposed planes
}

// ---------------------------------------------------------------------------
// The same two-pass convolution on the matrix cores, for 32x32 -> nw x nh with nw, nh in {16, 8}:
// the classes whose windows are long AND whose outputs are many, i.e. where the dot2 form above
// spends the most vector instructions.  Integer-exact: v_mfma_i32_16x16x32_i8 multiplies signed bytes,
// so a pixel enters as p - 128 and an i16 weight as two signed bytes K = 256 K_hi + K_lo:
//     sum p K = 256 sum (p-128) K_hi + sum (p-128) K_lo + 128 sum K
// (two products per output block; the last term and the rounding half come in as the C operand).
//
//   horizontal  T[y][ox] = clip8(sum_x P[y][x] Kx[ox][x])   D = A B: A = pixel rows (16 per product),
//               B = weights [x][ox]; the accumulator holds column ox = lane & 15, rows 4g + r (g = lane >> 4)
//   vertical    O[oy][ox] = clip8(sum_y Ky[oy][y] T[y][ox]) D = A B: A = weights [oy][y], B = T.
// T never leaves the registers: after the two horizontal products (rows 0..15, 16..31) lane (ox, g)
// holds T[4g + r][ox] and T[16 + 4g + r][ox], which IS a B operand whose k slots are the rows
// src(g, j) (pxz_internal.h) -- so the weight operand is stored with its k slots in that order, and the
// pixel operand of the horizontal product reads its 8 source columns in that order too (one table
// serves both passes, the axes of a 32x32 tile being alike).
// ---------------------------------------------------------------------------
typedef int v4i32 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int BYTE>
__device__ __forceinline__ void put_byte_shr(uint32_t &dst, uint32_t value, uint32_t shift)
{
	// dst.byte[BYTE] = (value >> shift) & 0xff, other bytes kept: one SDWA shift
	if constexpr (BYTE == 0)
		asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(dst) : "v"(shift), "v"(value));
	else if constexpr (BYTE == 1)
		asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(dst) : "v"(shift), "v"(value));
	else if constexpr (BYTE == 2)
		asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(dst) : "v"(shift), "v"(value));
	else
		asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(dst) : "v"(shift), "v"(value));
}

__device__ __forceinline__ uint32_t clamp_fixed(int32_t hi, int32_t lo, int32_t top)
{
	const int32_t v = (int32_t)(((uint32_t)hi << 8) + (uint32_t)lo);  // 256 * hi + lo (the bias is already in lo)
	int32_t r;  // clip8 before its shift: median of (v, 0, top); top is a run-time value, so spell the instruction
	asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(v), "s"(top));
	return (uint32_t)r;
}

// NCH = 3: opaque tile, alpha from the weight sums.  NCH = 4: the planes hold premultiplied colours and the
// alpha plane; all four are convolved and every output pixel is un-premultiplied (fir's U8x4 path).
template <int NCH>
__device__ __forceinline__ void resample_mfma32(const uint32_t *s_tab, const AxisTab &tx, const AxisTab &ty, const uint32_t *s_pl,
                                                uint32_t lane, uint32_t nw, uint32_t nh, uint32_t *out)
{
	const uint32_t o = lane & 15u, g = lane >> 4;
	const uint32_t *mx = s_tab + tx.mf_off, *my = s_tab + ty.mf_off;
	const long kx_lo = *reinterpret_cast<const long *>(mx + 2u * lane), kx_hi = *reinterpret_cast<const long *>(mx + 128u + 2u * lane);
	const long ky_lo = *reinterpret_cast<const long *>(my + 2u * lane), ky_hi = *reinterpret_cast<const long *>(my + 128u + 2u * lane);
	const int32_t bx = (int32_t)mx[256u + o];
	const v4i32 cx = {bx, bx, bx, bx};
	const v4i32 cy = *reinterpret_cast<const v4i32 *>(my + 256u + 4u * g);
	const v4i32 zero = {0, 0, 0, 0};
	const uint32_t px_ = tx.precision, py = ty.precision;
	const int32_t top_x = (int32_t)((256u << px_) - 1u), top_y = (int32_t)((256u << py) - 1u);
	uint32_t pix[4] = {0xff000000u, 0xff000000u, 0xff000000u, 0xff000000u};
	const uint32_t *rowp = s_pl + o * kRS32 + 2u * g;
#pragma unroll
	for (uint32_t c = 0; c < (uint32_t)NCH; ++c) {
		uint32_t t[2];
#pragma unroll
		for (uint32_t mb = 0; mb < 2; ++mb) {
			const uint32_t *row = rowp + c * kPD32 + mb * (16u * kRS32);
			const uint2 d0 = *reinterpret_cast<const uint2 *>(row);       // columns 4g .. 4g+3
			const uint2 d1 = *reinterpret_cast<const uint2 *>(row + 8u);  // columns 16+4g .. 16+4g+3
			const uint32_t a0 = __builtin_amdgcn_perm(d0.y, d0.x, 0x06040200u) ^ 0x80808080u;
			const uint32_t a1 = __builtin_amdgcn_perm(d1.y, d1.x, 0x06040200u) ^ 0x80808080u;
			const long av = (long)(((unsigned long long)a1 << 32) | (unsigned long long)a0);
			const v4i32 lo = __builtin_amdgcn_mfma_i32_16x16x32_i8(av, kx_lo, cx, 0, 0, 0);
			const v4i32 hi = __builtin_amdgcn_mfma_i32_16x16x32_i8(av, kx_hi, zero, 0, 0, 0);
			uint32_t packed = 0;
			put_byte_shr<0>(packed, clamp_fixed(hi[0], lo[0], top_x), px_);
			put_byte_shr<1>(packed, clamp_fixed(hi[1], lo[1], top_x), px_);
			put_byte_shr<2>(packed, clamp_fixed(hi[2], lo[2], top_x), px_);
			put_byte_shr<3>(packed, clamp_fixed(hi[3], lo[3], top_x), px_);
			t[mb] = packed ^ 0x80808080u;
		}
		const long tv = (long)(((unsigned long long)t[1] << 32) | (unsigned long long)t[0]);
		const v4i32 lo = __builtin_amdgcn_mfma_i32_16x16x32_i8(ky_lo, tv, cy, 0, 0, 0);
		const v4i32 hi = __builtin_amdgcn_mfma_i32_16x16x32_i8(ky_hi, tv, zero, 0, 0, 0);
#pragma unroll
		for (int r = 0; r < 4; ++r) {
			const uint32_t v = clamp_fixed(hi[r], lo[r], top_y);
			if (c == 0) put_byte_shr<0>(pix[r], v, py);
			else if (c == 1) put_byte_shr<1>(pix[r], v, py);
			else if (c == 2) put_byte_shr<2>(pix[r], v, py);
			else put_byte_shr<3>(pix[r], v, py);
		}
	}
	if constexpr (NCH == 4) {
#pragma unroll
		for (int r = 0; r < 4; ++r) pix[r] = unpremultiply(pix[r]);
	} else if (!(mx[288] & my[288])) {
		// opaque tile: alpha is the convolution of the constant 255 = the windows' weight sums (fast32_v);
		// the table says when that is 255 for every output of the axis
		const int32_t ah = (int32_t)clip8((1 << (px_ - 1)) + 255 * (int32_t)mx[272u + o], (int)px_);
#pragma unroll
		for (int r = 0; r < 4; ++r) {
			const uint32_t al = clip8((1 << (py - 1)) + ah * (int32_t)my[272u + 4u * g + (uint32_t)r], (int)py);
			pix[r] = (pix[r] & 0x00ffffffu) | (al << 24);
			if (al != 255u) pix[r] = unpremultiply(pix[r]);
		}
	}
	if (o < nw) {
#pragma unroll
		for (uint32_t r = 0; r < 4; ++r) {
			const uint32_t oy = 4u * g + r;
			if (oy < nh) out[oy * nw + o] = pix[r];
		}
	}
}

// Fast-path eligibility of a tile (full 32x32 RGBA, 16-byte aligned rows) and its first byte.
template <class Args>
__device__ __forceinline__ bool fast32_tile_src(const Args &a, uint32_t tile_g, const uint8_t *&src)
{
	if (tile_g >= a.n_tiles) return false;
	const uint32_t frame = fastdiv(tile_g, a.div_tpf);
	const uint32_t t = tile_g - frame * a.tiles_per_frame;
	const uint32_t ty = fastdiv(t, a.div_cols), tx = t - ty * a.cols;
	src = a.src + (size_t)frame * a.frame_stride + (size_t)(ty * 32u) * a.pitch + (size_t)(tx * 32u) * 4u;
	return tx < a.full_cols && ty < a.full_rows;  // full size; the alignment of the batch is folded in by the host
}
// Issues the four 16-byte loads of a lane's share of a fast tile (rows l/8 + 8k, quad l%8).
template <class Args>
__device__ __forceinline__ void fast32_prefetch(const Args &a, uint32_t tile_g, uint32_t lane, uint4 (&pre)[4], bool &valid)
{
	const uint8_t *src;
	valid = fast32_tile_src(a, tile_g, src);
	if (valid) {
		const uint8_t *p = src + (size_t)(lane >> 3) * a.pitch + (lane & 7u) * 16u;
#pragma unroll
		for (int k = 0; k < 4; ++k) pre[k] = *reinterpret_cast<const uint4 *>(p + (size_t)(8 * k) * a.pitch);
	}
}

// Detector result of one tile -> stored block value (and the raw detector outputs for pxz_lod_*):
// the f64 part of get_block_variance_directionally (operations.rs:253-258), shrink_*'s closures
// (pixlzr.rs:177-178, :199) and reduce_image_section's value (operations.rs:154).
constexpr uint32_t kDeferredKey = 0xffffffffu;  // sums[] of a tile shrink32_kernel handed to the worklist
__device__ __forceinline__ void finish_tile(const uint2 key, uint32_t w, uint32_t h, uint32_t mode, float factor, float *value,
                                            float *lod0, float *lod1, uint32_t tile_g)
{
	float raw0, raw1, v0, v1;
	if (mode == 1) {
		const uint64_t fac = (uint64_t)(w - 2) * (uint64_t)(h - 2) * 4096ull;  // operations.rs:253-254
		if (fac == 0) {
			// 0/0 on the reference's x86-64 target is the negative default NaN:
			// parse_value turns it into max(1+NaN, 0) = 0 -> stored value 0
			raw0 = raw1 = __uint_as_float(0xFFC00000u);
			v0 = v1 = 0.0f;
		} else {
			const double d = (double)fac;
			raw0 = (float)((double)key.x / d);  // :256
			raw1 = (float)((double)key.y / d);  // :257
			v0 = parse_value(raw0 * factor);    // pixlzr.rs:199
			v1 = parse_value(raw1 * factor);
		}
	} else {
		raw0 = raw1 = __uint_as_float(key.x);
		v0 = v1 = parse_value(raw0);  // pixlzr.rs:177-178
	}
	if (value) value[tile_g] = hypot_f32(v0, v1);  // operations.rs:154
	if (lod0) lod0[tile_g] = raw0;
	if (lod1) lod1[tile_g] = raw1;
}

// ---------------------------------------------------------------------------
// the fused shrink kernel
// ---------------------------------------------------------------------------
// NW   waves cooperating on one tile (1: four independent tiles per 256-thread
//      block, no block barriers; >1: one tile per block of 64*NW threads)
// C    interleaved channels in HBM (3|4)
// MODE 0 shrink_by (Oklab MAD), 1 shrink_directionally
//
// LDS image of a tile: four planes (R,G,B,A) of u16 samples, two horizontally
// adjacent pixels per dword, row stride a.rs dwords.  Packed 16-bit VALU ops then
// process two pixels per instruction (detector) and v_dot2_i32_i16 two filter
// taps per instruction (resample).  The horizontal pass writes its u8 results
// transposed ([ox][y], two rows per dword) so the vertical pass is dot2-shaped too.
template <int NW, int C, int MODE>
__device__ __forceinline__ void process_tile(const ShrinkArgs &a, const uint32_t tile_g, uint32_t *s_pl, uint32_t *s_red,
                                             const uint32_t tid)
{
	constexpr uint32_t TPT = 64u * NW;                // threads per tile
	const uint32_t frame = fastdiv(tile_g, a.div_tpf);
	const uint32_t t = tile_g - frame * a.tiles_per_frame;
	const uint32_t ty = fastdiv(t, a.div_cols), tx = t - ty * a.cols;
	const uint32_t w = (tx == a.cols - 1) ? a.edge_w : a.bw;  // split.rs:18
	const uint32_t h = (ty == a.rows - 1) ? a.edge_h : a.bh;  // split.rs:19
	const uint32_t n = w * h;
	const uint32_t cls = (w != a.bw ? 1u : 0u) | (h != a.bh ? 2u : 0u);
	const uint8_t *src = a.src + (size_t)frame * a.frame_stride + (size_t)(ty * a.bh) * a.pitch + (size_t)(tx * a.bw) * C;

	const uint32_t rs = a.rs, PD = a.plane_dw;
	uint32_t *s_tmp = s_pl + 4 * PD;
	uint16_t *pl16 = reinterpret_cast<uint16_t *>(s_pl);
	float *s_lab = reinterpret_cast<float *>(s_tmp);
	(void)s_red;  // 4*NW dwords, only carved (and used) when NW > 1
	(void)s_lab;

	// ---- stage the tile: coalesced 16-B loads along image rows -> planar u16 pairs ----
	uint32_t alpha_and = 0xffu;
	const uint32_t qpr = w >> 2, nquad = qpr * h;
	const bool vec = C == 4 && ((w & 3u) == 0) && ((reinterpret_cast<uintptr_t>(src) & 15u) == 0) &&
	                 ((a.pitch & 15u) == 0) && nquad <= 4u * TPT;
	if (vec) {
		RowWalker rw(tid, TPT, qpr);
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const uint32_t i = tid + (uint32_t)k * TPT;
			if (i < nquad) {
				const uint4 v = *reinterpret_cast<const uint4 *>(src + (size_t)rw.row * a.pitch + rw.col * 16u);
				alpha_and &= (v.x & v.y & v.z & v.w) >> 24;
				uint32_t *d = s_pl + rw.row * rs + rw.col * 2u;
#pragma unroll
				for (uint32_t c = 0; c < 4; ++c) {
					const uint32_t sel = c | 0x0c000c00u | ((4u + c) << 16);
					uint2 pr;
					pr.x = __builtin_amdgcn_perm(v.y, v.x, sel);
					pr.y = __builtin_amdgcn_perm(v.w, v.z, sel);
					*reinterpret_cast<uint2 *>(d + c * PD) = pr;
				}
			}
			rw.next();
		}
	} else {
		RowWalker rw(tid, TPT, w);
		for (uint32_t i = tid; i < n; i += TPT, rw.next()) {
			const uint8_t *p = src + (size_t)rw.row * a.pitch + rw.col * (uint32_t)C;
			const uint32_t idx = rw.row * rs * 2u + rw.col;
			pl16[idx] = p[0];
			pl16[idx + 2u * PD] = p[1];
			pl16[idx + 4u * PD] = p[2];
			const uint32_t al = C == 4 ? p[3] : 255u;
			pl16[idx + 6u * PD] = (uint16_t)al;
			alpha_and &= al;
		}
	}
	tile_sync<NW>();

	// ---- level-of-detail value --------------------------------------------
	uint32_t key0, key1;  // what finish_kernel turns into the stored value: gradient sums | f32 value bits
	uint32_t m0, m1;      // level exponents: size = ceil(size / 2^m)
	if constexpr (MODE == 1) {
		// get_block_variance_directionally, operations.rs:192-259.  Separable form of the 3x3
		// operators: hz = r(y+2) - r(y) with r = 1-2-1 smoothing along x; vr = c(x+2) - c(x) with
		// c = 1-2-1 smoothing along y.  Two windows per lane and instruction (packed u16).
		uint32_t sum_hz = 0, sum_vr = 0;
		if (w > 2 && h > 2) {
			const uint32_t WR = h - 2;
			const uint32_t VP = (w >> 1) - 1;  // pixel pairs that start a valid window pair (w even)
			uint32_t G = ((w & 1u) == 0 && VP >= 1) ? small_div(TPT, VP) : 0u;
			if (G > WR) G = WR;
			if (G >= 1) {
				const uint32_t RG = small_div(WR + G - 1, G);  // window rows per lane group
				const uint32_t g = small_div(tid, VP), q = tid - g * VP;
				const uint32_t y0 = g * RG;
				if (g < G && y0 < WR) {
					const uint32_t y1 = (y0 + RG < WR) ? y0 + RG : WR;
					const uint32_t *p = s_pl + y0 * rs + q;
					uint32_t ra[3], rb[3], t0[3], t1[3], d0[3], d1[3];
#pragma unroll
					for (int c = 0; c < 3; ++c) {
						const uint32_t a0 = p[c * PD], a1 = p[c * PD + 1];
						const uint32_t b0 = p[c * PD + rs], b1 = p[c * PD + rs + 1];
						ra[c] = smooth121(a0, a1);
						rb[c] = smooth121(b0, b1);
						t0[c] = u32(us2(a0) + us2(b0));
						t1[c] = u32(us2(a1) + us2(b1));
						d0[c] = b0;
						d1[c] = b1;
					}
					p += 2 * rs;
					for (uint32_t y = y0; y < y1; ++y, p += rs) {
#pragma unroll
						for (int c = 0; c < 3; ++c) {
							const uint32_t n0 = p[c * PD], n1 = p[c * PD + 1];
							const uint32_t rn = smooth121(n0, n1);
							sum_hz = sad16(rn, ra[c], sum_hz);  // |hz| of windows (2q, 2q+1), :240-241,:247
							const uint32_t t0n = u32(us2(d0[c]) + us2(n0)), t1n = u32(us2(d1[c]) + us2(n1));
							const uint32_t c0 = u32(us2(t0[c]) + us2(t0n)), c1 = u32(us2(t1[c]) + us2(t1n));
							sum_vr = sad16(c1, c0, sum_vr);     // |vr|, :244-245,:248
							ra[c] = rb[c];
							rb[c] = rn;
							t0[c] = t0n;
							t1[c] = t1n;
							d0[c] = n0;
							d1[c] = n1;
						}
					}
				}
			} else {
				// odd widths / very wide tiles: one window per lane and step
				const uint32_t ww = w - 2, hh = h - 2;
				RowWalker rw(tid, TPT, ww);
				for (uint32_t i = tid; i < ww * hh; i += TPT, rw.next()) {
#pragma unroll
					for (uint32_t c = 0; c < 3; ++c) {
						const uint16_t *r0 = pl16 + c * 2u * PD + rw.row * rs * 2u + rw.col;
						const uint16_t *r1 = r0 + rs * 2u, *r2 = r1 + rs * 2u;
						const int v00 = r0[0], v01 = r0[1], v02 = r0[2], v10 = r1[0], v12 = r1[2];
						const int v20 = r2[0], v21 = r2[1], v22 = r2[2];
						const int ghz = -v00 - 2 * v01 - v02 + v20 + 2 * v21 + v22;
						const int gvr = -v00 - 2 * v10 - v20 + v02 + 2 * v12 + v22;
						sum_hz += (uint32_t)(ghz < 0 ? -ghz : ghz);
						sum_vr += (uint32_t)(gvr < 0 ? -gvr : gvr);
					}
				}
			}
		}
		sum_hz = wave_sum_sgpr(sum_hz);
		sum_vr = wave_sum_sgpr(sum_vr);
		if constexpr (NW > 1) {
			const uint32_t wv = threadIdx.x / 64u;
			if ((threadIdx.x & 63u) == 0) {
				s_red[2 * wv] = sum_hz;
				s_red[2 * wv + 1] = sum_vr;
			}
			__syncthreads();
			sum_hz = 0;
			sum_vr = 0;
#pragma unroll
			for (int q = 0; q < NW; ++q) {
				sum_hz += s_red[2 * q];
				sum_vr += s_red[2 * q + 1];
			}
		}
		// operations.rs:147-148 through host-built integer breakpoints on the sums (exactly equivalent:
		// the float pipeline sum -> value -> level is monotone in the sum).  The f32/f64 value math
		// itself (operations.rs:253-258, :154) runs lane-parallel over tiles in finish_kernel.
		key0 = sum_hz;
		key1 = sum_vr;
		if (w <= 2 || h <= 2) {
			// 0/0 = negative default NaN on the reference's x86-64 target -> parse_value gives 0 -> 1x1
			m0 = m1 = (uint32_t)kMaxLevel;
		} else {
			m0 = level_count(sum_hz, a.breaks[cls], a.breaks_asc[cls]);
			m1 = level_count(sum_vr, a.breaks[cls], a.breaks_asc[cls]);
		}
	} else if (a.oklab_given && tx < a.full_cols && ty < a.ok_rows) {
		// full tile of a batch the block-cooperative detector (oklab_kernel) has already been over
		const float value = __uint_as_float(a.sums[2u * tile_g]);
		key0 = key1 = __float_as_uint(value);
		m0 = m1 = level_count(__float_as_uint(parse_value(value)), a.breaks[cls], a.breaks_asc[cls]);
	} else {
		// get_block_variance, operations.rs:26-126 with shrink_by's closures
		// (pixlzr.rs:160-162).  Colours are computed once, in parallel, into LDS
		// planes [a | b | l]; the two f32 accumulations are then replayed in the
		// reference's sequential pixel order, one lane per channel chain.
		{
			RowWalker rw(tid, TPT, w);
			for (uint32_t i = tid; i < n; i += TPT, rw.next()) {
				const uint32_t idx = rw.row * rs * 2u + rw.col;
				const uint32_t px = (uint32_t)pl16[idx] | ((uint32_t)pl16[idx + 2u * PD] << 8) | ((uint32_t)pl16[idx + 4u * PD] << 16);
				float L, A, B;
				oklab_from_rgba(px, L, A, B);
				s_lab[i] = A;
				s_lab[n + i] = B;
				s_lab[2 * n + i] = L;
			}
		}
		tile_sync<NW>();
		constexpr uint32_t NCH = C == 4 ? 4u : 3u;
		const float count = (float)n;  // :51
		float delta = 0.0f;
		if (threadIdx.x % 64u < NCH && (NW == 1 || threadIdx.x < 64u)) {
			const uint32_t k = threadIdx.x % 64u;
			float s = 0.0f;
			// The adds of a chain depend on each other; the loads (and the alpha divisions) do not: eight
			// elements are fetched / prepared while the previous eight are added, in the reference's order.
			auto run8 = [&](auto &&elem, uint32_t len, float acc, const bool magnitude, const float avg) -> float {
				uint32_t p = 0;
				if (len >= 8u) {
					float cur[8], nxt[8];
#pragma unroll
					for (int j = 0; j < 8; ++j) cur[j] = elem(p + (uint32_t)j);
					for (p = 8u; p + 8u <= len; p += 8u) {
#pragma unroll
						for (int j = 0; j < 8; ++j) nxt[j] = elem(p + (uint32_t)j);
						__builtin_amdgcn_sched_barrier(0);
#pragma unroll
						for (int j = 0; j < 8; ++j) acc += magnitude ? fabsf(cur[j] - avg) : cur[j];
						__builtin_amdgcn_sched_barrier(0);
#pragma unroll
						for (int j = 0; j < 8; ++j) cur[j] = nxt[j];
					}
#pragma unroll
					for (int j = 0; j < 8; ++j) acc += magnitude ? fabsf(cur[j] - avg) : cur[j];
				}
				for (; p < len; ++p) {
					const float v = elem(p);
					acc += magnitude ? fabsf(v - avg) : v;
				}
				return acc;
			};
			if (k < 3) {
				const float *plane = s_lab + k * n;
				auto at = [&](uint32_t p) { return plane[p]; };
				s = run8(at, n, 0.0f, false, 0.0f);                 // :60-62
				const float avg = __fdiv_rn(s, count);               // :65-67
				delta = run8(at, n, 0.0f, true, avg);                // :80-82
			} else {
				const uint16_t *al = pl16 + 6u * PD;
				for (uint32_t y = 0; y < h; ++y) {
					const uint16_t *row = al + y * rs * 2u;
					auto at = [&](uint32_t x) { return __fdiv_rn((float)row[x], 255.0f); };
					s = run8(at, w, s, false, 0.0f);  // :63
				}
				const float avg = __fdiv_rn(s, count);
				for (uint32_t y = 0; y < h; ++y) {
					const uint16_t *row = al + y * rs * 2u;
					auto at = [&](uint32_t x) { return __fdiv_rn((float)row[x], 255.0f); };
					delta = run8(at, w, delta, true, avg);
				}
			}
		}
		float total;
		{
			const float d0 = __shfl(delta, 0, 64), d1 = __shfl(delta, 1, 64), d2 = __shfl(delta, 2, 64);
			total = d0 + d1 + d2;                      // :124
			if constexpr (C == 4) total = total + __shfl(delta, 3, 64);  // :89
		}
		float value = __fdiv_rn(total, count) * a.factor * a.scale2;  // pixlzr.rs:162
		if constexpr (NW > 1) {
			if (threadIdx.x == 0) s_red[0] = __float_as_uint(value);
			__syncthreads();
			value = __uint_as_float(s_red[0]);
		}
		key0 = key1 = __float_as_uint(value);
		// pixlzr.rs:177-178: (value, value); parsed value >= 0 or NaN, so its bit pattern orders like the float
		m0 = m1 = level_count(__float_as_uint(parse_value(value)), a.breaks[cls], a.breaks_asc[cls]);
		tile_sync<NW>();  // the Oklab scratch aliases the transposed planes
	}

	// ---- reduce_image_section: target size + stored value -------------------
	const uint32_t nw = reduced_size(w, m0);  // operations.rs:150
	const uint32_t nh = reduced_size(h, m1);  // :151
	if (tid == 0) {
		reinterpret_cast<uint2 *>(a.sums)[tile_g] = make_uint2(key0, key1);
		if (a.out_w) a.out_w[tile_g] = nw;
		if (a.out_h) a.out_h[tile_g] = nh;
		if (a.work) finish_tile(make_uint2(key0, key1), w, h, (uint32_t)MODE, a.factor, a.value, a.lod0, a.lod1, tile_g);
	}
	if (a.out_px == nullptr) return;

	// ---- PixlzrBlock::resize (block.rs:273-334) out of the LDS copy ----------
	uint8_t *dst = a.out_px + (size_t)tile_g * a.slot_bytes;
	auto gather_px = [&](uint32_t x, uint32_t y) -> uint32_t {
		const uint32_t idx = y * rs * 2u + x;
		return (uint32_t)pl16[idx] | ((uint32_t)pl16[idx + 2u * PD] << 8) | ((uint32_t)pl16[idx + 4u * PD] << 16) |
		       ((uint32_t)pl16[idx + 6u * PD] << 24);
	};
	if (nw == w && nh == h) {  // block.rs:279-281: clone
		if (C == 4 && (w & 1u) == 0 && (reinterpret_cast<uintptr_t>(dst) & 7u) == 0) {
			// two pixels per step: one dword of each plane, re-interleaved with byte permutes
			const uint32_t P2 = w >> 1;
			RowWalker rw(tid, TPT, P2);
			for (uint32_t i = tid; i < P2 * h; i += TPT, rw.next()) {
				const uint32_t *p = s_pl + rw.row * rs + rw.col;
				const uint32_t rg = __builtin_amdgcn_perm(p[PD], p[0], 0x06020400u);          // r0 g0 r1 g1
				const uint32_t ba = __builtin_amdgcn_perm(p[3 * PD], p[2 * PD], 0x06020400u);  // b0 a0 b1 a1
				uint2 o;
				o.x = __builtin_amdgcn_perm(ba, rg, 0x05040100u);
				o.y = __builtin_amdgcn_perm(ba, rg, 0x07060302u);
				reinterpret_cast<uint2 *>(dst)[i] = o;
			}
		} else {
			RowWalker rw(tid, TPT, w);
			for (uint32_t i = tid; i < n; i += TPT, rw.next()) store_pixel<C>(dst, i, gather_px(rw.col, rw.row));
		}
		return;
	}

	const uint32_t lx = m0 < (uint32_t)kMaxLevel ? m0 : (uint32_t)kMaxLevel - 1;
	const uint32_t ly = m1 < (uint32_t)kMaxLevel ? m1 : (uint32_t)kMaxLevel - 1;
	const AxisTab tab_x = a.tabs[(0 * 2 + (w == a.bw ? 0 : 1)) * kMaxLevel + lx];
	const AxisTab tab_y = a.tabs[(1 * 2 + (h == a.bh ? 0 : 1)) * kMaxLevel + ly];

	if (a.filter == 0) {  // ResizeAlg::Nearest (mod.rs:277): pick, no alpha handling
		const uint16_t *sx = a.bounds + tab_x.bounds_off;
		const uint16_t *sy = a.bounds + tab_y.bounds_off;
		RowWalker rw(tid, TPT, nw);
		for (uint32_t i = tid; i < nw * nh; i += TPT, rw.next()) {
			const uint32_t x = nw == w ? rw.col : sx[rw.col];
			const uint32_t y = nh == h ? rw.row : sy[rw.row];
			store_pixel<C>(dst, i, gather_px(x, y));
		}
		return;
	}

	// ResizeAlg::Convolution, default ResizeOptions: U8x4 is alpha-premultiplied first.  For a fully
	// opaque tile that is the identity and the alpha channel is a constant-input convolution, which
	// collapses to the per-output weight sums (same integer arithmetic, no taps).
	bool opaque = true;
	if constexpr (C == 4) {
		alpha_and = wave_and_sgpr(alpha_and);
		if constexpr (NW > 1) {
			const uint32_t wv = threadIdx.x / 64u;
			if ((threadIdx.x & 63u) == 0) s_red[2 * NW + wv] = alpha_and;
			__syncthreads();
#pragma unroll
			for (int q = 0; q < NW; ++q) alpha_and &= s_red[2 * NW + q];
		}
		opaque = alpha_and == 0xffu;
		if (!opaque) {
			const uint32_t P2 = (w + 1) >> 1;
			RowWalker rw(tid, TPT, P2);
			for (uint32_t i = tid; i < P2 * h; i += TPT, rw.next()) {
				uint32_t *p = s_pl + rw.row * rs + rw.col;
				const uint32_t al = p[3 * PD];
#pragma unroll
				for (int c = 0; c < 3; ++c) {
					const uint32_t v = p[c * PD];
					p[c * PD] = mul_div_255(v & 0xffffu, al & 0xffffu) | (mul_div_255(v >> 16, al >> 16) << 16);
				}
			}
			tile_sync<NW>();
		}
	}
	const uint32_t nch = opaque ? 3u : 4u;  // channels that need taps

	const bool need_h = nw != w, need_v = nh != h;
	const int prec_x = tab_x.precision, prec_y = tab_y.precision;
	const int32_t init_x = 1 << (prec_x - 1), init_y = 1 << (prec_y - 1);
	const uint16_t *bnd_x = a.bounds + tab_x.bounds_off, *bnd_y = a.bounds + tab_y.bounds_off;
	const uint32_t *cf_x = a.coeffs + tab_x.coeff_off, *cf_y = a.coeffs + tab_y.coeff_off;
	const int32_t *ks_x = a.ksums + tab_x.ksum_off, *ks_y = a.ksums + tab_y.ksum_off;
	const uint32_t hps = a.hps, TD = a.tmp_dw;

	if (need_h) {
		// horizontal pass: item = (output column, pair of rows); u8 results kept transposed
		const uint32_t HP = (h + 1) >> 1;
		RowWalker rw(tid, TPT, nw);
		for (uint32_t i = tid; i < nw * HP; i += TPT, rw.next()) {
			const uint32_t ox = rw.col, yp = rw.row;
			const uint32_t fq = bnd_x[2 * ox], nq = bnd_x[2 * ox + 1];
			const uint32_t *k = cf_x + ox * tab_x.wquads * 2u;
			const uint32_t *row = s_pl + (2 * yp) * rs + fq * 2u;
			int32_t acc[4][2];
#pragma unroll
			for (int c = 0; c < 4; ++c) acc[c][0] = acc[c][1] = init_x;
			for (uint32_t q = 0; q < nq; ++q) {
				const uint32_t k01 = k[2 * q], k23 = k[2 * q + 1];
#pragma unroll
				for (uint32_t c = 0; c < 4; ++c) {
					if (c < nch) {
						const uint2 da = *reinterpret_cast<const uint2 *>(row + c * PD + q * 2u);
						const uint2 db = *reinterpret_cast<const uint2 *>(row + c * PD + rs + q * 2u);
						acc[c][0] = dot2(da.y, k23, dot2(da.x, k01, acc[c][0]));
						acc[c][1] = dot2(db.y, k23, dot2(db.x, k01, acc[c][1]));
					}
				}
			}
			uint32_t o[4][2];
#pragma unroll
			for (int c = 0; c < 4; ++c) {
				o[c][0] = clip8(acc[c][0], prec_x);
				o[c][1] = clip8(acc[c][1], prec_x);
			}
			if (opaque) o[3][0] = o[3][1] = clip8(init_x + 255 * ks_x[ox], prec_x);
			if (need_v) {
#pragma unroll
				for (uint32_t c = 0; c < 4; ++c)
					if (c < nch) s_tmp[c * TD + ox * hps + yp] = o[c][0] | (o[c][1] << 16);
			} else {
#pragma unroll
				for (uint32_t r = 0; r < 2; ++r) {
					const uint32_t y = 2 * yp + r;
					if (y < h) {
						uint32_t px = o[0][r] | (o[1][r] << 8) | (o[2][r] << 16) | (o[3][r] << 24);
						if constexpr (C == 4) px = unpremultiply(px);
						store_pixel<C>(dst, y * nw + ox, px);
					}
				}
			}
		}
		if (!need_v) return;
		tile_sync<NW>();
		// vertical pass over the transposed planes: item = (output column, output row)
		RowWalker rv(tid, TPT, nw);
		for (uint32_t i = tid; i < nw * nh; i += TPT, rv.next()) {
			const uint32_t ox = rv.col, oy = rv.row;
			const uint32_t fq = bnd_y[2 * oy], nq = bnd_y[2 * oy + 1];
			const uint32_t *k = cf_y + oy * tab_y.wquads * 2u;
			const uint32_t *colp = s_tmp + ox * hps + fq * 2u;
			int32_t acc[4] = {init_y, init_y, init_y, init_y};
			for (uint32_t q = 0; q < nq; ++q) {
				const uint32_t k01 = k[2 * q], k23 = k[2 * q + 1];
#pragma unroll
				for (uint32_t c = 0; c < 4; ++c) {
					if (c < nch) {
						const uint2 d = *reinterpret_cast<const uint2 *>(colp + c * TD + q * 2u);
						acc[c] = dot2(d.y, k23, dot2(d.x, k01, acc[c]));
					}
				}
			}
			uint32_t al = clip8(acc[3], prec_y);
			if (opaque) {
				const int32_t ah = (int32_t)clip8(init_x + 255 * ks_x[ox], prec_x);
				al = clip8(init_y + ah * ks_y[oy], prec_y);
			}
			uint32_t px = clip8(acc[0], prec_y) | (clip8(acc[1], prec_y) << 8) | (clip8(acc[2], prec_y) << 16) | (al << 24);
			if constexpr (C == 4) px = unpremultiply(px);
			store_pixel<C>(dst, i, px);
		}
		return;
	}
	{
		// vertical pass only (width kept): item = (pair of columns, output row) on the [y][x] planes
		const uint32_t P2 = (w + 1) >> 1;
		RowWalker rv(tid, TPT, P2);
		for (uint32_t i = tid; i < P2 * nh; i += TPT, rv.next()) {
			const uint32_t qx = rv.col, oy = rv.row;
			const uint32_t fq = bnd_y[2 * oy], nq = bnd_y[2 * oy + 1];
			const uint32_t *k = cf_y + oy * tab_y.wquads * 2u;
			const uint32_t *colp = s_pl + (fq * 4u) * rs + qx;
			int32_t acc[4][2];
#pragma unroll
			for (int c = 0; c < 4; ++c) acc[c][0] = acc[c][1] = init_y;
			for (uint32_t q = 0; q < nq; ++q) {
				const uint32_t k01 = k[2 * q], k23 = k[2 * q + 1];
#pragma unroll
				for (uint32_t c = 0; c < 4; ++c) {
					if (c < nch) {
						const uint32_t *p = colp + c * PD + (q * 4u) * rs;
						const uint32_t r0 = p[0], r1 = p[rs], r2 = p[2 * rs], r3 = p[3 * rs];
						// (row j, row j+1) pairs of the left / right column
						const uint32_t l01 = __builtin_amdgcn_perm(r1, r0, 0x05040100u), l23 = __builtin_amdgcn_perm(r3, r2, 0x05040100u);
						const uint32_t h01 = __builtin_amdgcn_perm(r1, r0, 0x07060302u), h23 = __builtin_amdgcn_perm(r3, r2, 0x07060302u);
						acc[c][0] = dot2(l23, k23, dot2(l01, k01, acc[c][0]));
						acc[c][1] = dot2(h23, k23, dot2(h01, k01, acc[c][1]));
					}
				}
			}
			uint32_t al0 = clip8(acc[3][0], prec_y), al1 = clip8(acc[3][1], prec_y);
			if (opaque) al0 = al1 = clip8(init_y + 255 * ks_y[oy], prec_y);
#pragma unroll
			for (uint32_t r = 0; r < 2; ++r) {
				const uint32_t x = 2 * qx + r;
				if (x < w) {
					uint32_t px = clip8(acc[0][r], prec_y) | (clip8(acc[1][r], prec_y) << 8) | (clip8(acc[2][r], prec_y) << 16) |
					              ((r ? al1 : al0) << 24);
					if constexpr (C == 4) px = unpremultiply(px);
					store_pixel<C>(dst, oy * w + x, px);
				}
			}
		}
	}
}

// ---- diagnostic build only (-DPXZ_STAMPS): per-phase wave-cycle shares of shrink32_kernel.
// Stamp values leave through a buffer of their own (a.work, past the worklist); no output depends on them.
#ifdef PXZ_STAMPS
__device__ __forceinline__ unsigned long long stamp_now()
{
	unsigned long long t;
	__builtin_amdgcn_sched_barrier(0);
	asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
	__builtin_amdgcn_sched_barrier(0);
	return t;
}
#define PXZ_STAMP(i)                                   \
	do {                                               \
		const unsigned long long now_ = stamp_now();   \
		st_acc[i] += now_ - st_last;                   \
		st_last = now_;                                \
	} while (0)
#else
#define PXZ_STAMP(i) \
	do {             \
	} while (0)
#endif

// Worklist appends, batched: a global atomic on ONE address completes every ~12 ns chip-wide, so a frame whose
// every tile is listed (transparent frames: 259 200 tiles) would spend 3 ms on the counter alone.  Each wave
// (or block) parks up to kListBatch tile numbers in LDS and reserves their list slots with one atomic.
constexpr uint32_t kListBatch = 16;
__device__ __forceinline__ void list_flush(uint32_t *buf, uint32_t &cnt, uint32_t *list, uint32_t *counter, uint32_t lane)
{
	if (cnt == 0u) return;
	uint32_t base = 0;
	if (lane == 0u) base = atomicAdd(counter, cnt);
	base = __builtin_amdgcn_readfirstlane(base);
	asm volatile("" ::: "memory");
	if (lane < cnt) list[base + lane] = buf[lane];
	asm volatile("" ::: "memory");
	cnt = 0u;
}
// cnt is uniform over the lanes that call this (a wave, or thread 0 of a block with lane == 0 semantics)
__device__ __forceinline__ void list_push(uint32_t *buf, uint32_t &cnt, uint32_t tile, uint32_t *list, uint32_t *counter,
										   uint32_t lane)
{
	if (lane == 0u) buf[cnt] = tile;
	++cnt;
	if (cnt == kListBatch) list_flush(buf, cnt, list, counter, lane);
}

// ---------------------------------------------------------------------------
// shrink32_kernel: the common case on its own — full, 16-byte-aligned 32x32 RGBA tiles whose
// resample is a clone, a nearest pick, or a two-pass convolution of an opaque tile.  Everything
// else (ragged-edge tiles, tiles with transparency, one-pass resamples) is appended to a device
// worklist that the generic kernel processes afterwards.  Persistent waves, one LDS tile image
// each, next tile's pixels prefetched into registers, table rows in LDS, lookups in kernarg.
// MODE 1: directional detector here; MODE 0: value already in sums[] (oklab_kernel).
// ---------------------------------------------------------------------------
// FULL: out_px, out_w and out_h are all there (the shrink entry points): no run-time tests of them in the loop --
// kept as loop-invariant lane masks they cost scalar registers, and a spilled one two v_readlane per use.
template <int MODE, bool FULL>
__global__ void __launch_bounds__(1024) shrink32_kernel(const Fast32Args a)
{
	extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
	const uint32_t wpb = blockDim.x / 64u, sub = threadIdx.x / 64u, tid = threadIdx.x % 64u;
	for (uint32_t i = threadIdx.x; i < a.tab_dw / 4u; i += blockDim.x)
		reinterpret_cast<uint4 *>(lds)[i] = reinterpret_cast<const uint4 *>(a.trows)[i];
	// Tiles are dealt to the waves of a block on demand (an LDS ticket counter): tile costs differ by 2x
	// between size classes, and a static stride leaves the unluckiest wave of the chip running alone.
	// Block b owns tiles b, b + blocks, b + 2*blocks, ...; ticket t is tile b + t*blocks.
	uint32_t *s_ticket = lds + a.tab_dw + wpb * a.tile_dw;
	if (threadIdx.x == 0) *s_ticket = wpb;  // tickets 0..wpb-1 are the waves' first tiles
	__syncthreads();
	const uint32_t *s_tab = lds;
	// level breakpoints, one per lane (lanes >= kMaxLevel never count): the level exponent of a key is
	// one lane-parallel compare + ballot + popcount instead of a scalar compare chain
	const uint32_t brk_lane = tid < (uint32_t)kMaxLevel ? a.breaks[tid] : (a.breaks_asc ? 0xffffffffu : 0u);
	auto level_of = [&](uint32_t key) -> uint32_t {
		const unsigned long long lt = __builtin_amdgcn_ballot_w64(key < brk_lane);
		const unsigned long long live = (1ull << kMaxLevel) - 1ull;
		return (uint32_t)__builtin_popcountll((a.breaks_asc ? ~lt : lt) & live);
	};
	uint32_t *s_pl = lds + a.tab_dw + sub * a.tile_dw;
	uint32_t *s_tmp = s_pl + 3 * kPD32;  // R, G, B planes only: tiles with transparency go to the worklist
	uint32_t *s_batch = s_ticket + 4u + sub * (2u * kListBatch);  // this wave's pending list-B / list-A entries
	uint32_t n_listb = 0, n_lista = 0;
	auto tile_of_ticket = [&](uint32_t t) -> uint32_t {
		// runs of 2^chunk_lg adjacent tiles per block: successive tickets walk along an image row
		const unsigned long long run = (unsigned long long)(t >> a.chunk_lg) * gridDim.x + blockIdx.x;
		const unsigned long long g = (run << a.chunk_lg) + (t & ((1u << a.chunk_lg) - 1u));
		return g < (unsigned long long)a.n_tiles ? (uint32_t)g : 0xffffffffu;
	};
	auto next_ticket = [&]() -> uint32_t {
		uint32_t t = 0;
		if (tid == 0) t = atomicAdd(s_ticket, 1u);
		return tile_of_ticket(__builtin_amdgcn_readfirstlane(t));
	};
	// The pixels of the next tile are requested right after the current one has been staged, and a
	// tile's output pixels are parked in LDS and stored at the START of the next iteration, before
	// that prefetch: loads and stores share one in-order counter (vmcnt), so the wait for the
	// prefetched registers must not find younger stores or loads behind it.
	uint4 pre[4];
	bool pre_valid = false;
	const uint32_t first = tile_of_ticket(__builtin_amdgcn_readfirstlane(sub));
	fast32_prefetch(a, first, tid, pre, pre_valid);
#ifdef PXZ_STAMPS
	unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	unsigned long long st_last = stamp_now();
	const unsigned long long st_begin = wall_clock64();
#endif
	uint32_t pend_kind = 0;       // 0 nothing, 1 linear pixels in LDS, 2 clone (re-interleave the planes)
	uint32_t pend_px = 0;         // pixels parked in LDS (kind 1)
	const uint32_t *pend_src = nullptr;
	uint8_t *pend_dst = nullptr;
	auto flush = [&]() {
		if (pend_kind == 1) {
			uint32_t *d = reinterpret_cast<uint32_t *>(pend_dst);
			if (pend_px >= 4u) {  // whole 16-byte groups (pixel counts are powers of two)
				for (uint32_t i = tid; i < (pend_px >> 2); i += 64u)
					reinterpret_cast<uint4 *>(d)[i] = reinterpret_cast<const uint4 *>(pend_src)[i];
			} else if (tid < pend_px) {
				d[tid] = pend_src[tid];
			}
		} else if (pend_kind == 2) {
			// clone (block.rs:279-281): re-interleave the planes, one 16-byte store per 4 pixels
#pragma unroll
			for (int k = 0; k < 4; ++k) {
				const uint32_t i = tid + 64u * (uint32_t)k;
				const uint32_t *p = s_pl + (i >> 3) * kRS32 + (i & 7u) * 2u;
				const uint2 r = *reinterpret_cast<const uint2 *>(p), g = *reinterpret_cast<const uint2 *>(p + kPD32);
				const uint2 b = *reinterpret_cast<const uint2 *>(p + 2 * kPD32);
				const uint32_t opq = 0x00ff00ffu;  // the tile is opaque: alpha pair (255, 255)
				const uint32_t rg01 = __builtin_amdgcn_perm(g.x, r.x, 0x06020400u), ba01 = __builtin_amdgcn_perm(opq, b.x, 0x06020400u);
				const uint32_t rg23 = __builtin_amdgcn_perm(g.y, r.y, 0x06020400u), ba23 = __builtin_amdgcn_perm(opq, b.y, 0x06020400u);
				uint4 o;
				o.x = __builtin_amdgcn_perm(ba01, rg01, 0x05040100u);
				o.y = __builtin_amdgcn_perm(ba01, rg01, 0x07060302u);
				o.z = __builtin_amdgcn_perm(ba23, rg23, 0x05040100u);
				o.w = __builtin_amdgcn_perm(ba23, rg23, 0x07060302u);
				reinterpret_cast<uint4 *>(pend_dst)[i] = o;
			}
		}
		pend_kind = 0;
	};
	auto one_tile = [&](const uint32_t tile_g, const uint32_t tile_next) {
		auto defer = [&]() {
			list_push(s_batch, n_listb, tile_g, a.work + kWorkList, a.work + a.work_slot, tid);
			if (tid == 0) {
				bool keep = false;  // MODE 0: a value the block-cooperative detector left is final and stays
				if constexpr (MODE == 0) {
					const uint32_t t = tile_g - fastdiv(tile_g, a.div_tpf) * a.tiles_per_frame;
					const uint32_t ty = fastdiv(t, a.div_cols), tx = t - ty * a.cols;
					keep = tx < a.full_cols && ty < a.ok_rows;
				}
				if (!keep) reinterpret_cast<uint2 *>(a.sums)[tile_g] = make_uint2(kDeferredKey, kDeferredKey);  // not finished here
			}
		};
		if (!pre_valid) {  // ragged edge / unaligned rows: generic kernel
			flush();
			defer();
			fast32_prefetch(a, tile_next, tid, pre, pre_valid);
			return;
		}
		uint32_t given_bits = 0;
		if constexpr (MODE == 0) given_bits = a.sums[2 * tile_g];  // issued before the prefetch: its wait leaves the prefetch in flight
		// ---- wait for the prefetched registers (the opacity test is their first use), then emit the
		// previous tile's parked pixels, then stage: registers -> planar u16 pairs
		uint32_t alpha_and = 0xffu;
#pragma unroll
		for (int k = 0; k < 4; ++k) alpha_and &= (pre[k].x & pre[k].y & pre[k].z & pre[k].w) >> 24;
		const bool transparent = __builtin_amdgcn_ballot_w64(alpha_and != 0xffu) != 0ull;
		__builtin_amdgcn_sched_barrier(0);
		flush();
		__builtin_amdgcn_sched_barrier(0);
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const uint32_t row = (tid >> 3) + 8u * (uint32_t)k, col = tid & 7u;
			const uint4 v = pre[k];
			uint32_t *d = s_pl + row * kRS32 + col * 2u;
#pragma unroll
			for (uint32_t c = 0; c < 3; ++c) {
				const uint32_t sel = c | 0x0c000c00u | ((4u + c) << 16);
				uint2 pr;
				pr.x = __builtin_amdgcn_perm(v.y, v.x, sel);
				pr.y = __builtin_amdgcn_perm(v.w, v.z, sel);
				*reinterpret_cast<uint2 *>(d + c * kPD32) = pr;
			}
		}
		PXZ_STAMP(0);  // wait for the prefetched pixels + staging
		fast32_prefetch(a, tile_next, tid, pre, pre_valid);  // lands while this tile is processed
		if (transparent && (FULL || a.out_px != nullptr)) {
			// transparency: the premultiplied convolution needs the alpha plane -- shrink32a_kernel (list A) when the
			// caller announced transparent frames, else the generic kernel (list B).  (Detector-only launches do not
			// care: the detector never looks at alpha.)
			if (a.alpha_list) {
				list_push(s_batch + kListBatch, n_lista, tile_g, a.work + kWorkList + a.n_tiles, a.work + kWorkA + a.work_slot, tid);
				if constexpr (MODE == 1)
					if (tid == 0) reinterpret_cast<uint2 *>(a.sums)[tile_g] = make_uint2(kDeferredKey, kDeferredKey);
			} else {
				defer();
			}
			return;
		}
		tile_sync<1>();
		PXZ_STAMP(1);  // prefetch issue

		// ---- detector + level decision
		uint32_t m0, m1;
		uint32_t key0 = 0, key1 = 0;
		if constexpr (MODE == 1) {
			uint32_t sum_hz = 0, sum_vr = 0;
			// 16 lanes (pixel pairs) per row group, 4 groups of 8 window rows (the last one 6).  Per
			// channel and window row: r = 1-2-1 along x (perm, add, mad), |hz| (sad), column smoothing
			// c = t(y)+t(y+1) (2 adds), the neighbour pair's c by DPP, |vr| (sad).  Fully unrolled:
			// every LDS address is a per-channel base + immediate, no loop-carried register moves.
			const uint32_t q = tid & 15u, g = tid >> 4;
			// (compared afresh where it is used: hoisted out of the tile loop the lane mask would sit in two scalar
			// registers, and those are spilled -- two v_readlane per use instead of one v_cmp)
			auto last_rows = [](uint32_t grp) -> bool {
				asm volatile("" : "+v"(grp));
				return grp != 3u;
			};
			const uint32_t two = 0x00020002u;
			const uint32_t *pc[3];
			pc[0] = s_pl + g * (8u * kRS32) + q;
			pc[1] = pc[0] + kPD32;
			pc[2] = pc[1] + kPD32;
			uint32_t rA[3], rB[3], tP[3], dP[3];
#pragma unroll
			for (int c = 0; c < 3; ++c) {
				const uint32_t a0 = pc[c][0], a1 = pc[c][1], b0 = pc[c][kRS32], b1 = pc[c][kRS32 + 1];
				rA[c] = pk_mad_u16(__builtin_amdgcn_alignbit(a1, a0, 16), two, u32(us2(a0) + us2(a1)));
				rB[c] = pk_mad_u16(__builtin_amdgcn_alignbit(b1, b0, 16), two, u32(us2(b0) + us2(b1)));
				tP[c] = u32(us2(a0) + us2(b0));
				dP[c] = b0;
			}
#pragma unroll
			for (int st = 0; st < 4; ++st) {
				if (st < 3 || last_rows(g)) {  // the last group has 6 window rows = 3 steps
#pragma unroll
					for (int c = 0; c < 3; ++c) {
						const uint32_t *pr = pc[c] + (2 + 2 * st) * (int)kRS32;
						const uint32_t n0 = pr[0], n1 = pr[1], o0 = pr[kRS32], o1 = pr[kRS32 + 1];
						const uint32_t rN = pk_mad_u16(__builtin_amdgcn_alignbit(n1, n0, 16), two, u32(us2(n0) + us2(n1)));
						sum_hz = sad16(rN, rA[c], sum_hz);
						const uint32_t tN = u32(us2(dP[c]) + us2(n0));
						const uint32_t c0 = u32(us2(tP[c]) + us2(tN));
						sum_vr = sad16(dpp_mov<0x101>(c0), c0, sum_vr);  // row_shl:1 = the pair to the right
						const uint32_t rO = pk_mad_u16(__builtin_amdgcn_alignbit(o1, o0, 16), two, u32(us2(o0) + us2(o1)));
						sum_hz = sad16(rO, rB[c], sum_hz);
						const uint32_t tO = u32(us2(n0) + us2(o0));
						const uint32_t e0 = u32(us2(tN) + us2(tO));
						sum_vr = sad16(dpp_mov<0x101>(e0), e0, sum_vr);
						rA[c] = rN;
						rB[c] = rO;
						tP[c] = tO;
						dP[c] = o0;
					}
				}
			}
			if (q == 15u) sum_hz = sum_vr = 0;  // pair 15 starts no window (x = 30, 31)
			sum_hz = wave_sum_sgpr(sum_hz);
			sum_vr = wave_sum_sgpr(sum_vr);
			m0 = level_of(sum_hz);
			m1 = level_of(sum_vr);
			key0 = sum_hz;
			key1 = sum_vr;
		} else {
			const uint32_t vb = __builtin_amdgcn_readfirstlane(given_bits);
			m0 = m1 = level_of(__float_as_uint(parse_value(__uint_as_float(vb))));
		}
		const uint32_t nw = reduced_size(32u, m0), nh = reduced_size(32u, m1);
		{
			uint32_t lane = tid;
			asm volatile("" : "+v"(lane));  // (as above: a fresh compare instead of a spilled lane mask)
			if (lane == 0) {
				if constexpr (MODE == 1) reinterpret_cast<uint2 *>(a.sums)[tile_g] = make_uint2(key0, key1);
				if (FULL || a.out_w) a.out_w[tile_g] = nw;
				if (FULL || a.out_h) a.out_h[tile_g] = nh;
			}
		}
		PXZ_STAMP(2);  // detector + reduction + level decision + metadata
		if (FULL || a.out_px != nullptr) {
			uint32_t filt = a.filter;
			asm volatile("" : "+s"(filt));  // a scalar compare per use, not a hoisted (and spilled) mask
			pend_dst = a.out_px + (size_t)tile_g * 4096u;
			pend_px = nw * nh;
			if (nw == 32u && nh == 32u) {
				pend_kind = 2;  // the planes themselves, re-interleaved by the flush
			} else if (nw != 32u && nh != 32u && filt != 0) {
				const uint32_t lx = m0 < (uint32_t)kMaxLevel ? m0 : (uint32_t)kMaxLevel - 1;
				const uint32_t ly = m1 < (uint32_t)kMaxLevel ? m1 : (uint32_t)kMaxLevel - 1;
				if (a.tabs[lx].mf_off != 0 && a.tabs[ly].mf_off != 0 && nw >= 4u && nh >= 4u) {
					// (2- and 1-px outputs have tables too -- shrink32a_kernel uses them -- but the dot2 form is cheaper there)
					resample_mfma32<3>(s_tab, a.tabs[lx], a.tabs[ly], s_pl, tid, nw, nh, s_tmp);
					pend_src = s_tmp;
					pend_kind = 1;
				} else if (nw <= 8u) {
					// the vertical pass reads only the transposed planes: the R plane is free for the pixels
					resample_fast32_hv(s_tab, a.tabs[lx], a.tabs[ly], s_pl, s_tmp, tid, nw, nh, s_pl);
					pend_src = s_pl;
					pend_kind = 1;
				} else {
					defer();  // 16 x (2|1): its transposed planes would not fit the 16-waves-per-CU LDS image
				}
			} else if (filt == 0) {
				// ResizeAlg::Nearest (mod.rs:277): source index = floor((o + 0.5) * 2^m), no alpha handling
				const uint32_t lgx = 31u - (uint32_t)__builtin_clz(nw);
				const uint32_t hx = m0 ? (1u << (m0 < 6u ? m0 - 1u : 4u)) : 0u, hy = m1 ? (1u << (m1 < 6u ? m1 - 1u : 4u)) : 0u;
				const uint16_t *pl16 = reinterpret_cast<const uint16_t *>(s_pl);
				for (uint32_t i = tid; i < nw * nh; i += 64u) {
					const uint32_t ox = i & (nw - 1u), oy = i >> lgx;
					// 32 -> nw = 32 >> m (m <= 5): index (2o+1) * 2^(m-1); m >= 5 gives the single index 16
					const uint32_t x = m0 == 0 ? ox : (m0 < 6u ? (2u * ox + 1u) * hx : 16u);
					const uint32_t y = m1 == 0 ? oy : (m1 < 6u ? (2u * oy + 1u) * hy : 16u);
					const uint32_t idx = y * (2u * kRS32) + x;
					s_tmp[i] = (uint32_t)pl16[idx] | ((uint32_t)pl16[idx + 2u * kPD32] << 8) | ((uint32_t)pl16[idx + 4u * kPD32] << 16) | 0xff000000u;
				}
				pend_kind = 1;
				pend_src = s_tmp;
			} else if (nh != 32u) {
				const uint32_t ly = m1 < (uint32_t)kMaxLevel ? m1 : (uint32_t)kMaxLevel - 1;
				fast32_v_only(s_tab, a.tabs[ly], s_pl, tid, nh, s_tmp);  // width kept; the transposed planes are unused here
				pend_kind = 1;
				pend_src = s_tmp;
			} else {
				const uint32_t lx = m0 < (uint32_t)kMaxLevel ? m0 : (uint32_t)kMaxLevel - 1;
				fast32_h_only(s_tab, a.tabs[lx], s_pl, tid, nw, s_tmp);  // height kept
				pend_kind = 1;
				pend_src = s_tmp;
			}
		}
		tile_sync<1>();  // the next tile reuses this wave's LDS image
		PXZ_STAMP(3);  // clone / resample / defer
	};
#ifdef PXZ_STAMPS
	uint32_t st_tiles = 0;
#endif
	for (uint32_t tile_g = first; tile_g < a.n_tiles;) {
#ifdef PXZ_STAMPS
		++st_tiles;
#endif
		const uint32_t tile_next = next_ticket();
		one_tile(tile_g, tile_next);
		tile_g = tile_next;
	}
	flush();  // the last tile's pixels
	list_flush(s_batch, n_listb, a.work + kWorkList, a.work + a.work_slot, tid);
	list_flush(s_batch + kListBatch, n_lista, a.work + kWorkList + a.n_tiles, a.work + kWorkA + a.work_slot, tid);
#ifdef PXZ_STAMPS
	if (tid == 0) {
		unsigned long long *out = reinterpret_cast<unsigned long long *>(a.work + ((2u * a.n_tiles + kWorkList + 1u + 1u) & ~1u));
		for (int i = 0; i < 8; ++i) atomicAdd(out + i, st_acc[i]);
		// per-wave run time (100 MHz ticks) | tiles processed << 48; last launch wins
		out[8 + blockIdx.x * 16u + sub] = ((wall_clock64() - st_begin) & 0xffffffffffffull) | ((unsigned long long)st_tiles << 48);
		if (sub == 0) out[8 + blockIdx.x * 16u + 15u] = st_begin;
	}
#endif
}

// ---------------------------------------------------------------------------
// shrink32a_kernel: the full 32x32 RGBA tiles WITH transparency that shrink32_kernel set aside (list A of the
// worklist buffer).  Same staging / detector / level decision, but a fourth LDS plane keeps the alpha channel
// and the resample is fir's U8x4 path: colours premultiplied in place (packed u16 arithmetic), all four planes
// through the matrix-core resample (every output size <= 16), every output pixel un-premultiplied.  Clone and
// nearest pick from the four planes as they are; the one-pass classes (32 x n, n x 32) go on to the generic
// kernel (list B).  13 waves per CU (four planes); outputs are stored directly.
// ---------------------------------------------------------------------------
template <int MODE>
__global__ void __launch_bounds__(1024) shrink32a_kernel(const Fast32Args a)
{
	extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
	const uint32_t wpb = blockDim.x / 64u, sub = threadIdx.x / 64u, tid = threadIdx.x % 64u;
	for (uint32_t i = threadIdx.x; i < a.tab_dw / 4u; i += blockDim.x)
		reinterpret_cast<uint4 *>(lds)[i] = reinterpret_cast<const uint4 *>(a.trows)[i];
	uint32_t *s_ticket = lds + a.tab_dw + wpb * a.tile_dw;
	if (threadIdx.x == 0) *s_ticket = wpb;
	__syncthreads();
	const uint32_t *s_tab = lds;
	const uint32_t brk_lane = tid < (uint32_t)kMaxLevel ? a.breaks[tid] : (a.breaks_asc ? 0xffffffffu : 0u);
	auto level_of = [&](uint32_t key) -> uint32_t {
		const unsigned long long lt = __builtin_amdgcn_ballot_w64(key < brk_lane);
		const unsigned long long live = (1ull << kMaxLevel) - 1ull;
		return (uint32_t)__builtin_popcountll((a.breaks_asc ? ~lt : lt) & live);
	};
	uint32_t *s_pl = lds + a.tab_dw + sub * a.tile_dw;  // four planes: R, G, B, A
	uint32_t *s_batch = s_ticket + 4u + sub * kListBatch;  // this wave's pending list-B entries
	uint32_t n_listb = 0;
	const uint32_t count = __builtin_amdgcn_readfirstlane(a.work[kWorkA + a.work_slot]);
	const uint32_t *list = a.work + kWorkList + a.n_tiles;
	auto tile_of_ticket = [&](uint32_t t) -> uint32_t {
		const unsigned long long i = (unsigned long long)blockIdx.x + (unsigned long long)t * gridDim.x;
		return i < (unsigned long long)count ? list[(uint32_t)i] : 0xffffffffu;
	};
	auto next_ticket = [&]() -> uint32_t {
		uint32_t t = 0;
		if (tid == 0) t = atomicAdd(s_ticket, 1u);
		return __builtin_amdgcn_readfirstlane(tile_of_ticket(__builtin_amdgcn_readfirstlane(t)));
	};
	uint4 pre[4];
	bool pre_valid = false;
	const uint32_t first = __builtin_amdgcn_readfirstlane(tile_of_ticket(__builtin_amdgcn_readfirstlane(sub)));
	fast32_prefetch(a, first, tid, pre, pre_valid);
	for (uint32_t tile_g = first; tile_g < a.n_tiles;) {
		const uint32_t tile_next = next_ticket();
		auto defer = [&]() {  // on to the generic kernel (list B); the marker in sums[] is there already (MODE 1) / not wanted (MODE 0)
			list_push(s_batch, n_listb, tile_g, a.work + kWorkList, a.work + a.work_slot, tid);
		};
		uint32_t given_bits = 0;
		if constexpr (MODE == 0) given_bits = a.sums[2 * tile_g];
		// ---- stage: registers -> four planes of u16 pairs (only full, aligned tiles are ever listed)
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const uint32_t row = (tid >> 3) + 8u * (uint32_t)k, col = tid & 7u;
			const uint4 v = pre[k];
			uint32_t *d = s_pl + row * kRS32 + col * 2u;
#pragma unroll
			for (uint32_t c = 0; c < 4; ++c) {
				const uint32_t sel = c | 0x0c000c00u | ((4u + c) << 16);
				uint2 pr;
				pr.x = __builtin_amdgcn_perm(v.y, v.x, sel);
				pr.y = __builtin_amdgcn_perm(v.w, v.z, sel);
				*reinterpret_cast<uint2 *>(d + c * kPD32) = pr;
			}
		}
		fast32_prefetch(a, tile_next, tid, pre, pre_valid);
		tile_sync<1>();
		// ---- detector + level decision (as shrink32_kernel: the colour planes are still as loaded)
		uint32_t m0, m1;
		if constexpr (MODE == 1) {
			uint32_t sum_hz = 0, sum_vr = 0;
			const uint32_t q = tid & 15u, g = tid >> 4;
			const uint32_t two = 0x00020002u;
			const uint32_t *pc[3];
			pc[0] = s_pl + g * (8u * kRS32) + q;
			pc[1] = pc[0] + kPD32;
			pc[2] = pc[1] + kPD32;
			uint32_t rA[3], rB[3], tP[3], dP[3];
#pragma unroll
			for (int c = 0; c < 3; ++c) {
				const uint32_t a0 = pc[c][0], a1 = pc[c][1], b0 = pc[c][kRS32], b1 = pc[c][kRS32 + 1];
				rA[c] = pk_mad_u16(__builtin_amdgcn_alignbit(a1, a0, 16), two, u32(us2(a0) + us2(a1)));
				rB[c] = pk_mad_u16(__builtin_amdgcn_alignbit(b1, b0, 16), two, u32(us2(b0) + us2(b1)));
				tP[c] = u32(us2(a0) + us2(b0));
				dP[c] = b0;
			}
#pragma unroll
			for (int st = 0; st < 4; ++st) {
				if (st < 3 || g != 3u) {
#pragma unroll
					for (int c = 0; c < 3; ++c) {
						const uint32_t *pr = pc[c] + (2 + 2 * st) * (int)kRS32;
						const uint32_t n0 = pr[0], n1 = pr[1], o0 = pr[kRS32], o1 = pr[kRS32 + 1];
						const uint32_t rN = pk_mad_u16(__builtin_amdgcn_alignbit(n1, n0, 16), two, u32(us2(n0) + us2(n1)));
						sum_hz = sad16(rN, rA[c], sum_hz);
						const uint32_t tN = u32(us2(dP[c]) + us2(n0));
						const uint32_t c0 = u32(us2(tP[c]) + us2(tN));
						sum_vr = sad16(dpp_mov<0x101>(c0), c0, sum_vr);
						const uint32_t rO = pk_mad_u16(__builtin_amdgcn_alignbit(o1, o0, 16), two, u32(us2(o0) + us2(o1)));
						sum_hz = sad16(rO, rB[c], sum_hz);
						const uint32_t tO = u32(us2(n0) + us2(o0));
						const uint32_t e0 = u32(us2(tN) + us2(tO));
						sum_vr = sad16(dpp_mov<0x101>(e0), e0, sum_vr);
						rA[c] = rN;
						rB[c] = rO;
						tP[c] = tO;
						dP[c] = o0;
					}
				}
			}
			if (q == 15u) sum_hz = sum_vr = 0;
			sum_hz = wave_sum_sgpr(sum_hz);
			sum_vr = wave_sum_sgpr(sum_vr);
			m0 = level_of(sum_hz);
			m1 = level_of(sum_vr);
			if (tid == 0) reinterpret_cast<uint2 *>(a.sums)[tile_g] = make_uint2(sum_hz, sum_vr);
		} else {
			const uint32_t vb = __builtin_amdgcn_readfirstlane(given_bits);
			m0 = m1 = level_of(__float_as_uint(parse_value(__uint_as_float(vb))));
		}
		const uint32_t nw = reduced_size(32u, m0), nh = reduced_size(32u, m1);
		const bool one_pass = (nw == 32u) != (nh == 32u) && a.filter != 0;
		if (one_pass) {
			// 32 x n / n x 32 with transparency: generic kernel.  MODE 1: it must not be finished from these sums
			if constexpr (MODE == 1) {
				if (tid == 0) reinterpret_cast<uint2 *>(a.sums)[tile_g] = make_uint2(kDeferredKey, kDeferredKey);
			}
			defer();
			tile_sync<1>();
			tile_g = tile_next;
			continue;
		}
		if (tid == 0) {
			if (a.out_w) a.out_w[tile_g] = nw;
			if (a.out_h) a.out_h[tile_g] = nh;
		}
		uint32_t *dst = reinterpret_cast<uint32_t *>(a.out_px + (size_t)tile_g * 4096u);
		if (nw == 32u && nh == 32u) {
			// clone (block.rs:279-281): re-interleave the four planes
#pragma unroll
			for (int k = 0; k < 4; ++k) {
				const uint32_t i = tid + 64u * (uint32_t)k;
				const uint32_t *p = s_pl + (i >> 3) * kRS32 + (i & 7u) * 2u;
				const uint2 r = *reinterpret_cast<const uint2 *>(p), g = *reinterpret_cast<const uint2 *>(p + kPD32);
				const uint2 b = *reinterpret_cast<const uint2 *>(p + 2 * kPD32), al = *reinterpret_cast<const uint2 *>(p + 3 * kPD32);
				const uint32_t rg01 = __builtin_amdgcn_perm(g.x, r.x, 0x06020400u), ba01 = __builtin_amdgcn_perm(al.x, b.x, 0x06020400u);
				const uint32_t rg23 = __builtin_amdgcn_perm(g.y, r.y, 0x06020400u), ba23 = __builtin_amdgcn_perm(al.y, b.y, 0x06020400u);
				uint4 o;
				o.x = __builtin_amdgcn_perm(ba01, rg01, 0x05040100u);
				o.y = __builtin_amdgcn_perm(ba01, rg01, 0x07060302u);
				o.z = __builtin_amdgcn_perm(ba23, rg23, 0x05040100u);
				o.w = __builtin_amdgcn_perm(ba23, rg23, 0x07060302u);
				reinterpret_cast<uint4 *>(dst)[i] = o;
			}
		} else if (a.filter == 0) {
			// ResizeAlg::Nearest (mod.rs:277): pick, no alpha handling
			const uint32_t lgx = 31u - (uint32_t)__builtin_clz(nw);
			const uint32_t hx = m0 ? (1u << (m0 < 6u ? m0 - 1u : 4u)) : 0u, hy = m1 ? (1u << (m1 < 6u ? m1 - 1u : 4u)) : 0u;
			const uint16_t *pl16 = reinterpret_cast<const uint16_t *>(s_pl);
			for (uint32_t i = tid; i < nw * nh; i += 64u) {
				const uint32_t ox = i & (nw - 1u), oy = i >> lgx;
				const uint32_t x = m0 == 0 ? ox : (m0 < 6u ? (2u * ox + 1u) * hx : 16u);
				const uint32_t y = m1 == 0 ? oy : (m1 < 6u ? (2u * oy + 1u) * hy : 16u);
				const uint32_t idx = y * (2u * kRS32) + x;
				dst[i] = (uint32_t)pl16[idx] | ((uint32_t)pl16[idx + 2u * kPD32] << 8) | ((uint32_t)pl16[idx + 4u * kPD32] << 16) |
				         ((uint32_t)pl16[idx + 6u * kPD32] << 24);
			}
		} else {
			// fir, U8x4: premultiply the colour planes in place -- mul_div_255 on both pixels of a dword at once:
			// t = v*a + 128 <= 65153, t + (t >> 8) <= 65407: nothing leaves its 16-bit half
#pragma unroll
			for (uint32_t it = 0; it < 8; ++it) {
				const uint32_t i = tid + 64u * it;
				uint32_t *p = s_pl + (i >> 4) * kRS32 + (i & 15u);
				const ushort2v al = us2(p[3 * kPD32]);
#pragma unroll
				for (uint32_t c = 0; c < 3; ++c) {
					ushort2v t = us2(p[c * kPD32]) * al + (ushort2v)(128);
					t = t + (t >> (ushort2v)(8));
					p[c * kPD32] = u32(t >> (ushort2v)(8));
				}
			}
			tile_sync<1>();
			const uint32_t lx = m0 < (uint32_t)kMaxLevel ? m0 : (uint32_t)kMaxLevel - 1;
			const uint32_t ly = m1 < (uint32_t)kMaxLevel ? m1 : (uint32_t)kMaxLevel - 1;
			resample_mfma32<4>(s_tab, a.tabs[lx], a.tabs[ly], s_pl, tid, nw, nh, dst);
		}
		tile_sync<1>();  // the next tile reuses this wave's LDS image
		tile_g = tile_next;
	}
	list_flush(s_batch, n_listb, a.work + kWorkList, a.work + a.work_slot, tid);
}

// ---------------------------------------------------------------------------
// shrink16_kernel: 16x16 RGBA tiles, four at a time -- a 2x2 group of tiles is one 32x32 region, loaded,
// staged and scanned by the detector exactly like a tile of shrink32_kernel; only the windows that would
// straddle two tiles are left out, the sums are kept per tile (segmented reduction), and each of the four
// tiles then gets its own level decision and its own small resample out of the shared LDS image.
// Groups are dealt to the waves of a block through the LDS ticket counter.  Groups with a ragged or missing
// tile, tiles with transparency and the one-pass classes (16 x n, n x 16) go to the worklist.
// MODE 1: directional detector here; MODE 0: values already in sums[] (oklab_kernel<16>).
// ---------------------------------------------------------------------------
// FULL: as in shrink32_kernel (all three output arrays are there; no run-time tests of them in the loop).
template <int MODE, bool FULL>
__global__ void __launch_bounds__(1024) shrink16_kernel(const Fast32Args a)
{
	extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
	const uint32_t wpb = blockDim.x / 64u, sub = threadIdx.x / 64u, tid = threadIdx.x % 64u;
	for (uint32_t i = threadIdx.x; i < a.tab_dw / 4u; i += blockDim.x)
		reinterpret_cast<uint4 *>(lds)[i] = reinterpret_cast<const uint4 *>(a.trows)[i];
	uint32_t *s_ticket = lds + a.tab_dw + wpb * a.tile_dw;
	if (threadIdx.x == 0) *s_ticket = wpb;
	__syncthreads();
	const uint32_t *s_tab = lds;
	const uint32_t brk_lane = tid < (uint32_t)kMaxLevel ? a.breaks[tid] : (a.breaks_asc ? 0xffffffffu : 0u);
	auto level_of = [&](uint32_t key) -> uint32_t {
		const unsigned long long lt = __builtin_amdgcn_ballot_w64(key < brk_lane);
		const unsigned long long live = (1ull << kMaxLevel) - 1ull;
		return (uint32_t)__builtin_popcountll((a.breaks_asc ? ~lt : lt) & live);
	};
	uint32_t *s_pl = lds + a.tab_dw + sub * a.tile_dw;
	uint32_t *s_tmp = s_pl + 3 * kPD32;
	uint32_t *s_batch = s_ticket + 4u + sub * kListBatch;  // this wave's pending list-B entries
	uint32_t n_listb = 0;
	// groups of 2x2 tiles: gcols x grows per frame
	const uint32_t gcols = (a.cols + 1u) >> 1, grows = (a.rows + 1u) >> 1, gpf = gcols * grows;
	auto group_of_ticket = [&](uint32_t t) -> uint32_t {
		const unsigned long long run = (unsigned long long)(t >> a.chunk_lg) * gridDim.x + blockIdx.x;
		const unsigned long long g = (run << a.chunk_lg) + (t & ((1u << a.chunk_lg) - 1u));
		return g < (unsigned long long)a.n_groups ? (uint32_t)g : 0xffffffffu;
	};
	auto next_ticket = [&]() -> uint32_t {
		uint32_t t = 0;
		if (tid == 0) t = atomicAdd(s_ticket, 1u);
		return group_of_ticket(__builtin_amdgcn_readfirstlane(t));
	};
	// a group's place: frame, (gx, gy); full: all four tiles exist, are full-size and the batch is aligned
	struct Place {
		uint32_t frame, gx, gy;
		bool full;
		const uint8_t *src;
	};
	auto place_of = [&](uint32_t grp) -> Place {
		Place p{0, 0, 0, false, nullptr};
		if (grp >= a.n_groups) return p;
		p.frame = fastdiv(grp, a.div_gpf);
		const uint32_t r = grp - p.frame * gpf;
		p.gy = fastdiv(r, a.div_gcols);
		p.gx = r - p.gy * gcols;
		p.src = a.src + (size_t)p.frame * a.frame_stride + (size_t)(p.gy * 32u) * a.pitch + (size_t)(p.gx * 32u) * 4u;
		p.full = 2u * p.gx + 1u < a.full_cols && 2u * p.gy + 1u < a.full_rows;
		return p;
	};
	uint4 pre[4];
	bool pre_valid = false;
	auto prefetch = [&](uint32_t grp) {
		const Place p = place_of(grp);
		pre_valid = p.full;
		if (pre_valid) {
			const uint8_t *q = p.src + (size_t)(tid >> 3) * a.pitch + (tid & 7u) * 16u;
#pragma unroll
			for (int k = 0; k < 4; ++k) pre[k] = *reinterpret_cast<const uint4 *>(q + (size_t)(8 * k) * a.pitch);
		}
	};
	const uint32_t first = group_of_ticket(__builtin_amdgcn_readfirstlane(sub));
	prefetch(first);
	for (uint32_t grp = first; grp < a.n_groups;) {
		const uint32_t grp_next = next_ticket();
		const Place pl = place_of(grp);
		// tile ids of the group: t(dx, dy) = frame * tiles_per_frame + (2 gy + dy) * cols + 2 gx + dx
		const uint32_t t00 = pl.frame * a.tiles_per_frame + (2u * pl.gy) * a.cols + 2u * pl.gx;
		auto tile_id = [&](uint32_t k) -> uint32_t { return t00 + (k & 1u) + (k >> 1) * a.cols; };
		auto defer_tile = [&](uint32_t t) {
			list_push(s_batch, n_listb, t, a.work + kWorkList, a.work + a.work_slot, tid);
			if (tid == 0) {
				bool keep = false;  // MODE 0: a value the block-cooperative detector left is final and stays
				if constexpr (MODE == 0) {
					const uint32_t tt = t - pl.frame * a.tiles_per_frame;
					const uint32_t ty = fastdiv(tt, a.div_cols), tx = tt - ty * a.cols;
					keep = tx < a.full_cols && ty < a.ok_rows;
				}
				if (!keep) reinterpret_cast<uint2 *>(a.sums)[t] = make_uint2(kDeferredKey, kDeferredKey);
			}
		};
		if (!pre_valid) {
			// a ragged or incomplete group (or an unaligned batch): its tiles one by one to the generic kernel
#pragma unroll
			for (uint32_t k = 0; k < 4; ++k)
				if (2u * pl.gx + (k & 1u) < a.cols && 2u * pl.gy + (k >> 1) < a.rows) defer_tile(tile_id(k));
			prefetch(grp_next);
			grp = grp_next;
			continue;
		}
		uint32_t given[4] = {0, 0, 0, 0};
		if constexpr (MODE == 0) {
#pragma unroll
			for (uint32_t k = 0; k < 4; ++k) given[k] = a.sums[2 * tile_id(k)];
		}
		// ---- stage: registers -> planar u16 pairs (as shrink32_kernel)
		uint32_t alpha_and = 0xffu;
#pragma unroll
		for (int k = 0; k < 4; ++k) alpha_and &= (pre[k].x & pre[k].y & pre[k].z & pre[k].w) >> 24;
		const bool transparent = __builtin_amdgcn_ballot_w64(alpha_and != 0xffu) != 0ull;
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const uint32_t row = (tid >> 3) + 8u * (uint32_t)k, col = tid & 7u;
			const uint4 v = pre[k];
			uint32_t *d = s_pl + row * kRS32 + col * 2u;
#pragma unroll
			for (uint32_t c = 0; c < 3; ++c) {
				const uint32_t sel = c | 0x0c000c00u | ((4u + c) << 16);
				uint2 pr;
				pr.x = __builtin_amdgcn_perm(v.y, v.x, sel);
				pr.y = __builtin_amdgcn_perm(v.w, v.z, sel);
				*reinterpret_cast<uint2 *>(d + c * kPD32) = pr;
			}
		}
		prefetch(grp_next);
		if (transparent) {
			// (one transparent tile sends the whole group: the generic kernel has the alpha plane)
#pragma unroll
			for (uint32_t k = 0; k < 4; ++k) defer_tile(tile_id(k));
			grp = grp_next;
			continue;
		}
		tile_sync<1>();
		// ---- detector: as shrink32_kernel, minus the windows that would straddle two tiles
		uint32_t m0[4], m1[4], key0[4], key1[4];
		if constexpr (MODE == 1) {
			uint32_t sum_hz = 0, sum_vr = 0;
			const uint32_t q = tid & 15u, g = tid >> 4;
			const uint32_t two = 0x00020002u;
			const uint32_t *pc[3];
			pc[0] = s_pl + g * (8u * kRS32) + q;
			pc[1] = pc[0] + kPD32;
			pc[2] = pc[1] + kPD32;
			uint32_t rA[3], rB[3], tP[3], dP[3];
#pragma unroll
			for (int c = 0; c < 3; ++c) {
				const uint32_t a0 = pc[c][0], a1 = pc[c][1], b0 = pc[c][kRS32], b1 = pc[c][kRS32 + 1];
				rA[c] = pk_mad_u16(__builtin_amdgcn_alignbit(a1, a0, 16), two, u32(us2(a0) + us2(a1)));
				rB[c] = pk_mad_u16(__builtin_amdgcn_alignbit(b1, b0, 16), two, u32(us2(b0) + us2(b1)));
				tP[c] = u32(us2(a0) + us2(b0));
				dP[c] = b0;
			}
#pragma unroll
			for (int st = 0; st < 4; ++st) {
				if (st < 3 || (g & 1u) == 0u) {  // groups 1 and 3 hold window rows 8 .. 13 of their tiles: 3 steps
#pragma unroll
					for (int c = 0; c < 3; ++c) {
						const uint32_t *pr = pc[c] + (2 + 2 * st) * (int)kRS32;
						const uint32_t n0 = pr[0], n1 = pr[1], o0 = pr[kRS32], o1 = pr[kRS32 + 1];
						const uint32_t rN = pk_mad_u16(__builtin_amdgcn_alignbit(n1, n0, 16), two, u32(us2(n0) + us2(n1)));
						sum_hz = sad16(rN, rA[c], sum_hz);
						const uint32_t tN = u32(us2(dP[c]) + us2(n0));
						const uint32_t c0 = u32(us2(tP[c]) + us2(tN));
						sum_vr = sad16(dpp_mov<0x101>(c0), c0, sum_vr);
						const uint32_t rO = pk_mad_u16(__builtin_amdgcn_alignbit(o1, o0, 16), two, u32(us2(o0) + us2(o1)));
						sum_hz = sad16(rO, rB[c], sum_hz);
						const uint32_t tO = u32(us2(n0) + us2(o0));
						const uint32_t e0 = u32(us2(tN) + us2(tO));
						sum_vr = sad16(dpp_mov<0x101>(e0), e0, sum_vr);
						rA[c] = rN;
						rB[c] = rO;
						tP[c] = tO;
						dP[c] = o0;
					}
				}
			}
			if ((q & 7u) == 7u) sum_hz = sum_vr = 0;  // pairs 7 and 15 start no window inside their tile
			// per tile: 8 lanes (q >> 3) of two 16-lane rows (g >> 1): sum inside the 8-lane groups, then pick
			sum_hz = (uint32_t)group_sum<8>((int32_t)sum_hz);
			sum_vr = (uint32_t)group_sum<8>((int32_t)sum_vr);
#pragma unroll
			for (uint32_t k = 0; k < 4; ++k) {
				const uint32_t l0 = 32u * (k >> 1) + 8u * (k & 1u);  // first lane of tile k's first row; its second row is 16 on
				key0[k] = (uint32_t)__builtin_amdgcn_readlane((int)sum_hz, l0) + (uint32_t)__builtin_amdgcn_readlane((int)sum_hz, l0 + 16u);
				key1[k] = (uint32_t)__builtin_amdgcn_readlane((int)sum_vr, l0) + (uint32_t)__builtin_amdgcn_readlane((int)sum_vr, l0 + 16u);
				m0[k] = level_of(key0[k]);
				m1[k] = level_of(key1[k]);
			}
		} else {
#pragma unroll
			for (uint32_t k = 0; k < 4; ++k) {
				const uint32_t vb = __builtin_amdgcn_readfirstlane(given[k]);
				key0[k] = key1[k] = vb;
				m0[k] = m1[k] = level_of(__float_as_uint(parse_value(__uint_as_float(vb))));
			}
		}
		uint32_t nw[4], nh[4];
#pragma unroll
		for (uint32_t k = 0; k < 4; ++k) {
			nw[k] = reduced_size(16u, m0[k]);
			nh[k] = reduced_size(16u, m1[k]);
		}
		// ---- per tile: metadata, then clone / nearest / two-pass resample straight into its slot
		uint32_t filt = a.filter;
		asm volatile("" : "+s"(filt));  // scalar compares per use instead of a hoisted mask
#pragma unroll
		for (uint32_t k = 0; k < 4; ++k) {
			const uint32_t t = tile_id(k);
			const bool one_pass = (FULL || a.out_px != nullptr) && (nw[k] == 16u) != (nh[k] == 16u) && filt != 0;
			if (one_pass) {  // 16 x n, n x 16: generic kernel (it writes the tile's metadata itself)
				defer_tile(t);
				continue;
			}
			{
				uint32_t lane = tid;
				asm volatile("" : "+v"(lane));  // a fresh compare, not a hoisted (and spilled) lane mask
				if (lane == 0) {
					reinterpret_cast<uint2 *>(a.sums)[t] = make_uint2(key0[k], key1[k]);
					if (FULL || a.out_w) a.out_w[t] = nw[k];
					if (FULL || a.out_h) a.out_h[t] = nh[k];
				}
			}
			if (!FULL && a.out_px == nullptr) continue;
			uint32_t *dst = reinterpret_cast<uint32_t *>(a.out_px + (size_t)t * 1024u);
			const uint32_t *tile_pl = s_pl + (16u * (k >> 1)) * kRS32 + 8u * (k & 1u);  // first pixel pair of the tile
			if (nw[k] == 16u && nh[k] == 16u) {
				// clone (block.rs:279-281): 64 groups of 4 pixels, one per lane
				const uint32_t row = tid >> 2, c4 = tid & 3u;
				const uint32_t *p = tile_pl + row * kRS32 + c4 * 2u;
				const uint2 r = *reinterpret_cast<const uint2 *>(p), gch = *reinterpret_cast<const uint2 *>(p + kPD32);
				const uint2 b = *reinterpret_cast<const uint2 *>(p + 2 * kPD32);
				const uint32_t opq = 0x00ff00ffu;
				const uint32_t rg01 = __builtin_amdgcn_perm(gch.x, r.x, 0x06020400u), ba01 = __builtin_amdgcn_perm(opq, b.x, 0x06020400u);
				const uint32_t rg23 = __builtin_amdgcn_perm(gch.y, r.y, 0x06020400u), ba23 = __builtin_amdgcn_perm(opq, b.y, 0x06020400u);
				uint4 o;
				o.x = __builtin_amdgcn_perm(ba01, rg01, 0x05040100u);
				o.y = __builtin_amdgcn_perm(ba01, rg01, 0x07060302u);
				o.z = __builtin_amdgcn_perm(ba23, rg23, 0x05040100u);
				o.w = __builtin_amdgcn_perm(ba23, rg23, 0x07060302u);
				reinterpret_cast<uint4 *>(dst)[tid] = o;
			} else if (filt == 0) {
				// ResizeAlg::Nearest: source index = floor((o + 0.5) * 2^m); any (nw, nh)
				const uint32_t mx = m0[k], my = m1[k];
				const uint32_t lgx = 31u - (uint32_t)__builtin_clz(nw[k]);
				const uint16_t *pl16 = reinterpret_cast<const uint16_t *>(tile_pl);
				for (uint32_t i = tid; i < nw[k] * nh[k]; i += 64u) {
					const uint32_t ox = i & (nw[k] - 1u), oy = i >> lgx;
					const uint32_t x = mx == 0 ? ox : (mx < 5u ? (2u * ox + 1u) << (mx - 1u) : 8u);
					const uint32_t y = my == 0 ? oy : (my < 5u ? (2u * oy + 1u) << (my - 1u) : 8u);
					const uint32_t idx = y * (2u * kRS32) + x;
					dst[i] = (uint32_t)pl16[idx] | ((uint32_t)pl16[idx + 2u * kPD32] << 8) | ((uint32_t)pl16[idx + 4u * kPD32] << 16) | 0xff000000u;
				}
			} else {
				const uint32_t lx = m0[k] < (uint32_t)kMaxLevel ? m0[k] : (uint32_t)kMaxLevel - 1;
				const uint32_t ly = m1[k] < (uint32_t)kMaxLevel ? m1[k] : (uint32_t)kMaxLevel - 1;
				resample_fast16_hv(s_tab, a.tabs[lx], a.tabs[ly], tile_pl, s_tmp, tid, nw[k], nh[k], dst);
			}
		}
		tile_sync<1>();  // the next group reuses this wave's LDS image
		grp = grp_next;
	}
	list_flush(s_batch, n_listb, a.work + kWorkList, a.work + a.work_slot, tid);
}

// ---------------------------------------------------------------------------
// shrink64_kernel: full, 16-byte-aligned, opaque 64x64 RGBA tiles -- the reference CLI's default block size
// (src/bin/main.rs:19) -- directional detector + clone / two-pass matrix-core resample.  One tile per block
// of four waves; every phase splits four ways:
//   stage      wave w loads and stages rows 16w .. 16w+15 (prefetched into registers during the previous tile)
//   detector   wave w sums the windows whose top row is 16w .. 16w+15 (32 column pairs x 2 groups of 8 rows;
//              the neighbour pair comes through a wave-wide DPP shift); partial sums meet in LDS
//   horizontal wave w = one 16-row block of A operands (pixels - 128 as bytes, 64 per row = ONE
//              v_mfma_i32_16x16x64_i8 per output block and weight byte); results (u8) go to LDS as [ox][y]
//   vertical   wave w = one (16 output rows, 16 output columns) block: weights x the LDS columns
// Three block barriers per tile.  Ragged-edge tiles, tiles with transparency and the one-pass classes
// (64 x n, n x 64) go to the worklist of the generic kernel.
// ---------------------------------------------------------------------------
constexpr uint32_t kRS64 = 36, kPD64 = 36 * 64;  // plane row stride (32 + 4 dwords: bank skew, rows stay 16-byte aligned), plane size
constexpr uint32_t kTS64 = 20;                   // dwords per column of the horizontal pass: 16 (64 bytes of y) + 4 of bank skew
constexpr uint32_t kRed64 = 48;                  // partial sums, flags, ticket, two worklist batches
constexpr uint32_t lds64_dwords(uint32_t nch) { return nch * kPD64 + nch * 32u * kTS64 + kRed64; }  // planes + [channel][ox < 32][kTS64] + s_red

template <class Args>
__device__ __forceinline__ bool fast64_tile_src(const Args &a, uint32_t tile_g, const uint8_t *&src)
{
	if (tile_g >= a.n_tiles) return false;
	const uint32_t frame = fastdiv(tile_g, a.div_tpf);
	const uint32_t t = tile_g - frame * a.tiles_per_frame;
	const uint32_t ty = fastdiv(t, a.div_cols), tx = t - ty * a.cols;
	src = a.src + (size_t)frame * a.frame_stride + (size_t)(ty * 64u) * a.pitch + (size_t)(tx * 64u) * 4u;
	return tx < a.full_cols && ty < a.full_rows;
}

// ALPHA: the full tiles WITH transparency the opaque kernel put on list A -- a fourth plane keeps the alpha
// channel, the colours are premultiplied in place once the detector is done with them (fir's U8x4 path), all
// four planes go through the passes and every output pixel is un-premultiplied.
// FULL: as in shrink32_kernel (out_px, out_w, out_h all there: no run-time tests of them in the tile loop).
template <int MODE, bool ALPHA, bool FULL>
__global__ void __launch_bounds__(256) shrink64_kernel(const Fast64Args a)
{
	constexpr uint32_t NCH = ALPHA ? 4u : 3u;
	extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
	uint32_t *s_pl = lds;                       // NCH planes of u16 pairs
	uint32_t *s_t = lds + NCH * kPD64;          // horizontal-pass results
	uint32_t *s_red = s_t + NCH * 32u * kTS64;  // [0..7] partial sums, [8..11] alpha, [13] ticket, [16..31] list-B batch, [32..47] list-A batch
	const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
	const uint32_t brk_lane = lane < (uint32_t)kMaxLevel ? a.breaks[lane] : (a.breaks_asc ? 0xffffffffu : 0u);
	auto level_of = [&](uint32_t key) -> uint32_t {
		const unsigned long long lt = __builtin_amdgcn_ballot_w64(key < brk_lane);
		const unsigned long long live = (1ull << kMaxLevel) - 1ull;
		return (uint32_t)__builtin_popcountll((a.breaks_asc ? ~lt : lt) & live);
	};
	// this lane's share of a tile: rows 16w + (lane >> 4) + 4k, 16 bytes at column quad lane & 15
	uint4 pre[4];
	bool pre_valid = false;
	auto prefetch = [&](uint32_t tile_g) {
		const uint8_t *src;
		pre_valid = fast64_tile_src(a, tile_g, src);
		if (pre_valid) {
			const uint8_t *p = src + (size_t)(16u * wave + (lane >> 4)) * a.pitch + (lane & 15u) * 16u;
#pragma unroll
			for (int k = 0; k < 4; ++k) pre[k] = *reinterpret_cast<const uint4 *>(p + (size_t)(4 * k) * a.pitch);
		}
	};
	// Tiles are dealt on demand: tile costs differ several times between size classes, and with ~64 tiles per
	// block a static stride leaves the unluckiest of a thousand blocks far behind.  kTicketCounters global
	// counters (one per residue class of the block index, so ~16 blocks share one and an address sees a few
	// atomics per microsecond); counter c owns tiles c, c + n_ctr, c + 2 n_ctr, ...  A block's first two tiles
	// are fixed; every iteration draws the ticket of the tile after next at its start and hands it to the
	// other waves across a barrier the iteration has anyway.
	// (ALPHA: the items are the entries of list A, in a fixed rotation over the blocks)
	const uint32_t n_ctr = ALPHA ? gridDim.x : (gridDim.x < kTicketCounters ? gridDim.x : kTicketCounters);
	const uint32_t cid = blockIdx.x % n_ctr, nb_c = (gridDim.x - cid + n_ctr - 1u) / n_ctr;
	uint32_t *ctr = a.work + 2u + kTicketCounters * a.work_slot + cid;
	const uint32_t n_items = ALPHA ? a.work[kWorkA + a.work_slot] : a.n_tiles;
	auto tile_of = [&](uint32_t k) -> uint32_t {
		const unsigned long long t = (unsigned long long)k * n_ctr + cid;
		if (t >= (unsigned long long)n_items) return 0xffffffffu;
		if constexpr (ALPHA) return a.work[kWorkList + a.n_tiles + (uint32_t)t];
		return (uint32_t)t;
	};
	uint32_t tile_g = tile_of(blockIdx.x / n_ctr), tile_next = tile_of(blockIdx.x / n_ctr + nb_c);
	prefetch(tile_g);
	// detector-only launches: equal cost per tile, and an iteration is shorter than an atomic's round trip:
	// there the "tickets" are simply this block's turn in a fixed rotation
	const bool dynamic = !ALPHA && (FULL || a.out_px != nullptr);
	uint32_t turn = blockIdx.x / n_ctr;
	uint32_t n_listb = 0, n_lista = 0;  // pending list-B entries in s_red[16..31], list-A entries in s_red[32..47]
	for (; tile_g < a.n_tiles;) {
		uint32_t drawn = turn;
		turn += nb_c;
		if (dynamic && threadIdx.x == 0) drawn = atomicAdd(ctr, 1u);
		// (call once per iteration, before a block barrier; the value is read after that barrier)
		auto publish_ticket = [&]() {
			if (threadIdx.x == 0) s_red[13] = drawn;
		};
		auto advance = [&]() {  // after the barrier that followed publish_ticket()
			tile_g = tile_next;
			tile_next = tile_of(2u * nb_c + s_red[13]);
		};
		auto defer = [&]() {
			// (block-uniform; only wave 0's lanes 0..15 ever touch the batch)
			list_push(s_red + 16, n_listb, tile_g, a.work + kWorkList, a.work + a.work_slot, threadIdx.x);
			if (threadIdx.x == 0) {
				bool keep = false;  // MODE 0: a value the block-cooperative detector left is final and stays
				if constexpr (MODE == 0) {
					const uint32_t t = tile_g - fastdiv(tile_g, a.div_tpf) * a.tiles_per_frame;
					const uint32_t ty = fastdiv(t, a.div_cols), tx = t - ty * a.cols;
					keep = tx < a.full_cols && ty < a.ok_rows;
				}
				if (!keep) reinterpret_cast<uint2 *>(a.sums)[tile_g] = make_uint2(kDeferredKey, kDeferredKey);
			}
		};
		if (!pre_valid) {  // ragged edge / unaligned batch (block-uniform)
			defer();
			prefetch(tile_next);
			publish_ticket();
			__syncthreads();
			advance();
			continue;
		}
		// ---- stage: registers -> planar u16 pairs
		uint32_t alpha_and = 0xffu;
#pragma unroll
		for (int k = 0; k < 4; ++k) alpha_and &= (pre[k].x & pre[k].y & pre[k].z & pre[k].w) >> 24;
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const uint32_t row = 16u * wave + (lane >> 4) + 4u * (uint32_t)k, col = lane & 15u;
			const uint4 v = pre[k];
			uint32_t *d = s_pl + row * kRS64 + col * 2u;
#pragma unroll
			for (uint32_t c = 0; c < NCH; ++c) {
				const uint32_t sel = c | 0x0c000c00u | ((4u + c) << 16);
				uint2 pr;
				pr.x = __builtin_amdgcn_perm(v.y, v.x, sel);
				pr.y = __builtin_amdgcn_perm(v.w, v.z, sel);
				*reinterpret_cast<uint2 *>(d + c * kPD64) = pr;
			}
		}
		const bool wave_transparent = __builtin_amdgcn_ballot_w64(alpha_and != 0xffu) != 0ull;
		if (lane == 0) s_red[8 + wave] = wave_transparent ? 1u : 0u;
		prefetch(tile_next);  // lands while this tile is processed
		__syncthreads();      // B1: the whole tile is staged
		if (!ALPHA && (FULL || a.out_px != nullptr) && (s_red[8] | s_red[9] | s_red[10] | s_red[11]) != 0u) {
			// transparency: the premultiplied convolution needs the alpha plane -- list A (the ALPHA instance of this
			// kernel, or the generic kernel when that one is not launched).  Detector-only launches do not care.
			list_push(s_red + 32, n_lista, tile_g, a.work + kWorkList + a.n_tiles, a.work + kWorkA + a.work_slot, threadIdx.x);
			if constexpr (MODE == 1)
				if (threadIdx.x == 0) reinterpret_cast<uint2 *>(a.sums)[tile_g] = make_uint2(kDeferredKey, kDeferredKey);
			publish_ticket();
			__syncthreads();
			advance();
			continue;
		}
		// ---- detector: window rows 16w + 8gg .. +7, column pair q (windows 2q, 2q+1)
		uint32_t sum_hz = 0, sum_vr = 0;
		uint32_t given_bits = 0;
		if constexpr (MODE == 0) given_bits = a.sums[2 * tile_g];  // MODE 0: the value is there already (oklab_kernel)
		if constexpr (MODE == 1) {
			const uint32_t q = lane & 31u, gg = lane >> 5;
			const uint32_t two = 0x00020002u;
			const uint32_t *pc[3];
			pc[0] = s_pl + (16u * wave + 8u * gg) * kRS64 + q;
			pc[1] = pc[0] + kPD64;
			pc[2] = pc[1] + kPD64;
			const bool short_group = wave == 3u && gg == 1u;  // window rows 56 .. 61 only
			uint32_t rA[3], rB[3], tP[3], dP[3];
#pragma unroll
			for (int c = 0; c < 3; ++c) {
				const uint32_t a0 = pc[c][0], a1 = pc[c][1], b0 = pc[c][kRS64], b1 = pc[c][kRS64 + 1];
				rA[c] = pk_mad_u16(__builtin_amdgcn_alignbit(a1, a0, 16), two, u32(us2(a0) + us2(a1)));
				rB[c] = pk_mad_u16(__builtin_amdgcn_alignbit(b1, b0, 16), two, u32(us2(b0) + us2(b1)));
				tP[c] = u32(us2(a0) + us2(b0));
				dP[c] = b0;
			}
#pragma unroll
			for (int st = 0; st < 4; ++st) {
				if (st < 3 || !short_group) {
#pragma unroll
					for (int c = 0; c < 3; ++c) {
						const uint32_t *pr = pc[c] + (2 + 2 * st) * (int)kRS64;
						const uint32_t n0 = pr[0], n1 = pr[1], o0 = pr[kRS64], o1 = pr[kRS64 + 1];
						const uint32_t rN = pk_mad_u16(__builtin_amdgcn_alignbit(n1, n0, 16), two, u32(us2(n0) + us2(n1)));
						sum_hz = sad16(rN, rA[c], sum_hz);
						const uint32_t tN = u32(us2(dP[c]) + us2(n0));
						const uint32_t c0 = u32(us2(tP[c]) + us2(tN));
						sum_vr = sad16(dpp_mov<0x130>(c0), c0, sum_vr);  // wave_shl:1 = the pair to the right
						const uint32_t rO = pk_mad_u16(__builtin_amdgcn_alignbit(o1, o0, 16), two, u32(us2(o0) + us2(o1)));
						sum_hz = sad16(rO, rB[c], sum_hz);
						const uint32_t tO = u32(us2(n0) + us2(o0));
						const uint32_t e0 = u32(us2(tN) + us2(tO));
						sum_vr = sad16(dpp_mov<0x130>(e0), e0, sum_vr);
						rA[c] = rN;
						rB[c] = rO;
						tP[c] = tO;
						dP[c] = o0;
					}
				}
			}
			if (q == 31u) sum_hz = sum_vr = 0;  // pair 31 starts no window (x = 62, 63)
		}
		sum_hz = wave_sum_sgpr(sum_hz);
		sum_vr = wave_sum_sgpr(sum_vr);
		if (lane == 0) {
			s_red[2 * wave] = sum_hz;
			s_red[2 * wave + 1] = sum_vr;
		}
		publish_ticket();
		__syncthreads();  // B2: partial sums are in; every wave is done with its neighbours' rows
		const uint32_t this_tile = tile_g;
		advance();
		sum_hz = s_red[0] + s_red[2] + s_red[4] + s_red[6];
		sum_vr = s_red[1] + s_red[3] + s_red[5] + s_red[7];
		sum_hz = __builtin_amdgcn_readfirstlane(sum_hz);
		sum_vr = __builtin_amdgcn_readfirstlane(sum_vr);
		uint32_t m0, m1;
		if constexpr (MODE == 1) {
			m0 = level_of(sum_hz);
			m1 = level_of(sum_vr);
		} else {
			const uint32_t vb = __builtin_amdgcn_readfirstlane(given_bits);
			sum_hz = sum_vr = vb;  // stays what it was
			m0 = m1 = level_of(__float_as_uint(parse_value(__uint_as_float(vb))));
		}
		const uint32_t nw = reduced_size(64u, m0), nh = reduced_size(64u, m1);
		if (threadIdx.x == 0) {
			reinterpret_cast<uint2 *>(a.sums)[this_tile] = make_uint2(sum_hz, sum_vr);
			if (FULL || a.out_w) a.out_w[this_tile] = nw;
			if (FULL || a.out_h) a.out_h[this_tile] = nh;
		}
		if (!FULL && a.out_px == nullptr) {
			__syncthreads();  // s_red is rewritten by the next tile
			continue;
		}
		uint8_t *dst = a.out_px + (size_t)this_tile * (64u * 64u * 4u);
		if (nw == 64u && nh == 64u) {
			// clone (block.rs:279-281): re-interleave this wave's 16 rows, 16 bytes per lane and step
#pragma unroll
			for (int k = 0; k < 4; ++k) {
				const uint32_t i = lane + 64u * (uint32_t)k;  // 256 groups of 4 pixels
				const uint32_t row = 16u * wave + (i >> 4), c4 = i & 15u;
				const uint32_t *p = s_pl + row * kRS64 + c4 * 2u;
				const uint2 r = *reinterpret_cast<const uint2 *>(p), g = *reinterpret_cast<const uint2 *>(p + kPD64);
				const uint2 b = *reinterpret_cast<const uint2 *>(p + 2 * kPD64);
				uint2 al = make_uint2(0x00ff00ffu, 0x00ff00ffu);
				if constexpr (ALPHA) al = *reinterpret_cast<const uint2 *>(p + 3 * kPD64);
				const uint32_t rg01 = __builtin_amdgcn_perm(g.x, r.x, 0x06020400u), ba01 = __builtin_amdgcn_perm(al.x, b.x, 0x06020400u);
				const uint32_t rg23 = __builtin_amdgcn_perm(g.y, r.y, 0x06020400u), ba23 = __builtin_amdgcn_perm(al.y, b.y, 0x06020400u);
				uint4 o;
				o.x = __builtin_amdgcn_perm(ba01, rg01, 0x05040100u);
				o.y = __builtin_amdgcn_perm(ba01, rg01, 0x07060302u);
				o.z = __builtin_amdgcn_perm(ba23, rg23, 0x05040100u);
				o.w = __builtin_amdgcn_perm(ba23, rg23, 0x07060302u);
				reinterpret_cast<uint4 *>(dst)[row * 16u + c4] = o;
			}
			__syncthreads();
			continue;
		}
		if constexpr (ALPHA) {
			// fir, U8x4: premultiply this wave's 16 rows in place (every wave is past B2: nobody reads them for the
			// detector any more, and both passes' first reads are of the wave's own rows).  mul_div_255 on both
			// pixels of a dword: t = v*a + 128 <= 65153, t + (t >> 8) <= 65407 -- nothing leaves its 16-bit half
#pragma unroll
			for (uint32_t it = 0; it < 8; ++it) {
				const uint32_t i = lane + 64u * it;
				uint32_t *p = s_pl + (16u * wave + (i >> 5)) * kRS64 + (i & 31u);
				const ushort2v al = us2(p[3 * kPD64]);
#pragma unroll
				for (uint32_t c = 0; c < 3; ++c) {
					ushort2v t = us2(p[c * kPD64]) * al + (ushort2v)(128);
					t = t + (t >> (ushort2v)(8));
					p[c * kPD64] = u32(t >> (ushort2v)(8));
				}
			}
			tile_sync<1>();
		}
		// ---- resample on the matrix cores: two passes, or one when an axis keeps its 64 samples
		const uint32_t lx = m0 < (uint32_t)kMaxLevel ? m0 : (uint32_t)kMaxLevel - 1;
		const uint32_t ly = m1 < (uint32_t)kMaxLevel ? m1 : (uint32_t)kMaxLevel - 1;
		const bool need_h = nw != 64u, need_v = nh != 64u;
		const uint32_t *mx = a.mf64 + a.mf_off[need_h ? lx : ly], *my = a.mf64 + a.mf_off[need_v ? ly : lx];
		const uint32_t nbx = nw > 16u ? 2u : 1u, nby = nh > 16u ? 2u : 1u;  // (of the axes that are resampled)
		const uint32_t *mx_tail = mx + nbx * 512u, *my_tail = my + nby * 512u;  // bias[32], ksum[32], flag
		const uint32_t px_ = a.precision[need_h ? lx : ly], py = a.precision[need_v ? ly : lx];
		const int32_t top_x = (int32_t)((256u << px_) - 1u), top_y = (int32_t)((256u << py) - 1u);
		const uint32_t o = lane & 15u, g = lane >> 4;
		const v4i32 zero = {0, 0, 0, 0};
		// horizontal pass of this wave's 16 rows into s_t[c][ox][y]; A = pixels of row 16w + o, columns 16g .. 16g+15
		auto hpass = [&]() {
			v4i32 wlo[2], whi[2];
			int32_t bx[2];
#pragma unroll
			for (uint32_t nb = 0; nb < 2; ++nb) {
				if (nb < nbx) {
					wlo[nb] = *reinterpret_cast<const v4i32 *>(mx + nb * 512u + lane * 4u);
					whi[nb] = *reinterpret_cast<const v4i32 *>(mx + nb * 512u + 256u + lane * 4u);
					bx[nb] = (int32_t)mx_tail[16u * nb + o];
				}
			}
			const uint32_t *rowp = s_pl + (16u * wave + o) * kRS64 + 8u * g;
#pragma unroll
			for (uint32_t c = 0; c < NCH; ++c) {
				const uint4 d0 = *reinterpret_cast<const uint4 *>(rowp + c * kPD64);
				const uint4 d1 = *reinterpret_cast<const uint4 *>(rowp + c * kPD64 + 4);
				v4i32 av;
				av[0] = (int)(__builtin_amdgcn_perm(d0.y, d0.x, 0x06040200u) ^ 0x80808080u);
				av[1] = (int)(__builtin_amdgcn_perm(d0.w, d0.z, 0x06040200u) ^ 0x80808080u);
				av[2] = (int)(__builtin_amdgcn_perm(d1.y, d1.x, 0x06040200u) ^ 0x80808080u);
				av[3] = (int)(__builtin_amdgcn_perm(d1.w, d1.z, 0x06040200u) ^ 0x80808080u);
#pragma unroll
				for (uint32_t nb = 0; nb < 2; ++nb) {
					if (nb < nbx) {
						const v4i32 cx = {bx[nb], bx[nb], bx[nb], bx[nb]};
						const v4i32 lo = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, wlo[nb], cx, 0, 0, 0);
						const v4i32 hi = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, whi[nb], zero, 0, 0, 0);
						uint32_t packed = 0;
						put_byte_shr<0>(packed, clamp_fixed(hi[0], lo[0], top_x), px_);
						put_byte_shr<1>(packed, clamp_fixed(hi[1], lo[1], top_x), px_);
						put_byte_shr<2>(packed, clamp_fixed(hi[2], lo[2], top_x), px_);
						put_byte_shr<3>(packed, clamp_fixed(hi[3], lo[3], top_x), px_);
						// rows 16w + 4g .. +3 of column ox = 16nb + o
						s_t[(c * 32u + 16u * nb + o) * kTS64 + 4u * wave + g] = packed;
					}
				}
			}
		};
		// vertical pass over the 32 columns held in s_t: wave = (output row block mb, column block nb);
		// ox0 = first output column of s_t, row_w = output row length, nbc = column blocks present
		auto vpass = [&](const uint32_t ox0, const uint32_t row_w, const uint32_t nbc, const bool through_h) {
			const uint32_t mb = wave >> 1, nb = wave & 1u;
			if (mb < nby && nb < nbc) {
				const v4i32 klo = *reinterpret_cast<const v4i32 *>(my + mb * 512u + lane * 4u);
				const v4i32 khi = *reinterpret_cast<const v4i32 *>(my + mb * 512u + 256u + lane * 4u);
				const v4i32 cy = *reinterpret_cast<const v4i32 *>(my_tail + 16u * mb + 4u * g);
				uint32_t pix[4] = {0xff000000u, 0xff000000u, 0xff000000u, 0xff000000u};
				if constexpr (ALPHA) pix[0] = pix[1] = pix[2] = pix[3] = 0u;
#pragma unroll
				for (uint32_t c = 0; c < NCH; ++c) {
					const uint4 tv = *reinterpret_cast<const uint4 *>(s_t + (c * 32u + 16u * nb + o) * kTS64 + 4u * g);
					v4i32 bv;
					bv[0] = (int)(tv.x ^ 0x80808080u);
					bv[1] = (int)(tv.y ^ 0x80808080u);
					bv[2] = (int)(tv.z ^ 0x80808080u);
					bv[3] = (int)(tv.w ^ 0x80808080u);
					const v4i32 lo = __builtin_amdgcn_mfma_i32_16x16x64_i8(klo, bv, cy, 0, 0, 0);
					const v4i32 hi = __builtin_amdgcn_mfma_i32_16x16x64_i8(khi, bv, zero, 0, 0, 0);
#pragma unroll
					for (int r = 0; r < 4; ++r) {
						const uint32_t v = clamp_fixed(hi[r], lo[r], top_y);
						if (c == 0) put_byte_shr<0>(pix[r], v, py);
						else if (c == 1) put_byte_shr<1>(pix[r], v, py);
						else if (c == 2) put_byte_shr<2>(pix[r], v, py);
						else put_byte_shr<3>(pix[r], v, py);
					}
				}
				const uint32_t oxl = 16u * nb + o;
				if constexpr (ALPHA) {
#pragma unroll
					for (int r = 0; r < 4; ++r) pix[r] = unpremultiply(pix[r]);
				} else if (!((through_h ? mx_tail[64] : 1u) & my_tail[64])) {
					// opaque tile: alpha is the convolution of the constant 255 = the windows' weight sums
					const int32_t ah = through_h ? (int32_t)clip8((1 << (px_ - 1)) + 255 * (int32_t)mx_tail[32u + oxl], (int)px_) : 255;
#pragma unroll
					for (int r = 0; r < 4; ++r) {
						const uint32_t al = clip8((1 << (py - 1)) + ah * (int32_t)my_tail[32u + 16u * mb + 4u * g + (uint32_t)r], (int)py);
						pix[r] = (pix[r] & 0x00ffffffu) | (al << 24);
						if (al != 255u) pix[r] = unpremultiply(pix[r]);
					}
				}
				if (ox0 + oxl < row_w) {
#pragma unroll
					for (uint32_t r = 0; r < 4; ++r) {
						const uint32_t oy = 16u * mb + 4u * g + r;
						if (oy < nh) reinterpret_cast<uint32_t *>(dst)[oy * row_w + ox0 + oxl] = pix[r];
					}
				}
			}
		};
		if (need_h && need_v) {
			hpass();
			__syncthreads();  // B3: all 64 rows of the horizontal pass are in LDS
			vpass(0u, nw, nbx, true);
			// no barrier here: the next tile's B1/B2 separate this vertical pass from the next horizontal one
		} else if (need_v) {
			// width kept: the pixels themselves, 32 columns at a time, as bytes [x][y] in s_t
			for (uint32_t half = 0; half < 2; ++half) {
				const uint32_t xl = lane & 31u, jj = lane >> 5;
#pragma unroll
				for (uint32_t c = 0; c < NCH; ++c) {
					const uint16_t *p16 = reinterpret_cast<const uint16_t *>(s_pl + c * kPD64) + 32u * half + xl;
#pragma unroll
					for (uint32_t it = 0; it < 2; ++it) {
						const uint32_t j = jj + 2u * it, y0 = 16u * wave + 4u * j;
						const uint32_t b0 = p16[(y0 + 0u) * (2u * kRS64)], b1 = p16[(y0 + 1u) * (2u * kRS64)];
						const uint32_t b2 = p16[(y0 + 2u) * (2u * kRS64)], b3 = p16[(y0 + 3u) * (2u * kRS64)];
						s_t[(c * 32u + xl) * kTS64 + 4u * wave + j] = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
					}
				}
				__syncthreads();
				vpass(32u * half, 64u, 2u, false);
				__syncthreads();  // s_t is refilled (next half, or the next tile's horizontal pass after only B1/B2)
			}
		} else {
			// height kept: the horizontal pass is the result; gather [c][ox][y] bytes into pixels
			hpass();
			__syncthreads();
			const bool opaque_stays = (mx_tail[64] & 1u) != 0u;
			for (uint32_t i = threadIdx.x; i < nw * 16u; i += 256u) {
				const uint32_t ox = i % nw, yq = i / nw;  // nw is a power of two here
				const uint32_t r4 = s_t[(0u * 32u + ox) * kTS64 + yq], g4 = s_t[(1u * 32u + ox) * kTS64 + yq], b4 = s_t[(2u * 32u + ox) * kTS64 + yq];
				uint32_t al = 255u, a4 = 0u;
				if constexpr (ALPHA) a4 = s_t[(3u * 32u + ox) * kTS64 + yq];
				else if (!opaque_stays) al = clip8((1 << (px_ - 1)) + 255 * (int32_t)mx_tail[32u + ox], (int)px_);
#pragma unroll
				for (uint32_t r = 0; r < 4; ++r) {
					if constexpr (ALPHA) al = (a4 >> (8u * r)) & 255u;
					uint32_t px = ((r4 >> (8u * r)) & 255u) | (((g4 >> (8u * r)) & 255u) << 8) | (((b4 >> (8u * r)) & 255u) << 16) | (al << 24);
					if (al != 255u) px = unpremultiply(px);
					reinterpret_cast<uint32_t *>(dst)[(4u * yq + r) * nw + ox] = px;
				}
			}
			__syncthreads();
		}
	}
	list_flush(s_red + 16, n_listb, a.work + kWorkList, a.work + a.work_slot, threadIdx.x);
	list_flush(s_red + 32, n_lista, a.work + kWorkList + a.n_tiles, a.work + kWorkA + a.work_slot, threadIdx.x);
}

// ---------------------------------------------------------------------------
// generic kernel: persistent over tiles (or over the worklist left by shrink32_kernel).  NW == 1:
// every wave of the block owns one LDS tile image and walks tiles wave_id, wave_id + total_waves, ...;
// NW > 1: one tile per block iteration.
// ---------------------------------------------------------------------------
template <int NW, int C, int MODE>
__global__ void __launch_bounds__(NW == 1 ? 768 : 64 * NW) shrink_kernel(const ShrinkArgs a)
{
	extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
	if constexpr (NW == 1) {
		const uint32_t wpb = blockDim.x / 64u, sub = threadIdx.x / 64u, tid = threadIdx.x % 64u;
		uint32_t *s_pl = lds + sub * a.tile_dw;
		// tiles are dealt to the waves of a block on demand (LDS ticket counter, as in shrink32_kernel):
		// block b owns items b, b + blocks, ...; ticket t is item b + t*blocks
		uint32_t *s_ticket = lds + wpb * a.tile_dw;
		if (threadIdx.x == 0) *s_ticket = wpb;
		__syncthreads();
		// with a worklist (left by shrink32_kernel) only the listed tiles are processed
		const uint32_t count_b = a.work ? __builtin_amdgcn_readfirstlane(a.work[a.work_slot]) : a.n_tiles;
		const uint32_t count_a = a.work ? __builtin_amdgcn_readfirstlane(a.work[kWorkA + a.work_slot]) : 0u;
		const uint32_t count = count_b + (a.list_a_too ? count_a : 0u);  // (each list holds a tile at most once: <= n_tiles)
		uint32_t ticket = sub;
		for (;;) {
			const unsigned long long i = (unsigned long long)blockIdx.x + (unsigned long long)ticket * gridDim.x;
			if (i >= (unsigned long long)count) break;
			uint32_t tile_g = (uint32_t)i;
			if (a.work) {
				const uint32_t at = (uint32_t)i < count_b ? (uint32_t)i : a.n_tiles + ((uint32_t)i - count_b);  // list B, then list A
				tile_g = __builtin_amdgcn_readfirstlane(a.work[kWorkList + at]);
			}
			process_tile<NW, C, MODE>(a, tile_g, s_pl, nullptr, tid);
			uint32_t t = 0;
			if (tid == 0) t = atomicAdd(s_ticket, 1u);
			ticket = __builtin_amdgcn_readfirstlane(t);
			tile_sync<1>();  // the next tile reuses this wave's LDS image
		}
		if (a.work) {
			// worklist mode = second and last launch of the 32x32 flow: finish, lane-parallel, the tiles
			// shrink32_kernel completed (it left kDeferredKey in the others, which process_tile finishes
			// itself), and zero the worklist counter of the NEXT launch
			for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < a.n_tiles; t += gridDim.x * blockDim.x) {
				const uint2 key = reinterpret_cast<const uint2 *>(a.sums)[t];
				if (key.x == kDeferredKey && key.y == kDeferredKey) continue;
				const uint32_t tf = t % a.tiles_per_frame;
				const uint32_t ty = tf / a.cols, tx = tf - ty * a.cols;
				finish_tile(key, (tx == a.cols - 1) ? a.edge_w : a.bw, (ty == a.rows - 1) ? a.edge_h : a.bh, (uint32_t)MODE, a.factor,
				            a.value, a.lod0, a.lod1, t);
			}
			if (blockIdx.x == 0) {  // the other set of counters is the next launch's
				if (threadIdx.x == 0) {
					a.work[a.work_slot ^ 1u] = 0u;
					a.work[kWorkA + (a.work_slot ^ 1u)] = 0u;
					if (a.stats) *a.stats = count_a;  // steers the next launches' kernel choice (pxz_api.cpp)
				}
				if (threadIdx.x < kTicketCounters) a.work[2u + kTicketCounters * (a.work_slot ^ 1u) + threadIdx.x] = 0u;
			}
		}
	} else {
		// with a worklist (left by shrink64_kernel) only the listed tiles are processed
		const uint32_t count_b = a.work ? __builtin_amdgcn_readfirstlane(a.work[a.work_slot]) : a.n_tiles;
		const uint32_t count_a = a.work ? __builtin_amdgcn_readfirstlane(a.work[kWorkA + a.work_slot]) : 0u;
		const uint32_t count = count_b + (a.list_a_too ? count_a : 0u);
		for (uint32_t i = blockIdx.x; i < count; i += gridDim.x) {
			uint32_t tile_g = i;
			if (a.work) tile_g = __builtin_amdgcn_readfirstlane(a.work[kWorkList + (i < count_b ? i : a.n_tiles + (i - count_b))]);  // list B, then list A
			process_tile<NW, C, MODE>(a, tile_g, lds, lds + a.tile_dw, threadIdx.x);
			__syncthreads();
		}
		if (a.work) {
			// as in the single-wave form: finish the tiles the fast kernel completed, zero the next launch's counter
			for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < a.n_tiles; t += gridDim.x * blockDim.x) {
				const uint2 key = reinterpret_cast<const uint2 *>(a.sums)[t];
				if (key.x == kDeferredKey && key.y == kDeferredKey) continue;
				const uint32_t tf = t % a.tiles_per_frame;
				const uint32_t ty = tf / a.cols, tx = tf - ty * a.cols;
				finish_tile(key, (tx == a.cols - 1) ? a.edge_w : a.bw, (ty == a.rows - 1) ? a.edge_h : a.bh, (uint32_t)MODE, a.factor,
				            a.value, a.lod0, a.lod1, t);
			}
			if (blockIdx.x == 0) {  // the other set of counters is the next launch's
				if (threadIdx.x == 0) {
					a.work[a.work_slot ^ 1u] = 0u;
					a.work[kWorkA + (a.work_slot ^ 1u)] = 0u;
					if (a.stats) *a.stats = count_a;
				}
				if (threadIdx.x < kTicketCounters) a.work[2u + kTicketCounters * (a.work_slot ^ 1u) + threadIdx.x] = 0u;
			}
		}
	}
}

// ---------------------------------------------------------------------------
// Oklab-MAD detector, block-cooperative (full 32x32 RGBA tiles): get_block_variance
// (reference src/operations.rs:26-126) with shrink_by's closures (pixlzr.rs:160-162).
//
// The two f32 accumulations of the reference are sequential over the tile's pixels and
// end up in the bitstream, so they are replayed in exactly that order.  To keep the chip
// busy anyway, a block of 16 waves works on 15 tiles at once: waves 0..14 ("producers")
// each convert one tile to Oklab (f32 + the glibc-cbrtf double-precision steps) and keep
// the 16 pixels x 3 values of every lane in registers; wave 15 ("chain") walks 15 x 4 chains
// (channels a, b, l, alpha of every tile) in lock-step, 60 lanes wide, reading the values
// from LDS bands of 8 tile rows.
//
// Software pipeline over the batches b_0, b_1, ... of a block, one "period" per batch, four
// "intervals" (bands) per period, two barriers per interval:
//   convert phase   producers convert band k of batch p into spare registers and request the
//                   same band of batch p+1 from HBM; meanwhile the chain adds up the two bands
//                   written one interval earlier: pass 1 (sum) of batch p and pass 2 (sum of
//                   |x - mean|) of batch p-1, whose means it published when its pass 1 ended
//   -- barrier A --
//   write phase     producers write band k of pass 2 (old registers minus the means), move the
//                   spare registers in, write band k of pass 1
//   -- barrier B --
// so the conversion (the expensive part) and the dependent-add chains run side by side, and
// one band buffer per pass is enough.
// ---------------------------------------------------------------------------
constexpr uint32_t kOkTiles = 15;           // tiles per block and batch (one per producer wave)
constexpr uint32_t kOkPlane = 256 + 4;      // floats per (tile, channel) band: 8 rows x 32 px + bank skew
constexpr uint32_t kOkBand = kOkTiles * 4 * kOkPlane;  // floats per band buffer

// The quotient of the Halley step inside cbrt_f32_lut: v_rcp_f64 (2^-23 or better) and ONE Newton step, i.e. a
// quotient good to ~2^-46 instead of the correctly rounded one glibc's `/` produces.  That is enough here, and
// provably so: the result is rounded to f32 right after, the inputs of this path are the l, m, s of the 2^24
// possible RGB triples and nothing else, and tests/test_gpu_parity.py::test_oklab_conversion_of_every_colour runs
// every one of them through this very function against the oracle's exact division: no bit differs.  (The
// generic kernel's cbrt_f32, which sees the same inputs, keeps the exact division.)  4 instructions instead of 8.
__device__ __forceinline__ double div_f64_oklab_domain(double n, double d)
{
	double r = __builtin_amdgcn_rcp(d);
	const double e = __builtin_fma(-d, r, 1.0);
	r = __builtin_fma(r, e, r);
	return n * r;
}

// glibc 2.35 cbrtf for the inputs of the Oklab detector: same arithmetic as cbrt_f32 above with the quotient below.
// The tail `(float)(q * third[2 + xe % 3])` followed by `ldexpf(.., xe / 3)` is folded into ONE
// multiplication by 2^(xe/3) * third[..]: scaling a double by a power of two is exact and commutes
// with the rounding to float (no underflow in this range), so the bits are unchanged.  `scale` is the
// LDS table of those 132 doubles indexed by xe + 130 (xe in [-130, 1]).
template <bool ZERO_CHECK = true>
__device__ __forceinline__ float cbrt_f32_lut(float x, const double *scale)
{
	int xe;
	const float xm = frexpf(x, &xe);
	// glibc: (float)(0.4926.. + (0.6975.. - 0.1915.. * xm) * xm) with separate double operations.  The fused form
	// differs from it by at most a few 2^-53 before the rounding to float, and for none of this path's inputs
	// does that cross a rounding boundary (same exhaustive test as for the quotient below).
	const float u = (float)__builtin_fma(__builtin_fma(-0.191502161678719066, (double)xm, 0.697570460207922770), (double)xm, 0.492659620528969547);
	const float t2 = u * u * u;
	// t2 + 2 xm and 2 t2 + xm are exact in double (24-bit operands a few binades apart), so the fused forms give
	// the same values as glibc's separate multiplications and additions
	const double num = (double)u * __builtin_fma(2.0, (double)xm, (double)t2);
	const double den = __builtin_fma(2.0, (double)t2, (double)xm);
	const float y = (float)(div_f64_oklab_domain(num, den) * scale[xe + 130]);
	if constexpr (!ZERO_CHECK) return y;  // (x == 0 is the caller's business)
	return x == 0.0f ? 0.0f : y;
}

// byte BYTE of v, times 4: the byte offset of a 256-entry f32 table row, in one SDWA shift
template <int BYTE>
__device__ __forceinline__ uint32_t byte_times4(uint32_t v)
{
	uint32_t r;
	const uint32_t two = 2u;
	if constexpr (BYTE == 0)
		asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(two), "v"(v));
	else if constexpr (BYTE == 1)
		asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(two), "v"(v));
	else
		asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(two), "v"(v));
	return r;
}
__device__ __forceinline__ float table_at(const float *table, uint32_t byte_offset)
{
	return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(table) + byte_offset);
}

// Srgba<u8> -> linear -> Oklab of two pixels (operations.rs:56-59; palette 0.7.6): LUT, then Ottosson's matrices
// with left-to-right f32 sums; two pixels per packed-f32 instruction (same IEEE results per component).
// out[k] = {a, b, l} of pixel k, the order the reference sums them in.
__device__ __forceinline__ void oklab_pair(uint32_t v0, uint32_t v1, const float *s_srgb, const double *s_scale,
                                           float (&out0)[3], float (&out1)[3])
{
	const f32x2 r = {table_at(s_srgb, byte_times4<0>(v0)), table_at(s_srgb, byte_times4<0>(v1))};
	const f32x2 g = {table_at(s_srgb, byte_times4<1>(v0)), table_at(s_srgb, byte_times4<1>(v1))};
	const f32x2 b = {table_at(s_srgb, byte_times4<2>(v0)), table_at(s_srgb, byte_times4<2>(v1))};
	const f32x2 l = 0.4122214708f * r + 0.5363325363f * g + 0.0514459929f * b;
	const f32x2 m = 0.2119034982f * r + 0.6806995451f * g + 0.1073969566f * b;
	const f32x2 s3 = 0.0883024619f * r + 0.2817188376f * g + 0.6299787005f * b;
	// l, m, s are zero only for black (every coefficient is positive, the table is zero at 0 only), and then all
	// three are: one test per pixel on the colour bytes instead of one per cube root
	const f32x2 l_ = {cbrt_f32_lut<false>(l.x, s_scale), cbrt_f32_lut<false>(l.y, s_scale)};
	const f32x2 m_ = {cbrt_f32_lut<false>(m.x, s_scale), cbrt_f32_lut<false>(m.y, s_scale)};
	const f32x2 s_ = {cbrt_f32_lut<false>(s3.x, s_scale), cbrt_f32_lut<false>(s3.y, s_scale)};
	const f32x2 L = 0.2104542553f * l_ + 0.7936177850f * m_ - 0.0040720468f * s_;
	const f32x2 A = 1.9779984951f * l_ - 2.4285922050f * m_ + 0.4505937099f * s_;
	const f32x2 B = 0.0259040371f * l_ + 0.7827717662f * m_ - 0.8086757660f * s_;
	const bool black0 = (v0 & 0x00ffffffu) == 0u, black1 = (v1 & 0x00ffffffu) == 0u;  // cbrt(0) = 0 -> L = a = b = +0
	out0[2] = black0 ? 0.0f : L.x; out1[2] = black1 ? 0.0f : L.y;
	out0[0] = black0 ? 0.0f : A.x; out1[0] = black1 ? 0.0f : A.y;
	out0[1] = black0 ? 0.0f : B.x; out1[1] = black1 ? 0.0f : B.y;
}

// The conversion tables of the Oklab kernels in LDS: sRGB u8 -> linear (256), a / 255 (256), and the 132 doubles
// 2^(xe/3) * cbrt(2)^(xe%3), xe = i - 130.  Call from the first 256 threads of a block, then a block barrier.
__device__ __forceinline__ void oklab_fill_tables(float *s_srgb, float *s_alpha, double *s_scale, uint32_t t)
{
	if (t < 256) {
		s_srgb[t] = __uint_as_float(kSrgbToLinearBits[t]);
		s_alpha[t] = __fdiv_rn((float)t, 255.0f);
	}
	if (t < 132) {
		const int xe = (int)t - 130;
		const int q3 = xe / 3, r3 = xe - 3 * q3;  // C semantics: the remainder carries the sign of xe
		const double third = r3 == 0 ? 1.0
		                   : r3 == 1 ? 1.2599210498948731648
		                   : r3 == 2 ? 1.5874010519681994748
		                   : r3 == -1 ? 1.0 / 1.2599210498948731648
		                              : 1.0 / 1.5874010519681994748;
		s_scale[t] = ldexp(third, q3);  // exact
	}
}

// Per-pixel form of the same conversion (pxz_oklab_pixels_device): out[i] = {l, a, b, alpha} of RGBA pixel i.
__global__ void __launch_bounds__(256) oklab_pixels_kernel(const uint32_t *px, uint32_t n, float4 *out)
{
	__shared__ float s_srgb[256], s_alpha[256];
	__shared__ double s_scale[132];
	oklab_fill_tables(s_srgb, s_alpha, s_scale, threadIdx.x);
	__syncthreads();
	for (uint32_t i = 2u * (blockIdx.x * blockDim.x + threadIdx.x); i < n; i += 2u * gridDim.x * blockDim.x) {
		const uint32_t v0 = px[i], v1 = i + 1u < n ? px[i + 1u] : 0u;
		float o0[3], o1[3];
		oklab_pair(v0, v1, s_srgb, s_scale, o0, o1);
		out[i] = make_float4(o0[2], o0[0], o0[1], s_alpha[v0 >> 24]);
		if (i + 1u < n) out[i + 1u] = make_float4(o1[2], o1[0], o1[1], s_alpha[v1 >> 24]);
	}
}

// Tile geometry of the block-cooperative Oklab detector, T = 16 | 32 | 64 (square RGBA tiles).  A band is 256
// pixels of a tile in row-major order = 4 consecutive pixels per lane: 64 / (T/4) rows of T pixels.
template <int T>
struct OkGeom {
	static constexpr uint32_t kBands = T * T / 256;   // 1 | 4 | 16
	static constexpr uint32_t kLanesPerRow = T / 4;   // 4 | 8 | 16
	static constexpr uint32_t kRowsPerBand = 256 / T; // 16 | 8 | 4
	// up to 4 bands the converted tile stays in registers between the passes; a 64x64 tile (192 values per
	// lane) parks it in a scratch buffer in HBM instead (13 dwords per lane and band: 12 values + the alpha bytes)
	static constexpr bool kInRegs = kBands <= 4;
};

// Tiles the block-cooperative detector takes: full width, and a height of whole bands (every full tile; the
// ragged last row of the grid when its height happens to be one).  tile_h = 0 when it does not.
template <int T, class Args>
__device__ __forceinline__ uint32_t oklab_tile_src(const Args &a, uint32_t tile_g, const uint8_t *&src)
{
	if (tile_g >= a.n_tiles) return 0u;
	const uint32_t frame = fastdiv(tile_g, a.div_tpf);
	const uint32_t t = tile_g - frame * a.tiles_per_frame;
	const uint32_t ty = fastdiv(t, a.div_cols), tx = t - ty * a.cols;
	src = a.src + (size_t)frame * a.frame_stride + (size_t)(ty * (uint32_t)T) * a.pitch + (size_t)(tx * (uint32_t)T) * 4u;
	if (tx >= a.full_cols || ty >= a.ok_rows) return 0u;
	return ty == a.rows - 1u ? a.edge_h : (uint32_t)T;
}

template <int T>
__global__ void __launch_bounds__(1024) oklab_kernel(const ShrinkArgs a)
{
	using G = OkGeom<T>;
	constexpr uint32_t NB = G::kBands;
	extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
	float *s_srgb = reinterpret_cast<float *>(lds);          // 256: sRGB u8 -> linear
	float *s_alpha = s_srgb + 256;                           // 256: a / 255
	double *s_scale = reinterpret_cast<double *>(s_alpha + 256);  // 132: 2^(xe/3) * 2^((xe%3)/3), xe = i - 130
	float *s_mean = reinterpret_cast<float *>(s_scale + 132);     // 64: per (tile, channel) means of the batch in pass 2
	float *s_p1 = s_mean + 64;                               // pass-1 band: values
	float *s_p2 = s_p1 + kOkBand;                            // pass-2 band: values minus means
	const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
	oklab_fill_tables(s_srgb, s_alpha, s_scale, threadIdx.x);
	__syncthreads();

	const uint32_t n_batches = (a.n_tiles + kOkTiles - 1) / kOkTiles;
	// batches of this block: blockIdx.x + j * gridDim.x, j < own; periods 0 .. own + 1 drain the pipeline
	const uint32_t own = n_batches > blockIdx.x ? (n_batches - blockIdx.x + gridDim.x - 1) / gridDim.x : 0u;
	const uint32_t periods = own + 2u;

	if (wave < kOkTiles) {
		// ---------------- producers ----------------
		const uint32_t row_off = lane / G::kLanesPerRow, col_off = (lane % G::kLanesPerRow) * 16u;
		// source pointers (lane's first group of band 0) of this wave's tiles in batches p, p+1, p+2; null: nothing there
		auto batch_src = [&](uint32_t j, uint32_t &bands) -> const uint8_t * {
			const uint8_t *src;
			bands = 0;
			if (j >= own) return nullptr;
			const uint32_t th = oklab_tile_src<T>(a, (blockIdx.x + j * gridDim.x) * kOkTiles + wave, src);
			if (th == 0) return nullptr;
			bands = th / G::kRowsPerBand;  // (a ragged tile is only taken with a whole number of bands)
			return src + (size_t)row_off * a.pitch + col_off;
		};
		uint32_t nb0 = 0, nb1 = 0, nb2 = 0, nb_prev = 0;  // bands of this wave's tile in batches p, p+1, p+2, p-1
		const uint8_t *src0 = nullptr, *src1 = batch_src(0, nb1), *src2 = batch_src(1, nb2);
		const size_t band_step = (size_t)G::kRowsPerBand * a.pitch;
		// raw pixels: the band being converted and the one after it (requested one interval ahead)
		uint4 px_cur = make_uint4(0, 0, 0, 0), px_nxt = make_uint4(0, 0, 0, 0);
		if (src1) px_cur = *reinterpret_cast<const uint4 *>(src1);
		if (NB > 1) {
			if (src1 && nb1 > 1u) px_nxt = *reinterpret_cast<const uint4 *>(src1 + band_step);
		} else if (src2) {
			px_nxt = *reinterpret_cast<const uint4 *>(src2);
		}
		float lab[G::kInRegs ? NB : 1][4][3];   // [band][pixel][a, b, l] of the batch whose pass 2 is being staged
		uint32_t alpha_px[G::kInRegs ? NB : 1];  // its 4 alpha bytes per band
		bool have_prev = false, elig_cur = false;
		uint32_t tile_prev = 0, tile_cur = 0;
		for (uint32_t p = 0; p < periods; ++p) {
			tile_prev = tile_cur;
			tile_cur = (blockIdx.x + p * gridDim.x) * kOkTiles + wave;
			src0 = src1;  // batch p
			src1 = src2;  // batch p + 1
			nb_prev = nb0;
			nb0 = nb1;
			nb1 = nb2;
			src2 = batch_src(p + 2u, nb2);
			elig_cur = src0 != nullptr;  // (false past the last batch and for tiles the detector does not take)
			constexpr int kUnroll = G::kInRegs ? (int)NB : 1;  // register form: lab[k] must be a static index
#pragma unroll kUnroll
			for (uint32_t k = 0; k < NB; ++k) {
				// ---- convert phase
				float4 old[4];  // scratch form only: band k of the previous batch, back from HBM for pass 2
				if constexpr (!G::kInRegs) {
					if (have_prev && k < nb_prev) {
						// per (tile, band): three arrays of 64 float4 (the 12 values of a lane) + 64 alpha words = 3328 bytes
						const float *sb = a.ok_scratch + ((size_t)tile_prev * NB + k) * 832u;
						const float4 *sp = reinterpret_cast<const float4 *>(sb) + lane;
#pragma unroll
						for (int q = 0; q < 3; ++q) old[q] = sp[64 * q];
						old[3].x = sb[768u + lane];
					}
				}
				float fresh[4][3];
				uint32_t fresh_alpha = 0;
				if (elig_cur && k < nb0) {
					const uint32_t v[4] = {px_cur.x, px_cur.y, px_cur.z, px_cur.w};
					fresh_alpha = (v[0] >> 24) | ((v[1] >> 24) << 8) | ((v[2] >> 24) << 16) | ((v[3] >> 24) << 24);
#pragma unroll
					for (int j = 0; j < 4; j += 2) {
						oklab_pair(v[j], v[j + 1], s_srgb, s_scale, fresh[j], fresh[j + 1]);
						// two pixels (six cube-root chains) at a time: the register file also holds a whole tile of results
						__builtin_amdgcn_sched_barrier(0);
					}
				}
				// px_cur is consumed: move the window on by one step of the sequence
				px_cur = px_nxt;
				{
					// two steps ahead: band k+2 of this batch, or an early band of the next one / the one after
					const uint32_t ahead = k + 2u, bo = ahead / NB, band = ahead % NB;
					const uint8_t *base = bo == 0 ? src0 : (bo == 1 ? src1 : src2);
					const uint32_t nbb = bo == 0 ? nb0 : (bo == 1 ? nb1 : nb2);
					if (base && band < nbb) px_nxt = *reinterpret_cast<const uint4 *>(base + band * band_step);
				}
				__syncthreads();  // A: the chain has consumed the bands of the previous interval
				// ---- write phase
				const uint32_t slot = (wave * 4u) * kOkPlane + lane * 4u;  // 4 consecutive pixels of this lane
				if (have_prev && k < nb_prev) {
					// operations.rs:75-84: the chain only has to add |x| of these.  The means were published during
					// the first convert phase of this period.
					const float4 mean = *reinterpret_cast<const float4 *>(s_mean + wave * 4u);
					const float mean4[4] = {mean.x, mean.y, mean.z, mean.w};
					float x[4][3];
					uint32_t al4;
					if constexpr (G::kInRegs) {
#pragma unroll
						for (int j = 0; j < 4; ++j)
#pragma unroll
							for (int c = 0; c < 3; ++c) x[j][c] = lab[k][j][c];
						al4 = alpha_px[k];
					} else {
						const float o12[12] = {old[0].x, old[0].y, old[0].z, old[0].w, old[1].x, old[1].y,
						                       old[1].z, old[1].w, old[2].x, old[2].y, old[2].z, old[2].w};
#pragma unroll
						for (int j = 0; j < 4; ++j)
#pragma unroll
							for (int c = 0; c < 3; ++c) x[j][c] = o12[3 * j + c];
						al4 = __float_as_uint(old[3].x);
					}
					float *d = s_p2 + slot;
#pragma unroll
					for (int c = 0; c < 3; ++c)
						*reinterpret_cast<float4 *>(d + c * kOkPlane) =
						    make_float4(x[0][c] - mean4[c], x[1][c] - mean4[c], x[2][c] - mean4[c], x[3][c] - mean4[c]);
					*reinterpret_cast<float4 *>(d + 3 * kOkPlane) =
					    make_float4(s_alpha[al4 & 255u] - mean4[3], s_alpha[(al4 >> 8) & 255u] - mean4[3],
					                s_alpha[(al4 >> 16) & 255u] - mean4[3], s_alpha[al4 >> 24] - mean4[3]);
				}
				if (elig_cur && k < nb0) {
					float *d = s_p1 + slot;
#pragma unroll
					for (int c = 0; c < 3; ++c)
						*reinterpret_cast<float4 *>(d + c * kOkPlane) = make_float4(fresh[0][c], fresh[1][c], fresh[2][c], fresh[3][c]);
					*reinterpret_cast<float4 *>(d + 3 * kOkPlane) =
					    make_float4(s_alpha[fresh_alpha & 255u], s_alpha[(fresh_alpha >> 8) & 255u], s_alpha[(fresh_alpha >> 16) & 255u],
					                s_alpha[fresh_alpha >> 24]);
					if constexpr (G::kInRegs) {
#pragma unroll
						for (int j = 0; j < 4; ++j)
#pragma unroll
							for (int c = 0; c < 3; ++c) lab[k][j][c] = fresh[j][c];
						alpha_px[k] = fresh_alpha;
					} else {
						float *sb = a.ok_scratch + ((size_t)tile_cur * NB + k) * 832u;
						float4 *sp = reinterpret_cast<float4 *>(sb) + lane;
						sp[0] = make_float4(fresh[0][0], fresh[0][1], fresh[0][2], fresh[1][0]);
						sp[64] = make_float4(fresh[1][1], fresh[1][2], fresh[2][0], fresh[2][1]);
						sp[128] = make_float4(fresh[2][2], fresh[3][0], fresh[3][1], fresh[3][2]);
						sb[768u + lane] = __uint_as_float(fresh_alpha);
					}
				}
				__syncthreads();  // B: the bands of this interval are complete
			}
			have_prev = elig_cur;
		}
	} else {
		// ---------------- chain wave: lane = tile*4 + channel (a, b, l, alpha) ----------------
		const uint32_t ct = lane >> 2, cc = lane & 3u;
		const bool live = ct < kOkTiles;
		float acc1 = 0.0f, acc2 = 0.0f;
		__builtin_amdgcn_s_setprio(3);  // the serial part of every interval: first pick of its SIMD's issue slots
		// one dependent add chain per lane and pass; two register sets take turns so that 16 values are
		// in flight from LDS while 16 are added
		auto walk = [&](const float *band, float acc, const bool magnitude) -> float {
			const float4 *x = reinterpret_cast<const float4 *>(band + (ct * 4u + cc) * kOkPlane);
			auto add16 = [&](const float4 (&v)[4]) {
#pragma unroll
				for (int q = 0; q < 4; ++q) {
					if (magnitude) {
						acc += fabsf(v[q].x);  // operations.rs:80-83
						acc += fabsf(v[q].y);
						acc += fabsf(v[q].z);
						acc += fabsf(v[q].w);
					} else {
						acc += v[q].x;  // operations.rs:60-63, row-major pixel order
						acc += v[q].y;
						acc += v[q].z;
						acc += v[q].w;
					}
				}
			};
			float4 va[4] = {x[0], x[1], x[2], x[3]}, vb[4];
#pragma unroll 1
			for (uint32_t i = 0; i < 64; i += 8) {
#pragma unroll
				for (int q = 0; q < 4; ++q) vb[q] = x[i + 4u + (uint32_t)q];
				__builtin_amdgcn_sched_barrier(0);
				add16(va);
				__builtin_amdgcn_sched_barrier(0);
				const uint32_t n = i + 8u < 64u ? i + 8u : 0u;  // the last round re-reads the first (unused)
#pragma unroll
				for (int q = 0; q < 4; ++q) va[q] = x[n + (uint32_t)q];
				__builtin_amdgcn_sched_barrier(0);
				add16(vb);
				__builtin_amdgcn_sched_barrier(0);
			}
			return acc;
		};
		uint32_t h0 = 0, hm1 = 0, hm2 = 0;  // height of this lane's tile in batches p, p-1, p-2 (0: not taken)
		for (uint32_t p = 0; p < periods; ++p) {
			{
				const uint8_t *unused;
				hm2 = hm1;
				hm1 = h0;
				h0 = (live && p < own) ? oklab_tile_src<T>(a, (blockIdx.x + p * gridDim.x) * kOkTiles + ct, unused) : 0u;
			}
#pragma unroll 1
			for (uint32_t k = 0; k < NB; ++k) {
				// the bands written one interval ago: band kk of period pp
				const uint32_t pp = k > 0 ? p : p - 1u, kk = k > 0 ? k - 1u : NB - 1u;
				const bool any = k > 0 || p > 0;
				const bool p1_valid = any && pp < own;                   // pass 1 of batch pp
				const bool p2_valid = any && pp >= 1u && pp - 1u < own;  // pass 2 of batch pp - 1
				const uint32_t h1 = k > 0 ? h0 : hm1, h2 = k > 0 ? hm1 : hm2;  // tile heights of those two batches
				if (p1_valid && kk * G::kRowsPerBand < h1) acc1 = walk(s_p1, acc1, false);
				if (p2_valid && kk * G::kRowsPerBand < h2) acc2 = walk(s_p2, acc2, true);
				if (kk == NB - 1u) {
					if (p1_valid) {
						s_mean[lane] = __fdiv_rn(acc1, (float)((uint32_t)T * h1));  // operations.rs:65-68; read after barrier A
						acc1 = 0.0f;
					}
					if (p2_valid) {
						const float d0 = __shfl(acc2, (int)(lane & ~3u) + 0, 64), d1 = __shfl(acc2, (int)(lane & ~3u) + 1, 64);
						const float d2 = __shfl(acc2, (int)(lane & ~3u) + 2, 64), d3 = __shfl(acc2, (int)(lane & ~3u) + 3, 64);
						const float total = d0 + d1 + d2 + d3;  // :89
						const float value = __fdiv_rn(total, (float)((uint32_t)T * h2)) * a.factor * a.scale2;  // pixlzr.rs:162
						const uint32_t tg = (blockIdx.x + (pp - 1u) * gridDim.x) * kOkTiles + ct;
						if (live && cc == 0 && h2 != 0u)
							reinterpret_cast<uint2 *>(a.sums)[tg] = make_uint2(__float_as_uint(value), __float_as_uint(value));
						acc2 = 0.0f;
					}
				}
				__syncthreads();  // A
				__syncthreads();  // B
			}
		}
	}
}

hipError_t launch_oklab(const ShrinkArgs &a, uint32_t n_cus, hipStream_t stream)
{
	const uint32_t lds_bytes = (512u + 2u * 132u + 64u) * 4u + 2u * kOkBand * 4u;
	const uint32_t n_batches = (a.n_tiles + kOkTiles - 1) / kOkTiles;
	const uint32_t blocks = n_batches < n_cus ? n_batches : n_cus;
	hipError_t e;
	auto go = [&](auto kernel) -> hipError_t {
		if ((e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes)) != hipSuccess) return e;
		hipLaunchKernelGGL(kernel, dim3(blocks), dim3(1024), lds_bytes, stream, a);
		return hipGetLastError();
	};
	switch (a.bw) {
	case 16: return go(oklab_kernel<16>);
	case 32: return go(oklab_kernel<32>);
	case 64: return go(oklab_kernel<64>);
	default: return hipErrorInvalidValue;
	}
}

// ---------------------------------------------------------------------------
// finishing kernel: one lane per tile turns the detector result into the stored
// block value (and the raw detector outputs for pxz_lod_*).  All f64 work lives here.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) finish_kernel(const FinishArgs f)
{
	const uint32_t tile_g = blockIdx.x * 256u + threadIdx.x;
	if (tile_g >= f.n_tiles) return;
	const uint2 key = reinterpret_cast<const uint2 *>(f.sums)[tile_g];
	const uint32_t t = tile_g % f.tiles_per_frame;
	const uint32_t ty = t / f.cols, tx = t - ty * f.cols;
	finish_tile(key, (tx == f.cols - 1) ? f.edge_w : f.bw, (ty == f.rows - 1) ? f.edge_h : f.bh, f.mode, f.factor, f.value, f.lod0,
	            f.lod1, tile_g);
}

// ---------------------------------------------------------------------------
// Decode side: Pixlzr::expand (reference pixlzr.rs:77-122) + to_image
// (pixlzr_image.rs:24-74) in one pass: every stored tile is resized back to its full size with
// PixlzrBlock::resize (block.rs:273-334: clone, ResizeAlg::Nearest, or the two-pass convolution with
// u8 intermediate and alpha pre-/un-multiplication for RGBA) and written to its place in the frame.
// One wave per tile, persistent, tiles dealt by an LDS ticket counter.  LDS per wave: the source tile as
// one dword per pixel (premultiplied) and the horizontal pass's result [y][ox].  First version: scalar
// multiply-adds straight from the global tables; correctness and coalesced frame writes first.
// ---------------------------------------------------------------------------
template <int C>
__global__ void __launch_bounds__(256) expand_kernel(const ExpandArgs a)
{
	extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
	const uint32_t wpb = blockDim.x / 64u, sub = threadIdx.x / 64u, lane = threadIdx.x % 64u;
	uint32_t *s_ticket = lds + wpb * a.tile_dw;
	if (threadIdx.x == 0) *s_ticket = wpb;
	__syncthreads();
	uint32_t *s_src = lds + sub * a.tile_dw;
	uint32_t *s_tmp = s_src + a.bw * a.bh;
	uint32_t ticket = sub;
	for (;;) {
		const unsigned long long tl = (unsigned long long)blockIdx.x + (unsigned long long)ticket * gridDim.x;
		if (tl >= (unsigned long long)a.n_tiles) break;
		const uint32_t t = (uint32_t)tl;
		const uint32_t frame = t / a.tiles_per_frame, tf = t - frame * a.tiles_per_frame;
		const uint32_t ty = tf / a.cols, tx = tf - ty * a.cols;
		const uint32_t fw = (tx == a.cols - 1) ? a.edge_w : a.bw, fh = (ty == a.rows - 1) ? a.edge_h : a.bh;
		const uint32_t tw = a.tile_w[t], th = a.tile_h[t];
		const bool widen = C == 3 && a.out_channels == 4;  // RGB tiles into an RGBA frame (process())
		const uint32_t opx = widen ? 4u : (uint32_t)C;
		uint8_t *dst = a.dst + (size_t)frame * a.frame_stride + (size_t)(ty * a.bh) * a.pitch + (size_t)(tx * a.bw) * opx;
		auto put = [&](uint32_t ox, uint32_t oy, uint32_t px) {
			uint8_t *p = dst + (size_t)oy * a.pitch + ox * opx;
			if (C == 4 || widen) {
				*reinterpret_cast<uint32_t *>(p) = px;  // C == 3: alpha was set to 255 when the tile was staged
			} else {
				p[0] = (uint8_t)px;
				p[1] = (uint8_t)(px >> 8);
				p[2] = (uint8_t)(px >> 16);
			}
		};
		if (tw == 0 || th == 0 || tw > fw || th > fh) {
			if (lane == 0) atomicOr(a.status, 1u);
		} else {
			// ---- stored pixels -> one dword per pixel
			const uint8_t *src = a.slots + (size_t)t * a.slot_bytes;
			const uint32_t n = tw * th;
			const bool conv = a.filter != 0 && (tw != fw || th != fh);
			for (uint32_t i = lane; i < n; i += 64u) {
				uint32_t px;
				if constexpr (C == 4) {
					px = reinterpret_cast<const uint32_t *>(src)[i];
					if (conv) px = premultiply(px);  // fir: U8x4 is alpha-premultiplied before a convolution
				} else {
					px = (uint32_t)src[3 * i] | ((uint32_t)src[3 * i + 1] << 8) | ((uint32_t)src[3 * i + 2] << 16) | 0xff000000u;
				}
				s_src[i] = px;
			}
			tile_sync<1>();
			const uint32_t cls_x = fw == a.bw ? 0u : 1u, cls_y = fh == a.bh ? 0u : 1u;
			const ExpandTab tab_x = a.tabs[(0u * 2u + cls_x) * a.dir_stride + tw];
			const ExpandTab tab_y = a.tabs[(1u * 2u + cls_y) * a.dir_stride + th];
			// Vector form (RGBA tiles in RGBA frames whose full width is a multiple of 4): the windows of the tile
			// are staged into LDS once, a lane then makes 4 rows (horizontal pass) or 4 adjacent columns (vertical
			// pass, nearest, clone) per item, so weights are fetched once per 16 multiply-adds and the frame is
			// written 16 bytes per lane.  Same arithmetic as the scalar form below.
			if (C == 4 && (fw & 3u) == 0 && tab_x.window <= 8 && tab_y.window <= 8) {
				const uint32_t q4 = fw >> 2;
				auto put4 = [&](uint32_t q, uint32_t oy, uint4 px) {
					*reinterpret_cast<uint4 *>(dst + (size_t)oy * a.pitch + q * 16u) = px;
				};
				// per output sample 5 dwords: first | count << 16, then 8 weights (i16); x windows, then y windows
				uint32_t *s_wx = s_tmp + a.bw * a.bh, *s_wy = s_wx + 5u * a.bw;
				auto stage_windows = [&](uint32_t *w5, const ExpandTab &tab, uint32_t outs) {
					for (uint32_t o = lane; o < outs; o += 64u) {
						const uint32_t first = a.starts[tab.start_off + o];
						const uint32_t cnt = a.filter == 0 ? 1u : a.sizes[tab.start_off + o];
						uint32_t kk[4] = {0, 0, 0, 0};
						if (a.filter != 0) {
							const int16_t *k = a.coeffs + tab.coeff_off + o * tab.window;
							for (uint32_t j = 0; j < cnt; ++j) kk[j >> 1] |= (uint32_t)(uint16_t)k[j] << (16u * (j & 1u));
						}
						uint32_t *d = w5 + 5u * o;
						d[0] = first | (cnt << 16);
						d[1] = kk[0]; d[2] = kk[1]; d[3] = kk[2]; d[4] = kk[3];
					}
				};
				if (tw != fw) stage_windows(s_wx, tab_x, fw);
				if (th != fh) stage_windows(s_wy, tab_y, fh);
				tile_sync<1>();
				auto weight = [](const uint32_t (&kk)[4], uint32_t j) -> int32_t {
					return (int32_t)(int16_t)(kk[j >> 1] >> (16u * (j & 1u)));
				};
				if (tw == fw && th == fh) {  // block.rs:279-281: clone
					for (uint32_t i = lane; i < q4 * fh; i += 64u) {
						const uint32_t oy = small_div(i, q4), q = i - oy * q4;
						put4(q, oy, *reinterpret_cast<const uint4 *>(s_src + oy * fw + 4u * q));
					}
				} else if (a.filter == 0) {  // ResizeAlg::Nearest
					for (uint32_t i = lane; i < q4 * fh; i += 64u) {
						const uint32_t oy = small_div(i, q4), q = i - oy * q4;
						const uint32_t y = th == fh ? oy : (s_wy[5u * oy] & 0xffffu);
						const uint32_t *row = s_src + y * tw;
						uint4 px;
						if (tw == fw) {
							px = *reinterpret_cast<const uint4 *>(row + 4u * q);
						} else {
							px.x = row[s_wx[5u * (4u * q)] & 0xffffu];
							px.y = row[s_wx[5u * (4u * q + 1u)] & 0xffffu];
							px.z = row[s_wx[5u * (4u * q + 2u)] & 0xffffu];
							px.w = row[s_wx[5u * (4u * q + 3u)] & 0xffffu];
						}
						put4(q, oy, px);
					}
				} else {
					const bool need_h = tw != fw, need_v = th != fh;
					if (need_h) {
						// horizontal pass: item = (ox, 4 source rows); the rows beyond th repeat the last one (never stored)
						const int prec = tab_x.precision;
						const int32_t init = 1 << (prec - 1);
						const uint32_t groups = (th + 3u) >> 2;
						for (uint32_t i = lane; i < fw * groups; i += 64u) {
							const uint32_t yq = small_div(i, fw), ox = i - yq * fw;
							const uint32_t *wd = s_wx + 5u * ox;
							const uint32_t hdr = wd[0], first = hdr & 0xffffu, cnt = hdr >> 16;
							const uint32_t kk[4] = {wd[1], wd[2], wd[3], wd[4]};
							uint32_t yr[4];
#pragma unroll
							for (uint32_t r = 0; r < 4; ++r) yr[r] = 4u * yq + r < th ? 4u * yq + r : th - 1u;
							int32_t acc[4][4];
#pragma unroll
							for (int r = 0; r < 4; ++r)
#pragma unroll
								for (int c = 0; c < 4; ++c) acc[r][c] = init;
							for (uint32_t j = 0; j < cnt; ++j) {
								const int32_t w = weight(kk, j);
#pragma unroll
								for (int r = 0; r < 4; ++r) {
									const uint32_t p = s_src[yr[r] * tw + first + j];
									acc[r][0] += (int32_t)(p & 255u) * w;
									acc[r][1] += (int32_t)((p >> 8) & 255u) * w;
									acc[r][2] += (int32_t)((p >> 16) & 255u) * w;
									acc[r][3] += (int32_t)(p >> 24) * w;
								}
							}
#pragma unroll
							for (uint32_t r = 0; r < 4; ++r) {
								const uint32_t y = 4u * yq + r;
								if (y < th) {
									uint32_t px = clip8(acc[r][0], prec) | (clip8(acc[r][1], prec) << 8) | (clip8(acc[r][2], prec) << 16) |
									              (clip8(acc[r][3], prec) << 24);
									if (need_v) {
										s_tmp[y * fw + ox] = px;
									} else {
										put(ox, y, unpremultiply(px));
									}
								}
							}
						}
						tile_sync<1>();
					}
					if (need_v) {
						// vertical pass: item = (4 adjacent columns, oy); rows are fw wide (fw == tw when only this pass runs)
						const uint32_t *cur = need_h ? s_tmp : s_src;
						const int prec = tab_y.precision;
						const int32_t init = 1 << (prec - 1);
						for (uint32_t i = lane; i < q4 * fh; i += 64u) {
							const uint32_t oy = small_div(i, q4), q = i - oy * q4;
							const uint32_t *wd = s_wy + 5u * oy;
							const uint32_t hdr = wd[0], first = hdr & 0xffffu, cnt = hdr >> 16;
							const uint32_t kk[4] = {wd[1], wd[2], wd[3], wd[4]};
							int32_t acc[4][4];
#pragma unroll
							for (int r = 0; r < 4; ++r)
#pragma unroll
								for (int c = 0; c < 4; ++c) acc[r][c] = init;
							for (uint32_t j = 0; j < cnt; ++j) {
								const int32_t w = weight(kk, j);
								const uint4 v = *reinterpret_cast<const uint4 *>(cur + (first + j) * fw + 4u * q);
								const uint32_t p4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
								for (int r = 0; r < 4; ++r) {
									acc[r][0] += (int32_t)(p4[r] & 255u) * w;
									acc[r][1] += (int32_t)((p4[r] >> 8) & 255u) * w;
									acc[r][2] += (int32_t)((p4[r] >> 16) & 255u) * w;
									acc[r][3] += (int32_t)(p4[r] >> 24) * w;
								}
							}
							uint32_t o4[4];
#pragma unroll
							for (int r = 0; r < 4; ++r)
								o4[r] = unpremultiply(clip8(acc[r][0], prec) | (clip8(acc[r][1], prec) << 8) | (clip8(acc[r][2], prec) << 16) |
								                      (clip8(acc[r][3], prec) << 24));
							put4(q, oy, make_uint4(o4[0], o4[1], o4[2], o4[3]));
						}
					}
				}
			} else if (tw == fw && th == fh) {  // block.rs:279-281: clone
				RowWalker rw(lane, 64u, fw);
				for (uint32_t i = lane; i < fw * fh; i += 64u, rw.next()) put(rw.col, rw.row, s_src[i]);
			} else if (a.filter == 0) {  // ResizeAlg::Nearest
				const uint16_t *sx = a.starts + tab_x.start_off, *sy = a.starts + tab_y.start_off;
				RowWalker rw(lane, 64u, fw);
				for (uint32_t i = lane; i < fw * fh; i += 64u, rw.next()) {
					const uint32_t x = tw == fw ? rw.col : sx[rw.col], y = th == fh ? rw.row : sy[rw.row];
					put(rw.col, rw.row, s_src[y * tw + x]);
				}
			} else {
				const bool need_h = tw != fw, need_v = th != fh;
				const uint32_t *cur = s_src;
				if (need_h) {
					// horizontal pass: item = (ox, y) of the th source rows
					const uint16_t *st = a.starts + tab_x.start_off, *sz = a.sizes + tab_x.start_off;
					const int16_t *kf = a.coeffs + tab_x.coeff_off;
					const int prec = tab_x.precision;
					const int32_t init = 1 << (prec - 1);
					RowWalker rw(lane, 64u, fw);
					for (uint32_t i = lane; i < fw * th; i += 64u, rw.next()) {
						const uint32_t ox = rw.col, y = rw.row;
						const uint32_t first = st[ox], cnt = sz[ox];
						const int16_t *k = kf + ox * tab_x.window;
						const uint32_t *row = s_src + y * tw + first;
						int32_t acc[4] = {init, init, init, init};
						for (uint32_t j = 0; j < cnt; ++j) {
							const uint32_t p = row[j];
							const int32_t w = k[j];
							acc[0] += (int32_t)(p & 255u) * w;
							acc[1] += (int32_t)((p >> 8) & 255u) * w;
							acc[2] += (int32_t)((p >> 16) & 255u) * w;
							if constexpr (C == 4) acc[3] += (int32_t)(p >> 24) * w;
						}
						uint32_t px = clip8(acc[0], prec) | (clip8(acc[1], prec) << 8) | (clip8(acc[2], prec) << 16);
						px |= C == 4 ? clip8(acc[3], prec) << 24 : 0xff000000u;
						if (need_v) {
							s_tmp[y * fw + ox] = px;
						} else {
							if constexpr (C == 4) px = unpremultiply(px);
							put(ox, y, px);
						}
					}
					cur = s_tmp;
					tile_sync<1>();
				}
				if (need_v) {
					// vertical pass: item = (ox, oy); the rows of `cur` are fw wide when the horizontal pass ran
					const uint32_t cw = need_h ? fw : tw;
					const uint16_t *st = a.starts + tab_y.start_off, *sz = a.sizes + tab_y.start_off;
					const int16_t *kf = a.coeffs + tab_y.coeff_off;
					const int prec = tab_y.precision;
					const int32_t init = 1 << (prec - 1);
					RowWalker rw(lane, 64u, fw);
					for (uint32_t i = lane; i < fw * fh; i += 64u, rw.next()) {
						const uint32_t ox = rw.col, oy = rw.row;
						const uint32_t first = st[oy], cnt = sz[oy];
						const int16_t *k = kf + oy * tab_y.window;
						const uint32_t *col = cur + first * cw + ox;
						int32_t acc[4] = {init, init, init, init};
						for (uint32_t j = 0; j < cnt; ++j) {
							const uint32_t p = col[j * cw];
							const int32_t w = k[j];
							acc[0] += (int32_t)(p & 255u) * w;
							acc[1] += (int32_t)((p >> 8) & 255u) * w;
							acc[2] += (int32_t)((p >> 16) & 255u) * w;
							if constexpr (C == 4) acc[3] += (int32_t)(p >> 24) * w;
						}
						uint32_t px = clip8(acc[0], prec) | (clip8(acc[1], prec) << 8) | (clip8(acc[2], prec) << 16);
						px |= C == 4 ? clip8(acc[3], prec) << 24 : 0xff000000u;
						if constexpr (C == 4) px = unpremultiply(px);
						put(ox, oy, px);
					}
				}
			}
		}
		uint32_t nt = 0;
		if (lane == 0) nt = atomicAdd(s_ticket, 1u);
		ticket = __builtin_amdgcn_readfirstlane(nt);
		tile_sync<1>();  // the next tile reuses this wave's LDS
	}
}

hipError_t launch_expand(const ExpandArgs &a, uint32_t n_cus, hipStream_t stream)
{
	constexpr uint32_t kLds = 160u * 1024u;
	const uint32_t tile_bytes = a.tile_dw * 4u;
	uint32_t wpb = (kLds - 16u) / tile_bytes;
	if (wpb > 4u) wpb = 4u;
	if (wpb < 1u) return hipErrorInvalidValue;
	const uint32_t lds_bytes = wpb * tile_bytes + 16u;
	uint32_t per_cu = kLds / lds_bytes;
	if (per_cu > 4u) per_cu = 4u;
	if (per_cu < 1u) per_cu = 1u;
	const uint32_t need = (a.n_tiles + wpb - 1u) / wpb, resident = n_cus * per_cu;
	const uint32_t blocks = need < resident ? need : resident;
	hipError_t e;
	if (a.channels == 4) {
		auto k = expand_kernel<4>;
		if (lds_bytes > 64u * 1024u && (e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes)) != hipSuccess) return e;
		hipLaunchKernelGGL(k, dim3(blocks), dim3(64u * wpb), lds_bytes, stream, a);
	} else {
		auto k = expand_kernel<3>;
		if (lds_bytes > 64u * 1024u && (e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes)) != hipSuccess) return e;
		hipLaunchKernelGGL(k, dim3(blocks), dim3(64u * wpb), lds_bytes, stream, a);
	}
	return hipGetLastError();
}

// ---------------------------------------------------------------------------
// block-stream compaction: the valid bytes of the fixed output slots, tile order,
// into one contiguous stream (what a writer / the RCCL gather consumes).
//   offsets[t] = sum_{u<t} w[u]*h[u]*C  (exclusive scan, u64), offsets[n] = total
// Three launches: per-chunk scan (4096 tiles per block), scan of the chunk totals,
// copy (one wave per tile, 16-byte reads from the slot, dword or byte writes).
// ---------------------------------------------------------------------------
constexpr uint32_t kPackChunk = 4096;  // tiles per block in the scan: 256 threads x 16

__global__ void __launch_bounds__(256) pack_scan_local_kernel(const PackArgs a)
{
	__shared__ uint32_t s_wave[4];
	const uint32_t base = blockIdx.x * kPackChunk + threadIdx.x * 16u;
	uint32_t sz[16], run = 0;
#pragma unroll
	for (uint32_t i = 0; i < 16; ++i) {
		const uint32_t t = base + i;
		sz[i] = t < a.n_tiles ? (a.sizes ? a.sizes[t] : a.w[t] * a.h[t] * a.channels) : 0u;
		run += sz[i];
	}
	// exclusive scan of the per-thread totals inside the block
	uint32_t incl = run;
#pragma unroll
	for (int off = 1; off < 64; off <<= 1) {
		const uint32_t v = __shfl_up(incl, off, 64);
		if ((threadIdx.x & 63u) >= (uint32_t)off) incl += v;
	}
	if ((threadIdx.x & 63u) == 63u) s_wave[threadIdx.x >> 6] = incl;
	__syncthreads();
	uint32_t wave_off = 0;
	for (uint32_t q = 0; q < (threadIdx.x >> 6); ++q) wave_off += s_wave[q];
	uint32_t excl = wave_off + incl - run;
#pragma unroll
	for (uint32_t i = 0; i < 16; ++i) {
		const uint32_t t = base + i;
		if (t < a.n_tiles) a.offsets[t] = excl;  // chunk-local for now (a chunk holds < 2^32 bytes)
		excl += sz[i];
	}
	if (threadIdx.x == 255) a.chunk_totals[blockIdx.x] = excl;
}

__global__ void __launch_bounds__(1024) pack_scan_chunks_kernel(const PackArgs a)
{
	// one block: exclusive scan of the chunk totals (u64), in place
	__shared__ unsigned long long s_wave[16];
	__shared__ unsigned long long s_carry;
	if (threadIdx.x == 0) s_carry = 0;
	__syncthreads();
	for (uint32_t base = 0; base < a.n_chunks; base += 1024u) {
		const uint32_t i = base + threadIdx.x;
		const unsigned long long v = i < a.n_chunks ? a.chunk_totals[i] : 0ull;
		unsigned long long incl = v;
#pragma unroll
		for (int off = 1; off < 64; off <<= 1) {
			const unsigned long long u = __shfl_up(incl, off, 64);
			if ((threadIdx.x & 63u) >= (uint32_t)off) incl += u;
		}
		if ((threadIdx.x & 63u) == 63u) s_wave[threadIdx.x >> 6] = incl;
		__syncthreads();
		unsigned long long wave_off = s_carry;
		for (uint32_t q = 0; q < (threadIdx.x >> 6); ++q) wave_off += s_wave[q];
		if (i < a.n_chunks) a.chunk_totals[i] = wave_off + incl - v;
		__syncthreads();
		if (threadIdx.x == 1023) s_carry = wave_off + incl;
		__syncthreads();
	}
	if (threadIdx.x == 0) a.offsets[a.n_tiles] = s_carry;  // grand total
}

__global__ void __launch_bounds__(256) pack_copy_kernel(const PackArgs a)
{
	const uint32_t t = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
	if (t >= a.n_tiles) return;
	const unsigned long long off = a.chunk_totals[t / kPackChunk] + a.offsets[t];
	const uint32_t bytes = a.w[t] * a.h[t] * a.channels;
	const uint8_t *src = a.slots + (size_t)t * a.slot_bytes;
	uint8_t *dst = a.packed + off;
	if (lane == 0) a.offsets[t] = off;  // chunk-local -> global (each tile is owned by exactly one wave)
	if (off + bytes > a.capacity) return;
	if (((off | bytes) & 3ull) == 0 && (a.slot_bytes & 3u) == 0) {
		const uint32_t *s4 = reinterpret_cast<const uint32_t *>(src);
		uint32_t *d4 = reinterpret_cast<uint32_t *>(dst);
		for (uint32_t i = lane; i < (bytes >> 2); i += 64u) d4[i] = s4[i];
	} else {
		for (uint32_t i = lane; i < bytes; i += 64u) dst[i] = src[i];
	}
}

hipError_t launch_pack(const PackArgs &a, hipStream_t stream)
{
	hipLaunchKernelGGL(pack_scan_local_kernel, dim3(a.n_chunks), dim3(256), 0, stream, a);
	hipLaunchKernelGGL(pack_scan_chunks_kernel, dim3(1), dim3(1024), 0, stream, a);
	hipLaunchKernelGGL(pack_copy_kernel, dim3((a.n_tiles + 3) / 4), dim3(256), 0, stream, a);
	return hipGetLastError();
}

// ---------------------------------------------------------------------------
// .pixlzr bitstream on the GPU: Pixlzr::encode_to_vec (reference src/encoding/mod.rs:40-89) with
// encode_block (:168-200) and the `qoi` crate 0.4.1 encoder it calls (:181-189).
//
// QOI is sequential per tile (previous pixel, run length, 64-entry index), tiles are independent:
// one lane encodes one tile.  Tiles are first binned by pixel count (counting sort) so that the 64
// tiles of a wave have the same length and the lanes stay busy together.  Each lane keeps its index
// in LDS as table[slot][lane] (consecutive lanes -> consecutive banks), reads 4 pixels per 16-byte
// load from its slot and appends bytes through a 64-bit accumulator (aligned 8-byte stores) into a
// per-tile scratch record:  "block" | f32 BE value | u32 BE len | w,h BE | channels | 0 | ops | 0x00*7 0x01.
// Then: scan of the record lengths, splice into the final files, header + per-row length table.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) qoi_bin_count_kernel(const QoiArgs a)
{
	// per-block histogram in LDS, then one global atomic per non-empty class and block
	__shared__ uint32_t s_hist[32];
	if (threadIdx.x < 32) s_hist[threadIdx.x] = 0;
	__syncthreads();
	const uint32_t t = blockIdx.x * 256u + threadIdx.x;
	if (t < a.n_tiles) atomicAdd(&s_hist[31u - (uint32_t)__builtin_clz((a.w[t] * a.h[t]) | 1u)], 1u);
	__syncthreads();
	if (threadIdx.x < 32 && s_hist[threadIdx.x]) atomicAdd(&a.bins[threadIdx.x], s_hist[threadIdx.x]);
}

__global__ void qoi_bin_scan_kernel(const QoiArgs a)
{
	// largest tiles first (they set the tail): cursor[c] = start of class c in the permutation
	uint32_t run = 0;
	for (int c = 31; c >= 0; --c) {
		const uint32_t n = a.bins[c];
		a.bins[32 + c] = run;
		run += n;
	}
}

__global__ void __launch_bounds__(256) qoi_bin_scatter_kernel(const QoiArgs a)
{
	// the block reserves one range per class (one global atomic each), threads take slots inside it
	__shared__ uint32_t s_hist[32], s_base[32];
	if (threadIdx.x < 32) s_hist[threadIdx.x] = 0;
	__syncthreads();
	const uint32_t t = blockIdx.x * 256u + threadIdx.x;
	uint32_t cls = 0, local = 0;
	if (t < a.n_tiles) {
		cls = 31u - (uint32_t)__builtin_clz((a.w[t] * a.h[t]) | 1u);
		local = atomicAdd(&s_hist[cls], 1u);
	}
	__syncthreads();
	if (threadIdx.x < 32 && s_hist[threadIdx.x]) s_base[threadIdx.x] = atomicAdd(&a.bins[32u + threadIdx.x], s_hist[threadIdx.x]);
	__syncthreads();
	if (t < a.n_tiles) a.perm[s_base[cls] + local] = t;
}

struct ByteSink {
	unsigned long long acc;
	uint32_t cnt;        // bytes in acc
	unsigned long long *out;
	__device__ __forceinline__ void put(uint32_t b)
	{
		acc |= (unsigned long long)(b & 255u) << (8u * cnt);
		if (++cnt == 8u) {
			*out++ = acc;
			acc = 0;
			cnt = 0;
		}
	}
};

template <int C>
__global__ void __launch_bounds__(256) qoi_tiles_kernel(const QoiArgs a)
{
	__shared__ uint32_t s_index[4][64][64];  // [wave][slot][lane]
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	const bool live = i < a.n_tiles;
	const uint32_t t = live ? a.perm[i] : 0u;
	uint32_t(*index)[64] = s_index[wave];
#pragma unroll 8
	for (int sidx = 0; sidx < 64; ++sidx) index[sidx][lane] = 0u;  // qoi: index starts as zero pixels
	if (!live) return;
	const uint32_t w = a.w[t], h = a.h[t], n = w * h;
	const uint8_t *src = a.slots + (size_t)t * a.slot_bytes;
	uint8_t *rec = a.scratch + (size_t)t * a.stride;
	ByteSink s{0ull, 0u, reinterpret_cast<unsigned long long *>(rec)};
	// encode_block: magic, value, length placeholder (mod.rs:172-178,195)
	const uint32_t vb = __float_as_uint(a.value[t]);
	s.put('b'); s.put('l'); s.put('o'); s.put('c'); s.put('k');
	s.put(vb >> 24); s.put(vb >> 16); s.put(vb >> 8); s.put(vb);
	s.put(0); s.put(0); s.put(0); s.put(0);
	// qoi header minus its 4-byte magic (mod.rs:191): width, height BE, channels, colourspace 0
	s.put(w >> 24); s.put(w >> 16); s.put(w >> 8); s.put(w);
	s.put(h >> 24); s.put(h >> 16); s.put(h >> 8); s.put(h);
	s.put((uint32_t)C); s.put(0);

	uint32_t prev = 0xff000000u, run = 0, last_slot = 0;
	bool seen_op = false;
	const bool aligned = C == 4 ? true : ((a.slot_bytes & 3u) == 0);
	for (uint32_t base = 0; base < n; base += 4u) {
		uint32_t px4[4];
		if constexpr (C == 4) {
			const uint4 v = *reinterpret_cast<const uint4 *>(src + (size_t)base * 4u);  // slots are 16-byte aligned
			px4[0] = v.x; px4[1] = v.y; px4[2] = v.z; px4[3] = v.w;
		} else {
			if (aligned) {
				const uint32_t *p = reinterpret_cast<const uint32_t *>(src + (size_t)base * 3u);
				const uint32_t d0 = p[0], d1 = p[1], d2 = p[2];
				px4[0] = d0 & 0xffffffu;
				px4[1] = (d0 >> 24) | ((d1 & 0xffffu) << 8);
				px4[2] = (d1 >> 16) | ((d2 & 0xffu) << 16);
				px4[3] = d2 >> 8;
			} else {
				for (int j = 0; j < 4; ++j) {
					const uint8_t *p = src + (size_t)(base + j) * 3u;
					px4[j] = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16);
				}
			}
#pragma unroll
			for (int j = 0; j < 4; ++j) px4[j] |= 0xff000000u;
		}
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			const uint32_t pi = base + (uint32_t)j;
			if (pi >= n) break;
			const uint32_t px = px4[j];
			if (px == prev) {
				if (++run == 62u || pi + 1u == n) {
					s.put(0xc0u | (run - 1u));
					run = 0;
				}
				continue;
			}
			if (run) {
				// the crate writes a pending run of ONE as INDEX of the repeated pixel once any op was written
				s.put(run == 1u && seen_op ? last_slot : (0xc0u | (run - 1u)));
				run = 0;
			}
			seen_op = true;
			last_slot = ((px & 255u) * 3u + ((px >> 8) & 255u) * 5u + ((px >> 16) & 255u) * 7u + (px >> 24) * 11u) & 63u;
			if (index[last_slot][lane] == px) {
				s.put(last_slot);  // QOI_OP_INDEX
			} else {
				index[last_slot][lane] = px;
				const uint32_t dr = ((px & 255u) - (prev & 255u)) & 255u;
				const uint32_t dg = (((px >> 8) & 255u) - ((prev >> 8) & 255u)) & 255u;
				const uint32_t db = (((px >> 16) & 255u) - ((prev >> 16) & 255u)) & 255u;
				if (C == 4 && (px >> 24) != (prev >> 24)) {
					s.put(0xff); s.put(px); s.put(px >> 8); s.put(px >> 16); s.put(px >> 24);  // QOI_OP_RGBA
				} else if (((dr + 2u) & 255u) < 4u && ((dg + 2u) & 255u) < 4u && ((db + 2u) & 255u) < 4u) {
					s.put(0x40u | (((dr + 2u) & 3u) << 4) | (((dg + 2u) & 3u) << 2) | ((db + 2u) & 3u));  // QOI_OP_DIFF
				} else if (((dg + 32u) & 255u) < 64u && ((dr - dg + 8u) & 255u) < 16u && ((db - dg + 8u) & 255u) < 16u) {
					s.put(0x80u | ((dg + 32u) & 63u));  // QOI_OP_LUMA
					s.put((((dr - dg + 8u) & 15u) << 4) | ((db - dg + 8u) & 15u));
				} else {
					s.put(0xfe); s.put(px); s.put(px >> 8); s.put(px >> 16);  // QOI_OP_RGB
				}
			}
			prev = px;
		}
	}
	s.put(0); s.put(0); s.put(0); s.put(0); s.put(0); s.put(0); s.put(0); s.put(1);  // QOI end marker
	const uint32_t total = (uint32_t)(reinterpret_cast<uint8_t *>(s.out) - rec) + s.cnt;
	if (s.cnt) *s.out = s.acc;  // partial tail (the record stride leaves room)
	const uint32_t qlen = total - 13u;  // mod.rs:193-195
	rec[9] = (uint8_t)(qlen >> 24);
	rec[10] = (uint8_t)(qlen >> 16);
	rec[11] = (uint8_t)(qlen >> 8);
	rec[12] = (uint8_t)qlen;
	a.rec_len[t] = total;
}

// splice: one wave per tile copies its record to (frame+1)*hdr + offset[t]
__global__ void __launch_bounds__(256) qoi_splice_kernel(const QoiArgs a)
{
	const uint32_t t = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
	if (t >= a.n_tiles) return;
	const unsigned long long off = a.chunk_totals[t / kPackChunk] + a.offsets[t];
	if (lane == 0) a.offsets[t] = off;
	const uint32_t frame = t / a.tiles_per_frame;
	const unsigned long long dstoff = (unsigned long long)(frame + 1u) * a.hdr_bytes + off;
	const uint32_t len = a.rec_len[t];
	if (dstoff + len > a.capacity) return;
	const uint8_t *src = a.scratch + (size_t)t * a.stride;
	uint8_t *dst = a.out + dstoff;
	for (uint32_t i = lane; i < len; i += 64u) dst[i] = src[i];
}

// file header + line-length table (mod.rs:50-57,77-82): one thread per (frame, tile row)
__global__ void __launch_bounds__(256) qoi_headers_kernel(const QoiArgs a)
{
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	const uint32_t frames = a.n_tiles / a.tiles_per_frame;
	if (i >= frames * a.rows) return;
	const uint32_t f = i / a.rows, r = i - f * a.rows;
	const uint32_t t0 = f * a.tiles_per_frame + r * a.cols;
	// offsets[] already hold global record offsets (splice ran first); a row's length is a difference
	const unsigned long long lo = a.offsets[t0];
	const unsigned long long hi = (t0 + a.cols == a.n_tiles) ? a.offsets[a.n_tiles] : a.offsets[t0 + a.cols];
	const unsigned long long file0 = (unsigned long long)f * a.hdr_bytes + a.offsets[f * a.tiles_per_frame];
	if (file0 + a.hdr_bytes > a.capacity) return;
	uint8_t *hd = a.out + file0;
	const uint32_t len = (uint32_t)(hi - lo);
	uint8_t *lt = hd + 26 + 4 * r;
	lt[0] = (uint8_t)(len >> 24); lt[1] = (uint8_t)(len >> 16); lt[2] = (uint8_t)(len >> 8); lt[3] = (uint8_t)len;
	if (r == 0) {
		const uint8_t magic[10] = {'P', 'I', 'X', 'L', 'Z', 'R', 0, 0, 2, (uint8_t)a.filter_byte};
		for (int k = 0; k < 10; ++k) hd[k] = magic[k];
		const uint32_t v[4] = {a.width, a.height, a.bw, a.bh};
		for (int k = 0; k < 4; ++k) {
			hd[10 + 4 * k] = (uint8_t)(v[k] >> 24);
			hd[11 + 4 * k] = (uint8_t)(v[k] >> 16);
			hd[12 + 4 * k] = (uint8_t)(v[k] >> 8);
			hd[13 + 4 * k] = (uint8_t)v[k];
		}
		a.file_offsets[f] = file0;
		if (f + 1 == frames) a.file_offsets[frames] = (unsigned long long)frames * a.hdr_bytes + a.offsets[a.n_tiles];
	}
}

hipError_t launch_qoi(const QoiArgs &a, hipStream_t stream)
{
	hipError_t e = hipMemsetAsync(a.bins, 0, 64 * sizeof(uint32_t), stream);
	if (e != hipSuccess) return e;
	const uint32_t tb = (a.n_tiles + 255u) / 256u;
	hipLaunchKernelGGL(qoi_bin_count_kernel, dim3(tb), dim3(256), 0, stream, a);
	hipLaunchKernelGGL(qoi_bin_scan_kernel, dim3(1), dim3(1), 0, stream, a);
	hipLaunchKernelGGL(qoi_bin_scatter_kernel, dim3(tb), dim3(256), 0, stream, a);
	if (a.channels == 4) hipLaunchKernelGGL(qoi_tiles_kernel<4>, dim3(tb), dim3(256), 0, stream, a);
	else hipLaunchKernelGGL(qoi_tiles_kernel<3>, dim3(tb), dim3(256), 0, stream, a);
	// exclusive scan of the record lengths (same chunked scan as the pixel pack, sizes given)
	PackArgs p{};
	p.sizes = a.rec_len;
	p.offsets = a.offsets;
	p.chunk_totals = a.chunk_totals;
	p.n_tiles = a.n_tiles;
	p.n_chunks = a.n_chunks;
	hipLaunchKernelGGL(pack_scan_local_kernel, dim3(a.n_chunks), dim3(256), 0, stream, p);
	hipLaunchKernelGGL(pack_scan_chunks_kernel, dim3(1), dim3(1024), 0, stream, p);
	hipLaunchKernelGGL(qoi_splice_kernel, dim3((a.n_tiles + 3u) / 4u), dim3(256), 0, stream, a);
	const uint32_t frames = a.n_tiles / a.tiles_per_frame;
	hipLaunchKernelGGL(qoi_headers_kernel, dim3((frames * a.rows + 255u) / 256u), dim3(256), 0, stream, a);
	return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Decode side: Pixlzr::decode_from_vec (reference src/encoding/mod.rs:95-165) + decode_block (:202-242)
// + the `qoi` decoder it calls, on the device.
//   pixlzr_index_kernel  one wave per (file, tile row): header check, the row's start from the line-length
//                        table, then a walk over the row's records ("block", f32 BE value, u32 BE length,
//                        QOI minus its magic) -> per tile value, size and body position
//   qoi_decode_kernel    one lane per tile: the QOI op stream -> pixels in the tile's slot; the 64-entry
//                        index of every lane lives in LDS as in the encoder
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t be32(const uint8_t *p)
{
	return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3];
}

// One wave per (file, tile row).  The walk over a row's records is a dependent chain (each length gives the next
// record's position): it runs on bytes staged in LDS, chunk by chunk, so a step costs an LDS round trip instead of
// an HBM one.  Per chunk: all lanes load it (coalesced), lane 0 walks up to 64 records ahead using only the
// length fields, then the lanes check and publish those records in parallel.
constexpr uint32_t kIdxChunk = 8192;   // bytes of a row held in LDS at a time (per wave)
constexpr uint32_t kIdxHeader = 23;    // "block" + value + length + QOI header minus its magic, up to the channel byte
__global__ void __launch_bounds__(256) pixlzr_index_kernel(const DecodeArgs a)
{
	__shared__ __attribute__((aligned(16))) uint32_t s_chunk[4][kIdxChunk / 4u + 4u];
	__shared__ uint32_t s_pos[4][64];  // positions (relative to the chunk's first byte) of the records of a batch
	const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
	const uint32_t i = blockIdx.x * 4u + wave;
	if (i >= a.n_frames * a.rows) return;
	const uint32_t f = i / a.rows, r = i - f * a.rows;
	const unsigned long long f0 = a.file_offsets[f], f1 = a.file_offsets[f + 1];
	const uint8_t *file = a.files + f0;
	const unsigned long long flen = f1 - f0;
	const unsigned long long hdr = 26ull + 4ull * a.rows;
	const uint32_t t_row = f * a.tiles_per_frame + r * a.cols;
	auto bad_from = [&](uint32_t c0) {  // the row is unusable from column c0 on
		if (lane == 0) atomicOr(a.status, 2u);
		for (uint32_t c = c0 + lane; c < a.cols; c += 64u) {
			a.rec_len[t_row + c] = 0u;
			a.tile_w[t_row + c] = 0u;
			a.tile_h[t_row + c] = 0u;
		}
	};
	const uint8_t magic[9] = {'P', 'I', 'X', 'L', 'Z', 'R', 0, 0, 2};  // constants.rs:10-11: v0.0.2 (filter byte + line table)
	bool ok = flen >= hdr;
	if (ok) {
		for (int k = 0; k < 9; ++k) ok = ok && file[k] == magic[k];
		ok = ok && be32(file + 10) == a.width && be32(file + 14) == a.height && be32(file + 18) == a.bw && be32(file + 22) == a.bh;
	}
	if (!ok) {
		bad_from(0);
		return;
	}
	// the line-length table, lane-parallel: bytes before this row, and the length of all rows (mod.rs:141)
	unsigned long long before = 0, total = 0;
	for (uint32_t q = lane; q < a.rows; q += 64u) {
		const uint32_t len = be32(file + 26 + 4 * q);
		if (q < r) before += len;
		total += len;
	}
	for (int sh = 32; sh >= 1; sh >>= 1) {
		before += __shfl_xor(before, sh, 64);
		total += __shfl_xor(total, sh, 64);
	}
	if (hdr + total != flen) {
		bad_from(0);
		return;
	}
	unsigned long long p = hdr + before;
	const unsigned long long row_end = p + be32(file + 26 + 4 * r);
	const uint8_t *cb = reinterpret_cast<const uint8_t *>(s_chunk[wave]);
	uint32_t c = 0;
	while (c < a.cols) {
		// ---- stage file bytes [p, p + kIdxChunk) of the row (whole aligned dwords of the buffer, then the tail bytes)
		const unsigned long long want = row_end - p < (unsigned long long)kIdxChunk ? row_end - p : (unsigned long long)kIdxChunk;
		const uintptr_t g = reinterpret_cast<uintptr_t>(file + p);
		const uint32_t skew = (uint32_t)(g & 3u);  // the chunk starts at the aligned dword below p
		const uint32_t dwords = (skew + (uint32_t)want + 3u) / 4u;
		const uintptr_t buf_end = reinterpret_cast<uintptr_t>(a.files) + a.file_offsets[a.n_frames];
		for (uint32_t d = lane; d < dwords; d += 64u) {
			const uintptr_t ga = (g - skew) + 4ull * d;
			uint32_t v = 0;
			if (ga + 4u <= buf_end) {
				v = *reinterpret_cast<const uint32_t *>(ga);
			} else {
				for (uint32_t k = 0; k < 4u && ga + k < buf_end; ++k) v |= (uint32_t) * reinterpret_cast<const uint8_t *>(ga + k) << (8u * k);
			}
			s_chunk[wave][d] = v;
		}
		tile_sync<1>();
		const uint32_t have = (uint32_t)want;  // valid bytes behind cb + skew
		// ---- lane 0: positions of up to 64 records whose headers lie inside the chunk
		uint32_t n_rec = 0, walked = 0;  // walked: bytes of the chunk consumed by the records found
		bool broken = false;             // a record that cannot be walked past (it is the last of the batch)
		if (lane == 0) {
			uint32_t o = 0;
			while (n_rec < 64u && c + n_rec < a.cols) {
				if (p + o + 13ull + 10ull + 8ull > row_end) {  // no room for a record: broken row
					s_pos[wave][n_rec++] = o;
					broken = true;
					break;
				}
				if (o + kIdxHeader > have) break;  // header not in this chunk: restage from here
				const uint32_t qlen = be32(cb + skew + o + 9u);
				s_pos[wave][n_rec++] = o;
				if (qlen < 18u || p + o + 13ull + qlen > row_end) {
					broken = true;
					break;
				}
				o += 13u + qlen;
				if (o >= have && p + o < row_end && c + n_rec < a.cols) break;  // next record starts beyond the chunk
			}
			walked = o;
		}
		n_rec = __builtin_amdgcn_readfirstlane(n_rec);
		walked = __builtin_amdgcn_readfirstlane(walked);
		broken = __builtin_amdgcn_readfirstlane(broken ? 1u : 0u) != 0u;
		tile_sync<1>();
		// ---- all lanes: check and publish the batch
		bool good = true;
		if (lane < n_rec) {
			const uint32_t o = s_pos[wave][lane];
			const uint32_t cc = c + lane;
			const uint32_t fw = (cc == a.cols - 1) ? a.edge_w : a.bw, fh = (r == a.rows - 1) ? a.edge_h : a.bh;
			good = p + o + 13ull + 10ull + 8ull <= row_end;
			if (good) {
				const uint8_t *rec = cb + skew + o;
				good = rec[0] == 'b' && rec[1] == 'l' && rec[2] == 'o' && rec[3] == 'c' && rec[4] == 'k';
				const uint32_t qlen = be32(rec + 9);
				good = good && qlen >= 18u && p + o + 13ull + qlen <= row_end;
				if (good) {
					const uint32_t w = be32(rec + 13), h = be32(rec + 17), ch = rec[21];
					good = ch == a.channels && w >= 1 && h >= 1 && w <= fw && h <= fh;
					// (the walk used this record's length whether or not its other fields are sound, as the
					// sequential reader does not: a bad record ends the row there, see below)
					if (good) {
						const uint32_t t = t_row + cc;
						a.value[t] = __uint_as_float(be32(rec + 5));
						a.tile_w[t] = w;
						a.tile_h[t] = h;
						a.rec_off[t] = f0 + p + o + 13ull + 10ull;  // first op byte
						a.rec_len[t] = qlen - 10u - 8u;             // ops only: without the header and the end marker
					}
				}
			}
		}
		const unsigned long long bad = __builtin_amdgcn_ballot_w64(!good);
		if (bad != 0ull) {
			// the walk cannot continue past a broken record: the rest of the row is unusable (records of this batch
			// behind the first bad one were published above and are taken back here)
			bad_from(c + (uint32_t)__builtin_ctzll(bad));
			return;
		}
		(void)broken;  // (a broken record fails the checks above)
		if (n_rec == 0) {
			// a header that does not fit the rest of the row although a record is due: broken row
			bad_from(c);
			return;
		}
		c += n_rec;
		p += walked;
		tile_sync<1>();  // the chunk is restaged
	}
	if (p != row_end && lane == 0) atomicOr(a.status, 2u);
}

template <int C>
__global__ void __launch_bounds__(256) qoi_decode_kernel(const DecodeArgs a)
{
	__shared__ uint32_t s_index[4][64][64];  // [wave][slot][lane]
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	const uint32_t i0 = blockIdx.x * 256u + threadIdx.x;
	uint32_t(*index)[64] = s_index[wave];
#pragma unroll 8
	for (int sidx = 0; sidx < 64; ++sidx) index[sidx][lane] = 0u;  // qoi: index starts as zero pixels
	if (i0 >= a.n_tiles) return;
	const uint32_t t = a.perm[i0];  // tiles of similar pixel count share a wave (the walk is serial per lane)
	const uint32_t len = a.rec_len[t];
	if (len == 0) {
		a.tile_w[t] = 0;  // unusable record: the expand step skips and flags it
		a.tile_h[t] = 0;
		return;
	}
	// the op bytes, fetched 8 at a time through an aligned 64-bit window (a byte load per op byte would make
	// every op wait for a memory round trip).  The window may run up to 7 bytes past the last op: those are
	// bytes of the record's own 8-byte end marker, still inside the file.
	const unsigned long long first_byte = a.rec_off[t];
	const unsigned long long *wp = reinterpret_cast<const unsigned long long *>(a.files + (first_byte & ~7ull));
	unsigned long long acc = *wp++ >> (8u * (uint32_t)(first_byte & 7ull));
	uint32_t have = 8u - (uint32_t)(first_byte & 7ull);
	int32_t left = (int32_t)len;
	auto next_byte = [&]() -> uint32_t {
		if (have == 0u) {
			acc = *wp++;
			have = 8u;
		}
		const uint32_t b = (uint32_t)acc & 255u;
		acc >>= 8;
		--have;
		--left;
		return b;
	};
	const uint32_t n = a.tile_w[t] * a.tile_h[t];
	uint8_t *dst = a.slots + (size_t)t * a.slot_bytes;
	uint32_t px = 0xff000000u, run = 0;
	uint4 hold = make_uint4(0, 0, 0, 0);
	for (uint32_t i = 0; i < n; ++i) {
		if (run > 0) {
			--run;
		} else if (left > 0) {
			// (as in the qoi crate, only the op's first byte is checked against the end of the stream; a truncated
			// last op reads on into the end marker, which is inside the file)
			const uint32_t b1 = next_byte();
			if (b1 == 0xfeu) {  // QOI_OP_RGB
				const uint32_t r = next_byte(), g = next_byte(), b = next_byte();
				px = (px & 0xff000000u) | r | (g << 8) | (b << 16);
			} else if (b1 == 0xffu) {  // QOI_OP_RGBA
				const uint32_t r = next_byte(), g = next_byte(), b = next_byte(), al = next_byte();
				px = r | (g << 8) | (b << 16) | (al << 24);
			} else if ((b1 & 0xc0u) == 0x00u) {  // QOI_OP_INDEX
				px = index[b1][lane];
			} else if ((b1 & 0xc0u) == 0x40u) {  // QOI_OP_DIFF
				const uint32_t r = ((px & 255u) + ((b1 >> 4) & 3u) - 2u) & 255u;
				const uint32_t g = (((px >> 8) & 255u) + ((b1 >> 2) & 3u) - 2u) & 255u;
				const uint32_t b = (((px >> 16) & 255u) + (b1 & 3u) - 2u) & 255u;
				px = (px & 0xff000000u) | r | (g << 8) | (b << 16);
			} else if ((b1 & 0xc0u) == 0x80u) {  // QOI_OP_LUMA
				const uint32_t b2 = next_byte();
				const uint32_t vg = (b1 & 0x3fu) - 32u;
				const uint32_t r = ((px & 255u) + vg - 8u + ((b2 >> 4) & 15u)) & 255u;
				const uint32_t g = (((px >> 8) & 255u) + vg) & 255u;
				const uint32_t b = (((px >> 16) & 255u) + vg - 8u + (b2 & 15u)) & 255u;
				px = (px & 0xff000000u) | r | (g << 8) | (b << 16);
			} else {  // QOI_OP_RUN
				run = b1 & 0x3fu;
			}
			index[((px & 255u) * 3u + ((px >> 8) & 255u) * 5u + ((px >> 16) & 255u) * 7u + (px >> 24) * 11u) & 63u][lane] = px;
		} else {
			atomicOr(a.status, 2u);  // the op stream ended before the tile was full
			a.tile_w[t] = 0;
			a.tile_h[t] = 0;
			return;
		}
		if constexpr (C == 4) {
			// four pixels per 16-byte store (slots are 16-byte aligned: bw*bh*4 bytes each)
			const uint32_t k = i & 3u;
			if (k == 0) hold.x = px;
			else if (k == 1) hold.y = px;
			else if (k == 2) hold.z = px;
			else {
				hold.w = px;
				reinterpret_cast<uint4 *>(dst)[i >> 2] = hold;
			}
		} else {
			dst[3 * i] = (uint8_t)px;
			dst[3 * i + 1] = (uint8_t)(px >> 8);
			dst[3 * i + 2] = (uint8_t)(px >> 16);
		}
	}
	if constexpr (C == 4) {
		const uint32_t tail = n & 3u, base = n & ~3u;  // 1x1, 2x1 ... tiles
		if (tail >= 1) reinterpret_cast<uint32_t *>(dst)[base] = hold.x;
		if (tail >= 2) reinterpret_cast<uint32_t *>(dst)[base + 1] = hold.y;
		if (tail >= 3) reinterpret_cast<uint32_t *>(dst)[base + 2] = hold.z;
	}
}

hipError_t launch_decode(const DecodeArgs &a, hipStream_t stream)
{
	hipError_t e = hipMemsetAsync(a.bins, 0, 64 * sizeof(uint32_t), stream);
	if (e != hipSuccess) return e;
	hipLaunchKernelGGL(pixlzr_index_kernel, dim3((a.n_frames * a.rows + 3u) / 4u), dim3(256), 0, stream, a);
	const uint32_t tb = (a.n_tiles + 255u) / 256u;
	QoiArgs q{};  // the encoder's binning by pixel count, on the sizes the index kernel has just read
	q.w = a.tile_w;
	q.h = a.tile_h;
	q.n_tiles = a.n_tiles;
	q.bins = a.bins;
	q.perm = a.perm;
	hipLaunchKernelGGL(qoi_bin_count_kernel, dim3(tb), dim3(256), 0, stream, q);
	hipLaunchKernelGGL(qoi_bin_scan_kernel, dim3(1), dim3(1), 0, stream, q);
	hipLaunchKernelGGL(qoi_bin_scatter_kernel, dim3(tb), dim3(256), 0, stream, q);
	if (a.channels == 4) hipLaunchKernelGGL(qoi_decode_kernel<4>, dim3(tb), dim3(256), 0, stream, a);
	else hipLaunchKernelGGL(qoi_decode_kernel<3>, dim3(tb), dim3(256), 0, stream, a);
	return hipGetLastError();
}

// ---------------------------------------------------------------------------
// RGB input on the RGBA fast paths.  An RGB tile and the same tile with alpha 255 give the same detector values
// (the alpha chain of the Oklab detector sums exact ones: mean 1, deviation 0, and x + 0 = x) and the same
// resampled colours (premultiplying by 255 is the identity; the host checks that the constant-255 alpha comes
// back as 255 from every table in use, so nothing is un-premultiplied).  So RGB frames are widened once on the
// way in and the tile slots narrowed on the way out, instead of running the generic kernel on 3-byte pixels.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) rgb_to_rgba_kernel(const WidenArgs a)
{
	// four pixels per thread: 12 bytes in (three dwords when the row start allows), 16 bytes out
	const uint32_t x = (blockIdx.x * 256u + threadIdx.x) * 4u, y = blockIdx.y, f = blockIdx.z;
	if (x >= a.width) return;
	const uint8_t *p = a.src + (size_t)f * a.src_frame_stride + (size_t)y * a.src_pitch + (size_t)x * 3u;
	uint32_t *d = reinterpret_cast<uint32_t *>(a.dst + (size_t)f * a.dst_frame_stride + (size_t)y * a.dst_pitch + (size_t)x * 4u);
	if (x + 4u <= a.width && (reinterpret_cast<uintptr_t>(p) & 3u) == 0) {
		const uint32_t *q = reinterpret_cast<const uint32_t *>(p);
		const uint32_t d0 = q[0], d1 = q[1], d2 = q[2];
		uint4 o;
		o.x = (d0 & 0x00ffffffu) | 0xff000000u;
		o.y = (d0 >> 24) | ((d1 & 0xffffu) << 8) | 0xff000000u;
		o.z = (d1 >> 16) | ((d2 & 0xffu) << 16) | 0xff000000u;
		o.w = (d2 >> 8) | 0xff000000u;
		*reinterpret_cast<uint4 *>(d) = o;  // dst rows are 16-byte aligned (scratch pitch)
	} else {
		for (uint32_t i = 0; i < 4u && x + i < a.width; ++i)
			d[i] = (uint32_t)p[3u * i] | ((uint32_t)p[3u * i + 1u] << 8) | ((uint32_t)p[3u * i + 2u] << 16) | 0xff000000u;
	}
}

__global__ void __launch_bounds__(256) slots_rgba_to_rgb_kernel(const NarrowArgs a)
{
	const uint32_t t = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
	if (t >= a.n_tiles) return;
	const uint32_t n = a.w[t] * a.h[t];
	const uint32_t *s = reinterpret_cast<const uint32_t *>(a.slots4 + (size_t)t * a.slot4_bytes);
	uint8_t *d = a.slots3 + (size_t)t * a.slot3_bytes;
	const bool aligned = (reinterpret_cast<uintptr_t>(d) & 3u) == 0;
	const uint32_t n4 = aligned ? n & ~3u : 0u;
	for (uint32_t i = lane * 4u; i < n4; i += 256u) {  // four pixels: 16 bytes in, three dwords out
		const uint4 v = *reinterpret_cast<const uint4 *>(s + i);
		uint32_t *o = reinterpret_cast<uint32_t *>(d + 3u * i);
		o[0] = (v.x & 0x00ffffffu) | (v.y << 24);
		o[1] = ((v.y >> 8) & 0xffffu) | (v.z << 16);
		o[2] = ((v.z >> 16) & 0xffu) | (v.w << 8);
	}
	for (uint32_t i = n4 + lane; i < n; i += 64u) {
		const uint32_t px = s[i];
		d[3u * i] = (uint8_t)px;
		d[3u * i + 1u] = (uint8_t)(px >> 8);
		d[3u * i + 2u] = (uint8_t)(px >> 16);
	}
}

hipError_t launch_widen(const WidenArgs &a, hipStream_t stream)
{
	hipLaunchKernelGGL(rgb_to_rgba_kernel, dim3((a.width + 1023u) / 1024u, a.height, a.n_frames), dim3(256), 0, stream, a);
	return hipGetLastError();
}
hipError_t launch_narrow(const NarrowArgs &a, hipStream_t stream)
{
	hipLaunchKernelGGL(slots_rgba_to_rgb_kernel, dim3((a.n_tiles + 3u) / 4u), dim3(256), 0, stream, a);
	return hipGetLastError();
}

// ---------------------------------------------------------------------------
// synthetic frames (DESIGN.md "Synthetic frames"): integer-only, one pixel per thread
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t fmix32(uint32_t h)
{
	h ^= h >> 16;
	h *= 0x85ebca6bu;
	h ^= h >> 13;
	h *= 0xc2b2ae35u;
	h ^= h >> 16;
	return h;
}

__global__ void __launch_bounds__(256) synth_kernel(const SynthArgs s)
{
	const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t y = blockIdx.y;
	const uint32_t f = blockIdx.z;
	if (x >= s.width) return;
	const uint32_t amp_table[8] = {0, 0, 1, 2, 4, 16, 64, 255};
	const uint32_t seed = 0x5049584Cu + s.first_frame + f;
	uint32_t amp = amp_table[((x >> 5) * 7u + (y >> 5) * 13u + seed) & 7u];
	if (s.dist == 2) amp = 0;
	if (s.dist == 3) amp = 255;
	const uint32_t idx = (y * s.width + x) * 4u;
	uint8_t *p = s.dst + (size_t)f * s.frame_stride + (size_t)y * s.pitch + (size_t)x * s.channels;
	uint32_t px = 0;
#pragma unroll
	for (uint32_t c = 0; c < 3; ++c) {
		const int base = (int)(((3u * x + 5u * y + 85u * c) >> 3) & 255u);
		const int nz = (int)(fmix32((idx + c) ^ seed) % (amp + 1u));
		int v = base + nz - (int)(amp / 2u);
		v = v < 0 ? 0 : (v > 255 ? 255 : v);
		px |= (uint32_t)v << (8 * c);
	}
	if (s.channels == 4) {
		const uint32_t al = s.dist == 1 ? 128u + fmix32((idx + 3u) ^ seed) % 128u : 255u;
		*reinterpret_cast<uint32_t *>(p) = px | (al << 24);
	} else {
		p[0] = (uint8_t)px;
		p[1] = (uint8_t)(px >> 8);
		p[2] = (uint8_t)(px >> 16);
	}
}

// ---------------------------------------------------------------------------
// launchers (called from pxz_api.cpp)
// ---------------------------------------------------------------------------
template <int NW, int C, int MODE>
static hipError_t launch_one(const ShrinkArgs &a, const LaunchGeom &g, hipStream_t stream)
{
	auto kernel = shrink_kernel<NW, C, MODE>;
	if (g.lds_bytes > 64u * 1024u) {
		hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
		                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds_bytes);
		if (e != hipSuccess) return e;
	}
	hipLaunchKernelGGL(kernel, dim3(g.blocks), dim3(g.threads), g.lds_bytes, stream, a);
	return hipGetLastError();
}

template <int NW>
static hipError_t launch_nw(const ShrinkArgs &a, uint32_t channels, const LaunchGeom &g, hipStream_t stream)
{
	if (channels == 4)
		return a.mode == 1 ? launch_one<NW, 4, 1>(a, g, stream) : launch_one<NW, 4, 0>(a, g, stream);
	return a.mode == 1 ? launch_one<NW, 3, 1>(a, g, stream) : launch_one<NW, 3, 0>(a, g, stream);
}

// waves per tile: 1 up to 32x32, then one wave per 1024 px, capped at 16
uint32_t waves_per_tile(uint32_t bw, uint32_t bh)
{
	const uint32_t px = bw * bh;
	if (px <= 1024) return 1;
	if (px <= 2048) return 2;
	if (px <= 4096) return 4;
	if (px <= 8192) return 8;
	return 16;
}

// Launch geometry.  NW == 1: persistent blocks of up to 12 waves (one LDS tile image each) sized to
// the 160 KB of LDS, at most one resident set per CU; NW > 1: one tile per block, grid capped likewise.
LaunchGeom plan_launch(const ShrinkArgs &a, uint32_t channels, uint32_t n_cus)
{
	LaunchGeom g{};
	const uint32_t nw = waves_per_tile(a.bw, a.bh);
	const uint32_t tile_bytes = a.tile_dw * 4u;
	constexpr uint32_t kLds = 160u * 1024u;
	if (nw == 1) {
		uint32_t wpb = (kLds - 16u) / tile_bytes;  // 16 bytes: the ticket counter
		if (wpb > 12u) wpb = 12u;
		if (const char *e = getenv("PXZ_WPB")) {  // tuning knob: waves per block
			const uint32_t v = (uint32_t)atoi(e);
			if (v >= 1 && v < wpb) wpb = v;
		}
		if (wpb < 1u) wpb = 1u;
		g.threads = 64u * wpb;
		g.lds_bytes = wpb * tile_bytes + 16u;
		const uint32_t per_cu = kLds / g.lds_bytes > 0 ? kLds / g.lds_bytes : 1u;
		uint32_t resident = n_cus * (per_cu > 2u ? 2u : per_cu);
		const uint32_t need = (a.n_tiles + wpb - 1u) / wpb;
		g.blocks = need < resident ? need : resident;
	} else {
		g.threads = 64u * nw;
		g.lds_bytes = tile_bytes + 16u * nw;
		const uint32_t per_cu = kLds / g.lds_bytes > 0 ? kLds / g.lds_bytes : 1u;
		const uint32_t resident = n_cus * per_cu * 2u;
		g.blocks = a.n_tiles < resident ? a.n_tiles : resident;
	}
	return g;
}

bool fast64_applicable(const ShrinkArgs &a, uint32_t channels)
{
	return channels == 4 && a.bw == 64 && a.bh == 64 && (a.mode == 1 || a.oklab_given) && a.work != nullptr &&
	       (a.out_px == nullptr || (a.filter != 0 && a.mf64 != nullptr));
}

bool fast16_applicable(const ShrinkArgs &a, uint32_t channels)
{
	return channels == 4 && a.bw == 16 && a.bh == 16 && a.work != nullptr &&
	       (a.out_px == nullptr || a.filter == 0 || a.tab_dw != 0) && !(a.mode == 0 && !a.oklab_given);
}

bool fast32_applicable(const ShrinkArgs &a, uint32_t channels)
{
	return channels == 4 && a.bw == 32 && a.bh == 32 && a.work != nullptr &&
	       (a.out_px == nullptr || a.filter == 0 || a.tab_dw != 0) && !(a.mode == 0 && !a.oklab_given);
}

hipError_t launch_shrink(const ShrinkArgs &a, uint32_t channels, uint32_t n_cus, hipStream_t stream)
{
	ShrinkArgs ga = a;
	if (fast64_applicable(a, channels)) {
		// 64x64: the four-wave kernel for full opaque tiles; it leaves the rest in the worklist
		Fast64Args f{};
		f.src = a.src;
		f.frame_stride = a.frame_stride;
		f.pitch = a.pitch;
		f.cols = a.cols;
		f.rows = a.rows;
		f.tiles_per_frame = a.tiles_per_frame;
		f.n_tiles = a.n_tiles;
		f.div_tpf = a.div_tpf;
		f.div_cols = a.div_cols;
		f.full_cols = a.full_cols;
		f.full_rows = a.full_rows;
		f.ok_rows = a.ok_rows;
		f.filter = a.filter;
		f.sums = a.sums;
		f.out_w = a.out_w;
		f.out_h = a.out_h;
		f.out_px = a.out_px;
		f.work = a.work;
		f.work_slot = a.work_slot;
		f.mf64 = a.mf64;
		for (int j = 0; j < kMaxLevel; ++j) {
			f.mf_off[j] = a.tabs[j].mf_off;
			f.precision[j] = a.tabs[j].precision;
			f.breaks[j] = a.breaks[0][j];
		}
		f.breaks_asc = a.breaks_asc[0];
		const uint32_t lds_bytes = lds64_dwords(3) * 4u;
		constexpr uint32_t kLds = 160u * 1024u;
		const uint32_t per_cu = kLds / lds_bytes;
		const uint32_t resident = n_cus * per_cu;
		const uint32_t blocks = a.n_tiles < resident ? a.n_tiles : resident;
		hipError_t e;
		{
			const bool full = a.out_px != nullptr && a.out_w != nullptr && a.out_h != nullptr;
			void (*k)(const Fast64Args) = a.mode == 1 ? (full ? shrink64_kernel<1, false, true> : shrink64_kernel<1, false, false>)
			                                          : (full ? shrink64_kernel<0, false, true> : shrink64_kernel<0, false, false>);
			if (lds_bytes > 64u * 1024u && (e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes)) != hipSuccess) return e;
			hipLaunchKernelGGL(k, dim3(blocks), dim3(256), lds_bytes, stream, f);
		}
		if ((e = hipGetLastError()) != hipSuccess) return e;
		if (a.mid_event && (e = hipEventRecord(static_cast<hipEvent_t>(a.mid_event), stream)) != hipSuccess) return e;
		ga.mid_event = nullptr;
		// full tiles with transparency are on list A: the four-plane instance takes it when transparency was
		// announced or seen before, else the generic kernel walks it after list B
		const bool run_alpha = a.out_px != nullptr && a.out_w != nullptr && a.out_h != nullptr && a.alpha_kernel != 0;
		ga.list_a_too = a.out_px != nullptr && !run_alpha ? 1u : 0u;
		if (run_alpha) {
			const uint32_t lds_a = lds64_dwords(4) * 4u;
			const uint32_t blocks_a = n_cus * (kLds / lds_a);
			if (a.mode == 1) {
				auto k = shrink64_kernel<1, true, true>;
				if (lds_a > 64u * 1024u && (e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_a)) != hipSuccess) return e;
				hipLaunchKernelGGL(k, dim3(blocks_a), dim3(256), lds_a, stream, f);
			} else {
				auto k = shrink64_kernel<0, true, true>;
				if (lds_a > 64u * 1024u && (e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_a)) != hipSuccess) return e;
				hipLaunchKernelGGL(k, dim3(blocks_a), dim3(256), lds_a, stream, f);
			}
			if ((e = hipGetLastError()) != hipSuccess) return e;
		}
	} else if (fast32_applicable(a, channels) || fast16_applicable(a, channels)) {
		// 1) the lean kernel for full opaque tiles (32x32), or for 2x2 groups of them (16x16); it leaves the rest
		// in the worklist
		const bool groups16 = a.bw == 16;
		Fast32Args f{};
		f.n_groups = a.n_frames_x_groups;
		f.div_gpf = a.div_gpf;
		f.div_gcols = a.div_gcols;
		f.src = a.src;
		f.frame_stride = a.frame_stride;
		f.pitch = a.pitch;
		f.cols = a.cols;
		f.rows = a.rows;
		f.tiles_per_frame = a.tiles_per_frame;
		f.n_tiles = a.n_tiles;
		f.div_tpf = a.div_tpf;
		f.div_cols = a.div_cols;
		f.full_cols = a.full_cols;
		f.full_rows = a.full_rows;
		f.ok_rows = a.ok_rows;
		f.filter = a.filter;
		f.sums = a.sums;
		f.out_w = a.out_w;
		f.out_h = a.out_h;
		f.out_px = a.out_px;
		f.work = a.work;
		f.work_slot = a.work_slot;
		// full tiles with transparency always go to list A; shrink32a_kernel takes it when transparency was announced
		// or seen before, else the worklist kernel walks it after list B
		f.alpha_list = (!groups16 && a.out_px != nullptr && (a.filter == 0 || a.tab_dw != 0)) ? 1u : 0u;
		const bool run_alpha = f.alpha_list != 0 && a.alpha_kernel != 0;
		ga.list_a_too = f.alpha_list != 0 && !run_alpha ? 1u : 0u;
		f.trows = a.trows;
		f.tab_dw = a.out_px && a.filter != 0 ? a.tab_dw : 0u;
		// only the x-axis tables of full tiles are used (the y axis of a 32x32 tile is identical): they
		// are the first rows of the blob, up to where the y axis begins
		const uint32_t y_begin = a.tabs[2 * kMaxLevel + 1].rows_off;
		if (f.tab_dw != 0 && y_begin != 0 && y_begin < f.tab_dw) f.tab_dw = (y_begin + 3u) & ~3u;
		for (int j = 0; j < kMaxLevel; ++j) {
			f.breaks[j] = a.breaks[0][j];
			f.tabs[j] = a.tabs[j];  // x axis, full class; identical to the y axis for 32x32
		}
		f.breaks_asc = a.breaks_asc[0];
		// planes 3 x 576 dwords (R, G, B), output region (transposed planes 3 x 144 of the dot2 form / parked
		// pixels of nearest, one-pass and matrix-core outputs: at most 32x16), slack for zero-weight over-reads
		static_assert(3u * kTD32 <= kOut32, "transposed planes fit the output region");
		f.tile_dw = f.out_px ? 3u * kPD32 + kOut32 + 2u * kRS32 : 3u * kPD32 + 4u * kRS32;
		f.tile_dw = (f.tile_dw + 3u) & ~3u;
		constexpr uint32_t kLds = 160u * 1024u;
		constexpr uint32_t kTail = 16u + 16u * 2u * kListBatch * 4u;  // the ticket counter + every wave's list batches
		uint32_t wpb = (kLds - f.tab_dw * 4u - kTail) / (f.tile_dw * 4u);
		if (wpb > 16u) wpb = 16u;
		if (const char *e = getenv("PXZ_WPB")) {
			const uint32_t v = (uint32_t)atoi(e);
			if (v >= 1 && v < wpb) wpb = v;
		}
		const uint32_t lds_bytes = f.tab_dw * 4u + wpb * f.tile_dw * 4u + kTail;
		const uint32_t per_cu = kLds / lds_bytes > 0 ? kLds / lds_bytes : 1u;
		const uint32_t resident = n_cus * (per_cu > 2u ? 2u : per_cu);
		const uint32_t units = groups16 ? f.n_groups : a.n_tiles;
		const uint32_t need = (units + wpb - 1u) / wpb;
		const uint32_t blocks = need < resident ? need : resident;
		f.chunk_lg = 3;
		if (const char *e = getenv("PXZ_CHUNK_LG")) f.chunk_lg = (uint32_t)atoi(e) & 15u;
		hipError_t e = hipSuccess;  // the worklist counter of this launch was zeroed by the previous one (or at allocation)
		if (groups16) {
			const bool full = a.out_px != nullptr && a.out_w != nullptr && a.out_h != nullptr;
			void (*k)(const Fast32Args) = a.mode == 1 ? (full ? shrink16_kernel<1, true> : shrink16_kernel<1, false>)
			                                          : (full ? shrink16_kernel<0, true> : shrink16_kernel<0, false>);
			if (lds_bytes > 64u * 1024u && (e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes)) != hipSuccess) return e;
			hipLaunchKernelGGL(k, dim3(blocks), dim3(64u * wpb), lds_bytes, stream, f);
		} else {
			const bool full = a.out_px != nullptr && a.out_w != nullptr && a.out_h != nullptr;
			void (*k)(const Fast32Args) = a.mode == 1 ? (full ? shrink32_kernel<1, true> : shrink32_kernel<1, false>)
			                                          : (full ? shrink32_kernel<0, true> : shrink32_kernel<0, false>);
			if (lds_bytes > 64u * 1024u && (e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes)) != hipSuccess) return e;
			hipLaunchKernelGGL(k, dim3(blocks), dim3(64u * wpb), lds_bytes, stream, f);
		}
		if ((e = hipGetLastError()) != hipSuccess) return e;
		if (a.mid_event && (e = hipEventRecord(static_cast<hipEvent_t>(a.mid_event), stream)) != hipSuccess) return e;
		ga.mid_event = nullptr;
		if (run_alpha) {
			// 1b) the full tiles with transparency that shrink32_kernel listed: four planes, no output region
			Fast32Args fa = f;
			fa.tile_dw = (4u * kPD32 + 2u * kRS32 + 3u) & ~3u;
			uint32_t wa = (kLds - fa.tab_dw * 4u - kTail) / (fa.tile_dw * 4u);
			if (wa > 16u) wa = 16u;
			const uint32_t lds_a = fa.tab_dw * 4u + wa * fa.tile_dw * 4u + kTail;
			if (a.mode == 1) {
				auto k = shrink32a_kernel<1>;
				if (lds_a > 64u * 1024u && (e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_a)) != hipSuccess) return e;
				hipLaunchKernelGGL(k, dim3(n_cus), dim3(64u * wa), lds_a, stream, fa);
			} else {
				auto k = shrink32a_kernel<0>;
				if (lds_a > 64u * 1024u && (e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_a)) != hipSuccess) return e;
				hipLaunchKernelGGL(k, dim3(n_cus), dim3(64u * wa), lds_a, stream, fa);
			}
			if ((e = hipGetLastError()) != hipSuccess) return e;
		}
		// 2) the generic kernel walks the worklist (usually empty or a few percent of the tiles)
	} else {
		ga.work = nullptr;
	}
	// (the block-cooperative detector's values stand for every tile it took, deferred or not: the fast
	// kernels leave those tiles' sums alone)
	const LaunchGeom g = plan_launch(ga, channels, n_cus);
	switch (waves_per_tile(ga.bw, ga.bh)) {
	case 1: return launch_nw<1>(ga, channels, g, stream);
	case 2: return launch_nw<2>(ga, channels, g, stream);
	case 4: return launch_nw<4>(ga, channels, g, stream);
	case 8: return launch_nw<8>(ga, channels, g, stream);
	default: return launch_nw<16>(ga, channels, g, stream);
	}
}

hipError_t launch_oklab_pixels(const uint32_t *px, uint32_t n, float *out, uint32_t n_cus, hipStream_t stream)
{
	const uint32_t need = (n / 2u + 255u) / 256u + 1u, blocks = need < 8u * n_cus ? need : 8u * n_cus;
	hipLaunchKernelGGL(oklab_pixels_kernel, dim3(blocks), dim3(256), 0, stream, px, n, reinterpret_cast<float4 *>(out));
	return hipGetLastError();
}

hipError_t launch_finish(const FinishArgs &f, hipStream_t stream)
{
	hipLaunchKernelGGL(finish_kernel, dim3((f.n_tiles + 255) / 256), dim3(256), 0, stream, f);
	return hipGetLastError();
}

hipError_t launch_synth(const SynthArgs &s, hipStream_t stream)
{
	dim3 grid((s.width + 255) / 256, s.height, s.n_frames);
	hipLaunchKernelGGL(synth_kernel, grid, dim3(256), 0, stream, s);
	return hipGetLastError();
}

}  // namespace pxz
