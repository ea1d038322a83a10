// pxz_device.h -- device-side helpers shared by the kernel files (gfx950 only): constant tables, index math,
// packed 16-bit / dot2 / matrix-core resample helpers of the fast paths, the worklist batching, finish_tile.
// Included by every pxz_*.hip; everything here is __device__ __forceinline__ or a file-local table.
#pragma once
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
static __constant__ uint32_t kSrgbToLinearBits[256] = {
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
static __constant__ RecipAlphaTable kRecipAlpha = RecipAlphaTable();

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
	// the identity at alpha 255 (recip = 256): no table look-up when every lane that is here holds an opaque pixel
	if (__builtin_amdgcn_ballot_w64(al != 255u) == 0ull) return px;
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
// a + b on both u16 halves where the low halves cannot carry (sums of a few 8-bit samples): ONE plain 32-bit add.  Written
// as a packed add (v_pk_add_u16) it is a VOP3P instruction, and those -- like every three-source integer form, SDWA and DPP
// -- occupy the SIMD for 4.3 cycles against the 2 of a VOP2 add (profiles/r03_issue_probe.txt).
__device__ __forceinline__ uint32_t add2x16(uint32_t a, uint32_t b) { return a + b; }
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

// LDS accesses spelled one by one (base + 16-bit immediate): left to itself the compiler pairs neighbouring accesses
// into ds_read2 / ds_write2, whose 8-bit offsets cost a vector add per new base -- 16 of them per tile in the detector of
// shrink32_kernel, which is bound by vector instructions, not by LDS instructions.
typedef const volatile __attribute__((address_space(3))) uint32_t *lds_cptr;
__device__ __forceinline__ uint32_t lds_dword(const uint32_t *p) { return *(lds_cptr)p; }
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint2 lds_load2(const uint32_t *p)
{
	const u32x2 v = *(const volatile __attribute__((address_space(3))) u32x2 *)p;
	return make_uint2(v.x, v.y);
}
__device__ __forceinline__ void lds_store2(uint32_t *p, uint2 v)
{
	u32x2 w = {v.x, v.y};
	*(volatile __attribute__((address_space(3))) u32x2 *)p = w;
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
// C = 3: out is a slot of 3-byte pixels (shrink16_kernel on RGB frames)
template <int LPI, int C = 4>
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
			if constexpr (C == 4) {
				out[oy * nw + ox] = px;
			} else {
				uint8_t *o3 = reinterpret_cast<uint8_t *>(out) + 3u * (oy * nw + ox);
				o3[0] = (uint8_t)px;
				o3[1] = (uint8_t)(px >> 8);
				o3[2] = (uint8_t)(px >> 16);
			}
		}
	}
}

// single-pass cases of the fast path (one axis keeps its 32 samples), opaque tile.
// Vertical only: item = (pair of columns, output row) straight on the [y][x] planes; the (row j,
// row j+1) sample pairs dot2 needs are built with two perms per column pair and row pair.
// T = 32: a whole 32x32 LDS image, pixels parked as dwords (out).  T = 16: a 16x16 tile somewhere inside the image (s_pl points at
// its first pixel pair), pixels stored straight to its slot (C bytes each).
template <int T = 32, int C = 4>
__device__ __forceinline__ void fast32_v_only(const uint32_t *trows, const AxisTab &ty, const uint32_t *s_pl, uint32_t lane,
                                              uint32_t nh, uint32_t *out)
{
	constexpr uint32_t kPairs = T / 2;  // column pairs of a row
	const uint32_t lgy = 31u - (uint32_t)__builtin_clz(nh);
	const uint32_t oy = lane & (nh - 1u);  // invariant per lane: 64 is a multiple of nh
	RowRegs r;
	load_row(trows + ty.rows_off + oy * ty.row_stride, r);
	const int prec = ty.precision;
	const int32_t init = 1 << (prec - 1);
	const uint32_t al = clip8(init + 255 * r.ksum, prec);  // constant-255 alpha through the same window
	const uint32_t wq = ty.wquads;
	for (uint32_t i = lane; i < kPairs * nh; i += 64u) {
		const uint32_t qx = i >> lgy;  // column pair
		const uint32_t *colp = s_pl + (r.fq * 4u) * kRS32 + qx;
		int32_t acc[3][2];
#pragma unroll
		for (int c = 0; c < 3; ++c) acc[c][0] = acc[c][1] = init;
#pragma unroll
		for (int q = 0; q < T / 4; ++q) {
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
		// pixels (2qx, 2qx+1) of output row oy, row length T
		if constexpr (C == 4) {
			*reinterpret_cast<uint2 *>(out + (oy * kPairs + qx) * 2u) = o;
		} else {
			store_pixel<3>(reinterpret_cast<uint8_t *>(out), (oy * kPairs + qx) * 2u, o.x);
			store_pixel<3>(reinterpret_cast<uint8_t *>(out), (oy * kPairs + qx) * 2u + 1u, o.y);
		}
	}
}

// Horizontal only: item = (output column, pair of rows), results go straight to the slot.
template <int T = 32, int C = 4>
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
	for (uint32_t i = lane; i < nw * (uint32_t)(T / 2); i += 64u) {  // T / 2 pairs of rows
		const uint32_t yp = i >> lgx;
		const uint32_t *row = s_pl + yp * (2 * kRS32) + r.fq * 2u;
		int32_t acc[3][2];
#pragma unroll
		for (int c = 0; c < 3; ++c) acc[c][0] = acc[c][1] = init;
#pragma unroll
		for (int q = 0; q < T / 4; ++q) {
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
			store_pixel<C>(reinterpret_cast<uint8_t *>(out), (2u * yp + rr) * nw + ox, px);
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
template <int C = 4>
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
	if (items >= 64u) fast32_v<1, C>(trows, tx, ty, s_tmp, lane, nw, nh, out);
	else if (items >= 32u) fast32_v<2, C>(trows, tx, ty, s_tmp, lane, nw, nh, out);
	else if (items >= 16u) fast32_v<4, C>(trows, tx, ty, s_tmp, lane, nw, nh, out);
	else fast32_v<8, C>(trows, tx, ty, s_tmp, lane, nw, nh, out);
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
			const uint2 d0 = lds_load2(row);       // columns 4g .. 4g+3
			const uint2 d1 = lds_load2(row + 8u);  // columns 16+4g .. 16+4g+3
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

// The matrix-core form for narrow outputs, 32x32 -> nw x nh with nw in {4, 2, 1}, nh <= 16, opaque tile whose alpha
// stays 255 (both tables say so): the three channels share ONE accumulator.  Column n of the horizontal product is
// (channel n / nw, output column n % nw) -- the weight operand of channel c is the level's table in the columns of
// channel c and zero elsewhere (a lane picks its own row of the table or an all-zero one), and the three products
// accumulate: D[y][(c, ox)] = sum_c' sum_x P_c'[y][x] B_c'[x][(c, ox)].  One clamp per 16 rows instead of three, one
// vertical product instead of three, and the columns of a fourth "channel" carry the constant 255 (bias = the clamp's
// top) through the vertical product: the alpha byte.  Lane (n, g) ends with channel n / nw of the pixels (4g + r, n % nw)
// and stores them as bytes of the parked pixels.  Same integers as resample_mfma32, 14 MFMAs + ~85 vector instructions
// (the dot2 form: ~125 for a 2x1 tile; resample_mfma32: ~170).
__device__ __forceinline__ void resample_mfma32_narrow(const uint32_t *s_tab, const AxisTab &tx, const AxisTab &ty, const uint32_t *s_pl,
                                                       uint32_t lane, uint32_t nw, uint32_t nh, uint32_t *out)
{
	const uint32_t o = lane & 15u, g = lane >> 4;
	const uint32_t lgw = nw >> 1;  // 1, 2, 4 -> 0, 1, 2
	const uint32_t ch = o >> lgw, ox = o & (nw - 1u);
	const uint32_t *mx = s_tab + tx.mf_off, *my = s_tab + ty.mf_off;
	const uint32_t own = 2u * (16u * g + ox), none = 2u * (16u * g + 15u);  // (rows >= the output size hold zero weights)
	const long ky_lo = *reinterpret_cast<const long *>(my + 2u * lane), ky_hi = *reinterpret_cast<const long *>(my + 128u + 2u * lane);
	const uint32_t px_ = tx.precision, py = ty.precision;
	const int32_t top_x = (int32_t)((256u << px_) - 1u), top_y = (int32_t)((256u << py) - 1u);
	const int32_t bx = ch == 3u ? top_x : (int32_t)mx[256u + ox];
	const v4i32 cy = *reinterpret_cast<const v4i32 *>(my + 256u + 4u * g);
	const v4i32 zero = {0, 0, 0, 0};
	v4i32 lo[2] = {{bx, bx, bx, bx}, {bx, bx, bx, bx}}, hi[2] = {zero, zero};
	const uint32_t *rowp = s_pl + o * kRS32 + 2u * g;
#pragma unroll
	for (uint32_t c = 0; c < 3; ++c) {
		const uint32_t sel = ch == c ? own : none;
		const long k_lo = *reinterpret_cast<const long *>(mx + sel), k_hi = *reinterpret_cast<const long *>(mx + 128u + sel);
#pragma unroll
		for (uint32_t mb = 0; mb < 2; ++mb) {
			const uint32_t *row = rowp + c * kPD32 + mb * (16u * kRS32);
			const uint2 d0 = lds_load2(row);       // columns 4g .. 4g+3
			const uint2 d1 = lds_load2(row + 8u);  // columns 16+4g .. 16+4g+3
			const uint32_t a0 = __builtin_amdgcn_perm(d0.y, d0.x, 0x06040200u) ^ 0x80808080u;
			const uint32_t a1 = __builtin_amdgcn_perm(d1.y, d1.x, 0x06040200u) ^ 0x80808080u;
			const long av = (long)(((unsigned long long)a1 << 32) | (unsigned long long)a0);
			lo[mb] = __builtin_amdgcn_mfma_i32_16x16x32_i8(av, k_lo, lo[mb], 0, 0, 0);
			hi[mb] = __builtin_amdgcn_mfma_i32_16x16x32_i8(av, k_hi, hi[mb], 0, 0, 0);
		}
	}
	uint32_t t[2];
#pragma unroll
	for (uint32_t mb = 0; mb < 2; ++mb) {
		uint32_t packed = 0;
		put_byte_shr<0>(packed, clamp_fixed(hi[mb][0], lo[mb][0], top_x), px_);
		put_byte_shr<1>(packed, clamp_fixed(hi[mb][1], lo[mb][1], top_x), px_);
		put_byte_shr<2>(packed, clamp_fixed(hi[mb][2], lo[mb][2], top_x), px_);
		put_byte_shr<3>(packed, clamp_fixed(hi[mb][3], lo[mb][3], top_x), px_);
		t[mb] = packed ^ 0x80808080u;
	}
	const long tv = (long)(((unsigned long long)t[1] << 32) | (unsigned long long)t[0]);
	const v4i32 vlo = __builtin_amdgcn_mfma_i32_16x16x32_i8(ky_lo, tv, cy, 0, 0, 0);
	const v4i32 vhi = __builtin_amdgcn_mfma_i32_16x16x32_i8(ky_hi, tv, zero, 0, 0, 0);
	// rows 4g + r of column (ch, ox): byte ch of pixel (4g + r) * nw + ox.  nh is a power of two: a lane with 4g < nh has
	// min(nh, 4) rows (scalar tests); rows past nh are not produced.
	if (o < 4u * nw && 4u * g < nh) {
		uint8_t *dst = reinterpret_cast<uint8_t *>(out) + (4u * g * nw + ox) * 4u + ch;
		dst[0] = (uint8_t)(clamp_fixed(vhi[0], vlo[0], top_y) >> py);
		if (nh > 1u) dst[nw * 4u] = (uint8_t)(clamp_fixed(vhi[1], vlo[1], top_y) >> py);
		if (nh > 2u) {
			dst[2u * nw * 4u] = (uint8_t)(clamp_fixed(vhi[2], vlo[2], top_y) >> py);
			dst[3u * nw * 4u] = (uint8_t)(clamp_fixed(vhi[3], vlo[3], top_y) >> py);
		}
	}
}

// The two-pass convolution of a GROUP of four 16x16 tiles (a 32x32 region of LDS planes, shrink16_kernel) as ONE set of
// matrix-core products (round 4): the weight operands are block-diagonal.  A k slot (g, j) of v_mfma_i32_16x16x32_i8 stands for
// sample 4g + j of the FIRST 16 (j < 4) or of the SECOND 16 (j >= 4) of the 32, so
//   horizontal, half mb (rows 16 mb ..): B[k][n] = Wx of tile (dx = n >> 3, dy = mb) for output column n & 7 in the k slots of its
//               own 16 source columns and zero in the other 16: one product gives T of the left tile in columns 0..7 and of the
//               right tile in columns 8..15, each at its own level (a lane reads its own tile's table: levels are per lane);
//   vertical    A[m][k] = Wy of the tile in row dy = m >> 3 for output row m & 7, in the k slots of its own 16 source rows.  The
//               left and the right tiles of a row have different y levels, so there are two products into ONE accumulator:
//               A of the left tiles times T with columns 8..15 zeroed, plus A of the right tiles times T with columns 0..7 zeroed
//               (a zero byte is p - 128 = 0: it adds nothing, the bias of the lane's own tile carries its 128 * sum K).
// 24 MFMAs and ~230 vector instructions for the four tiles; the dot2 form (resample_fast16_hv) took ~118 per tile.  Same
// integers: the clamp after each pass, the precisions and the biases are those of the lane's tile.  Opaque tiles whose alpha
// stays 255 (every table involved says so); tiles outside `mask` (clone, one-pass, nothing) ride along and are not stored.
__device__ __forceinline__ uint32_t clamp_fixed_v(int32_t hi, int32_t lo, int32_t top)
{
	const int32_t v = (int32_t)(((uint32_t)hi << 8) + (uint32_t)lo);
	int32_t r;
	asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(v), "v"(top));
	return (uint32_t)r;
}

// Per-lane views of the group (set up by shrink16_kernel while it decides the tiles one by one, so that no per-tile scalar
// outlives its tile): tx[mb] = table of the tile in this lane's COLUMN half (n >> 3) and row half mb (the B operand of the
// horizontal product of half mb); ty[side] = table of the tile in column `side` and the ROW half this lane stands for as a row
// of the weight operand (m >> 3, m = lane & 15); tyo / nw / nh / dst / ok = the tile this lane's outputs belong to (column half
// n >> 3, row half g >> 1): its y table, reduced size, slot, and whether it is one of the two-pass tiles at all.  Tables of
// views without a two-pass tile: any valid one (those columns and rows are not stored).
template <int C>
__device__ __forceinline__ void resample_group16_mfma(const uint32_t *s_tab, const uint32_t (&tx)[2], const uint32_t (&ty)[2], uint32_t tyo,
                                                      const uint32_t *s_pl, uint32_t lane, uint32_t nw, uint32_t nh, bool ok, uint8_t *dst)
{
	const uint32_t n = lane & 15u, g = lane >> 4, o = n & 7u;
	const bool second = n >= 8u;  // as a column: a right tile; as a row of a weight operand: a bottom tile
	const v4i32 zero = {0, 0, 0, 0};
	auto operand = [&](uint32_t w) -> long {  // the lane's four weight bytes in the k slots of its own 16 samples
		return (long)(second ? (unsigned long long)w << 32 : (unsigned long long)w);
	};
	long kx_lo[2], kx_hi[2], ky_lo[2], ky_hi[2];
	int32_t bx[2], top_x[2];
	uint32_t px_[2];
#pragma unroll
	for (uint32_t mb = 0; mb < 2; ++mb) {
		const uint32_t *t = s_tab + tx[mb];
		kx_lo[mb] = operand(t[o * 4u + g]);
		kx_hi[mb] = operand(t[32u + o * 4u + g]);
		bx[mb] = (int32_t)t[64u + o];
		px_[mb] = t[81];
		top_x[mb] = (int32_t)((256u << px_[mb]) - 1u);
	}
#pragma unroll
	for (uint32_t side = 0; side < 2; ++side) {
		const uint32_t *t = s_tab + ty[side];
		ky_lo[side] = operand(t[o * 4u + g]);
		ky_hi[side] = operand(t[32u + o * 4u + g]);
	}
	// this lane's outputs: column n, rows 4g + r -> pixel (ox = o, oy = 4 (g & 1) + r) of its tile
	const uint32_t *to = s_tab + tyo;
	const v4i32 cy = *reinterpret_cast<const v4i32 *>(to + 64u + 4u * (g & 1u));
	const uint32_t py = to[81];
	const int32_t top_y = (int32_t)((256u << py) - 1u);
	uint32_t pix[4] = {0xff000000u, 0xff000000u, 0xff000000u, 0xff000000u};
	const uint32_t *rowp = s_pl + n * kRS32 + 2u * g;
#pragma unroll
	for (uint32_t c = 0; c < 3; ++c) {
		uint32_t t[2];
#pragma unroll
		for (uint32_t mb = 0; mb < 2; ++mb) {
			const uint32_t *row = rowp + c * kPD32 + mb * (16u * kRS32);
			const uint2 d0 = lds_load2(row);       // columns 4g .. 4g+3 (left tile)
			const uint2 d1 = lds_load2(row + 8u);  // columns 16+4g .. 16+4g+3 (right tile)
			const uint32_t a0 = __builtin_amdgcn_perm(d0.y, d0.x, 0x06040200u) ^ 0x80808080u;
			const uint32_t a1 = __builtin_amdgcn_perm(d1.y, d1.x, 0x06040200u) ^ 0x80808080u;
			const long av = (long)(((unsigned long long)a1 << 32) | (unsigned long long)a0);
			const v4i32 cx = {bx[mb], bx[mb], bx[mb], bx[mb]};
			const v4i32 lo = __builtin_amdgcn_mfma_i32_16x16x32_i8(av, kx_lo[mb], cx, 0, 0, 0);
			const v4i32 hi = __builtin_amdgcn_mfma_i32_16x16x32_i8(av, kx_hi[mb], zero, 0, 0, 0);
			uint32_t packed = 0;
			put_byte_shr<0>(packed, clamp_fixed_v(hi[0], lo[0], top_x[mb]), px_[mb]);
			put_byte_shr<1>(packed, clamp_fixed_v(hi[1], lo[1], top_x[mb]), px_[mb]);
			put_byte_shr<2>(packed, clamp_fixed_v(hi[2], lo[2], top_x[mb]), px_[mb]);
			put_byte_shr<3>(packed, clamp_fixed_v(hi[3], lo[3], top_x[mb]), px_[mb]);
			t[mb] = packed ^ 0x80808080u;
		}
		const uint32_t l0 = second ? 0u : t[0], l1 = second ? 0u : t[1], r0 = second ? t[0] : 0u, r1 = second ? t[1] : 0u;
		const long tl = (long)(((unsigned long long)l1 << 32) | (unsigned long long)l0);
		const long tr = (long)(((unsigned long long)r1 << 32) | (unsigned long long)r0);
		v4i32 lo = __builtin_amdgcn_mfma_i32_16x16x32_i8(ky_lo[0], tl, cy, 0, 0, 0);
		lo = __builtin_amdgcn_mfma_i32_16x16x32_i8(ky_lo[1], tr, lo, 0, 0, 0);
		v4i32 hi = __builtin_amdgcn_mfma_i32_16x16x32_i8(ky_hi[0], tl, zero, 0, 0, 0);
		hi = __builtin_amdgcn_mfma_i32_16x16x32_i8(ky_hi[1], tr, hi, 0, 0, 0);
#pragma unroll
		for (int r = 0; r < 4; ++r) {
			const uint32_t v = clamp_fixed_v(hi[r], lo[r], top_y);
			if (c == 0) put_byte_shr<0>(pix[r], v, py);
			else if (c == 1) put_byte_shr<1>(pix[r], v, py);
			else put_byte_shr<2>(pix[r], v, py);
		}
	}
	if (ok && o < nw) {
#pragma unroll
		for (uint32_t r = 0; r < 4; ++r) {
			const uint32_t oy = 4u * (g & 1u) + r;
			if (oy < nh) store_pixel<C>(dst, oy * nw + o, pix[r]);
		}
	}
}

// Source pixels are read once per launch: loads marked non-temporal ("nt": stream through the caches, do not displace
// what is reused).  On the headline kernel that alone is worth 6-8 % -- the tile reads no longer evict each other's
// not-yet-consumed lines and the output stream's lines from L2 (round 2, tools: A/B of two builds on one box).
__device__ __forceinline__ uint4 stream_load4(const uint8_t *p)
{
	typedef uint32_t u32q __attribute__((ext_vector_type(4)));
	const u32q v = __builtin_nontemporal_load(reinterpret_cast<const u32q *>(p));
	return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint2 stream_load2(const uint8_t *p)
{
	typedef uint32_t u32d __attribute__((ext_vector_type(2)));
	const u32d v = __builtin_nontemporal_load(reinterpret_cast<const u32d *>(p));
	return make_uint2(v.x, v.y);
}
__device__ __forceinline__ uint3 stream_load3(const uint8_t *p)
{
	typedef uint32_t u32t __attribute__((ext_vector_type(3)));
	const u32t v = __builtin_nontemporal_load(reinterpret_cast<const u32t *>(p));
	return make_uint3(v.x, v.y, v.z);
}

// Fast-path eligibility of a tile (full 32x32 RGBA, 16-byte aligned rows) and its first byte.
template <int C = 4, class Args>
__device__ __forceinline__ bool fast32_tile_src(const Args &a, uint32_t tile_g, const uint8_t *&src)
{
	if (tile_g >= a.n_tiles) return false;
	const uint32_t frame = fastdiv(tile_g, a.div_tpf);
	const uint32_t t = tile_g - frame * a.tiles_per_frame;
	const uint32_t ty = fastdiv(t, a.div_cols), tx = t - ty * a.cols;
	src = a.src + (size_t)frame * a.frame_stride + (size_t)(ty * 32u) * a.pitch + (size_t)(tx * 32u) * (uint32_t)C;
	return tx < a.full_cols && ty < a.full_rows;  // full size; the alignment of the batch is folded in by the host
}
// Issues the four loads of a lane's share of a fast tile: rows l/8 + 8k, pixel quad l%8 -- 16 bytes of an RGBA row, 12
// of an RGB one (pre[k].w unused).
template <int C = 4, class Args>
__device__ __forceinline__ void fast32_prefetch(const Args &a, uint32_t tile_g, uint32_t lane, uint4 (&pre)[4], bool &valid, bool skip_loads = false)
{
	const uint8_t *src;
	valid = fast32_tile_src<C>(a, tile_g, src);
	if (valid && !skip_loads) {  // (skip_loads: the caller knows it will not look at the pixels -- a tile the detector already copied)
		const uint8_t *p = src + (size_t)(lane >> 3) * a.pitch + (lane & 7u) * (4u * (uint32_t)C);
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			if constexpr (C == 4) {
				pre[k] = stream_load4(p + (size_t)(8 * k) * a.pitch);
			} else {
				const uint3 v = stream_load3(p + (size_t)(8 * k) * a.pitch);  // rows are 4-byte aligned
				pre[k] = make_uint4(v.x, v.y, v.z, 0u);
			}
		}
	}
}

// Does sums[] already hold this tile's Oklab value (left by an oklab_kernel launch)?  Interior: the full tiles, plus the
// ragged last row of the square sizes when its height is a whole number of bands (ok_rows).  Edges: the regions a
// run-time-geometry launch has taken (ok_edges: bit 0 right column, bit 1 bottom row, bit 2 corner tile).
template <class Args>
__device__ __forceinline__ bool oklab_value_given(const Args &a, uint32_t tx, uint32_t ty)
{
	const bool in_x = tx < a.full_cols, in_y = ty < a.full_rows;
	if (in_x && ty < a.ok_rows) return true;  // (ok_rows >= full_rows)
	const uint32_t bit = in_x ? 2u : (in_y ? 1u : 4u);
	return (a.ok_edges & bit) != 0u;
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

}  // namespace pxz
