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
struct RowWalker {
	uint32_t row, col, drow, dcol, width;
	__device__ RowWalker(uint32_t first, uint32_t step, uint32_t width_) : width(width_)
	{
		row = first / width_;
		col = first - row * width_;
		drow = step / width_;
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
// TW   compile-time tile side (32) enabling the fast path for full RGBA tiles, 0 = none
template <int NW, int C, int MODE, int TW>
__global__ void __launch_bounds__(NW == 1 ? 256 : 64 * NW) shrink_kernel(const ShrinkArgs a)
{
	extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
	constexpr uint32_t TPT = 64u * NW;                // threads per tile
	constexpr uint32_t TPB = NW == 1 ? 4u : 1u;       // tiles per block
	const uint32_t sub = NW == 1 ? threadIdx.x / 64u : 0u;
	const uint32_t tid = NW == 1 ? threadIdx.x % 64u : threadIdx.x;
	const uint32_t tile_g = blockIdx.x * TPB + sub;
	if (tile_g >= a.n_tiles) return;  // whole wave (NW==1) or whole block (NW>1)

	const uint32_t frame = tile_g / a.tiles_per_frame;
	const uint32_t t = tile_g - frame * a.tiles_per_frame;
	const uint32_t ty = t / a.cols, tx = t - ty * a.cols;
	const uint32_t w = (tx == a.cols - 1) ? a.edge_w : a.bw;  // split.rs:18
	const uint32_t h = (ty == a.rows - 1) ? a.edge_h : a.bh;  // split.rs:19
	const uint32_t n = w * h;
	const uint32_t cls = (w != a.bw ? 1u : 0u) | (h != a.bh ? 2u : 0u);
	const uint8_t *src = a.src + (size_t)frame * a.frame_stride + (size_t)(ty * a.bh) * a.pitch + (size_t)(tx * a.bw) * C;

	const uint32_t rs = a.rs, PD = a.plane_dw;
	uint32_t *s_pl = lds + sub * a.tile_dw;
	uint32_t *s_tmp = s_pl + 4 * PD;
	uint16_t *pl16 = reinterpret_cast<uint16_t *>(s_pl);
	float *s_lab = reinterpret_cast<float *>(s_tmp);
	uint32_t *s_red = lds + TPB * a.tile_dw;  // 4*NW dwords, only carved (and used) when NW > 1
	(void)s_red;
	(void)s_lab;

	// ---- stage the tile: coalesced 16-B loads along image rows -> planar u16 pairs ----
	uint32_t alpha_and = 0xffu;
	uint4 raw[4];
	const uint32_t qpr = w >> 2, nquad = qpr * h;
	const bool vec = C == 4 && ((w & 3u) == 0) && ((reinterpret_cast<uintptr_t>(src) & 15u) == 0) &&
	                 ((a.pitch & 15u) == 0) && nquad <= 4u * TPT;
	// full 32x32 RGBA tile of the fast path: every index below is a shift/mask
	const bool fast = TW == 32 && NW == 1 && C == 4 && vec && w == 32 && h == 32 && rs == 16;
	if (fast) {
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const uint32_t row = (tid >> 3) + 8u * (uint32_t)k, col = tid & 7u;
			const uint4 v = *reinterpret_cast<const uint4 *>(src + (size_t)row * a.pitch + col * 16u);
			raw[k] = v;
			alpha_and &= (v.x & v.y & v.z & v.w) >> 24;
			uint32_t *d = s_pl + row * 16u + col * 2u;
#pragma unroll
			for (uint32_t c = 0; c < 4; ++c) {
				const uint32_t sel = c | 0x0c000c00u | ((4u + c) << 16);
				uint2 pr;
				pr.x = __builtin_amdgcn_perm(v.y, v.x, sel);
				pr.y = __builtin_amdgcn_perm(v.w, v.z, sel);
				*reinterpret_cast<uint2 *>(d + c * PD) = pr;
			}
		}
	} else if (vec) {
		RowWalker rw(tid, TPT, qpr);
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const uint32_t i = tid + (uint32_t)k * TPT;
			if (i < nquad) {
				const uint4 v = *reinterpret_cast<const uint4 *>(src + (size_t)rw.row * a.pitch + rw.col * 16u);
				raw[k] = v;
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
		if (fast) {
			// 16 lanes (pixel pairs) per row group, 4 groups of 8 window rows (the last one 6).
			// Per channel and row step: r = 1-2-1 along x (perm, add, mad), |hz| (sad), column
			// smoothing c = t(y)+t(y+1) (2 adds), the neighbour pair's c by DPP, |vr| (sad).
			const uint32_t q = tid & 15u, g = tid >> 4;
			const uint32_t y0 = g * 8u, steps = g == 3 ? 3u : 4u;  // two window rows per step
			const uint32_t *p = s_pl + y0 * 16u + q;
			const uint32_t two = 0x00020002u;
			uint32_t rA[3], rB[3], tP[3], dP[3];
#pragma unroll
			for (int c = 0; c < 3; ++c) {
				const uint32_t a0 = p[c * PD], a1 = p[c * PD + 1], b0 = p[c * PD + 16], b1 = p[c * PD + 17];
				rA[c] = pk_mad_u16(__builtin_amdgcn_alignbit(a1, a0, 16), two, u32(us2(a0) + us2(a1)));
				rB[c] = pk_mad_u16(__builtin_amdgcn_alignbit(b1, b0, 16), two, u32(us2(b0) + us2(b1)));
				tP[c] = u32(us2(a0) + us2(b0));
				dP[c] = b0;
			}
			p += 32;
			for (uint32_t st = 0; st < steps; ++st, p += 32) {
#pragma unroll
				for (int c = 0; c < 3; ++c) {
					const uint32_t n0 = p[c * PD], n1 = p[c * PD + 1], o0 = p[c * PD + 16], o1 = p[c * PD + 17];
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
			if (q == 15u) sum_hz = sum_vr = 0;  // pair 15 starts no window (x = 30, 31)
		} else if (w > 2 && h > 2) {
			const uint32_t WR = h - 2;
			const uint32_t VP = (w >> 1) - 1;  // pixel pairs that start a valid window pair (w even)
			uint32_t G = ((w & 1u) == 0 && VP >= 1) ? TPT / VP : 0u;
			if (G > WR) G = WR;
			if (G >= 1) {
				const uint32_t RG = (WR + G - 1) / G;  // window rows per lane group
				const uint32_t g = tid / VP, q = tid - g * VP;
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
			if (k < 3) {
				const float *plane = s_lab + k * n;
				for (uint32_t p = 0; p < n; ++p) s += plane[p];  // :60-62
				const float avg = __fdiv_rn(s, count);           // :65-67
				for (uint32_t p = 0; p < n; ++p) delta += fabsf(plane[p] - avg);  // :80-82
			} else {
				const uint16_t *al = pl16 + 6u * PD;
				for (uint32_t y = 0; y < h; ++y)
					for (uint32_t x = 0; x < w; ++x) s += __fdiv_rn((float)al[y * rs * 2u + x], 255.0f);  // :63
				const float avg = __fdiv_rn(s, count);
				for (uint32_t y = 0; y < h; ++y)
					for (uint32_t x = 0; x < w; ++x) delta += fabsf(__fdiv_rn((float)al[y * rs * 2u + x], 255.0f) - avg);
			}
		}
		float total;
		{
			const float d0 = __shfl(delta, 0, 64), d1 = __shfl(delta, 1, 64), d2 = __shfl(delta, 2, 64);
			total = d0 + d1 + d2;                      // :124
			if constexpr (C == 4) total = total + __shfl(delta, 3, 64);  // :89
		}
		float value = __fdiv_rn(total, count) * a.factor * 10.0f;  // pixlzr.rs:162
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
		if (vec && (a.slot_bytes & 15u) == 0 && (reinterpret_cast<uintptr_t>(a.out_px) & 15u) == 0) {
#pragma unroll
			for (int k = 0; k < 4; ++k) {
				const uint32_t i = tid + (uint32_t)k * TPT;
				if (i < nquad) reinterpret_cast<uint4 *>(dst)[i] = raw[k];
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

// ---------------------------------------------------------------------------
// finishing kernel: one lane per tile turns the detector result into the stored
// block value (and the raw detector outputs for pxz_lod_*).  All f64 work lives here.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) finish_kernel(const FinishArgs f)
{
	const uint32_t tile_g = blockIdx.x * 256u + threadIdx.x;
	if (tile_g >= f.n_tiles) return;
	const uint2 key = reinterpret_cast<const uint2 *>(f.sums)[tile_g];
	float raw0, raw1, v0, v1;
	if (f.mode == 1) {
		const uint32_t t = tile_g % f.tiles_per_frame;
		const uint32_t ty = t / f.cols, tx = t - ty * f.cols;
		const uint32_t w = (tx == f.cols - 1) ? f.edge_w : f.bw;
		const uint32_t h = (ty == f.rows - 1) ? f.edge_h : f.bh;
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
			v0 = parse_value(raw0 * f.factor);  // pixlzr.rs:199
			v1 = parse_value(raw1 * f.factor);
		}
	} else {
		raw0 = raw1 = __uint_as_float(key.x);
		v0 = v1 = parse_value(raw0);  // pixlzr.rs:177-178
	}
	if (f.value) f.value[tile_g] = hypot_f32(v0, v1);  // operations.rs:154
	if (f.lod0) f.lod0[tile_g] = raw0;
	if (f.lod1) f.lod1[tile_g] = raw1;
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
template <int NW, int C, int MODE, int TW>
static hipError_t launch_one(const ShrinkArgs &a, uint32_t lds_bytes, hipStream_t stream)
{
	constexpr uint32_t TPB = NW == 1 ? 4u : 1u;
	const uint32_t blocks = (a.n_tiles + TPB - 1) / TPB;
	const uint32_t threads = NW == 1 ? 256u : 64u * NW;
	auto kernel = shrink_kernel<NW, C, MODE, TW>;
	if (lds_bytes > 64u * 1024u) {
		hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
		                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
		if (e != hipSuccess) return e;
	}
	hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), lds_bytes, stream, a);
	return hipGetLastError();
}

template <int NW>
static hipError_t launch_nw(const ShrinkArgs &a, uint32_t channels, uint32_t lds_bytes, hipStream_t stream)
{
	if constexpr (NW == 1) {
		if (channels == 4 && a.bw == 32 && a.bh == 32)  // the headline geometry: compile-time fast path
			return a.mode == 1 ? launch_one<1, 4, 1, 32>(a, lds_bytes, stream) : launch_one<1, 4, 0, 32>(a, lds_bytes, stream);
	}
	if (channels == 4)
		return a.mode == 1 ? launch_one<NW, 4, 1, 0>(a, lds_bytes, stream) : launch_one<NW, 4, 0, 0>(a, lds_bytes, stream);
	return a.mode == 1 ? launch_one<NW, 3, 1, 0>(a, lds_bytes, stream) : launch_one<NW, 3, 0, 0>(a, lds_bytes, stream);
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

hipError_t launch_shrink(const ShrinkArgs &a, uint32_t channels, hipStream_t stream)
{
	const uint32_t nw = waves_per_tile(a.bw, a.bh);
	const uint32_t lds_bytes = a.tile_dw * 4u * (nw == 1 ? 4u : 1u) + (nw > 1 ? 16u * nw : 0u);
	switch (nw) {
	case 1: return launch_nw<1>(a, channels, lds_bytes, stream);
	case 2: return launch_nw<2>(a, channels, lds_bytes, stream);
	case 4: return launch_nw<4>(a, channels, lds_bytes, stream);
	case 8: return launch_nw<8>(a, channels, lds_bytes, stream);
	default: return launch_nw<16>(a, channels, lds_bytes, stream);
	}
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
