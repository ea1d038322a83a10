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
// the fused shrink kernel
// ---------------------------------------------------------------------------
// NW   waves cooperating on one tile (1: four independent tiles per 256-thread
//      block, no block barriers; >1: one tile per block of 64*NW threads)
// C    interleaved channels in HBM (3|4); LDS always holds RGBA dwords (A=255 for RGB)
// MODE 0 shrink_by (Oklab MAD), 1 shrink_directionally
template <int NW, int C, int MODE>
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
	const uint8_t *src = a.src + (size_t)frame * a.frame_stride + (size_t)(ty * a.bh) * a.pitch + (size_t)(tx * a.bw) * C;

	const uint32_t per_tile_dw = a.lds_src_dw + a.lds_tmp_dw + a.lds_lab_dw;
	uint32_t *s_src = lds + sub * per_tile_dw;
	uint32_t *s_tmp = s_src + a.lds_src_dw;
	float *s_lab = reinterpret_cast<float *>(s_tmp + a.lds_tmp_dw);
	uint32_t *s_red = lds + TPB * per_tile_dw;  // 4*NW dwords, only carved (and used) when NW > 1
	(void)s_red;

	// ---- stage the tile: coalesced 16-B loads along image rows -------------
	uint32_t alpha_and = 0xffu;
	if constexpr (C == 4) {
		const bool vec = ((w & 3u) == 0) && ((reinterpret_cast<uintptr_t>(src) & 15u) == 0) && ((a.pitch & 15u) == 0);
		if (vec) {
			const uint32_t qpr = w >> 2;
			RowWalker rw(tid, TPT, qpr);
			for (uint32_t i = tid; i < h * qpr; i += TPT, rw.next()) {
				uint4 v = *reinterpret_cast<const uint4 *>(src + (size_t)rw.row * a.pitch + rw.col * 16u);
				*reinterpret_cast<uint4 *>(s_src + rw.row * w + rw.col * 4u) = v;
				alpha_and &= (v.x & v.y & v.z & v.w) >> 24;
			}
		} else {
			RowWalker rw(tid, TPT, w);
			for (uint32_t i = tid; i < n; i += TPT, rw.next()) {
				const uint8_t *p = src + (size_t)rw.row * a.pitch + rw.col * 4u;
				uint32_t v = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
				s_src[i] = v;
				alpha_and &= v >> 24;
			}
		}
	} else {
		RowWalker rw(tid, TPT, w);
		for (uint32_t i = tid; i < n; i += TPT, rw.next()) {
			const uint8_t *p = src + (size_t)rw.row * a.pitch + rw.col * 3u;
			s_src[i] = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | 0xff000000u;
		}
	}
	tile_sync<NW>();

	// ---- level-of-detail value --------------------------------------------
	float v0, v1;      // parsed values (operations.rs:145)
	float raw0, raw1;  // detector outputs
	if constexpr (MODE == 1) {
		// get_block_variance_directionally, operations.rs:192-259: 3x3 Sobel-like
		// absolute gradient sums over the tile interior, channels R,G,B only.
		uint32_t sum_hz = 0, sum_vr = 0;
		if (w > 2 && h > 2) {
			const uint32_t ww = w - 2, hh = h - 2;
			RowWalker rw(tid, TPT, ww);
			for (uint32_t i = tid; i < ww * hh; i += TPT, rw.next()) {
				const uint32_t *r0 = s_src + rw.row * w + rw.col;
				const uint32_t *r1 = r0 + w, *r2 = r1 + w;
				const uint32_t p00 = r0[0], p01 = r0[1], p02 = r0[2];
				const uint32_t p10 = r1[0], p12 = r1[2];
				const uint32_t p20 = r2[0], p21 = r2[1], p22 = r2[2];
#pragma unroll
				for (int k = 0; k < 3; ++k) {
					const int sh = 8 * k;
					const int v00 = (p00 >> sh) & 255, v01 = (p01 >> sh) & 255, v02 = (p02 >> sh) & 255;
					const int v10 = (p10 >> sh) & 255, v12 = (p12 >> sh) & 255;
					const int v20 = (p20 >> sh) & 255, v21 = (p21 >> sh) & 255, v22 = (p22 >> sh) & 255;
					const int ghz = -v00 - 2 * v01 - v02 + v20 + 2 * v21 + v22;  // :240-241
					const int gvr = -v00 - 2 * v10 - v20 + v02 + 2 * v12 + v22;  // :244-245
					sum_hz += (uint32_t)(ghz < 0 ? -ghz : ghz);
					sum_vr += (uint32_t)(gvr < 0 ? -gvr : gvr);
				}
			}
		}
		sum_hz = wave_sum_u32(sum_hz);
		sum_vr = wave_sum_u32(sum_vr);
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
		const uint64_t fac = (uint64_t)(w - 2) * (uint64_t)(h - 2) * 4096ull;  // :253-254
		if (fac == 0) {
			// 0/0 on the reference's x86-64 target is the negative default NaN:
			// parse_value turns it into max(1+NaN, 0) = 0 -> 1x1, stored value 0
			raw0 = raw1 = __uint_as_float(0xFFC00000u);
			v0 = v1 = 0.0f;
		} else {
			const double f = (double)fac;
			raw0 = (float)((double)sum_hz / f);
			raw1 = (float)((double)sum_vr / f);
			v0 = parse_value(raw0 * a.factor);  // pixlzr.rs:199
			v1 = parse_value(raw1 * a.factor);
		}
	} else {
		// get_block_variance, operations.rs:26-126 with shrink_by's closures
		// (pixlzr.rs:160-162).  Colours are computed once, in parallel, into LDS
		// planes [a | b | l]; the two f32 accumulations are then replayed in the
		// reference's sequential pixel order, one lane per channel chain.
		for (uint32_t i = tid; i < n; i += TPT) {
			float L, A, B;
			oklab_from_rgba(s_src[i], L, A, B);
			s_lab[i] = A;
			s_lab[n + i] = B;
			s_lab[2 * n + i] = L;
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
				for (uint32_t p = 0; p < n; ++p) s += __fdiv_rn((float)(s_src[p] >> 24), 255.0f);  // :63
				const float avg = __fdiv_rn(s, count);
				for (uint32_t p = 0; p < n; ++p) delta += fabsf(__fdiv_rn((float)(s_src[p] >> 24), 255.0f) - avg);
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
		raw0 = raw1 = value;
		v0 = v1 = parse_value(value);  // pixlzr.rs:177-178: (value, value)
	}

	// ---- reduce_image_section: target size + stored value -------------------
	const uint32_t m0 = level_exponent(v0, a.thresholds);  // operations.rs:147
	const uint32_t m1 = level_exponent(v1, a.thresholds);  // :148
	const uint32_t nw = reduced_size(w, m0);               // :150
	const uint32_t nh = reduced_size(h, m1);               // :151
	if (tid == 0) {
		if (a.value) a.value[tile_g] = hypot_f32(v0, v1);  // :154
		if (a.out_w) a.out_w[tile_g] = nw;
		if (a.out_h) a.out_h[tile_g] = nh;
		if (a.lod0) a.lod0[tile_g] = raw0;
		if (a.lod1) a.lod1[tile_g] = raw1;
	}
	if (a.out_px == nullptr) return;

	// ---- PixlzrBlock::resize (block.rs:273-334) out of the LDS copy ----------
	uint8_t *dst = a.out_px + (size_t)tile_g * a.slot_bytes;
	if (nw == w && nh == h) {  // block.rs:279-281: clone
		if constexpr (C == 4) {
			if ((n & 3u) == 0 && (a.slot_bytes & 15u) == 0 && (reinterpret_cast<uintptr_t>(a.out_px) & 15u) == 0) {
				for (uint32_t i = tid; i < (n >> 2); i += TPT)
					reinterpret_cast<uint4 *>(dst)[i] = reinterpret_cast<const uint4 *>(s_src)[i];
			} else {
				for (uint32_t i = tid; i < n; i += TPT) store_pixel<4>(dst, i, s_src[i]);
			}
		} else {
			for (uint32_t i = tid; i < n; i += TPT) store_pixel<3>(dst, i, s_src[i]);
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
			store_pixel<C>(dst, i, s_src[y * w + x]);
		}
		return;
	}

	// ResizeAlg::Convolution, default ResizeOptions: U8x4 is alpha-premultiplied
	// first (identity when the whole tile is opaque, so skipped then)
	if constexpr (C == 4) {
		alpha_and = wave_and_u32(alpha_and);
		if constexpr (NW > 1) {
			const uint32_t wv = threadIdx.x / 64u;
			if ((threadIdx.x & 63u) == 0) s_red[2 * NW + wv] = alpha_and;
			__syncthreads();
#pragma unroll
			for (int q = 0; q < NW; ++q) alpha_and &= s_red[2 * NW + q];
		}
		if (alpha_and != 0xffu) {
			for (uint32_t i = tid; i < n; i += TPT) s_src[i] = premultiply(s_src[i]);
			tile_sync<NW>();
		}
	}

	const bool need_h = nw != w, need_v = nh != h;
	if (need_h) {  // horizontal pass: w -> nw, all h rows, u8 intermediate
		const uint16_t *bnd = a.bounds + tab_x.bounds_off;
		const int16_t *cf = a.coeffs + tab_x.coeff_off;
		const int prec = tab_x.precision;
		const int32_t init = 1 << (prec - 1);
		RowWalker rw(tid, TPT, nw);
		for (uint32_t i = tid; i < nw * h; i += TPT, rw.next()) {
			const uint32_t ox = rw.col, y = rw.row;
			const uint32_t first = bnd[2 * ox], taps = bnd[2 * ox + 1];
			const int16_t *k = cf + ox * tab_x.window;
			const uint32_t *row = s_src + y * w + first;
			int32_t a0 = init, a1 = init, a2 = init, a3 = init;
			for (uint32_t j = 0; j < taps; ++j) {
				const uint32_t p = row[j];
				const int32_t kk = k[j];
				a0 += (int32_t)(p & 255u) * kk;
				a1 += (int32_t)((p >> 8) & 255u) * kk;
				a2 += (int32_t)((p >> 16) & 255u) * kk;
				if constexpr (C == 4) a3 += (int32_t)(p >> 24) * kk;
			}
			uint32_t px = clip8(a0, prec) | (clip8(a1, prec) << 8) | (clip8(a2, prec) << 16);
			if constexpr (C == 4) px |= clip8(a3, prec) << 24; else px |= 0xff000000u;
			if (need_v) {
				s_tmp[i] = px;
			} else {
				if constexpr (C == 4) px = unpremultiply(px);
				store_pixel<C>(dst, i, px);
			}
		}
		if (!need_v) return;
		tile_sync<NW>();
	}
	{  // vertical pass: h -> nh over nw columns
		const uint32_t *in = need_h ? s_tmp : s_src;
		const uint16_t *bnd = a.bounds + tab_y.bounds_off;
		const int16_t *cf = a.coeffs + tab_y.coeff_off;
		const int prec = tab_y.precision;
		const int32_t init = 1 << (prec - 1);
		RowWalker rw(tid, TPT, nw);
		for (uint32_t i = tid; i < nw * nh; i += TPT, rw.next()) {
			const uint32_t ox = rw.col, oy = rw.row;
			const uint32_t first = bnd[2 * oy], taps = bnd[2 * oy + 1];
			const int16_t *k = cf + oy * tab_y.window;
			const uint32_t *col = in + first * nw + ox;
			int32_t a0 = init, a1 = init, a2 = init, a3 = init;
			for (uint32_t j = 0; j < taps; ++j) {
				const uint32_t p = col[j * nw];
				const int32_t kk = k[j];
				a0 += (int32_t)(p & 255u) * kk;
				a1 += (int32_t)((p >> 8) & 255u) * kk;
				a2 += (int32_t)((p >> 16) & 255u) * kk;
				if constexpr (C == 4) a3 += (int32_t)(p >> 24) * kk;
			}
			uint32_t px = clip8(a0, prec) | (clip8(a1, prec) << 8) | (clip8(a2, prec) << 16);
			if constexpr (C == 4) {
				px |= clip8(a3, prec) << 24;
				px = unpremultiply(px);
			}
			store_pixel<C>(dst, i, px);
		}
	}
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
static hipError_t launch_one(const ShrinkArgs &a, uint32_t lds_bytes, hipStream_t stream)
{
	constexpr uint32_t TPB = NW == 1 ? 4u : 1u;
	const uint32_t blocks = (a.n_tiles + TPB - 1) / TPB;
	const uint32_t threads = NW == 1 ? 256u : 64u * NW;
	auto kernel = shrink_kernel<NW, C, MODE>;
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
	if (channels == 4)
		return a.mode == 1 ? launch_one<NW, 4, 1>(a, lds_bytes, stream) : launch_one<NW, 4, 0>(a, lds_bytes, stream);
	return a.mode == 1 ? launch_one<NW, 3, 1>(a, lds_bytes, stream) : launch_one<NW, 3, 0>(a, lds_bytes, stream);
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
	const uint32_t per_tile = (a.lds_src_dw + a.lds_tmp_dw + a.lds_lab_dw) * 4u;
	const uint32_t lds_bytes = per_tile * (nw == 1 ? 4u : 1u) + (nw > 1 ? 16u * nw : 0u);
	switch (nw) {
	case 1: return launch_nw<1>(a, channels, lds_bytes, stream);
	case 2: return launch_nw<2>(a, channels, lds_bytes, stream);
	case 4: return launch_nw<4>(a, channels, lds_bytes, stream);
	case 8: return launch_nw<8>(a, channels, lds_bytes, stream);
	default: return launch_nw<16>(a, channels, lds_bytes, stream);
	}
}

hipError_t launch_synth(const SynthArgs &s, hipStream_t stream)
{
	dim3 grid((s.width + 255) / 256, s.height, s.n_frames);
	hipLaunchKernelGGL(synth_kernel, grid, dim3(256), 0, stream, s);
	return hipGetLastError();
}

}  // namespace pxz
