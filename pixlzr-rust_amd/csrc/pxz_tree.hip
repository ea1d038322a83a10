// pxz_tree.hip -- tree::process (reference src/process/tree.rs:23-83) on LISTS OF RECTANGLES: one level of the
// recursion per launch, one block per tile of that level, wherever the tile lies and whatever its size (up to 128x128,
// what src/bin/tree.rs:6 uses).  The per-level regular grids of pxz_tree_process_frames_device's first form only exist
// while every block is exactly two of the next; a 50-px block splits into 25 + 25, those into 12 + 12 + 1 (the
// recursion cuts every tile from its own corner, tree.rs:70-79), and a 128-px tile does not fit the LDS image of the
// fused shrink kernel.  Here a tile is:
//   1. measured    get_block_variance with |x - avg| and the identity (tree.rs:56-57, operations.rs:26-126): the block
//                  converts 1024 pixels at a time into LDS (the detector kernels' own conversion, pxz_oklab_math.h),
//                  four lanes add them up in pixel order; twice (sums, then deviations from the means);
//   2. decided     (value >= threshold) ^ is_positive (tree.rs:60);
//   3a. pixelised  reduce_image_section((v, v)) and the resize back (tree.rs:62-68): fir's two-pass convolution (or
//                  nearest) down, then up, every intermediate image in LDS, written as RGBA;
//   3b. or split   its (bw >> 1, bh >> 1) tiles are appended to the next level's list -- or, when that block is not
//                  above the minimum, the tile keeps its pixels (tree.rs:34-36).
// Not a fast path: work is proportional to the tiles that are still open, which is what the recursion does.
//
// Compiled with -ffp-contract=off (the detector's f32 arithmetic follows the reference's unfused operations).
#include "pxz_device.h"
#include "pxz_oklab_math.h"

namespace pxz {

constexpr uint32_t kTreeChunk = 1024;                      // pixels converted per round (four per thread)
constexpr uint32_t kTreeTables = 3072u + 256u + 2u * 128u;  // dwords: matrix-column products, alpha / 255, scale factors
// (tile sides up to 128: the LDS images below; the host refuses larger blocks)
// LDS after the tables: [X] 64 KB | [A] 32 KB | [S] 16 KB.  Detector: X = the chunk's four planes.  Resample: X = the tile
// (up to 128 x 128 x 4 bytes) and, at the end, the tile again; A = the first pass of the way down (h x nw, nw <= w / 2 or
// w = 1) and later of the way up (nh x w); S = the reduced tile (nh x nw).
constexpr uint32_t kTreeLdsBytes = kTreeTables * 4u + 65536u + 32768u + 16384u;

__device__ __forceinline__ uint8_t tree_mul_div_255(uint32_t a, uint32_t b)
{
	const uint32_t t = a * b + 128u;
	return (uint8_t)(((t >> 8) + t) >> 8);
}
__device__ __forceinline__ uint32_t tree_recip_alpha(uint32_t alpha) { return alpha == 0u ? 0u : ((255u * 512u) / alpha + 1u) >> 1; }
__device__ __forceinline__ uint8_t tree_div_and_clip(uint32_t v, uint32_t recip)
{
	const uint32_t r = (v * recip + 128u) >> 8;
	return (uint8_t)(r > 255u ? 255u : r);
}
__device__ __forceinline__ uint8_t tree_clip8(int32_t v, uint32_t precision)
{
	v >>= precision;
	return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// One pass of fir's convolution (or the nearest pick) along one axis of an image in LDS: `lines` lines of `in` samples
// of C bytes, sample stride sstep and line stride lstep on the way in; the result tightly packed the same way round
// (dst[line * out * C + o * C + c] for a horizontal pass, dst[o * lines * C + line * C + c] for a vertical one).
template <int C>
__device__ __forceinline__ void tree_axis_pass(const TreeRectArgs &a, const TreeAxisEntry &e, bool nearest, const uint8_t *src, uint32_t sstep,
                                               uint32_t lstep, uint8_t *dst, uint32_t dst_ostep, uint32_t dst_lstep, uint32_t lines, uint32_t tid)
{
	const uint32_t out = e.out;
	for (uint32_t i = tid; i < lines * out; i += 256u) {
		const uint32_t line = i / out, o = i - line * out;
		const uint8_t *s = src + line * lstep;
		uint8_t *d = dst + line * dst_lstep + o * dst_ostep;
		if (nearest) {
			const uint8_t *p = s + (uint32_t)a.starts[e.starts_off + o] * sstep;
#pragma unroll
			for (int c = 0; c < C; ++c) d[c] = p[c];
			continue;
		}
		const int32_t first = a.starts[e.starts_off + o], n = a.sizes[e.starts_off + o];
		const int16_t *k = a.coeffs + e.coeff_off + (size_t)o * e.window;
		int32_t acc[C];
#pragma unroll
		for (int c = 0; c < C; ++c) acc[c] = 1 << (e.precision - 1u);
		for (int32_t j = 0; j < n; ++j) {
			const uint8_t *p = s + (uint32_t)(first + j) * sstep;
			const int32_t kj = k[j];
#pragma unroll
			for (int c = 0; c < C; ++c) acc[c] += (int32_t)p[c] * kj;
		}
#pragma unroll
		for (int c = 0; c < C; ++c) d[c] = tree_clip8(acc[c], e.precision);
	}
}

// PixlzrBlock::resize (block.rs:273-334) of an image in LDS (tightly packed, C bytes per pixel) into another one: fir's
// premultiply -> horizontal pass -> vertical pass -> un-premultiply for U8x4, the two passes alone for U8x3, the nearest
// pick for FilterType::Nearest.  The source is overwritten (premultiplied in place); tmp holds the horizontal pass.
template <int C>
__device__ __forceinline__ void tree_resize(const TreeRectArgs &a, uint8_t *src, uint32_t w, uint32_t h, uint8_t *tmp, uint8_t *dst, uint32_t nw,
                                            uint32_t nh, uint32_t filter, uint32_t up, const TreeAxisEntry *s_ent, uint32_t tid)
{
	const bool nearest = filter == 0u;
	const TreeAxisEntry ex = s_ent[0], ey = s_ent[1];
	if (C == 4 && !nearest) {
		for (uint32_t i = tid; i < w * h; i += 256u) {
			uint8_t *p = src + i * 4u;
			const uint32_t al = p[3];
			p[0] = tree_mul_div_255(p[0], al);
			p[1] = tree_mul_div_255(p[1], al);
			p[2] = tree_mul_div_255(p[2], al);
		}
		__syncthreads();
	}
	const bool need_h = nw != w, need_v = nh != h;
	const uint8_t *cur = src;
	if (need_h) {
		uint8_t *o = need_v ? tmp : dst;
		tree_axis_pass<C>(a, ex, nearest, src, (uint32_t)C, w * (uint32_t)C, o, (uint32_t)C, nw * (uint32_t)C, h, tid);
		__syncthreads();
		cur = o;
	}
	if (need_v) {
		// lines = the nw * C byte columns' pixels: one "line" per column, samples a row apart
		tree_axis_pass<C>(a, ey, nearest, cur, nw * (uint32_t)C, (uint32_t)C, dst, nw * (uint32_t)C, (uint32_t)C, nw, tid);
		__syncthreads();
	}
	if (C == 4 && !nearest) {
		for (uint32_t i = tid; i < nw * nh; i += 256u) {
			uint8_t *p = dst + i * 4u;
			const uint32_t rc = tree_recip_alpha(p[3]);
			p[0] = tree_div_and_clip(p[0], rc);
			p[1] = tree_div_and_clip(p[1], rc);
			p[2] = tree_div_and_clip(p[2], rc);
		}
		__syncthreads();
	}
	(void)up;
}

template <int C>
__global__ void __launch_bounds__(256) tree_rect_kernel(const TreeRectArgs a)
{
	extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
	float4 *s_lms = reinterpret_cast<float4 *>(lds);
	float *s_alpha = reinterpret_cast<float *>(s_lms + 768);
	double *s_scale = reinterpret_cast<double *>(s_alpha + 256);
	uint8_t *s_x = reinterpret_cast<uint8_t *>(lds + kTreeTables);
	uint8_t *s_a = s_x + 65536u, *s_s = s_a + 32768u;
	float *s_plane = reinterpret_cast<float *>(s_x);  // [4][kTreeChunk]
	__shared__ float s_acc[4];
	__shared__ TreeAxisEntry s_ent[4];  // down x, down y, up x, up y
	__shared__ uint32_t s_found, s_base;
	const uint32_t tid = threadIdx.x;
	const TreeRect r = a.rects[blockIdx.x];
	const uint32_t w = r.w, h = r.h, n = w * h;
	const uint8_t *src = a.src + (size_t)r.frame * a.src_frame_stride + (size_t)r.y * a.src_pitch + (size_t)r.x * (uint32_t)C;
	uint8_t *dst = a.dst + (size_t)r.frame * a.dst_frame_stride + (size_t)r.y * a.dst_pitch + (size_t)r.x * 4u;
	oklab_fill_tables(s_lms, s_alpha, s_scale, tid);
	__syncthreads();

	// ---- 1. get_block_variance (operations.rs:26-126): two passes over the pixels in row-major order
	float mean = 0.0f;  // (lanes 0..3 of the first wave: the chains a, b, l, alpha)
	float value = 0.0f;
	const float count = (float)n;  // :51
	for (int pass = 0; pass < 2; ++pass) {
		float acc = 0.0f;
		for (uint32_t base = 0; base < n; base += kTreeChunk) {
			// four consecutive pixels per thread (two conversions of a pair); pixels past the end count as nothing
			const uint32_t first = base + tid * 4u;
			uint32_t px[4] = {0u, 0u, 0u, 0u};
#pragma unroll
			for (int j = 0; j < 4; ++j) {
				const uint32_t i = first + (uint32_t)j;
				if (i < n) {
					const uint32_t y = i / w, x = i - y * w;
					const uint8_t *p = src + (size_t)y * a.src_pitch + (size_t)x * (uint32_t)C;
					px[j] = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((C == 4 ? (uint32_t)p[3] : 255u) << 24);
				}
			}
#pragma unroll
			for (int j = 0; j < 4; j += 2) {
				float o0[3], o1[3];
				oklab_pair(px[j], px[j + 1], s_lms, s_scale, o0, o1);
				const uint32_t k = tid * 4u + (uint32_t)j;
#pragma unroll
				for (int c = 0; c < 3; ++c) {
					s_plane[c * kTreeChunk + k] = o0[c];
					s_plane[c * kTreeChunk + k + 1u] = o1[c];
				}
				s_plane[3 * kTreeChunk + k] = s_alpha[px[j] >> 24];
				s_plane[3 * kTreeChunk + k + 1u] = s_alpha[px[j + 1] >> 24];
			}
			__syncthreads();
			if (tid < 4u) {
				const uint32_t m = n - base < kTreeChunk ? n - base : kTreeChunk;
				const float *v = s_plane + tid * kTreeChunk;
				if (pass == 0) {
					for (uint32_t i = 0; i < m; ++i) acc += v[i];  // :60-63
				} else {
					for (uint32_t i = 0; i < m; ++i) acc += fabsf(v[i] - mean);  // :80-83 with pixlzr.rs:160-161
				}
			}
			__syncthreads();
		}
		if (tid < 4u) {
			if (pass == 0) mean = __fdiv_rn(acc, count);  // :65-68
			else s_acc[tid] = acc;
		}
	}
	__syncthreads();
	{
		const float total = C == 4 ? ((s_acc[0] + s_acc[1]) + s_acc[2]) + s_acc[3] : (s_acc[0] + s_acc[1]) + s_acc[2];  // :89 / :124
		value = __fdiv_rn(total, count);  // the identity closure: (x * 1) * 1
	}

	// ---- 2. tree.rs:60
	const bool pixelise = (value >= a.threshold) != (a.positive != 0u);
	if (!pixelise && !a.next_is_leaf) {
		// tree.rs:70-79: the tile's own grid of (bw >> 1, bh >> 1) tiles, cut from its corner (split.rs:10-27)
		const uint32_t cw = (w + a.next_bw - 1u) / a.next_bw, ch = (h + a.next_bh - 1u) / a.next_bh;
		if (tid == 0u) s_base = atomicAdd(a.next_count, cw * ch);
		__syncthreads();
		const uint32_t b0 = s_base;
		for (uint32_t i = tid; i < cw * ch; i += 256u) {
			if (b0 + i >= a.next_capacity) break;  // (the host sizes the list for every tile splitting; it checks the count)
			const uint32_t cy = i / cw, cx = i - cy * cw;
			TreeRect c;
			c.x = r.x + cx * a.next_bw;
			c.y = r.y + cy * a.next_bh;
			c.w = (uint16_t)(cx + 1u == cw ? w - cx * a.next_bw : a.next_bw);
			c.h = (uint16_t)(cy + 1u == ch ? h - cy * a.next_bh : a.next_bh);
			c.frame = r.frame;
			a.next_rects[b0 + i] = c;
		}
		return;
	}
	uint32_t nw = w, nh = h;
	if (pixelise) {
		// reduce_image_section((v, v)), operations.rs:140-156
		const uint32_t m = level_exponent(parse_value(value), a.thresholds);
		nw = reduced_size(w, m);
		nh = reduced_size(h, m);
	}
	if (nw == w && nh == h) {
		// nothing is reduced (block.rs:279-281, twice), or the tile keeps its pixels (tree.rs:34-36): as RGBA
		for (uint32_t i = tid; i < n; i += 256u) {
			const uint32_t y = i / w, x = i - y * w;
			const uint8_t *p = src + (size_t)y * a.src_pitch + (size_t)x * (uint32_t)C;
			uint8_t *d = dst + (size_t)y * a.dst_pitch + (size_t)x * 4u;
			d[0] = p[0];
			d[1] = p[1];
			d[2] = p[2];
			d[3] = C == 4 ? p[3] : 255u;
		}
		return;
	}
	// ---- 3a. the four axis tables of this tile: (w -> nw), (h -> nh) down, (nw -> w), (nh -> h) up
	if (tid == 0u) s_found = 0u;
	__syncthreads();
	for (uint32_t i = tid; i < a.n_dir; i += 256u) {
		const TreeAxisEntry e = a.dir[i];
		if (e.up == 0u && e.in == w && e.out == nw) { s_ent[0] = e; atomicOr(&s_found, 1u); }
		if (e.up == 0u && e.in == h && e.out == nh) { s_ent[1] = e; atomicOr(&s_found, 2u); }
		if (e.up == 1u && e.in == nw && e.out == w) { s_ent[2] = e; atomicOr(&s_found, 4u); }
		if (e.up == 1u && e.in == nh && e.out == h) { s_ent[3] = e; atomicOr(&s_found, 8u); }
	}
	__syncthreads();
	{
		const uint32_t need = (nw != w ? 5u : 0u) | (nh != h ? 10u : 0u);
		if ((s_found & need) != need) {
			// (cannot happen while the host's enumeration of size pairs and the kernel's child sizing agree; if it ever does, the
			// tile's region stays unwritten and the host turns the flag into PXZ_ERR_INTERNAL after the level's read-back)
			if (tid == 0u) atomicOr(a.next_count + 1, 1u);
			return;
		}
	}
	// the tile into LDS
	uint8_t *s_src = s_x;
	for (uint32_t i = tid; i < n; i += 256u) {
		const uint32_t y = i / w, x = i - y * w;
		const uint8_t *p = src + (size_t)y * a.src_pitch + (size_t)x * (uint32_t)C;
#pragma unroll
		for (int c = 0; c < C; ++c) s_src[i * (uint32_t)C + (uint32_t)c] = p[c];
	}
	__syncthreads();
	// down: X (w x h) -> A (h x nw) -> S (nh x nw)
	tree_resize<C>(a, s_src, w, h, s_a, s_s, nw, nh, a.filter_down, 0u, s_ent, tid);
	// up: S (nw x nh) -> A (nh x w) -> X (h x w)
	tree_resize<C>(a, s_s, nw, nh, s_a, s_x, w, h, a.filter_up, 1u, s_ent + 2, tid);
	for (uint32_t i = tid; i < n; i += 256u) {
		const uint32_t y = i / w, x = i - y * w;
		const uint8_t *p = s_x + i * (uint32_t)C;
		uint8_t *d = dst + (size_t)y * a.dst_pitch + (size_t)x * 4u;
		d[0] = p[0];
		d[1] = p[1];
		d[2] = p[2];
		d[3] = C == 4 ? p[3] : 255u;
	}
}

hipError_t launch_tree_rects(const TreeRectArgs &a, hipStream_t stream)
{
	if (a.n_rects == 0u) return hipSuccess;
	hipError_t e;
	auto go = [&](auto kernel) -> hipError_t {
		if ((e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kTreeLdsBytes)) != hipSuccess) return e;
		hipLaunchKernelGGL(kernel, dim3(a.n_rects), dim3(256), kTreeLdsBytes, stream, a);
		return hipGetLastError();
	};
	return a.channels == 4u ? go(tree_rect_kernel<4>) : go(tree_rect_kernel<3>);
}

}  // namespace pxz
