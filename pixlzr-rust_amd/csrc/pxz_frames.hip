// pxz_frames.hip -- whole-frame helpers: RGB <-> RGBA widening / slot narrowing for the RGBA fast paths, and the
// synthetic frame generator of bench and tests.
//
// Compiled with -ffp-contract=off: the f32 results of the Oklab detector are
// written into the bitstream, and the reference (Rust) never fuses a*b+c.
#include "pxz_device.h"

namespace pxz {


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
// tree::process (reference src/process/tree.rs:23-83), one level: a quarter wave per tile of the level's grid.
// A tile takes part if its parent went on to this level; it is pixelised here if (value >= threshold) ^ is_positive
// (:60) -- then the expand step that follows writes it -- and otherwise goes on to the next level (:70-79), or, when
// there is none, keeps its own pixels (the recursion's first lines, :34-36, return the block as it is).
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) tree_decide_kernel(const TreeArgs a)
{
	const uint32_t t = blockIdx.x * 16u + (threadIdx.x >> 4), ql = threadIdx.x & 15u;
	if (t >= a.n_tiles) return;
	const uint32_t frame = t / a.tiles_per_frame, tf = t - frame * a.tiles_per_frame;
	const uint32_t ty = tf / a.cols, tx = tf - ty * a.cols;
	bool active = true;
	if (a.parent_open) active = a.parent_open[(size_t)frame * a.parent_tiles_per_frame + (size_t)(ty >> 1) * a.parent_cols + (tx >> 1)] != 0;
	const float value = a.value[t];
	const bool pixelise = active && ((value >= a.threshold) != (a.positive != 0u));
	if (ql == 0) {
		if (!pixelise) {
			a.tile_w[t] = 0u;
			a.tile_h[t] = 0u;
		}
		a.open[t] = (active && !pixelise && !a.last) ? 1u : 0u;
	}
	if (!(active && !pixelise && a.last)) return;
	const uint32_t w = (tx == a.cols - 1) ? a.edge_w : a.bw, h = (ty == a.rows - 1) ? a.edge_h : a.bh;
	const uint8_t *src = a.src + (size_t)frame * a.src_frame_stride + (size_t)(ty * a.bh) * a.src_pitch + (size_t)(tx * a.bw) * a.channels;
	uint8_t *dst = a.dst + (size_t)frame * a.dst_frame_stride + (size_t)(ty * a.bh) * a.dst_pitch + (size_t)(tx * a.bw) * 4u;
	for (uint32_t i = ql; i < w * h; i += 16u) {
		const uint32_t y = i / w, x = i - y * w;
		const uint8_t *p = src + (size_t)y * a.src_pitch + (size_t)x * a.channels;
		const uint32_t px = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((a.channels == 4 ? (uint32_t)p[3] : 255u) << 24);
		*reinterpret_cast<uint32_t *>(dst + (size_t)y * a.dst_pitch + (size_t)x * 4u) = px;
	}
}

hipError_t launch_tree_decide(const TreeArgs &a, hipStream_t stream)
{
	hipLaunchKernelGGL(tree_decide_kernel, dim3((a.n_tiles + 15u) / 16u), dim3(256), 0, stream, a);
	return hipGetLastError();
}

hipError_t launch_synth(const SynthArgs &s, hipStream_t stream)
{
	dim3 grid((s.width + 255) / 256, s.height, s.n_frames);
	hipLaunchKernelGGL(synth_kernel, grid, dim3(256), 0, stream, s);
	return hipGetLastError();
}


}  // namespace pxz
