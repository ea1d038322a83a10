// pxz_stream.hip -- the byte-stream kernels: tile compaction (pack_*), the device .pixlzr writer (qoi_* encode kernels:
// binning, one QOI stream per lane, splice, headers) and reader (pixlzr_index_kernel, qoi_decode_kernel).
//
// Compiled with -ffp-contract=off: the f32 results of the Oklab detector are
// written into the bitstream, and the reference (Rust) never fuses a*b+c.
#include "pxz_device.h"

namespace pxz {


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
	// (indices clamped, loads unconditional and all sixteen in one block: behind a condition each sat in a block of its own and was
	// waited for before the next was issued)
	uint32_t sz[16], run = 0;
	if (a.sizes) {
#pragma unroll
		for (uint32_t i = 0; i < 16; ++i) sz[i] = a.sizes[base + i < a.n_tiles ? base + i : a.n_tiles - 1u];
	} else {
#pragma unroll
		for (uint32_t i = 0; i < 16; ++i) {
			const uint32_t tc = base + i < a.n_tiles ? base + i : a.n_tiles - 1u;
			sz[i] = a.w[tc] * a.h[tc] * a.channels;
		}
	}
#pragma unroll
	for (uint32_t i = 0; i < 16; ++i) {
		sz[i] = base + i < a.n_tiles ? sz[i] : 0u;
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
// QOI is sequential per tile (previous pixel, run length, 64-entry index) and a lane's pixel loop is one long
// dependent chain (~900 cycles per pixel), so the longest tile of a launch used to set its duration: 1024 pixels on
// one lane.  Round 2: a tile of n pixels is cut into G = 2^(class - 6) SEGMENTS (class = floor(log2 n); 64..128 pixels
// each, G = 1 below 128 pixels) that consecutive lanes encode side by side.  What a segment needs from the pixels in
// front of it is reconstructed exactly:
//   previous pixel     the pixel before the segment;
//   index table        a DRY pass over every segment writes its non-repeating pixels into the lane's table and notes the
//                      written slots; then lane s, for slot s, walks the lanes of each tile in order and hands every
//                      segment the last value written to that slot before it (the crate stores a pixel in the index
//                      whenever it is not a repeat of its predecessor, hit or not);
//   pending run        (pixels repeated so far, modulo the flush at 62) and "an op was written before" (the crate's
//                      run-of-one INDEX quirk): a segment that is one repeat from end to end passes the count through,
//                      any other ends with its trailing repeats -- a fixed point reached in as many rounds as there
//                      are all-repeat segments in a row.
// A pending run is written by the segment in which it ends, the end marker by the segment holding the last pixel.
// Tiles are binned by class (counting sort, largest first) so that a wave holds segments of one class; the lanes' tables
// live in LDS as table[lane][slot] (stride 65: conflict-free for the per-slot walk, random for the pixel loops).
// What a segment writes (its PIECE of the tile's record): segment 0 starts with "block" | f32 BE value | u32 BE len | w,h BE |
// channels | 0, every segment has its ops, the last one ends with 0x00*7 0x01.  The pieces go into the wave's unit of the
// scratch (qoi_class_rows below).  Then: scan of the record lengths, splice of the units into the final files (every piece behind
// the one before it), header + per-row length table.
// ---------------------------------------------------------------------------
constexpr uint32_t kBinCounts = 0, kBinCursor = 32, kBinUnits = 64, kBinTiles = 96, kBinTotal = 128, kBinArrive = 129;  // u32 offsets in bins[]
constexpr uint32_t kBinBase = 130;   // 32 x u64 (as dword pairs): where the units of a class start in the scratch (encode side)
constexpr uint32_t kBinN = 194;      // 32: tiles per class, for the kernels behind the binning (kBinCounts are the accumulators: the scan zeroes them again)
constexpr uint32_t kBinDwords = 256;  // what a caller allocates (and zeroes once) for bins[]

__device__ __host__ __forceinline__ uint32_t qoi_class_segments(uint32_t cls) { return cls <= 6u ? 1u : (cls >= 12u ? 64u : 1u << (cls - 6u)); }
// pixels per segment (a multiple of 4: the RGBA loads stay 16-byte aligned, the RGB ones dword aligned)
__device__ __host__ __forceinline__ uint32_t qoi_segment_pixels(uint32_t n, uint32_t g) { return ((n + g - 1u) / g + 3u) & ~3u; }
// scratch record: bytes of piece 0 (header + segment 0) and of every further piece (8-byte length header + segment)
__device__ __host__ __forceinline__ uint32_t qoi_piece0_bytes(uint32_t seg_px, uint32_t c) { return (23u + seg_px * (c + 1u) + 8u + 8u + 7u) & ~7u; }
__device__ __host__ __forceinline__ uint32_t qoi_piece_bytes(uint32_t seg_px, uint32_t c) { return (8u + seg_px * (c + 1u) + 8u + 8u + 7u) & ~7u; }
// Scratch of the encoder (round 3): a UNIT is the 64 segments one wave encodes side by side, stored as rows of 512 bytes --
// qword k of lane l at (64 k + l) * 8.  Row 0 holds the 64 lengths, the rows behind it the segments' bytes, as many as the longest
// piece of a tile of the class can need (a record header and the largest segment).  The units of a class lie back to back, the
// classes in descending order (bins[kBinBase ..]).  What the lanes of a wave store together is then one run of 512 bytes, and what
// the splice reads is whole rows -- with the lanes' pieces each in a place of its own every store touched 64 cache lines, which
// was 0.07 ms of the encoder's 0.27.
__device__ __host__ __forceinline__ uint32_t qoi_class_rows(uint32_t c, uint32_t slot_px, uint32_t channels)
{
	const uint32_t g = qoi_class_segments(c);
	uint32_t n_max = c >= 31u ? slot_px : (2u << c) - 1u;  // the class holds 2^c .. 2^(c+1) - 1 pixels
	if (n_max > slot_px) n_max = slot_px;
	return 1u + qoi_piece0_bytes(qoi_segment_pixels(n_max ? n_max : 1u, g), channels) / 8u;
}

// Binning (counting sort by class, largest first): two launches of ceil(n / 4096) blocks, 16 tiles per thread.
//   qoi_bin_count_kernel    per-block histogram in LDS, one global atomic per non-empty class and block; the block that
//                           arrives last turns the counts into the classes' places in the permutation of tiles and in the
//                           run of segments ("units": a class starts a fresh wave) -- round 3: this was a launch of one thread
//   qoi_bin_scatter_kernel  the block reserves one range per class (one global atomic each), threads take slots inside it
// (Round 2 ran one tile per thread: a thousand blocks whose atomics queued on a handful of addresses, 13 us each for 2 MB of
// sizes.)  bins[kBinCounts ..] must be zero at the start: the scan leaves them so (the kernels behind it read the copy at kBinN).
constexpr uint32_t kBinChunk = 4096;  // tiles per block: 256 threads x 16
__device__ __forceinline__ uint32_t qoi_tile_class(const QoiArgs &a, uint32_t t) { return 31u - (uint32_t)__builtin_clz((a.w[t] * a.h[t]) | 1u); }

__global__ void __launch_bounds__(256) qoi_bin_count_kernel(const QoiArgs a)
{
	__shared__ uint32_t s_hist[32];
	__shared__ uint32_t s_last;
	if (threadIdx.x < 32) s_hist[threadIdx.x] = 0;
	__syncthreads();
	{
		// (the sizes of the block's tiles first -- indices clamped, loads unconditional, all in flight together --, then the counting)
		uint32_t cls[kBinChunk / 256u];
#pragma unroll
		for (uint32_t i = 0; i < kBinChunk / 256u; ++i) {
			const uint32_t t = blockIdx.x * kBinChunk + i * 256u + threadIdx.x;
			cls[i] = qoi_tile_class(a, t < a.n_tiles ? t : a.n_tiles - 1u);
		}
#pragma unroll
		for (uint32_t i = 0; i < kBinChunk / 256u; ++i) {
			const uint32_t t = blockIdx.x * kBinChunk + i * 256u + threadIdx.x;
			if (t < a.n_tiles) atomicAdd(&s_hist[cls[i]], 1u);
		}
	}
	__syncthreads();
	if (threadIdx.x < 32) {
		if (s_hist[threadIdx.x]) atomicAdd(&a.bins[kBinCounts + threadIdx.x], s_hist[threadIdx.x]);
		__threadfence();  // the counts before the arrival
	}
	__syncthreads();
	if (threadIdx.x == 0) s_last = atomicAdd(&a.bins[kBinArrive], 1u) == gridDim.x - 1u ? 1u : 0u;
	__syncthreads();
	if (!s_last) return;
	// the last block: every count is in (device-scope atomics, read as such)
	if (threadIdx.x < 32) s_hist[threadIdx.x] = __hip_atomic_load(&a.bins[kBinCounts + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	__syncthreads();
	// a lane per class: what lies in front of it (the larger classes) in the permutation of tiles, in the run of units and --
	// the encoder -- in the scratch
	__shared__ uint32_t s_units[32];
	__shared__ unsigned long long s_bytes[32];
	if (threadIdx.x < 32) {
		const uint32_t c = threadIdx.x, units = (s_hist[c] * qoi_class_segments(c) + 63u) & ~63u;
		s_units[c] = units;
		s_bytes[c] = a.slot_bytes ? (unsigned long long)(units / 64u) * 512ull * qoi_class_rows(c, a.slot_bytes / a.channels, a.channels) : 0ull;
	}
	__syncthreads();
	if (threadIdx.x < 32) {
		const uint32_t c = threadIdx.x;
		uint32_t tiles = 0, units = 0;
		unsigned long long base = 0;
		for (uint32_t o = c + 1u; o < 32u; ++o) {
			tiles += s_hist[o];
			units += s_units[o];
			base += s_bytes[o];
		}
		a.bins[kBinN + c] = s_hist[c];
		a.bins[kBinCounts + c] = 0u;  // (every block has added its share: ready for the next launch)
		a.bins[kBinCursor + c] = tiles;
		a.bins[kBinTiles + c] = tiles;
		a.bins[kBinUnits + c] = units;
		a.bins[kBinBase + 2u * c] = (uint32_t)base;
		a.bins[kBinBase + 2u * c + 1u] = (uint32_t)(base >> 32);
		if (c == 0u) {
			a.bins[kBinTotal] = units + s_units[0];
			a.bins[kBinArrive] = 0u;  // for the next launch
		}
	}
}

__global__ void __launch_bounds__(256) qoi_bin_scatter_kernel(const QoiArgs a)
{
	__shared__ uint32_t s_hist[32], s_base[32];
	if (threadIdx.x < 32) s_hist[threadIdx.x] = 0;
	__syncthreads();
	uint32_t where[kBinChunk / 256u];  // class | place inside the block's share of the class << 5
#pragma unroll
	for (uint32_t i = 0; i < kBinChunk / 256u; ++i) {  // (loads first, as in the counting kernel)
		const uint32_t t = blockIdx.x * kBinChunk + i * 256u + threadIdx.x;
		where[i] = qoi_tile_class(a, t < a.n_tiles ? t : a.n_tiles - 1u);
	}
#pragma unroll
	for (uint32_t i = 0; i < kBinChunk / 256u; ++i) {
		const uint32_t t = blockIdx.x * kBinChunk + i * 256u + threadIdx.x;
		if (t < a.n_tiles) where[i] |= atomicAdd(&s_hist[where[i]], 1u) << 5;
	}
	__syncthreads();
	if (threadIdx.x < 32 && s_hist[threadIdx.x]) s_base[threadIdx.x] = atomicAdd(&a.bins[kBinCursor + threadIdx.x], s_hist[threadIdx.x]);
	__syncthreads();
#pragma unroll
	for (uint32_t i = 0; i < kBinChunk / 256u; ++i) {
		const uint32_t t = blockIdx.x * kBinChunk + i * 256u + threadIdx.x;
		if (t < a.n_tiles) a.perm[s_base[where[i] & 31u] + (where[i] >> 5)] = t;
	}
}

typedef __attribute__((address_space(1))) unsigned long long *global_qword_ptr;
typedef uint32_t u32q_a1 __attribute__((ext_vector_type(4), aligned(1)));
struct ByteSink {
	unsigned long long acc;
	uint32_t cnt;        // bytes in acc (0..7)
	uint32_t rows;       // qwords written
	// (a pointer the compiler knows to be global: through a generic one the stores are flat_store, which also count as LDS
	// operations -- every wait for an index read then waited for the stores in flight as well)
	global_qword_ptr out;  // this lane's qword of the unit's next row (rows are 64 qwords)
	// n <= 8 bytes at once, first byte in the low bits of v (bits above 8n must be zero)
	__device__ __forceinline__ void append(unsigned long long v, uint32_t n)
	{
		acc |= v << (8u * cnt);
		const uint32_t total = cnt + n;
		if (total >= 8u) {
			*out = acc;
			out += 64;
			++rows;
			acc = cnt ? v >> (64u - 8u * cnt) : 0ull;  // the bytes that did not fit
			cnt = total - 8u;
		} else {
			cnt = total;
		}
	}
};

// which tile and which of its segments lane `lane` of the wave at unit0 has: the class whose run of units holds unit0 (classes in
// descending order, runs padded to whole waves; lane c looks at class c: one round trip to the counters instead of a walk), then
// the tile by its rank in the class.  The same in the encoder and in the splice.
struct QoiLane {
	uint32_t cls, G, rank, seg, t;
	bool live;
	uint8_t *unit;  // the wave's unit in the scratch
};
__device__ __forceinline__ QoiLane qoi_lane(const QoiArgs &a, uint32_t unit0, uint32_t lane)
{
	QoiLane q;
	{
		const uint32_t c = lane & 31u;
		const uint32_t ub = a.bins[kBinUnits + c], nu = (a.bins[kBinN + c] * qoi_class_segments(c) + 63u) & ~63u;
		const unsigned long long mine = __ballot(lane < 32u && unit0 >= ub && unit0 < ub + nu);
		q.cls = (uint32_t)__builtin_ctzll(mine | (1ull << 63));
	}
	q.G = qoi_class_segments(q.cls);  // lanes per tile (a power of two)
	const uint32_t first = a.bins[kBinUnits + q.cls], rel = unit0 - first + lane;
	q.rank = rel / q.G;
	q.seg = rel & (q.G - 1u);
	q.live = q.rank < a.bins[kBinN + q.cls];
	q.t = q.live ? a.perm[a.bins[kBinTiles + q.cls] + q.rank] : 0u;
	const unsigned long long base = (unsigned long long)a.bins[kBinBase + 2u * q.cls] | ((unsigned long long)a.bins[kBinBase + 2u * q.cls + 1u] << 32);
	q.unit = a.scratch + base + (size_t)((unit0 - first) / 64u) * (512u * qoi_class_rows(q.cls, a.slot_bytes / a.channels, a.channels));
	return q;
}

__device__ __forceinline__ uint32_t qoi_hash(uint32_t px)
{
	return __builtin_amdgcn_udot4(px, 0x0b070503u, 0u, false) & 63u;  // r*3 + g*5 + b*7 + a*11: one v_dot4_u32_u8
}

// pixel i of a tile's slot (RGB: alpha 255 added)
template <int C>
__device__ __forceinline__ uint32_t qoi_pixel(const uint8_t *src, uint32_t i)
{
	if constexpr (C == 4) return reinterpret_cast<const uint32_t *>(src)[i];
	const uint8_t *p = src + (size_t)i * 3u;
	return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | 0xff000000u;
}

// four pixels from pixel `base` on (base a multiple of 4); may run past the tile's pixels, never past its slot of
// `slot_px` pixels (RGBA slots are whole pixel quads)
template <int C>
__device__ __forceinline__ void qoi_load4(const uint8_t *src, uint32_t base, uint32_t slot_px, uint32_t (&px4)[4])
{
	if (C == 3 && base + 4u > slot_px) {
		for (int j = 0; j < 4; ++j) px4[j] = base + (uint32_t)j < slot_px ? qoi_pixel<3>(src, base + (uint32_t)j) : 0u;
		return;
	}
	if constexpr (C == 4) {
		const uint4 v = *reinterpret_cast<const uint4 *>(src + (size_t)base * 4u);
		px4[0] = v.x; px4[1] = v.y; px4[2] = v.z; px4[3] = v.w;
	} else {
		// twelve bytes as three dwords at whatever address the slot has (gfx950 loads them from any byte address; the byte-wise
		// form for slots that are not dword multiples was twelve loads at each of the encoder's call sites)
		typedef uint32_t u32_a1 __attribute__((aligned(1)));
		const u32_a1 *p = reinterpret_cast<const u32_a1 *>(src + (size_t)base * 3u);
		const uint32_t d0 = p[0], d1 = p[1], d2 = p[2];
		px4[0] = d0 & 0xffffffu;
		px4[1] = (d0 >> 24) | ((d1 & 0xffffu) << 8);
		px4[2] = (d1 >> 16) | ((d2 & 0xffu) << 16);
		px4[3] = d2 >> 8;
#pragma unroll
		for (int j = 0; j < 4; ++j) px4[j] |= 0xff000000u;
	}
}

// sixteen pixels from pixel `base` on, all inside the tile (base + 16 <= its pixels).  RGB: the 48 bytes as three 16-byte
// loads from whatever byte address they have (three vector memory instructions instead of twelve dwords), cut into pixels
// with byte shifts.
template <int C>
__device__ __forceinline__ void qoi_load16(const uint8_t *src, uint32_t base, uint32_t slot_px, uint32_t (&px)[16])
{
	if constexpr (C == 4) {
#pragma unroll
		for (int q = 0; q < 4; ++q) {
			uint32_t t4[4];
			qoi_load4<4>(src, base + 4u * (uint32_t)q, slot_px, t4);
#pragma unroll
			for (int j = 0; j < 4; ++j) px[4 * q + j] = t4[j];
		}
	} else {
		typedef uint32_t u32q_a1 __attribute__((ext_vector_type(4), aligned(1)));
		const u32q_a1 *p = reinterpret_cast<const u32q_a1 *>(src + (size_t)base * 3u);
		const u32q_a1 v0 = p[0], v1 = p[1], v2 = p[2];
		const uint32_t d[13] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w, 0u};
#pragma unroll
		for (int j = 0; j < 16; ++j) {
			const int k = (3 * j) >> 2, sh = (3 * j) & 3;
			const uint32_t v = sh == 0 ? d[k] : __builtin_amdgcn_alignbyte(d[k + 1], d[k], (uint32_t)sh);
			px[j] = v | 0xff000000u;
		}
	}
}

// One wave per block: the 16 KB index table of a wave is what limits residency (nine waves per CU), and the lanes'
// pixel loops are latency chains that only other waves can hide.
constexpr uint32_t kQoiWaves = 1;  // waves per block of the decoder (one 16 KB table per wave)
constexpr uint32_t kQoiRow = 65;  // table[lane][slot] row stride (dwords / bytes)
template <int C>
__global__ void __launch_bounds__(64) qoi_tiles_kernel(const QoiArgs a)
{
	__shared__ uint32_t s_index[64 * kQoiRow];
	__shared__ uint32_t s_written[64][2];  // per lane: which slots its dry pass wrote (bit s of the 64-bit mask)
	const uint32_t lane = threadIdx.x;
	const uint32_t unit0 = blockIdx.x * 64u;
	if (unit0 >= a.bins[kBinTotal]) return;  // (the grid covers the worst case: every tile in the largest class)
	const QoiLane ql_ = qoi_lane(a, unit0, lane);
	const uint32_t G = ql_.G, seg = ql_.seg, t = ql_.t;
	const bool live = ql_.live;
	const uint32_t w = live ? a.w[t] : 0u, h = live ? a.h[t] : 0u, n = w * h;
	const uint32_t seg_px = qoi_segment_pixels(n ? n : 1u, G);  // (the same value the splice kernel derives from n)
	const uint32_t start = seg * seg_px, end = start + seg_px < n ? start + seg_px : n;
	const uint32_t len = start < n ? end - start : 0u;
	const uint8_t *src = a.slots + (size_t)t * a.slot_bytes;
	const uint32_t slot_px = a.slot_bytes / (uint32_t)C;
	uint32_t *index = s_index + lane * kQoiRow;
#pragma unroll 8
	for (int sidx = 0; sidx < 64; ++sidx) index[sidx] = 0u;  // qoi: the index starts as zero pixels
	uint32_t wr_lo = 0u, wr_hi = 0u;
	const uint32_t first_prev = (len && start) ? qoi_pixel<C>(src, start - 1u) : 0xff000000u;  // qoi: previous pixel starts as opaque black

	// ---- dry pass (G > 1): what this segment leaves in the index, its trailing repeats, whether it is all repeats
	uint32_t trail = 0;
	bool all_same = true;
	if (G > 1u) {
		uint32_t prev = first_prev;
		auto dry = [&](const uint32_t px) __attribute__((always_inline)) {
			if (px != prev) {
				const uint32_t slot = qoi_hash(px);
				index[slot] = px;
				const uint32_t bit = 1u << (slot & 31u);
				wr_lo |= slot < 32u ? bit : 0u;
				wr_hi |= slot < 32u ? 0u : bit;
				trail = 0;
				all_same = false;
			} else {
				++trail;
			}
			prev = px;
		};
		// (sixteen pixels per round with the next sixteen on their way, as in the real pass below: four at a time without
		// prefetch this loop was one memory round trip per four pixels)
		uint32_t base = start;
		if (base + 16u <= end) {
			uint32_t ahead[16];
			auto load16 = [&](uint32_t from) __attribute__((always_inline)) { qoi_load16<C>(src, from, slot_px, ahead); };
			load16(base);
			for (; base + 16u <= end; base += 16u) {
				uint32_t cur[16];
#pragma unroll
				for (int j = 0; j < 16; ++j) cur[j] = ahead[j];
				if (base + 32u <= end) load16(base + 16u);
#pragma unroll
				for (int j = 0; j < 16; ++j) dry(cur[j]);
			}
		}
		for (; base < end; base += 4u) {
			uint32_t px4[4];
			qoi_load4<C>(src, base, slot_px, px4);
#pragma unroll
			for (int j = 0; j < 4; ++j) {
				if (base + (uint32_t)j >= end) break;
				dry(px4[j]);
			}
		}
	}
	s_written[lane][0] = wr_lo;
	s_written[lane][1] = wr_hi;
	__syncthreads();
	// ---- the index every segment starts from: lane s walks slot s over the lanes in order
	if (G > 1u) {
		uint32_t cur = 0u;
		const uint32_t half = lane >> 5, bit = lane & 31u;
#pragma unroll 8
		for (uint32_t j = 0; j < 64u; ++j) {
			if ((j & (G - 1u)) == 0u) cur = 0u;  // a new tile: zero pixels
			const uint32_t v = s_index[j * kQoiRow + lane];
			const uint32_t f = (s_written[j][half] >> bit) & 1u;
			s_index[j * kQoiRow + lane] = cur;
			cur = f ? v : cur;
		}
	}
	__syncthreads();
	// ---- pending run and "an op was written" at the start of every segment
	uint32_t run_in = 0;
	bool seen_in = false;
	if (G > 1u) {
		uint32_t run_out = all_same ? len % 62u : trail % 62u;
		for (uint32_t round = 0; round < G; ++round) {
			const uint32_t left = __shfl_up(run_out, 1, 64);
			run_in = seg ? left : 0u;
			const uint32_t next = all_same ? (run_in + len) % 62u : trail % 62u;
			const bool changed = next != run_out;
			run_out = next;
			if (!__any(changed)) break;
		}
		const unsigned long long ops = __ballot(!all_same);  // segments that write at least one op
		const uint32_t first_lane = lane - seg;
		const unsigned long long before = (ops >> first_lane) & ((1ull << seg) - 1ull);
		seen_in = before != 0ull;
	}
	// row 0 of the unit: the 64 lengths (lanes without a segment: 0)
	uint32_t *my_len = reinterpret_cast<uint32_t *>(ql_.unit) + 2u * lane;
	if (!live) {
		*my_len = 0u;
		return;
	}

	// ---- the segment's ops, into this lane's qwords of the unit's rows 1, 2, ...
	ByteSink s{0ull, 0u, 0u, (global_qword_ptr)ql_.unit + 64 + lane};
	if (seg == 0u) {
		// encode_block: magic, value, length placeholder (mod.rs:172-178,195), then the qoi header minus its 4-byte magic
		// (mod.rs:191): width, height BE, channels, colourspace 0 -- 23 bytes
		const uint32_t vb = __float_as_uint(a.value[t]);
		auto be = [](uint32_t v) -> unsigned long long { return (unsigned long long)__builtin_bswap32(v); };  // BE bytes, first one lowest
		s.append(0x6b636f6c62ull | (be(vb) << 40), 8);                   // "block", value bytes 0..2
		s.append((be(vb) >> 24) | (be(w) << 40), 8);                      // value byte 3, four zero bytes (length, patched below), w bytes 0..2
		s.append((be(w) >> 24) | (be(h) << 8) | ((unsigned long long)C << 40), 7);  // w byte 3, h, channels, colourspace 0
	}
	uint32_t prev = first_prev, run = run_in, last_slot = qoi_hash(first_prev);
	bool seen_op = seen_in;
	// One pixel (px = pixel pi of the tile):
	auto step = [&](const uint32_t px, const uint32_t pi, const uint32_t slot, const bool hit) __attribute__((always_inline)) {
		// One pixel, without branches: the lanes of a wave sit in different ops at every pixel, and a wave that
		// takes every branch in turn spends its time in the ones its lanes did not want.  Every candidate op is
		// worked out, selects pick the bytes (a pending run byte first), one append writes them.
		const bool same = px == prev;
		// (a) the pixel repeats: count it; the run is written at 62 or at the end of the tile
		const uint32_t run_if_same = run + 1u;
		const bool flush_same = run_if_same == 62u || pi + 1u == n;
		// (b) it differs: a pending run first -- as INDEX of the repeated pixel when it is a run of ONE and any op was
		// written before (the crate's quirk), else as RUN
		const uint32_t pre_byte = (run == 1u && seen_op) ? last_slot : (0xc0u | (run - 1u));
		const uint32_t pre_len = run ? 1u : 0u;
		const uint32_t dr = ((px & 255u) - (prev & 255u)) & 255u;
		const uint32_t dg = (((px >> 8) & 255u) - ((prev >> 8) & 255u)) & 255u;
		const uint32_t db = (((px >> 16) & 255u) - ((prev >> 16) & 255u)) & 255u;
		const bool alpha_moves = C == 4 && (px >> 24) != (prev >> 24);
		const bool diff_ok = ((dr + 2u) & 255u) < 4u && ((dg + 2u) & 255u) < 4u && ((db + 2u) & 255u) < 4u;
		const bool luma_ok = ((dg + 32u) & 255u) < 64u && ((dr - dg + 8u) & 255u) < 16u && ((db - dg + 8u) & 255u) < 16u;
		const unsigned long long rgb = (unsigned long long)(px & 0x00ffffffu);
		unsigned long long op = 0xfeull | (rgb << 8);  // QOI_OP_RGB
		uint32_t op_len = 4u;
		if (luma_ok) {
			op = (0x80u | ((dg + 32u) & 63u)) | (((((dr - dg + 8u) & 15u) << 4) | ((db - dg + 8u) & 15u)) << 8);  // QOI_OP_LUMA
			op_len = 2u;
		}
		if (diff_ok) {
			op = 0x40u | (((dr + 2u) & 3u) << 4) | (((dg + 2u) & 3u) << 2) | ((db + 2u) & 3u);  // QOI_OP_DIFF
			op_len = 1u;
		}
		if (alpha_moves) {
			op = 0xffull | ((unsigned long long)px << 8);  // QOI_OP_RGBA
			op_len = 5u;
		}
		if (hit) {
			op = slot;  // QOI_OP_INDEX
			op_len = 1u;
		}
		unsigned long long bytes = pre_len ? ((unsigned long long)pre_byte | (op << 8)) : op;
		uint32_t n_bytes = pre_len + op_len;
		if (same) {
			bytes = flush_same ? (0xc0u | (run_if_same - 1u)) : 0u;
			n_bytes = flush_same ? 1u : 0u;
		}
		s.append(bytes, n_bytes);
		run = same ? (flush_same ? 0u : run_if_same) : 0u;
		seen_op = seen_op || !same;
		last_slot = same ? last_slot : slot;
		prev = px;  // (unchanged when the pixel repeats)
	};
	// Sixteen pixels per round, the next sixteen requested before the current ones are encoded -- a lane's pixel loop is one
	// dependent chain, and loads and stores share one in-order counter: waiting for a load also waits for every store issued
	// before it, so the loop waits once per sixteen pixels instead of once per four.  The last pixels of a segment (fewer
	// than sixteen) go four at a time.
	constexpr int kGroup = 16;
	uint32_t base = start;
	if (base + (uint32_t)kGroup <= end) {
		uint32_t ahead[kGroup];
		auto load_group = [&](uint32_t from, uint32_t (&dst)[kGroup]) __attribute__((always_inline)) { qoi_load16<C>(src, from, slot_px, dst); };
		load_group(base, ahead);
		for (; base + (uint32_t)kGroup <= end; base += (uint32_t)kGroup) {
			uint32_t cur[kGroup];
#pragma unroll
			for (int j = 0; j < kGroup; ++j) cur[j] = ahead[j];
			if (base + 2u * (uint32_t)kGroup <= end) load_group(base + (uint32_t)kGroup, ahead);
			// the index traffic of the whole round first: LDS works a wave's instructions off in order, so every look-up
			// sees the table as the pixels before it left it, and the sixteen round trips overlap instead of standing one
			// in front of every pixel (a hit rewrites the same value; slot 64 takes the writes of repeats)
			uint32_t slot16[kGroup], was16[kGroup];
#pragma unroll
			for (int j = 0; j < kGroup; ++j) {
				slot16[j] = qoi_hash(cur[j]);
				was16[j] = lds_dword(index + slot16[j]);
				const bool rep = cur[j] == (j ? cur[j - 1] : prev);
				*(volatile __attribute__((address_space(3))) uint32_t *)(index + (rep ? 64u : slot16[j])) = cur[j];
			}
#pragma unroll
			for (int j = 0; j < kGroup; ++j) step(cur[j], base + (uint32_t)j, slot16[j], was16[j] == cur[j]);
		}
	}
	for (; base < end; base += 4u) {
		uint32_t px4[4];
		qoi_load4<C>(src, base, slot_px, px4);
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			if (base + (uint32_t)j >= end) break;
			const uint32_t slot = qoi_hash(px4[j]);
			const bool hit = index[slot] == px4[j];
			index[px4[j] == prev ? 64u : slot] = px4[j];
			step(px4[j], base + (uint32_t)j, slot, hit);
		}
	}
	if (len && end == n) s.append(0x0100000000000000ull, 8);  // QOI end marker: seven zero bytes and a one
	const uint32_t bytes_here = 8u * s.rows + s.cnt;
	if (s.cnt) *s.out = s.acc;  // partial tail (the unit has a row for it)
	*my_len = bytes_here;
	// the record's length: every piece of the tile (a butterfly over the tile's lanes; all of them are here)
	uint32_t total = bytes_here;
	for (uint32_t d = 1; d < G; d <<= 1) total += __shfl_xor(total, (int)d, 64);
	if (seg == 0u) {
		// mod.rs:193-195: the record's length field, bytes 9..12 = bytes 1..4 of this lane's qword in row 2 (stored above, same lane)
		typedef uint32_t u32_a1 __attribute__((aligned(1)));
		*reinterpret_cast<u32_a1 *>(ql_.unit + (size_t)(2u * 64u + lane) * 8u + 1u) = __builtin_bswap32(total - 13u);
		a.rec_len[t] = total;
	}
}

// ---------------------------------------------------------------------------
// splice: the units of the encoder, one wave each, into the files behind the record scan.  Lane l stands for the same segment as
// in qoi_tiles_kernel (qoi_lane): row 0 gives it its piece's length, a scan over the tile's lanes the piece's place in the
// record, the two scans of the record lengths the record's place in the file.  The rows then come through LDS -- whole rows of
// 512 bytes from the scratch, 16 bytes per lane (two neighbouring pieces' qwords), into an image [piece][row] -- LPP rows at a
// time, and LPP lanes per piece write the bytes out with byte-aligned 16-byte moves (gfx950 runs in unaligned access mode: hipcc
// emits global_store_dwordx4 ... offset:3 for an align-1 vector), the bytes that are left of a piece one per lane.
// LPP = 16: segments of tiles of 128 pixels and more (up to 680 bytes a piece, 32 rows at a time); LPP = 4: units whose longest
// piece is at most 64 bytes (the records of the smallest tiles, 40 % of a typical frame: 16 of them per store).
// History: round 2 kept records as pieces per tile and moved them dword-wise (13 memory instructions per 344-byte piece, 0.19 ms);
// what bounds a splice is the NUMBER of its vector memory instructions (loads alone 0.08 ms, stores alone 0.07, look-ups 0.007).
// ---------------------------------------------------------------------------
template <int LPP>
__device__ __forceinline__ void qoi_splice_rows(const uint8_t *unit, uint32_t *s_img32, uint32_t lane, uint32_t bytes, uint8_t *my_dst, uint32_t k_unit)
{
	constexpr uint32_t kRows = 2u * LPP;        // rows per round: LPP lanes x 16 bytes of every piece
	constexpr uint32_t kSlot = kQoiRow * 4u;    // image of a piece: 260 bytes apart (bank skew)
	constexpr uint32_t kPer = 64u / LPP;        // pieces per store
	uint8_t *s_img = reinterpret_cast<uint8_t *>(s_img32);
	const uint32_t sub = lane / LPP, ql = lane % LPP;
	// step st: piece kPer st + sub -- its length and where it goes, fetched once
	const uint32_t dlo_mine = (uint32_t)reinterpret_cast<uintptr_t>(my_dst), dhi_mine = (uint32_t)(reinterpret_cast<uintptr_t>(my_dst) >> 32);
	uint32_t pb[LPP];
	uint8_t *pdst[LPP];
#pragma unroll
	for (uint32_t st = 0; st < (uint32_t)LPP; ++st) {
		const int p = (int)(kPer * st + sub);
		pb[st] = __shfl(bytes, p, 64);
		const uint32_t dlo = __shfl(dlo_mine, p, 64), dhi = __shfl(dhi_mine, p, 64);
		pdst[st] = reinterpret_cast<uint8_t *>(((uintptr_t)dhi << 32) | dlo);
	}
	for (uint32_t k0 = 0; k0 < k_unit; k0 += kRows) {
		const uint32_t rows = k_unit - k0 < kRows ? k_unit - k0 : kRows;
		tile_sync<1>();  // (the moves of the round before are done with the image)
		// rows k0 .. k0 + rows (data rows: the unit's row 1 + ...): lane l takes the qwords of pieces 2 (l & 31), + 1 of row 2 j + (l >> 5);
		// every load is issued before the first one is used (rows past the end repeat row k0: nothing is conditional)
		uint4 v[LPP];
#pragma unroll
		for (uint32_t j = 0; j < (uint32_t)LPP; ++j) {
			const uint32_t r = 2u * j + (lane >> 5), row = r < rows ? k0 + r : k0;
			v[j] = *reinterpret_cast<const uint4 *>(unit + ((size_t)(1u + row) * 64u + 2u * (lane & 31u)) * 8u);
		}
#pragma unroll
		for (uint32_t j = 0; j < (uint32_t)LPP; ++j) {
			const uint32_t r = 2u * j + (lane >> 5);
			if (r < rows) {
				uint32_t *d0 = reinterpret_cast<uint32_t *>(s_img + (2u * (lane & 31u)) * kSlot + r * 8u);
				uint32_t *d1 = reinterpret_cast<uint32_t *>(s_img + (2u * (lane & 31u) + 1u) * kSlot + r * 8u);
				d0[0] = v[j].x; d0[1] = v[j].y;
				d1[0] = v[j].z; d1[1] = v[j].w;
			}
		}
		tile_sync<1>();
		// piece p's bytes [8 k0, min(bytes, 8 (k0 + rows))): one 16-byte move per lane, then the bytes that are left, one per lane
		const uint32_t lo = 8u * k0, top = 8u * (k0 + rows);
#pragma unroll
		for (uint32_t st = 0; st < (uint32_t)LPP; ++st) {
			const uint32_t hi = pb[st] < top ? pb[st] : top;
			const uint32_t span = hi > lo ? hi - lo : 0u, n_q = span >> 4;
			const uint8_t *img = s_img + (kPer * st + sub) * kSlot;
			// (a pointer put together from two shuffled dwords has lost its address space: said again, or the stores are flat_store)
			typedef __attribute__((address_space(1))) uint8_t *global_byte_ptr;
			typedef __attribute__((address_space(1))) u32q_a1 *global_u32q_a1_ptr;
			global_byte_ptr dst = (global_byte_ptr)(pdst[st] + lo);
			if (ql < n_q) {
				const uint32_t *w = reinterpret_cast<const uint32_t *>(img + 16u * ql);
				const u32q_a1 o = {w[0], w[1], w[2], w[3]};
				*(global_u32q_a1_ptr)(dst + 16u * ql) = o;
			}
#pragma unroll
			for (uint32_t b = ql; b < 15u; b += (uint32_t)LPP)  // (one round at LPP = 16, up to four at LPP = 4)
				if (b < (span & 15u)) dst[16u * n_q + b] = img[16u * n_q + b];
		}
	}
}

// The units a wave takes (a fixed grid of waves, each walking the units with a stride): what a unit needs before its first
// row moves is a chain of dependent round trips -- the tile of every lane (perm), its record's place (the two scans), the
// lengths (row 0) -- so the chain of the unit after next and of the next one are in flight while the current one is copied
// (a wave per unit spent most of its 15 us in that chain).  The classes' counters are read once per wave: lane c keeps class c.
struct SpliceClasses {
	uint32_t first_unit, n_units64, n_tiles, first_tile, base_lo, base_hi;  // of class lane & 31
};
struct SpliceUnit {
	bool valid;
	uint32_t cls, G, seg, t, bytes0;
	bool live;
	const uint8_t *unit;
	unsigned long long off_chunk, off_tile;  // the record's place: the two scans (kept apart: adding them here would wait for them here)
};
__device__ __forceinline__ void splice_locate(const QoiArgs &a, const SpliceClasses &k, uint32_t unit0, uint32_t total_units, uint32_t lane, SpliceUnit &u)
{
	u.valid = unit0 < total_units;
	u.live = false;
	u.t = 0u;
	u.bytes0 = 0u;
	u.unit = a.scratch;
	u.cls = 0u; u.G = 1u; u.seg = 0u;
	if (!u.valid) return;
	const unsigned long long mine = __ballot(lane < 32u && unit0 >= k.first_unit && unit0 < k.first_unit + k.n_units64);
	u.cls = (uint32_t)__builtin_ctzll(mine | (1ull << 63));
	u.G = qoi_class_segments(u.cls);
	const uint32_t first = (uint32_t)__builtin_amdgcn_readlane((int)k.first_unit, (int)u.cls), rel = unit0 - first + lane;
	const uint32_t rank = rel / u.G;
	u.seg = rel & (u.G - 1u);
	u.live = rank < (uint32_t)__builtin_amdgcn_readlane((int)k.n_tiles, (int)u.cls);
	const unsigned long long base = (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)k.base_lo, (int)u.cls) |
	                                ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)k.base_hi, (int)u.cls) << 32);
	u.unit = a.scratch + base + (size_t)((unit0 - first) / 64u) * (512u * qoi_class_rows(u.cls, a.slot_bytes / a.channels, a.channels));
	if (u.live) u.t = a.perm[(uint32_t)__builtin_amdgcn_readlane((int)k.first_tile, (int)u.cls) + rank];
	u.bytes0 = reinterpret_cast<const uint32_t *>(u.unit)[2u * lane];  // row 0 (lanes without a segment: 0)
}

__device__ __forceinline__ void qoi_splice_units(const QoiArgs &a, uint32_t wave, uint32_t n_waves, uint32_t *s_img32)
{
	const uint32_t lane = threadIdx.x;
	const uint32_t total_units = a.bins[kBinTotal];
	SpliceClasses k;
	{
		const uint32_t c = lane & 31u;
		k.first_unit = a.bins[kBinUnits + c];
		k.n_tiles = a.bins[kBinN + c];
		k.n_units64 = (k.n_tiles * qoi_class_segments(c) + 63u) & ~63u;
		k.first_tile = a.bins[kBinTiles + c];
		k.base_lo = a.bins[kBinBase + 2u * c];
		k.base_hi = a.bins[kBinBase + 2u * c + 1u];
	}
	const uint32_t stride = n_waves * 64u;
	SpliceUnit u0, u1, u2;  // current (place known), next (tile known), the one after (being located)
	splice_locate(a, k, wave * 64u, total_units, lane, u0);
	u0.off_chunk = a.chunk_totals[u0.t / kPackChunk];  // (lanes without a segment: tile 0's, unused)
	u0.off_tile = a.offsets[u0.t];
	splice_locate(a, k, wave * 64u + stride, total_units, lane, u1);
	for (uint32_t unit0 = wave * 64u; u0.valid; unit0 += stride) {
		// ---- the look-ups of the units to come (issued first: they travel while the copy below runs)
		splice_locate(a, k, unit0 + 2u * stride, total_units, lane, u2);
		u1.off_chunk = a.chunk_totals[u1.t / kPackChunk];
		u1.off_tile = a.offsets[u1.t];
		// ---- the current unit: the piece's place in its record (a scan over the tile's lanes) and the record's length
		uint32_t bytes = u0.bytes0;
		uint32_t total = bytes, incl = bytes;
		for (uint32_t d = 1; d < u0.G; d <<= 1) {
			total += __shfl_xor(total, (int)d, 64);
			const uint32_t up = __shfl_up(incl, d, 64);
			if (u0.seg >= d) incl += up;
		}
		const unsigned long long dstoff = (unsigned long long)(u0.t / a.tiles_per_frame + 1u) * a.hdr_bytes + u0.off_chunk + u0.off_tile;
		if (!u0.live || dstoff + total > a.capacity) bytes = 0u;  // (a record that does not fit is left out whole: total is the same on its lanes)
		uint8_t *my_dst = a.out + dstoff + (incl - u0.bytes0);
		uint32_t k_unit = (bytes + 7u) >> 3;
		for (uint32_t d = 1; d < 64u; d <<= 1) {
			const uint32_t o = __shfl_xor(k_unit, (int)d, 64);
			k_unit = o > k_unit ? o : k_unit;
		}
		k_unit = __builtin_amdgcn_readfirstlane(k_unit);
		if (k_unit <= 8u) qoi_splice_rows<4>(u0.unit, s_img32, lane, bytes, my_dst, k_unit);
		else qoi_splice_rows<16>(u0.unit, s_img32, lane, bytes, my_dst, k_unit);
		u0 = u1;
		u1 = u2;
	}
}

// file header + line-length table (mod.rs:50-57,77-82): one thread per (frame, tile row)
__device__ __forceinline__ void qoi_headers(const QoiArgs &a, uint32_t block)
{
	const uint32_t i = block * 64u + threadIdx.x;
	const uint32_t frames = a.n_tiles / a.tiles_per_frame;
	if (i >= frames * a.rows) return;
	const uint32_t f = i / a.rows, r = i - f * a.rows;
	const uint32_t t0 = f * a.tiles_per_frame + r * a.cols;
	// record offsets: the chunk's start + the place inside the chunk (the two scans); a row's length is a difference
	auto record_offset = [&](uint32_t t) -> unsigned long long {
		return t == a.n_tiles ? a.offsets[a.n_tiles] : a.chunk_totals[t / kPackChunk] + a.offsets[t];
	};
	const unsigned long long lo = record_offset(t0);
	const unsigned long long hi = record_offset(t0 + a.cols);
	const unsigned long long file0 = (unsigned long long)f * a.hdr_bytes + record_offset(f * a.tiles_per_frame);
	if (r == 0) {
		// the offsets are valid whatever the room: only the byte stores below are guarded (a caller sizes its next
		// buffer, and dist.gather_files its sends, from file_offsets[frames])
		a.file_offsets[f] = file0;
		if (f + 1 == frames) a.file_offsets[frames] = (unsigned long long)frames * a.hdr_bytes + a.offsets[a.n_tiles];
	}
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
	}
}

// One launch for what follows the scan of the record lengths: the waves that walk the units, then the blocks of the headers.
__global__ void __launch_bounds__(64) qoi_splice_kernel(const QoiArgs a)
{
	__shared__ uint32_t s_img[64 * kQoiRow];
	if (blockIdx.x < a.splice_unit_blocks) {
		qoi_splice_units(a, blockIdx.x, a.splice_unit_blocks, s_img);
	} else {
		qoi_headers(a, blockIdx.x - a.splice_unit_blocks);
	}
}

// bytes of scratch the encoder can need for n_tiles slots of slot_px pixels: every tile in the class that costs most, plus
// every class's run rounded up to a whole unit
size_t qoi_scratch_bytes(uint32_t n_tiles, uint32_t slot_px, uint32_t channels)
{
	const uint32_t top = 31u - (uint32_t)__builtin_clz(slot_px | 1u);
	size_t per_tile = 0, pad = 0;
	for (uint32_t c = 0; c <= top; ++c) {
		const size_t lane_bytes = 8u * (size_t)qoi_class_rows(c, slot_px, channels);
		const size_t t = lane_bytes * qoi_class_segments(c);
		per_tile = t > per_tile ? t : per_tile;
		pad += 64u * lane_bytes;
	}
	return (size_t)n_tiles * per_tile + pad;
}
uint32_t qoi_bins_dwords() { return kBinDwords; }

hipError_t launch_qoi(const QoiArgs &args, bool bins_clean, uint32_t n_cus, hipStream_t stream)
{
	QoiArgs a = args;
	hipError_t e;
	// (the counters are left zeroed by the previous launch on the same buffer: bins_clean)
	if (!bins_clean && (e = hipMemsetAsync(a.bins, 0, kBinDwords * sizeof(uint32_t), stream)) != hipSuccess) return e;
	const uint32_t tb = (a.n_tiles + kBinChunk - 1u) / kBinChunk;
	hipLaunchKernelGGL(qoi_bin_count_kernel, dim3(tb), dim3(256), 0, stream, a);
	hipLaunchKernelGGL(qoi_bin_scatter_kernel, dim3(tb), dim3(256), 0, stream, a);
	// one wave per 64 segments; the number of segments is only known on the device, so the grid covers the worst case
	// (every tile in the class of a full slot, every class padded to a whole wave) and the surplus waves leave at once
	const uint32_t slot_px = a.slot_bytes / a.channels, top = 31u - (uint32_t)__builtin_clz(slot_px | 1u);
	const unsigned long long worst = ((unsigned long long)a.n_tiles * qoi_class_segments(top) + 63ull) / 64ull + 32ull;
	if (worst > 0x7fffffffull) return hipErrorInvalidValue;
	const uint32_t qb = (uint32_t)worst;
	if (a.channels == 4) hipLaunchKernelGGL(qoi_tiles_kernel<4>, dim3(qb), dim3(64u), 0, stream, a);
	else hipLaunchKernelGGL(qoi_tiles_kernel<3>, dim3(qb), dim3(64u), 0, stream, a);
	// exclusive scan of the record lengths (same chunked scan as the pixel pack, sizes given)
	PackArgs p{};
	p.sizes = a.rec_len;
	p.offsets = a.offsets;
	p.chunk_totals = a.chunk_totals;
	p.n_tiles = a.n_tiles;
	p.n_chunks = a.n_chunks;
	hipLaunchKernelGGL(pack_scan_local_kernel, dim3(a.n_chunks), dim3(256), 0, stream, p);
	hipLaunchKernelGGL(pack_scan_chunks_kernel, dim3(1), dim3(1024), 0, stream, p);
	// the waves that walk the units (as many as fit the chip: eight per CU -- or fewer, if even the worst case has fewer units),
	// a thread per (frame, tile row) for the headers
	const uint32_t frames = a.n_tiles / a.tiles_per_frame;
	const uint32_t per_cu = 8u;  // (its registers allow two waves per SIMD; 6 per CU: +2 %, 4: +20 %)
	a.splice_unit_blocks = qb < n_cus * per_cu ? qb : n_cus * per_cu;
	const unsigned long long grid = (unsigned long long)a.splice_unit_blocks + (frames * a.rows + 63u) / 64u;
	if (grid > 0x7fffffffull) return hipErrorInvalidValue;
	hipLaunchKernelGGL(qoi_splice_kernel, dim3((uint32_t)grid), dim3(64), 0, stream, a);
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
// the same on pointers that say where they point (through generic ones every access is a flat_load: the index kernel had 150)
typedef const __attribute__((address_space(3))) uint8_t *lds_byte_cptr;
typedef const __attribute__((address_space(1))) uint8_t *global_byte_cptr;
__device__ __forceinline__ uint32_t be32(lds_byte_cptr p)
{
	return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3];
}
__device__ __forceinline__ uint32_t be32(global_byte_cptr p)
{
	return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3];
}

// One wave per (file, tile row).  The walk over a row's records is a dependent chain (each length gives the next
// record's position): it runs on bytes staged in LDS, chunk by chunk, so a step costs an LDS round trip instead of
// an HBM one.  Per chunk: all lanes load it (coalesced), lane 0 walks up to 64 records ahead using only the
// length fields, then the lanes check and publish those records in parallel.
constexpr uint32_t kIdxChunk = 8192;   // bytes of a row held in LDS at a time (per wave; 0.078 ms -- 4096: 0.096, 16384: 0.090)
constexpr uint32_t kIdxHeader = 23;    // "block" + value + length + QOI header minus its magic, up to the channel byte
__global__ void __launch_bounds__(256) pixlzr_index_kernel(const DecodeArgs a)
{
	__shared__ __attribute__((aligned(16))) uint32_t s_chunk[4][kIdxChunk / 4u + 8u];
	// (the wave's number and everything derived from it as scalars: the walk below then runs on scalar branches)
	const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;
	const uint32_t i = blockIdx.x * 4u + wave;
	if (i >= a.n_frames * a.rows) return;
#ifdef PXZ_STAMPS
	unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	unsigned long long st_last = stamp_now();
	auto st_flush = [&]() {
		if (lane == 0) {
			unsigned long long *out = reinterpret_cast<unsigned long long *>(a.status + 2);
			for (int k = 0; k < 8; ++k) atomicAdd(out + k, st_acc[k]);
		}
	};
#endif
	const uint32_t f = i / a.rows, r = i - f * a.rows;
	const unsigned long long f0 = a.file_offsets[f], f1 = a.file_offsets[f + 1];
	global_byte_cptr file = (global_byte_cptr)(a.files + f0);
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
	// the fixed header, a byte per lane: "PIXLZR", 0, 0, 2 (constants.rs:10-11: v0.0.2, filter byte + line table), the filter byte
	// (any), then width, height and the block sides as BE dwords -- ONE round trip (as `a && file[k] == ..` chains the 25 byte
	// loads were 25 dependent round trips in front of every row's walk)
	bool ok = flen >= hdr;
	{
		uint32_t got = 0, want = 0;
		if (ok && lane < 26u) got = file[lane];
		if (lane < 8u) want = (uint32_t)(0x0000525a4c584950ull >> (8u * lane)) & 255u;
		else if (lane == 8u) want = 2u;
		else if (lane == 9u) want = got;
		else if (lane < 26u) {
			const uint32_t k = lane - 10u, field = k < 4u ? a.width : (k < 8u ? a.height : (k < 12u ? a.bw : a.bh));
			want = (field >> (8u * (3u - (k & 3u)))) & 255u;
		}
		ok = ok && __builtin_amdgcn_ballot_w64(got != want) == 0ull;
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
	auto uniform64 = [](unsigned long long v) -> unsigned long long {
		// (the builtin returns a signed int: without the casts a low half of 2^31 and more is sign-extended over the high one)
		return ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(v >> 32)) << 32) |
		       (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)v);
	};
	unsigned long long p = uniform64(hdr + before);
	const unsigned long long row_end = p + (uint32_t)__builtin_amdgcn_readfirstlane(be32(file + 26 + 4 * r));
	uint32_t c = 0;
	PXZ_STAMP(0);  // file header, line table
	while (c < a.cols) {
		// ---- stage file bytes [p, p + kIdxChunk) of the row: whole aligned 16-byte granules of the buffer, every load of the
		// chunk issued before the first one is written to LDS (round 2 staged dwords, one load -> store round per 256 bytes: a
		// wave walks a row of ~170 KB alone, and the kernel was 0.33 ms of chained round trips)
		const unsigned long long want = row_end - p < (unsigned long long)kIdxChunk ? row_end - p : (unsigned long long)kIdxChunk;
		const uintptr_t g = reinterpret_cast<uintptr_t>(file + p);
		const uint32_t skew = (uint32_t)(g & 15u);  // the chunk starts at the aligned granule below p
		const uint32_t granules = (skew + (uint32_t)want + 15u) / 16u;  // <= kIdxChunk / 16 + 1
		const uintptr_t buf_end = reinterpret_cast<uintptr_t>(a.files) + a.file_offsets[a.n_frames];
		constexpr uint32_t kRounds = (kIdxChunk / 16u + 1u + 63u) / 64u;
		uint4 gv[kRounds];
		typedef uint32_t u32q __attribute__((ext_vector_type(4)));
		// (every load unconditional, at a clamped address -- the first bytes of the buffer stand in for granules that are not
		// wanted or not whole: a load under a condition is waited for where it is issued, round by round)
#pragma unroll
		for (uint32_t k = 0; k < kRounds; ++k) {
			const uint32_t d = lane + 64u * k;
			const uintptr_t ga = (g - skew) + 16ull * d;
			const bool whole = d < granules && ga + 16u <= buf_end;
			const u32q q4 = *(const __attribute__((address_space(1))) u32q *)(whole ? ga : reinterpret_cast<uintptr_t>(a.files));
			gv[k] = make_uint4(q4.x, q4.y, q4.z, q4.w);
		}
#pragma unroll
		for (uint32_t k = 0; k < kRounds; ++k) {
			const uint32_t d = lane + 64u * k;
			const uintptr_t ga = (g - skew) + 16ull * d;
			if (d < granules && ga + 16u > buf_end) {  // (the last granule of the buffer: byte by byte)
				uint32_t w4[4] = {0, 0, 0, 0};
				for (uint32_t b = 0; b < 16u && ga + b < buf_end; ++b) w4[b >> 2] |= (uint32_t) * (global_byte_cptr)(ga + b) << (8u * (b & 3u));
				gv[k] = make_uint4(w4[0], w4[1], w4[2], w4[3]);
			}
		}
#pragma unroll
		for (uint32_t k = 0; k < kRounds; ++k) {
			const uint32_t d = lane + 64u * k;
			if (d < granules) reinterpret_cast<uint4 *>(s_chunk[wave])[d] = gv[k];
		}
		tile_sync<1>();
		PXZ_STAMP(1);  // chunk staged
		const uint32_t have = (uint32_t)want;  // valid bytes behind byte `skew` of the chunk
		// ---- the walk: positions of up to 64 records whose headers lie inside the chunk.  Every lane runs it with the same
		// values (the length field is read by all lanes and made scalar), so its tests are scalar branches, and lane n keeps the
		// position of record n in a register.  (As `if (lane == 0) { .. }` a step was sixty instructions, half of them on the
		// execution mask of its four nested branches, and the positions went through LDS: the walk was most of the kernel.)
		const unsigned long long rem64 = row_end - p;
		const uint32_t rem = rem64 < 0xffffffffull ? (uint32_t)rem64 : 0xffffffffu;  // bytes left of the row, saturated
		uint32_t n_rec = 0, walked = 0;  // walked: bytes of the chunk consumed by the records found
		uint32_t pos = 0;
		{
			uint32_t o = 0;
			const uint32_t cols_left = a.cols - c, n_max = cols_left < 64u ? cols_left : 64u;
			// a record can be walked from o if it has room (13 + 10 + 8 bytes: o + 31 <= rem) and its header lies in the chunk
			// (o + 23 <= have; have <= rem): o <= lim, one test
			const bool none = rem < 31u;
			const uint32_t lim = none ? 0u : (rem - 31u < have - kIdxHeader ? rem - 31u : have - kIdxHeader);
			bool at_top = false;
			while (n_rec < n_max) {
				pos = lane == n_rec ? o : pos;
				if (none | (o > lim)) {
					at_top = true;
					break;
				}
				// the length field: the two aligned dwords around it, shifted (one LDS instruction instead of four byte reads)
				const uint32_t x = skew + o + 9u;
				const uint32_t *w2 = s_chunk[wave] + (x >> 2);
				const uint32_t qlen = __builtin_amdgcn_readfirstlane(__builtin_bswap32(__builtin_amdgcn_alignbyte(w2[1], w2[0], x & 3u)));
				++n_rec;
				const unsigned long long next = (unsigned long long)o + 13ull + qlen;
				if ((qlen < 18u) | (next > (unsigned long long)rem)) break;  // broken: found again by the checks below
				o = (uint32_t)next;  // (a next record beyond the chunk leaves through the test at the top)
			}
			// no room for a record although one is due: a broken row, counted so that the checks below flag it
			if (at_top && (none || (unsigned long long)o + 31ull > (unsigned long long)rem)) ++n_rec;
			walked = o;
		}
		PXZ_STAMP(2);  // walk
		// ---- all lanes: check and publish the batch (header bytes read without conditions, compared afterwards)
		bool good = true;
		if (lane < n_rec) {
			const uint32_t o = pos;
			const uint32_t cc = c + lane;
			const uint32_t fw = (cc == a.cols - 1) ? a.edge_w : a.bw, fh = (r == a.rows - 1) ? a.edge_h : a.bh;
			good = o + 31u <= rem;
			if (good) {
				// the 22 header bytes as the seven aligned dwords around them, shifted into place ("block", value, length, and of
				// the QOI header behind its magic: width, height, channels)
				const uint32_t x = skew + o, sh = x & 3u;
				const uint32_t *d = s_chunk[wave] + (x >> 2);
				uint32_t dw[7], wd[6];
#pragma unroll
				for (int k = 0; k < 7; ++k) dw[k] = d[k];
#pragma unroll
				for (int k = 0; k < 6; ++k) wd[k] = __builtin_amdgcn_alignbyte(dw[k + 1], dw[k], sh);  // bytes 4k .. 4k + 3
				auto be_at1 = [&](int k) -> uint32_t { return __builtin_bswap32(__builtin_amdgcn_alignbyte(wd[k + 1], wd[k], 1u)); };  // bytes 4k + 1 ..
				const uint32_t value_bits = be_at1(1), qlen = be_at1(2), w = be_at1(3), h = be_at1(4), ch = (wd[5] >> 8) & 255u;
				const bool magic = (wd[0] == 0x636f6c62u) & ((wd[1] & 255u) == (uint32_t)'k');  // "bloc", "k"
				// (the walk used this record's length whether or not its other fields are sound, as the
				// sequential reader does not: a bad record ends the row there, see below)
				good = magic & (qlen >= 18u) & ((unsigned long long)o + 13ull + qlen <= (unsigned long long)rem) & (ch == a.channels) &
				       (w >= 1u) & (h >= 1u) & (w <= fw) & (h <= fh);
				if (good) {
					const uint32_t t = t_row + cc;
					a.value[t] = __uint_as_float(value_bits);
					a.tile_w[t] = w;
					a.tile_h[t] = h;
					a.rec_off[t] = f0 + p + o + 13ull + 10ull;  // first op byte
					a.rec_len[t] = qlen - 10u - 8u;             // ops only: without the header and the end marker
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
		if (n_rec == 0) {
			// a header that does not fit the rest of the row although a record is due: broken row
			bad_from(c);
			return;
		}
		c += n_rec;
		p += walked;
		tile_sync<1>();  // the chunk is restaged
		PXZ_STAMP(3);  // checks, publishing, closing sync
#ifdef PXZ_STAMPS
		st_acc[4] += 1;      // batches
		st_acc[5] += n_rec;  // records
#endif
	}
	if (p != row_end && lane == 0) atomicOr(a.status, 2u);
#ifdef PXZ_STAMPS
	st_flush();
#endif
}

// What an op's first byte says (qoi_decode_kernel's look-up table, one per channel count; built at compile time):
//   bits 0-23   what a DIFF op adds to r, g, b (bytes, mod 256); for a LUMA op (vg - 8, vg, vg - 8), to which its second byte's
//               nibbles are added; 0 otherwise
//   bits 24-26  the bytes the op has (3 channels: 0 for 0xff -- see the kernel)
//   bits 27-29  its kind
struct QoiLut {
	static constexpr uint32_t kIndex = 0, kDiff = 1, kLuma = 2, kRun = 3, kRgb = 4, kRgba = 5, kKeep = 6;
	uint32_t v[256];
	constexpr explicit QoiLut(int channels) : v()
	{
		for (uint32_t b = 0; b < 256u; ++b) {
			uint32_t kind = kRun, used = 1u, d = 0u;
			if (b == 0xfeu) { kind = kRgb; used = 4u; }
			else if (b == 0xffu) { kind = channels == 4 ? kRgba : kKeep; used = channels == 4 ? 5u : 0u; }
			else if (b < 0x40u) kind = kIndex;
			else if (b < 0x80u) { kind = kDiff; d = ((((b >> 4) & 3u) - 2u) & 255u) | (((((b >> 2) & 3u) - 2u) & 255u) << 8) | ((((b & 3u) - 2u) & 255u) << 16); }
			else if (b < 0xc0u) { kind = kLuma; used = 2u; const uint32_t vg = (b & 0x3fu) - 32u; d = ((vg - 8u) & 255u) | ((vg & 255u) << 8) | (((vg - 8u) & 255u) << 16); }
			v[b] = d | (used << 24) | (kind << 27);
		}
	}
};
__device__ const QoiLut kQoiLut4(4), kQoiLut3(3);

template <int C>
__global__ void __launch_bounds__(64 * kQoiWaves) qoi_decode_kernel(const DecodeArgs a)
{
	__shared__ uint32_t s_index[kQoiWaves][65][64];  // [wave][slot][lane]; row 64 takes the writes of lanes that have none
	// What an op's first byte says, looked up (round 4; it was worked out with ~35 compares, shifts and masks per pixel):
	//   bits 0-23   what a DIFF op adds to r, g, b (bytes, mod 256); for a LUMA op (vg - 8, vg, vg - 8), to which its second byte's
	//               nibbles are added; 0 otherwise
	//   bits 24-26  the bytes the op has (C = 3: 0 for 0xff -- see below)
	//   bits 27-29  its kind
	__shared__ __attribute__((aligned(16))) uint32_t s_lut[256];
	constexpr uint32_t kIndex = QoiLut::kIndex, kDiff = QoiLut::kDiff, kLuma = QoiLut::kLuma, kRun = QoiLut::kRun, kRgb = QoiLut::kRgb, kRgba = QoiLut::kRgba;
	// (copied, not worked out: a wave of 2x2 tiles -- half the tiles of a 16x16 grid -- has less to do than the fifty instructions the
	// table took to build; 8 x 8K at 16x16 tiles: 0.53 ms with the table built per block, 0.43 without a table at all)
	for (uint32_t i = threadIdx.x; i < 64u; i += 64u * kQoiWaves)
		reinterpret_cast<uint4 *>(s_lut)[i] = reinterpret_cast<const uint4 *>(C == 4 ? kQoiLut4.v : kQoiLut3.v)[i];
	__syncthreads();
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	const uint32_t i0 = blockIdx.x * (64u * kQoiWaves) + threadIdx.x;
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
	// The op bytes come through aligned 64-bit windows: kWin of them in registers (from the window the next op starts in), the (up
	// to) five bytes an op can have cut out of w[0] | w[1] at the stream position with one funnel shift.  A used-up window is
	// shifted out on the spot -- moves of data that is there -- and the buffer is filled up again once per kGroup pixels (which
	// use 5 kGroup bytes at most): the kReq windows that may be missing then are requested at the start of the group and taken at
	// its end, one wait per group with a group's time for the loads to arrive (4 pixels: 0.39 ms; 8: 0.34; 16: 0.35 -- more windows to move).
	// Rounds 2 and 3 (first half) requested a window whenever one was used up: every lane of a wave does that at its own pixels,
	// so the wave did it at nearly every pixel, and a request is waited for when its register is next moved -- in a wave whose
	// lanes take turns, at the very next pixel: one memory round trip per pixel step, 1024 of them in a row for a 32x32 tile
	// (0.69 ms; 0.52 once the request no longer went through a temporary that was copied, and waited for, on the spot).
	// Windows are requested without a condition but never past the last one of the files (what lies behind a record's ops is its
	// own end marker).  As in the encoder there are no branches per op: every op's result is worked out and selects keep the right one.
	// (the windows are 8-byte words at ABSOLUTE aligned addresses, whatever the alignment of `files`: the first may begin up to 7
	// bytes before the record and the last end up to 7 bytes behind the files' last byte, but an aligned word never crosses a page,
	// so an exactly sized buffer that ends on a page boundary is never read past its page)
	const uintptr_t first_addr = reinterpret_cast<uintptr_t>(a.files) + a.rec_off[t];
	const uintptr_t last_addr = reinterpret_cast<uintptr_t>(a.files) + (a.file_offsets[a.n_frames] - 1ull);
	// (pointers the compiler knows to be global, round 4: made from integers they were generic and the windows came through
	// flat_load, which counts as an LDS operation too -- every wait for an index or table look-up also waited for the windows
	// requested at the start of the group, the very loads the group's eight pixels are there to hide)
	typedef const __attribute__((address_space(1))) unsigned long long *global_qwords;
	global_qwords wp = (global_qwords)(first_addr & ~(uintptr_t)7);  // where w0 is from
	const global_qwords w_last = (global_qwords)(last_addr & ~(uintptr_t)7);
	auto window = [&](global_qwords p) __attribute__((always_inline)) { return *(p < w_last ? p : w_last); };
	constexpr uint32_t kGroup = 8;                               // pixels between two refills (4, tried again with the global loads: 0.44 against 0.40 ms)
	constexpr uint32_t kReq = (5u * kGroup + 7u) / 8u;           // windows they can use up (5 bytes a pixel)
	constexpr uint32_t kWin = (7u + 5u * (kGroup - 1u) + 8u + 7u) / 8u + 1u;  // windows the last of them can reach into (+1: whole moves)
	unsigned long long w[kWin];
#pragma unroll
	for (uint32_t k = 0; k < kWin; ++k) w[k] = window(wp + k);
	uint32_t used_up = 0;  // windows shifted out since the buffer was last filled (0 .. kReq)
	uint32_t pos = (uint32_t)(first_addr & 7u);             // byte position of the next op inside w0 (0..7)
	uint32_t left = len;                                    // op bytes not yet consumed
	const uint32_t n = a.tile_w[t] * a.tile_h[t];
	uint8_t *dst = a.slots + (size_t)t * a.slot_bytes;
	uint32_t px = 0xff000000u, run = 0;
	bool starved = false;
	uint4 hold = make_uint4(0, 0, 0, 0);
	// one pixel: the next op (or the run in progress) -> px
	auto next_pixel = [&]() __attribute__((always_inline)) {
		// the (up to) 8 bytes at the stream position
		const unsigned long long at = pos ? (w[0] >> (8u * pos)) | (w[1] << (64u - 8u * pos)) : w[0];
		const uint32_t b1 = (uint32_t)at & 255u;
		const uint32_t next4 = (uint32_t)(at >> 8);  // the four bytes behind the tag
		const bool in_run = run > 0u;
		starved = starved || (!in_run && left == 0u);  // the op stream ended before the tile was full
		const bool take = !in_run && left != 0u;       // this pixel consumes an op
		// (as in the qoi crate, only the op's first byte is checked against the end of the stream; a truncated last op
		// reads on into the end marker)
		const uint32_t e = s_lut[b1];
		// (the op's length from the table too: worked out by arithmetic -- to keep the LDS round trip out of the chain the next
		// op's position hangs on -- the kernel was SLOWER, 1.23 against 1.17 ms at 64x64: it is the instructions that count)
		const uint32_t kind = e >> 27, used = (e >> 24) & 7u;
		const uint32_t from_index = index[b1 & 63u][lane];
		// QOI_OP_DIFF / QOI_OP_LUMA: r and b in the two halves of one dword, g in another -- sums of bytes that cannot reach
		// the neighbouring field; a LUMA op's second byte adds its nibbles to r and b
		const uint32_t nib = kind == kLuma ? ((next4 >> 4) & 15u) | ((next4 & 15u) << 16) : 0u;
		const uint32_t rb = ((px & 0x00ff00ffu) + (e & 0x00ff00ffu) + nib) & 0x00ff00ffu;
		const uint32_t g_ = ((px >> 8) + (e >> 8)) & 0x000000ffu;
		uint32_t cand = px;  // QOI_OP_RUN: the pixel repeats
		if (kind - kDiff < 2u) cand = (px & 0xff000000u) | rb | (g_ << 8);
		if (kind == kIndex) cand = from_index;                                  // QOI_OP_INDEX
		if (kind == kRgb) cand = (px & 0xff000000u) | (next4 & 0x00ffffffu);    // QOI_OP_RGB
		if (C == 4 && kind == kRgba) cand = next4;                              // QOI_OP_RGBA (3 channels: see below)
		const uint32_t new_run = kind == kRun ? b1 & 0x3fu : 0u;
		px = take ? cand : px;
		run = in_run ? run - 1u : (take ? new_run : 0u);
		// The qoi crate stores the pixel in the index after RGB / RGBA / DIFF / LUMA ops only: its RUN and INDEX arms go
		// on to the next op before the store (an INDEX op's pixel is in its slot already; a RUN's is too, except when the
		// stream OPENS with a run of the implicit opaque black, which is never stored -- the encoder's run-of-one quirk
		// has the same root).  A 3-channel stream has no RGBA op: the crate's 3-channel decoder does not match 0xff (its
		// RGBA arm is guarded by the channel count) and falls into its catch-all arm, which fails only when fewer than
		// eight bytes are left -- never in front of the end marker -- and otherwise consumes NOTHING, stores the unchanged
		// pixel in the index and writes it: the same byte is met again by every pixel that follows, so the rest of the tile
		// repeats the last pixel and the decode succeeds [qoi 0.4.1 decode_impl_slice, from memory: the crate's source is
		// not in this environment; round 2 flagged such a record as malformed].
		const bool is_run_op = kind == kRun;
		index[(take && !is_run_op) ? qoi_hash(px) : 64u][lane] = px;
		const uint32_t step = take ? used : 0u;
		left = left > step ? left - step : 0u;
		pos += step;
		if (pos >= 8u) {  // the window is used up: the others move down (the last ones are refilled below)
			pos -= 8u;
#pragma unroll
			for (uint32_t k = 0; k + 1u < kWin; ++k) w[k] = w[k + 1u];
			++used_up;
		}
	};
	// kGroup pixels with the refill around them
	unsigned long long l[kReq];
	auto request = [&]() __attribute__((always_inline)) {
#pragma unroll
		for (uint32_t k = 0; k < kReq; ++k) l[k] = window(wp + kWin + k);
	};
	auto refill = [&]() __attribute__((always_inline)) {
		// u = used_up windows were shifted out: w[kWin - u + j] <- l[j] for j < u
#pragma unroll
		for (uint32_t k = kWin - kReq; k < kWin; ++k) {
			unsigned long long v = w[k];
#pragma unroll
			for (uint32_t j = 0; j < kReq; ++j)
				if (j + kWin >= k + 1u && j + kWin - k <= kReq) v = used_up == j + kWin - k ? l[j] : v;  // k = kWin - u + j  <=>  u = kWin + j - k
			w[k] = v;
		}
		wp += used_up;
		used_up = 0u;
	};
	if constexpr (C == 4) {
		// four pixels per 16-byte store (slots are 16-byte aligned: bw*bh*4 bytes each), the loop unrolled by those four: a
		// lane's pixel loop is one dependent chain, and every instruction on it counts (the kernel lasts as long as its longest
		// lanes; the per-pixel tests of which quarter of the store a pixel is were scalar instructions on that chain)
		uint32_t i = 0;
		for (; i + kGroup <= n; i += kGroup) {
			request();
#pragma unroll
			for (uint32_t q = 0; q < kGroup / 4u; ++q) {
				next_pixel(); hold.x = px;
				next_pixel(); hold.y = px;
				next_pixel(); hold.z = px;
				next_pixel(); hold.w = px;
				reinterpret_cast<uint4 *>(dst)[(i >> 2) + q] = hold;
			}
			refill();
		}
		// (what is left of the tile: fewer than kGroup pixels, within what the buffer holds after a refill)
		for (; i + 4u <= n; i += 4u) {
			next_pixel(); hold.x = px;
			next_pixel(); hold.y = px;
			next_pixel(); hold.z = px;
			next_pixel(); hold.w = px;
			reinterpret_cast<uint4 *>(dst)[i >> 2] = hold;
		}
		hold = make_uint4(0, 0, 0, 0);
		if (i < n) { next_pixel(); hold.x = px; }
		if (i + 1u < n) { next_pixel(); hold.y = px; }
		if (i + 2u < n) { next_pixel(); hold.z = px; }
	} else {
		uint32_t i = 0;
		for (; i + kGroup <= n; i += kGroup) {
			request();
			// the group's 24 bytes as six dwords, two stores (three byte stores per pixel before)
			static_assert(kGroup == 8, "the packing below is for groups of eight pixels");
			uint32_t g8[8];
#pragma unroll
			for (uint32_t k = 0; k < kGroup; ++k) {
				next_pixel();
				g8[k] = px & 0xffffffu;
			}
			typedef uint32_t u32q_a1 __attribute__((ext_vector_type(4), aligned(1)));
			typedef uint32_t u32d_a1 __attribute__((ext_vector_type(2), aligned(1)));
			const u32q_a1 lo = {g8[0] | (g8[1] << 24), (g8[1] >> 8) | (g8[2] << 16), (g8[2] >> 16) | (g8[3] << 8), g8[4] | (g8[5] << 24)};
			const u32d_a1 hi = {(g8[5] >> 8) | (g8[6] << 16), (g8[6] >> 16) | (g8[7] << 8)};
			*reinterpret_cast<u32q_a1 *>(dst + 3u * i) = lo;
			*reinterpret_cast<u32d_a1 *>(dst + 3u * i + 16u) = hi;
			refill();
		}
		for (; i < n; ++i) {
			next_pixel();
			dst[3 * i] = (uint8_t)px;
			dst[3 * i + 1] = (uint8_t)(px >> 8);
			dst[3 * i + 2] = (uint8_t)(px >> 16);
		}
	}
	if (starved) {
		atomicOr(a.status, 2u);
		a.tile_w[t] = 0;
		a.tile_h[t] = 0;
		return;
	}
	if constexpr (C == 4) {
		const uint32_t tail = n & 3u, base = n & ~3u;  // 1x1, 2x1 ... tiles
		if (tail >= 1) reinterpret_cast<uint32_t *>(dst)[base] = hold.x;
		if (tail >= 2) reinterpret_cast<uint32_t *>(dst)[base + 1] = hold.y;
		if (tail >= 3) reinterpret_cast<uint32_t *>(dst)[base + 2] = hold.z;
	}
}

hipError_t launch_decode(const DecodeArgs &a, bool bins_clean, hipStream_t stream)
{
	// (the binning counters are left zeroed by the previous launch on the same buffer: bins_clean, as in launch_qoi)
	hipError_t e;
	if (!bins_clean && (e = hipMemsetAsync(a.bins, 0, kBinDwords * sizeof(uint32_t), stream)) != hipSuccess) return e;
	hipLaunchKernelGGL(pixlzr_index_kernel, dim3((a.n_frames * a.rows + 3u) / 4u), dim3(256), 0, stream, a);
	const uint32_t tb = (a.n_tiles + kBinChunk - 1u) / kBinChunk;
	QoiArgs q{};  // the encoder's binning by pixel count, on the sizes the index kernel has just read
	q.w = a.tile_w;
	q.h = a.tile_h;
	q.n_tiles = a.n_tiles;
	q.bins = a.bins;
	q.perm = a.perm;
	hipLaunchKernelGGL(qoi_bin_count_kernel, dim3(tb), dim3(256), 0, stream, q);
	hipLaunchKernelGGL(qoi_bin_scatter_kernel, dim3(tb), dim3(256), 0, stream, q);
	const uint32_t qb = (a.n_tiles + 64u * kQoiWaves - 1u) / (64u * kQoiWaves);
	if (a.channels == 4) hipLaunchKernelGGL(qoi_decode_kernel<4>, dim3(qb), dim3(64u * kQoiWaves), 0, stream, a);
	else hipLaunchKernelGGL(qoi_decode_kernel<3>, dim3(qb), dim3(64u * kQoiWaves), 0, stream, a);
	return hipGetLastError();
}

}  // namespace pxz
