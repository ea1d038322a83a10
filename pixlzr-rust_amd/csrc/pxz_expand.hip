// pxz_expand.hip -- decode side: Pixlzr::expand + to_image on the device (expand_kernel) and its launcher.
//
// Compiled with -ffp-contract=off: the f32 results of the Oklab detector are
// written into the bitstream, and the reference (Rust) never fuses a*b+c.
#include "pxz_device.h"

namespace pxz {

typedef int v16i32 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------
// The convolutions of a full 32x32 RGBA tile stored as tw x th, both in {1, 2, 4, 8, 16}, on the matrix cores
// (v_mfma_i32_32x32x16_i8; same integers as the vector form in expand_kernel, which follows block.rs:273-334):
//   horizontal  T[(c, y)][ox] = clip8(sum_x P_c[y][x] Kx[x][ox])   A = pixel bytes of the channel planes, rows (channel, y):
//               all four channels in ONE product when th <= 8, two products of two channels at 16;  B = weights
//   vertical    O_c[oy][ox]   = clip8(sum_y Ky[oy][y] T_c[y][ox])   A = weights, B = T_c
// T never leaves the registers: the accumulator of the horizontal product holds, for column ox = lane & 31, the rows
// xmf_row(g, reg) -- bytes y = 4g + j (and 8 + 4g + j) of a channel, which IS a B operand whose k slots stand for the
// rows xmf_src(g, j); the weight tables are stored with their k slots in that order (pxz_internal.h) and the pixel operand of
// the horizontal product reads its source columns in that order too, so one table per stored size serves both passes.
// Weights are 16-bit: two products (low bytes, high bytes) per block, pixels as signed bytes (p - 128, the bias
// carries 128 * weight sum + the rounding half), recombined and clamped by clamp_fixed.
// Output: lane (ox, g) ends with the 16 pixels of column ox in the rows xmf_row(g, reg): 16 dword stores whose 32 lanes
// of a half-wave cover 128 contiguous bytes of a frame row.  A stored height of 1 whose table says "copied" skips the
// vertical product (40 % of the tiles of a typical frame are 2x1): the row goes through LDS and is written 32 times.
// s_wave: the wave's LDS (planes: 1 KB; the row: 128 B behind them).
// ---------------------------------------------------------------------------
__device__ __forceinline__ void expand_tile_mfma(const ExpandArgs &a, const uint32_t *s_xmf, uint32_t *s_wave, uint32_t lane,
                                                 uint32_t tw, uint32_t th, const uint8_t *src, uint32_t first_px, uint8_t *dst
#ifdef PXZ_STAMPS
                                                 , unsigned long long (&st_acc)[8], unsigned long long &st_last
#endif
)
{
	const uint32_t n = lane & 31u, g = lane >> 5;
	const uint32_t lw = 31u - (uint32_t)__builtin_clz(tw), lh = 31u - (uint32_t)__builtin_clz(th);
	const uint32_t *mx = s_xmf + lw * kXmfDw, *my = s_xmf + lh * kXmfDw;
	const bool fold = th <= 8u;
	const uint32_t plane = fold ? 128u : 256u;  // bytes per channel plane: rows of 16 bytes
	uint8_t *s_pl = reinterpret_cast<uint8_t *>(s_wave);
	// ---- stored pixels -> premultiplied channel planes [c][y][16]
	const uint32_t npx = tw * th;
	for (uint32_t i = lane; i < npx; i += 64u) {
		uint32_t px = i == lane ? first_px : reinterpret_cast<const uint32_t *>(src)[i];
		if (__builtin_amdgcn_ballot_w64((px >> 24) != 255u) != 0ull) px = premultiply(px);  // fir: U8x4 is alpha-premultiplied before a convolution
		uint8_t *d = s_pl + (i >> lw) * 16u + (i & (tw - 1u));
		d[0] = (uint8_t)px;
		d[plane] = (uint8_t)(px >> 8);
		d[2u * plane] = (uint8_t)(px >> 16);
		d[3u * plane] = (uint8_t)(px >> 24);
	}
	tile_sync<1>();
	PXZ_STAMP(3);  // planes staged
	// ---- horizontal product(s)
	const uint32_t px_ = __builtin_amdgcn_readfirstlane(mx[320]), py = __builtin_amdgcn_readfirstlane(my[320]);
	const int32_t top_x = (int32_t)((256u << px_) - 1u), top_y = (int32_t)((256u << py) - 1u);
	const long kx_lo = *reinterpret_cast<const long *>(mx + 2u * lane), kx_hi = *reinterpret_cast<const long *>(mx + 128u + 2u * lane);
	const int32_t bx = (int32_t)mx[256u + n];
	v16i32 cx, zero;
#pragma unroll
	for (int r = 0; r < 16; ++r) { cx[r] = bx; zero[r] = 0; }
	uint32_t t0[4] = {0, 0, 0, 0}, t1[4] = {0, 0, 0, 0};  // per channel: rows 4g + j | rows 8 + 4g + j of column n, as bytes
	auto product = [&](uint32_t blk, auto place) __attribute__((always_inline)) {
		const uint32_t *row = reinterpret_cast<const uint32_t *>(s_pl + (32u * blk + n) * 16u + 4u * g);
		const uint32_t a0 = row[0] ^ 0x80808080u, a1 = row[2] ^ 0x80808080u;  // columns 4g + j | 8 + 4g + j of row (channel, y) = n
		const long av = (long)(((unsigned long long)a1 << 32) | (unsigned long long)a0);
		const v16i32 lo = __builtin_amdgcn_mfma_i32_32x32x16_i8(av, kx_lo, cx, 0, 0, 0);
		const v16i32 hi = __builtin_amdgcn_mfma_i32_32x32x16_i8(av, kx_hi, zero, 0, 0, 0);
#pragma unroll
		for (int r = 0; r < 16; ++r) place(r, clamp_fixed(hi[r], lo[r], top_x));
	};
	auto put_byte = [&](uint32_t &d, uint32_t j, uint32_t v, uint32_t sh) __attribute__((always_inline)) {
		if (j == 0) put_byte_shr<0>(d, v, sh);
		else if (j == 1) put_byte_shr<1>(d, v, sh);
		else if (j == 2) put_byte_shr<2>(d, v, sh);
		else put_byte_shr<3>(d, v, sh);
	};
	if (fold) {
		product(0, [&](int r, uint32_t v) __attribute__((always_inline)) { put_byte(t0[r >> 2], (uint32_t)r & 3u, v, px_); });  // reg = 4 c + j
	} else {
#pragma unroll
		for (uint32_t blk = 0; blk < 2; ++blk)  // reg = 8 (c & 1) + 4 (second group of rows) + j
			product(blk, [&](int r, uint32_t v) __attribute__((always_inline)) {
				put_byte(((r >> 2) & 1) ? t1[2u * blk + ((uint32_t)r >> 3)] : t0[2u * blk + ((uint32_t)r >> 3)], (uint32_t)r & 3u, v, px_);
			});
	}
	PXZ_STAMP(4);  // horizontal products + clamp
	uint32_t *s_row = s_wave + 256u;  // behind the planes
	if (th == 1u && __builtin_amdgcn_readfirstlane(my[321]) != 0u) {
		// ---- one stored row, copied by every window of the way up: un-premultiply it once, write it 32 times
		if (g == 0u) {
			uint32_t px = (t0[0] & 255u) | ((t0[1] & 255u) << 8) | ((t0[2] & 255u) << 16) | (t0[3] << 24);
			s_row[n] = unpremultiply(px);
		}
		tile_sync<1>();
		typedef uint32_t u32q __attribute__((ext_vector_type(4)));
		const uint4 v = *reinterpret_cast<const uint4 *>(s_row + 4u * (lane & 7u));
		const u32q w = {v.x, v.y, v.z, v.w};
#pragma unroll
		for (uint32_t k = 0; k < 4; ++k)
			__builtin_nontemporal_store(w, reinterpret_cast<u32q *>(dst + (size_t)(8u * k + (lane >> 3)) * a.pitch + 16u * (lane & 7u)));
		PXZ_STAMP(5);  // one-row replicate
		return;
	}
	// ---- vertical products, channel by channel
	const long ky_lo = *reinterpret_cast<const long *>(my + 2u * lane), ky_hi = *reinterpret_cast<const long *>(my + 128u + 2u * lane);
	v16i32 cy;
	{
		const uint4 *bp = reinterpret_cast<const uint4 *>(my + 288u + 16u * g);
#pragma unroll
		for (int q = 0; q < 4; ++q) {
			const uint4 b = bp[q];
			cy[4 * q] = (int)b.x; cy[4 * q + 1] = (int)b.y; cy[4 * q + 2] = (int)b.z; cy[4 * q + 3] = (int)b.w;
		}
	}
	uint32_t pix[16];
#pragma unroll
	for (int r = 0; r < 16; ++r) pix[r] = 0;
#pragma unroll
	for (uint32_t c = 0; c < 4; ++c) {
		const long tv = (long)(((unsigned long long)(t1[c] ^ 0x80808080u) << 32) | (unsigned long long)(t0[c] ^ 0x80808080u));
		const v16i32 lo = __builtin_amdgcn_mfma_i32_32x32x16_i8(ky_lo, tv, cy, 0, 0, 0);
		const v16i32 hi = __builtin_amdgcn_mfma_i32_32x32x16_i8(ky_hi, tv, zero, 0, 0, 0);
#pragma unroll
		for (int r = 0; r < 16; ++r) put_byte(pix[r], c, clamp_fixed(hi[r], lo[r], top_y), py);
	}
	// un-premultiplying is the identity at alpha 255: skipped when no lane of the wave holds anything else
	uint32_t alpha_and = 0xffffffffu;
#pragma unroll
	for (int r = 0; r < 16; ++r) alpha_and &= pix[r];
	if (__builtin_amdgcn_ballot_w64((alpha_and >> 24) != 255u) != 0ull) {
#pragma unroll
		for (int r = 0; r < 16; ++r) pix[r] = unpremultiply(pix[r]);
	}
	const uint32_t lane_off = 4u * g * a.pitch + 4u * n;
#pragma unroll
	for (uint32_t r = 0; r < 16; ++r)
		__builtin_nontemporal_store(pix[r], reinterpret_cast<uint32_t *>(dst + (size_t)xmf_row(0, r) * a.pitch + lane_off));
	PXZ_STAMP(6);  // vertical products + clamp + stores
}

// ResizeAlg::Nearest of a full 32x32 RGBA tile stored as tw x th, both powers of two: the source index
// floor((o + 0.5) * tw / 32) is o >> (5 - log2 tw), so no table is needed; a lane writes 4 adjacent pixels of 4 rows.
__device__ __forceinline__ void expand_tile_nearest_pow2(const ExpandArgs &a, uint32_t *s_src, uint32_t lane, uint32_t tw, uint32_t th,
                                                         const uint8_t *src, uint32_t first_px, uint8_t *dst)
{
	const uint32_t lw = 31u - (uint32_t)__builtin_clz(tw), sx = 5u - lw, sy = 5u - (31u - (uint32_t)__builtin_clz(th));
	const uint32_t npx = tw * th;
	for (uint32_t i = lane; i < npx; i += 64u) s_src[i] = i == lane ? first_px : reinterpret_cast<const uint32_t *>(src)[i];
	tile_sync<1>();
	typedef uint32_t u32q __attribute__((ext_vector_type(4)));
	const uint32_t q = lane & 7u;
#pragma unroll
	for (uint32_t k = 0; k < 4; ++k) {
		const uint32_t oy = 8u * k + (lane >> 3);
		const uint32_t *row = s_src + ((oy >> sy) << lw);
		u32q w;
		if (sx >= 2u) {
			const uint32_t v = row[q >> (sx - 2u)];
			w = u32q{v, v, v, v};
		} else if (sx == 1u) {
			const uint2 v = *reinterpret_cast<const uint2 *>(row + 2u * q);
			w = u32q{v.x, v.x, v.y, v.y};
		} else {
			const uint4 v = *reinterpret_cast<const uint4 *>(row + 4u * q);
			w = u32q{v.x, v.y, v.z, v.w};
		}
		__builtin_nontemporal_store(w, reinterpret_cast<u32q *>(dst + (size_t)oy * a.pitch + 16u * q));
	}
}

// A full 32x32 RGBA tile stored at full size (block.rs:279-281: clone): the slot's 4 KB straight into the frame, four 16-byte
// moves per lane, all four requested before the first is stored.  (Through the general form -- a dword per lane and round into
// LDS, sixteen load -> store rounds, then out again -- these tiles, an eighth of a typical frame, were 40 % of the kernel's time:
// tools/stamps_expand.py.)
__device__ __forceinline__ void expand_tile_clone32(const ExpandArgs &a, uint32_t lane, const uint8_t *src, uint8_t *dst)
{
	typedef uint32_t u32q __attribute__((ext_vector_type(4)));
	u32q v[4];
#pragma unroll
	for (uint32_t k = 0; k < 4; ++k) v[k] = __builtin_nontemporal_load(reinterpret_cast<const u32q *>(src) + 64u * k + lane);
#pragma unroll
	for (uint32_t k = 0; k < 4; ++k) {
		const uint32_t c = 64u * k + lane;  // 16-byte chunk c of the tile: row c / 8, columns 4 (c % 8) ..
		__builtin_nontemporal_store(v[k], reinterpret_cast<u32q *>(dst + (size_t)(c >> 3) * a.pitch + 16u * (c & 7u)));
	}
}

// ---------------------------------------------------------------------------
// expand16_kernel (round 4): Pixlzr::expand + to_image for 16x16 RGBA tiles, FOUR at a time -- a 2x2 group of full tiles is one
// 32x32 region of the frame, written with the store pattern of the 32x32 instance (16 dwords per lane, half-waves covering 128-byte
// row segments).  One wave per group, persistent, groups dealt by an LDS ticket counter; sizes and the first 64 pixels of each of
// the four slots are requested one group ahead.  Per tile of the group:
//   stored 16x16 (block.rs:279-281, clone)           its 1 KB slot straight into the frame, one 16-byte move per lane
//   stored tw x th, both in {1, 2, 4, 8}, Nearest    source index o >> (4 - log2 size), from the staged pixels
//   the same, a convolution                          ALL such tiles of the group in one set of matrix-core products (below)
//   anything else (16 x n, n x 16, sizes a foreign file may hold, empty, tiles of partial groups)  appended to a list that
//                                                    expand_kernel takes in a second launch (status[1] counts it)
// The convolutions (same integers as expand_kernel's vector form, block.rs:273-334): the four stored tiles are ONE virtual 16x16
// source -- tile (dx, dy) at columns 8 dx .., rows 8 dy .. of premultiplied byte planes [c][16][16] -- and the weight operands are
// block-diagonal, as in resample_group16_mfma on the encode side: a k slot (kg, j) of v_mfma_i32_32x32x16_i8 stands for sample
// 4 kg + j of the FIRST tile (j < 4) or of the SECOND (j >= 4).
//   horizontal, tile row dy: rows (c, y < 8) of that row's planes times B[k][n] = Kx of tile (n >> 4, dy) for output column n & 15 in
//               the slots of its own tile: a lane reads its own tile's table, bias and precision.  The accumulator of lane (n, g)
//               holds, per channel, rows y = 4 g + j of that tile row as four bytes -- for both tile rows together exactly a B
//               operand of the vertical product (first tile = top, second = bottom): T never leaves the registers;
//   vertical, per channel: A[m][k] = Ky of the tile in row m >> 4 for output row m & 15.  Left and right tiles have their own
//               stored heights, so two products into one accumulator: the left tiles' weights times T with columns 16..31 zeroed,
//               plus the right tiles' times T with columns 0..15 zeroed (a zero byte is p - 128 = 0: it adds nothing).
// 20 MFMAs per group.  Tiles outside the mask ride along with any valid table; their part of the region is not stored.
// ---------------------------------------------------------------------------
constexpr uint32_t kX16Wave = 256;  // dwords of LDS per wave: the byte planes [c][16][16] / the staged pixels [tile][64] of Nearest

__global__ void __launch_bounds__(1024) expand16_kernel(const ExpandArgs a)
{
	extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
	const uint32_t wpb = blockDim.x / 64u, sub = threadIdx.x / 64u, lane = threadIdx.x % 64u;
	const uint32_t xmf_dw = a.xmf16 ? kXmf16Levels * kXmf16Dw : 0u;
	for (uint32_t i = threadIdx.x; i < xmf_dw / 4u; i += blockDim.x)
		reinterpret_cast<uint4 *>(lds)[i] = reinterpret_cast<const uint4 *>(a.xmf16)[i];
	const uint32_t *s_xmf = lds;
	uint32_t *s_ticket = lds + xmf_dw + wpb * kX16Wave;
	if (threadIdx.x == 0) *s_ticket = wpb;
	__syncthreads();
	uint32_t *s_wave = lds + xmf_dw + sub * kX16Wave;
	const uint32_t gcols = (a.cols + 1u) >> 1, grows = (a.rows + 1u) >> 1, gpf = gcols * grows;
	const uint32_t n_frames = a.n_tiles / a.tiles_per_frame, n_groups = n_frames * gpf;
	const uint32_t full_cols = a.edge_w == 16u ? a.cols : a.cols - 1u, full_rows = a.edge_h == 16u ? a.rows : a.rows - 1u;
	struct Place {
		uint32_t t00;  // tile (0, 0) of the group
		uint32_t gx, gy, frame;
		bool full;     // all four tiles exist and are 16x16
	};
	auto place_of = [&](uint32_t grp) -> Place {
		Place p{0, 0, 0, 0, false};
		if (grp >= n_groups) return p;
		p.frame = fastdiv(grp, a.div_gpf);
		const uint32_t r = grp - p.frame * gpf;
		p.gy = fastdiv(r, a.div_gcols);
		p.gx = r - p.gy * gcols;
		p.t00 = p.frame * a.tiles_per_frame + (2u * p.gy) * a.cols + 2u * p.gx;
		p.full = 2u * p.gx + 1u < full_cols && 2u * p.gy + 1u < full_rows;
		return p;
	};
	auto group_of = [&](uint32_t tk) -> uint32_t {
		const unsigned long long g = (unsigned long long)blockIdx.x + (unsigned long long)tk * gridDim.x;
		return g < (unsigned long long)n_groups ? (uint32_t)g : 0xffffffffu;
	};
	auto list_tile = [&](uint32_t t) {  // (wave-uniform; rare: one atomic per tile)
		if (lane == 0) a.list[atomicAdd(a.status + 1, 1u)] = t;
	};
	// requested one group ahead: lane k < 4 the stored size of tile k, every lane dword `lane` of each of the four slots
	uint32_t p_tw = 0, p_th = 0, p_px[4] = {0, 0, 0, 0};
	auto prefetch = [&](uint32_t grp) {
		if (grp == 0xffffffffu) return;
		const Place p = place_of(grp);
		if (!p.full) return;
		const uint32_t tk = p.t00 + (lane & 1u) + ((lane >> 1) & 1u) * a.cols;  // (lanes >= 4 repeat the four)
		p_tw = a.tile_w[tk];
		p_th = a.tile_h[tk];
#pragma unroll
		for (uint32_t k = 0; k < 4; ++k)
			p_px[k] = reinterpret_cast<const uint32_t *>(a.slots + (size_t)(p.t00 + (k & 1u) + (k >> 1) * a.cols) * a.slot_bytes)[lane];
	};
	uint32_t grp = group_of(sub);
	prefetch(grp);
	auto put_byte = [&](uint32_t &d, uint32_t j, uint32_t v, uint32_t sh) __attribute__((always_inline)) {
		if (j == 0) put_byte_shr<0>(d, v, sh);
		else if (j == 1) put_byte_shr<1>(d, v, sh);
		else if (j == 2) put_byte_shr<2>(d, v, sh);
		else put_byte_shr<3>(d, v, sh);
	};
	typedef uint32_t u32q __attribute__((ext_vector_type(4), aligned(4)));  // (frame rows and slots: dword aligned)
	while (grp != 0xffffffffu) {
		uint32_t nt = 0;
		if (lane == 0) nt = atomicAdd(s_ticket, 1u);
		const uint32_t grp_next = group_of(__builtin_amdgcn_readfirstlane(nt));
		const Place pl = place_of(grp);
		if (!pl.full) {
			// a partial group (ragged edge, odd tile counts): its tiles one by one to expand_kernel
#pragma unroll
			for (uint32_t k = 0; k < 4; ++k)
				if (2u * pl.gx + (k & 1u) < a.cols && 2u * pl.gy + (k >> 1) < a.rows) list_tile(pl.t00 + (k & 1u) + (k >> 1) * a.cols);
			prefetch(grp_next);
			grp = grp_next;
			continue;
		}
		uint32_t tw[4], th[4], px[4];
#pragma unroll
		for (uint32_t k = 0; k < 4; ++k) {
			tw[k] = (uint32_t)__builtin_amdgcn_readlane((int)p_tw, k);
			th[k] = (uint32_t)__builtin_amdgcn_readlane((int)p_th, k);
			px[k] = p_px[k];
		}
		uint8_t *dst = a.dst + (size_t)pl.frame * a.frame_stride + (size_t)(pl.gy * 32u) * a.pitch + (size_t)(pl.gx * 32u) * 4u;
		// ---- per tile: clone / eligible for the group forms / listed
		uint32_t mask = 0, clones = 0;  // tiles that take the group form (stored tw x th, both in {1, 2, 4, 8}); tiles stored at 16x16
#pragma unroll
		for (uint32_t k = 0; k < 4; ++k) {
			const bool small = tw[k] >= 1u && tw[k] <= 8u && th[k] >= 1u && th[k] <= 8u && (tw[k] & (tw[k] - 1u)) == 0u && (th[k] & (th[k] - 1u)) == 0u;
			if (tw[k] == 16u && th[k] == 16u) clones |= 1u << k;
			else if (small && (a.filter == 0u || xmf_dw != 0u)) mask |= 1u << k;
			else list_tile(pl.t00 + (k & 1u) + (k >> 1) * a.cols);
		}
		const uint32_t near = a.filter == 0u ? mask : 0u;
		const uint32_t dxl = (lane >> 2) & 1u, q = lane & 7u;
		// the clones' bytes are requested BEFORE the next group's prefetch: loads come back in order, so behind it their wait would be
		// a wait for the whole prefetch -- a full memory round trip per group with a clone in it
		u32q wc[4] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
		if (clones != 0u) {
#pragma unroll
			for (uint32_t k = 0; k < 4; ++k) {
				const uint32_t dy = k >> 1, oy = 8u * k + (lane >> 3), tk = dxl + 2u * dy;
				if ((clones >> tk) & 1u) {
					const uint32_t t = pl.t00 + dxl + dy * a.cols;
					wc[k] = __builtin_nontemporal_load(reinterpret_cast<const u32q *>(a.slots + (size_t)t * a.slot_bytes) + (oy & 15u) * 4u + (q & 3u));
				}
			}
		}
		prefetch(grp_next);  // in flight while this group is expanded
		if ((clones | near) != 0u) {
			// ---- clones (block.rs:279-281: the slot's own bytes) and ResizeAlg::Nearest (source index floor((o + 0.5) * size / 16) =
			// o >> (4 - log2 size), from the staged pixels), written in the pattern of whole frame rows: a lane makes the 16 bytes
			// of quad q = lane & 7 of row 8 k + (lane >> 3) of the 32x32 region -- the tile is a matter of the lane (a store
			// instruction covers eight whole 128-byte rows; tile by tile it was sixteen half rows)
			if (near != 0u) {
#pragma unroll
				for (uint32_t k = 0; k < 4; ++k) s_wave[64u * k + lane] = px[k];
				tile_sync<1>();
			}
			uint32_t lwk[4], syk[4];  // (scalar: log2 of the stored width, row shift)
#pragma unroll
			for (uint32_t k = 0; k < 4; ++k) {
				lwk[k] = 31u - (uint32_t)__builtin_clz(tw[k] | 1u);
				syk[k] = 4u - (31u - (uint32_t)__builtin_clz(th[k] | 1u));
			}
#pragma unroll
			for (uint32_t k = 0; k < 4; ++k) {
				const uint32_t dy = k >> 1, oy = 8u * k + (lane >> 3), tk = dxl + 2u * dy;
				const bool is_clone = ((clones >> tk) & 1u) != 0u, is_near = ((near >> tk) & 1u) != 0u;
				u32q w = wc[k];
				if (is_near) {
					const uint32_t lw = dxl ? lwk[2u * dy + 1u] : lwk[2u * dy], sy = dxl ? syk[2u * dy + 1u] : syk[2u * dy], sx = 4u - lw;
					const uint32_t *row = s_wave + 64u * tk + (((oy & 15u) >> sy) << lw);
					const uint32_t x = 4u * (q & 3u);
					w = u32q{row[x >> sx], row[(x + 1u) >> sx], row[(x + 2u) >> sx], row[(x + 3u) >> sx]};
				}
				if (is_clone || is_near) __builtin_nontemporal_store(w, reinterpret_cast<u32q *>(dst + (size_t)oy * a.pitch + 16u * q));
			}
			if (near != 0u) tile_sync<1>();
		}
		if (mask != 0u && a.filter != 0u) {
			// ---- the convolutions of the group's eligible tiles on the matrix cores
			// stored pixels -> premultiplied byte planes [c][8 dy + y][8 dx + x]
			uint8_t *s_pl = reinterpret_cast<uint8_t *>(s_wave);
			bool any_alpha = false;
#pragma unroll
			for (uint32_t k = 0; k < 4; ++k)
				any_alpha = any_alpha || (((mask >> k) & 1u) && lane < tw[k] * th[k] && (px[k] >> 24) != 255u);
			const bool premul = __builtin_amdgcn_ballot_w64(any_alpha) != 0ull;
#pragma unroll
			for (uint32_t k = 0; k < 4; ++k) {
				if (((mask >> k) & 1u) && lane < tw[k] * th[k]) {
					uint32_t v = px[k];
					if (premul) v = premultiply(v);  // fir: U8x4 is alpha-premultiplied before a convolution
					const uint32_t lw = 31u - (uint32_t)__builtin_clz(tw[k]);
					uint8_t *d = s_pl + (8u * (k >> 1) + (lane >> lw)) * 16u + 8u * (k & 1u) + (lane & (tw[k] - 1u));
					d[0] = (uint8_t)v;
					d[256] = (uint8_t)(v >> 8);
					d[512] = (uint8_t)(v >> 16);
					d[768] = (uint8_t)(v >> 24);
				}
			}
			tile_sync<1>();
			// a valid table for every view (tiles outside the mask: that of any eligible tile)
			uint32_t lx[4], ly[4];
			{
				uint32_t ax = 0, ay = 0;
#pragma unroll
				for (uint32_t k = 0; k < 4; ++k)
					if ((mask >> k) & 1u) { ax = 31u - (uint32_t)__builtin_clz(tw[k]); ay = 31u - (uint32_t)__builtin_clz(th[k]); }
#pragma unroll
				for (uint32_t k = 0; k < 4; ++k) {
					const bool in = ((mask >> k) & 1u) != 0u;
					lx[k] = (in ? 31u - (uint32_t)__builtin_clz(tw[k]) : ax) * kXmf16Dw;
					ly[k] = (in ? 31u - (uint32_t)__builtin_clz(th[k]) : ay) * kXmf16Dw;
				}
			}
			const uint32_t n = lane & 31u, g = lane >> 5, o = n & 15u;
			const bool second = n >= 16u;  // as a column: a right tile; as a row of a weight operand: a bottom tile
			auto operand = [&](uint32_t w) -> long { return (long)(second ? (unsigned long long)w << 32 : (unsigned long long)w); };
			v16i32 zero;
#pragma unroll
			for (int r = 0; r < 16; ++r) zero[r] = 0;
			// ---- horizontal products, one per tile row: t[dy][c] = rows y = 4 g + j of column n as bytes
			uint32_t t[2][4];
#pragma unroll
			for (uint32_t dy = 0; dy < 2; ++dy) {
				const uint32_t *tb = s_xmf + (second ? lx[2u * dy + 1u] : lx[2u * dy]);
				const long k_lo = operand(tb[o * 2u + g]), k_hi = operand(tb[32u + o * 2u + g]);
				const int32_t bx = (int32_t)tb[64u + o];
				const uint32_t px_ = tb[96];
				const int32_t top_x = (int32_t)((256u << px_) - 1u);
				v16i32 cx;
#pragma unroll
				for (int r = 0; r < 16; ++r) cx[r] = bx;
				// row (c, y) = n of this tile row's planes: columns 4 g .. (left tile), 8 + 4 g .. (right tile)
				const uint32_t *row = reinterpret_cast<const uint32_t *>(s_pl + (n >> 3) * 256u + (8u * dy + (n & 7u)) * 16u + 4u * g);
				const uint32_t a0 = row[0] ^ 0x80808080u, a1 = row[2] ^ 0x80808080u;
				const long av = (long)(((unsigned long long)a1 << 32) | (unsigned long long)a0);
				const v16i32 lo = __builtin_amdgcn_mfma_i32_32x32x16_i8(av, k_lo, cx, 0, 0, 0);
				const v16i32 hi = __builtin_amdgcn_mfma_i32_32x32x16_i8(av, k_hi, zero, 0, 0, 0);
#pragma unroll
				for (uint32_t c = 0; c < 4; ++c) t[dy][c] = 0u;
#pragma unroll
				for (int r = 0; r < 16; ++r) put_byte(t[dy][r >> 2], (uint32_t)r & 3u, clamp_fixed_v(hi[r], lo[r], top_x), px_);  // reg = 4 c + j
			}
			// ---- vertical products, channel by channel
			long ky_lo[2], ky_hi[2];
#pragma unroll
			for (uint32_t side = 0; side < 2; ++side) {
				const uint32_t *tb = s_xmf + (second ? ly[2u + side] : ly[side]);
				ky_lo[side] = operand(tb[o * 2u + g]);
				ky_hi[side] = operand(tb[32u + o * 2u + g]);
			}
			// this lane's outputs: column n, rows (r & 3) + 8 (r >> 2) + 4 g: regs 0..7 the top tile of its column half, 8..15 the bottom one
			const uint32_t *tt = s_xmf + (second ? ly[1] : ly[0]), *tbm = s_xmf + (second ? ly[3] : ly[2]);
			const uint32_t py_t = tt[96], py_b = tbm[96];
			const int32_t top_t = (int32_t)((256u << py_t) - 1u), top_b = (int32_t)((256u << py_b) - 1u);
			v16i32 cy;
			{
				const uint4 b0 = *reinterpret_cast<const uint4 *>(tt + 80u + 8u * g), b1 = *reinterpret_cast<const uint4 *>(tt + 84u + 8u * g);
				const uint4 b2 = *reinterpret_cast<const uint4 *>(tbm + 80u + 8u * g), b3 = *reinterpret_cast<const uint4 *>(tbm + 84u + 8u * g);
				cy[0] = (int)b0.x; cy[1] = (int)b0.y; cy[2] = (int)b0.z; cy[3] = (int)b0.w;
				cy[4] = (int)b1.x; cy[5] = (int)b1.y; cy[6] = (int)b1.z; cy[7] = (int)b1.w;
				cy[8] = (int)b2.x; cy[9] = (int)b2.y; cy[10] = (int)b2.z; cy[11] = (int)b2.w;
				cy[12] = (int)b3.x; cy[13] = (int)b3.y; cy[14] = (int)b3.z; cy[15] = (int)b3.w;
			}
			uint32_t pix[16];
#pragma unroll
			for (int r = 0; r < 16; ++r) pix[r] = 0;
#pragma unroll
			for (uint32_t c = 0; c < 4; ++c) {
				const uint32_t b0 = t[0][c] ^ 0x80808080u, b1 = t[1][c] ^ 0x80808080u;
				const long tl = (long)(((unsigned long long)(second ? 0u : b1) << 32) | (unsigned long long)(second ? 0u : b0));
				const long tr = (long)(((unsigned long long)(second ? b1 : 0u) << 32) | (unsigned long long)(second ? b0 : 0u));
				v16i32 lo = __builtin_amdgcn_mfma_i32_32x32x16_i8(ky_lo[0], tl, cy, 0, 0, 0);
				lo = __builtin_amdgcn_mfma_i32_32x32x16_i8(ky_lo[1], tr, lo, 0, 0, 0);
				v16i32 hi = __builtin_amdgcn_mfma_i32_32x32x16_i8(ky_hi[0], tl, zero, 0, 0, 0);
				hi = __builtin_amdgcn_mfma_i32_32x32x16_i8(ky_hi[1], tr, hi, 0, 0, 0);
#pragma unroll
				for (int r = 0; r < 16; ++r) put_byte(pix[r], c, clamp_fixed_v(hi[r], lo[r], r < 8 ? top_t : top_b), r < 8 ? py_t : py_b);
			}
			// un-premultiplying is the identity at alpha 255: skipped when no lane of the wave holds anything else
			uint32_t alpha_and = 0xffffffffu;
#pragma unroll
			for (int r = 0; r < 16; ++r) alpha_and &= pix[r];
			if (premul || __builtin_amdgcn_ballot_w64((alpha_and >> 24) != 255u) != 0ull) {
#pragma unroll
				for (int r = 0; r < 16; ++r) pix[r] = unpremultiply(pix[r]);
			}
			const bool ok_t = ((mask >> (second ? 1u : 0u)) & 1u) != 0u, ok_b = ((mask >> (second ? 3u : 2u)) & 1u) != 0u;
			uint8_t *lane_dst = dst + (size_t)(4u * g) * a.pitch + 4u * n;
#pragma unroll
			for (uint32_t r = 0; r < 16; ++r) {
				if (r < 8 ? ok_t : ok_b)
					__builtin_nontemporal_store(pix[r], reinterpret_cast<uint32_t *>(lane_dst + (size_t)xmf_row(0, r) * a.pitch));
			}
			tile_sync<1>();  // the next group restages the planes
		}
		grp = grp_next;
	}
}

// ---------------------------------------------------------------------------
// expand64_kernel (round 4): Pixlzr::expand + to_image for 64x64 RGBA tiles -- the reference CLI's default block size
// (src/bin/main.rs:19) -- one wave per tile, persistent, tiles dealt by an LDS ticket counter, stored size and first pixels
// requested one tile ahead.  Per full tile:
//   stored 64x64 (block.rs:279-281, clone)               its 16 KB slot straight into the frame rows, four 16-byte moves in flight
//   stored tw x th, powers of two, Nearest               source index o >> (6 - log2 size) from the staged pixels
//   stored tw x th, both in {1 .. 32}, a convolution     the matrix cores, as expand_tile_mfma does it for 32x32 tiles, looped:
//       for each half qx of the 64 output columns: T[(c, y)][ox] = clip8(sum_x P_c[y][x] Kx[x][ox]) in NBLK row blocks of 32
//       (1: th <= 8, rows (c, y < 8); 2: th = 16; 4: th = 32) and one or two steps of 16 source columns; the clamped bytes stay in
//       registers in the k-slot order of the vertical product (xmf_src), whose weights are the same table; then for each half qy
//       of the output rows and each channel: O_c[oy][ox] = clip8(sum_y Ky[oy][y] T_c[y][ox]), one or two steps of 16 stored rows,
//       un-premultiplied and written as 16 dwords per lane (half-waves cover 128-byte row segments).  Same integers as
//       expand_kernel's vector form (block.rs:273-334).
//   stored 64 x th or tw x 64 (one axis kept)             the one product that axis needs (expand64_vonly / expand64_honly)
//   anything else (sizes a foreign file may hold, empty, ragged-edge tiles)
//                                                        appended to the list expand_kernel takes in a second launch
// ---------------------------------------------------------------------------
// dwords of LDS per wave.  RGBA: byte planes [c][max(th, 8)][max(tw, 16)] / staged pixels of Nearest (<= 64 x 32).  RGB (C = 3, round 4:
// what image::open yields for most photographs): three planes, and 1024 dwords through which a quadrant's pixels are turned round
// so that a lane writes four adjacent ones as twelve bytes
constexpr uint32_t kX64Wave = 2048, kX64Wave3 = 1536 + 1024;

// One 32x32 quadrant of a 64x64 tile, first column col0 and first row row0: lane (n, g) holds column n of the rows xmf_row(g, r).
// RGBA: 16 dword stores per lane (half-waves cover 128-byte row segments).  RGB: through s_out.
template <int C>
__device__ __forceinline__ void x64_store_quadrant(const ExpandArgs &a, const uint32_t (&pix)[16], uint32_t lane, uint32_t col0, uint32_t row0,
                                                   uint8_t *dst, uint32_t *s_out)
{
	const uint32_t n = lane & 31u, g = lane >> 5;
	if constexpr (C == 4) {
		uint8_t *lane_dst = dst + (size_t)(row0 + 4u * g) * a.pitch + 4u * (col0 + n);
#pragma unroll
		for (uint32_t r = 0; r < 16; ++r)
			__builtin_nontemporal_store(pix[r], reinterpret_cast<uint32_t *>(lane_dst + (size_t)xmf_row(0, r) * a.pitch));
	} else {
#pragma unroll
		for (uint32_t r = 0; r < 16; ++r) s_out[(xmf_row(0, r) + 4u * g) * 32u + n] = pix[r];
		tile_sync<1>();
		typedef uint32_t u32_a1 __attribute__((aligned(1)));
#pragma unroll
		for (uint32_t k = 0; k < 4; ++k) {
			const uint32_t i = lane + 64u * k, row = i >> 3, q = i & 7u;
			const uint4 v = *reinterpret_cast<const uint4 *>(s_out + row * 32u + 4u * q);
			u32_a1 *o = reinterpret_cast<u32_a1 *>(dst + (size_t)(row0 + row) * a.pitch + 3u * (col0 + 4u * q));
			o[0] = (v.x & 0xffffffu) | (v.y << 24);
			o[1] = ((v.y >> 8) & 0xffffu) | (v.z << 16);
			o[2] = ((v.z >> 16) & 0xffu) | (v.w << 8);
		}
		tile_sync<1>();  // the next quadrant reuses s_out
	}
}

template <int NBLK, int C>
__device__ __forceinline__ void expand64_conv(const ExpandArgs &a, const uint32_t *s_xmf, const uint8_t *s_pl, uint32_t lane, uint32_t lw,
                                              uint32_t lh, uint32_t P, uint8_t *dst, uint32_t *s_out)
{
	constexpr uint32_t KSY = NBLK == 4 ? 2u : 1u;                          // steps of 16 stored rows
	constexpr uint32_t LTHP = NBLK == 1 ? 3u : (NBLK == 2 ? 4u : 5u);     // log2 of the rows a channel plane holds
	const uint32_t n = lane & 31u, g = lane >> 5;
	const uint32_t *mx = s_xmf + lw * kXmf64Dw, *my = s_xmf + lh * kXmf64Dw;
	const uint32_t ksx = lw == 5u ? 2u : 1u;                              // steps of 16 stored columns
	const uint32_t plane = P << LTHP;
	const uint32_t px_ = __builtin_amdgcn_readfirstlane(mx[1152]), py = __builtin_amdgcn_readfirstlane(my[1152]);
	const int32_t top_x = (int32_t)((256u << px_) - 1u), top_y = (int32_t)((256u << py) - 1u);
	auto put_byte = [&](uint32_t &d, uint32_t j, uint32_t v, uint32_t sh) __attribute__((always_inline)) {
		if (j == 0) put_byte_shr<0>(d, v, sh);
		else if (j == 1) put_byte_shr<1>(d, v, sh);
		else if (j == 2) put_byte_shr<2>(d, v, sh);
		else put_byte_shr<3>(d, v, sh);
	};
	v16i32 zero;
#pragma unroll
	for (int r = 0; r < 16; ++r) zero[r] = 0;
#pragma unroll 1
	for (uint32_t qx = 0; qx < 2; ++qx) {
		// ---- horizontal: this half's 32 output columns
		long kxl[2], kxh[2];
#pragma unroll
		for (uint32_t st = 0; st < 2; ++st) {
			const uint32_t *w = mx + ((qx * 2u + (st < ksx ? st : 0u)) * 2u) * 128u + 2u * lane;
			kxl[st] = *reinterpret_cast<const long *>(w);
			kxh[st] = *reinterpret_cast<const long *>(w + 128u);
		}
		const int32_t bx = (int32_t)mx[1024u + 32u * qx + n];
		uint32_t T[4][KSY][2];  // [channel][step of 16 rows][rows 4 g + j | 8 + 4 g + j] as bytes: the vertical product's B operand
#pragma unroll
		for (uint32_t c = 0; c < 4; ++c)
#pragma unroll
			for (uint32_t st = 0; st < KSY; ++st) T[c][st][0] = T[c][st][1] = 0u;
#pragma unroll
		for (uint32_t b = 0; b < (uint32_t)NBLK; ++b) {
			if (C == 3 && NBLK == 4 && b == 3u) continue;  // (RGB: no fourth channel)
			const uint32_t G = 32u * b + n, c = G >> LTHP, y = G & ((1u << LTHP) - 1u);  // row n of this block: channel c, stored row y
			const uint32_t *row = reinterpret_cast<const uint32_t *>(s_pl + c * plane + y * P + 4u * g);
			v16i32 lo, hi = zero;  // (the bias: sixteen moves per block, not sixteen registers held across the tile)
#pragma unroll
			for (int r = 0; r < 16; ++r) lo[r] = bx;
#pragma unroll
			for (uint32_t st = 0; st < 2; ++st) {
				if (st < ksx) {
					const uint32_t a0 = row[4u * st] ^ 0x80808080u, a1 = row[4u * st + 2u] ^ 0x80808080u;  // columns 16 st + 4 g + j | + 8
					const long av = (long)(((unsigned long long)a1 << 32) | (unsigned long long)a0);
					lo = __builtin_amdgcn_mfma_i32_32x32x16_i8(av, kxl[st], lo, 0, 0, 0);
					hi = __builtin_amdgcn_mfma_i32_32x32x16_i8(av, kxh[st], hi, 0, 0, 0);
				}
			}
#pragma unroll
			for (uint32_t r = 0; r < 16; ++r) {
				const uint32_t q4 = r >> 2;  // four accumulator registers = rows 8 q4 + 4 g + j of the block
				const uint32_t cc = NBLK == 1 ? q4 : (NBLK == 2 ? 2u * b + (q4 >> 1) : b);
				const uint32_t st = NBLK == 4 ? q4 >> 1 : 0u, h = NBLK == 1 ? 0u : q4 & 1u;
				if (cc < (uint32_t)C) put_byte(T[cc][st][h], r & 3u, clamp_fixed(hi[r], lo[r], top_x), px_);
			}
		}
		// ---- vertical: the two 32x32 quadrants of this column half
#pragma unroll 1
		for (uint32_t qy = 0; qy < 2; ++qy) {
			long kyl[KSY], kyh[KSY];
#pragma unroll
			for (uint32_t st = 0; st < KSY; ++st) {
				const uint32_t *w = my + ((qy * 2u + st) * 2u) * 128u + 2u * lane;
				kyl[st] = *reinterpret_cast<const long *>(w);
				kyh[st] = *reinterpret_cast<const long *>(w + 128u);
			}
			const uint4 *bp = reinterpret_cast<const uint4 *>(my + 1088u + (qy * 2u + g) * 16u);  // the biases in accumulator order
			uint32_t pix[16];
#pragma unroll
			for (int r = 0; r < 16; ++r) pix[r] = 0;
#pragma unroll
			for (uint32_t c = 0; c < (uint32_t)C; ++c) {
				v16i32 lo, hi = zero;  // (the biases: four LDS reads per channel, not sixteen registers held across the quadrant)
#pragma unroll
				for (int q = 0; q < 4; ++q) {
					const uint4 bb = bp[q];
					lo[4 * q] = (int)bb.x; lo[4 * q + 1] = (int)bb.y; lo[4 * q + 2] = (int)bb.z; lo[4 * q + 3] = (int)bb.w;
				}
#pragma unroll
				for (uint32_t st = 0; st < KSY; ++st) {
					const long tv = (long)(((unsigned long long)(T[c][st][1] ^ 0x80808080u) << 32) | (unsigned long long)(T[c][st][0] ^ 0x80808080u));
					lo = __builtin_amdgcn_mfma_i32_32x32x16_i8(kyl[st], tv, lo, 0, 0, 0);
					hi = __builtin_amdgcn_mfma_i32_32x32x16_i8(kyh[st], tv, hi, 0, 0, 0);
				}
#pragma unroll
				for (uint32_t r = 0; r < 16; ++r) put_byte(pix[r], c, clamp_fixed(hi[r], lo[r], top_y), py);
			}
			if constexpr (C == 4) {
				// un-premultiplying is the identity at alpha 255: skipped when no lane of the wave holds anything else
				uint32_t alpha_and = 0xffffffffu;
#pragma unroll
				for (int r = 0; r < 16; ++r) alpha_and &= pix[r];
				if (__builtin_amdgcn_ballot_w64((alpha_and >> 24) != 255u) != 0ull) {
#pragma unroll
					for (int r = 0; r < 16; ++r) pix[r] = unpremultiply(pix[r]);
				}
			}
			x64_store_quadrant<C>(a, pix, lane, 32u * qx, 32u * qy, dst, s_out);
		}
	}
}

// The one-pass classes of a 64x64 tile (fir resizes an axis only where its size changes, block.rs:292-322).
// Width kept (stored 64 x th, th <= 32): the vertical product alone; its B operand -- four stored rows of a column as bytes -- is
// gathered from the premultiplied planes [c][th][64].
template <int KSY, int C>
__device__ __forceinline__ void expand64_vonly(const ExpandArgs &a, const uint32_t *s_xmf, const uint8_t *s_pl, uint32_t lane, uint32_t lh,
                                               uint32_t plane, uint8_t *dst, uint32_t *s_out)
{
	const uint32_t n = lane & 31u, g = lane >> 5;
	const uint32_t *my = s_xmf + lh * kXmf64Dw;
	const uint32_t py = __builtin_amdgcn_readfirstlane(my[1152]);
	const int32_t top_y = (int32_t)((256u << py) - 1u);
	auto put_byte = [&](uint32_t &d, uint32_t j, uint32_t v, uint32_t sh) __attribute__((always_inline)) {
		if (j == 0) put_byte_shr<0>(d, v, sh);
		else if (j == 1) put_byte_shr<1>(d, v, sh);
		else if (j == 2) put_byte_shr<2>(d, v, sh);
		else put_byte_shr<3>(d, v, sh);
	};
	v16i32 zero;
#pragma unroll
	for (int r = 0; r < 16; ++r) zero[r] = 0;
#pragma unroll 1
	for (uint32_t qx = 0; qx < 2; ++qx) {
		uint32_t T[4][KSY][2];
#pragma unroll
		for (uint32_t c = 0; c < (uint32_t)C; ++c)
#pragma unroll
			for (uint32_t st = 0; st < (uint32_t)KSY; ++st)
#pragma unroll
				for (uint32_t h = 0; h < 2; ++h) {
					const uint8_t *col = s_pl + c * plane + (16u * st + 8u * h + 4u * g) * 64u + 32u * qx + n;  // rows .. + 3 of column 32 qx + n
					T[c][st][h] = ((uint32_t)col[0] | ((uint32_t)col[64] << 8) | ((uint32_t)col[128] << 16) | ((uint32_t)col[192] << 24)) ^ 0x80808080u;
				}
#pragma unroll 1
		for (uint32_t qy = 0; qy < 2; ++qy) {
			long kyl[KSY], kyh[KSY];
#pragma unroll
			for (uint32_t st = 0; st < (uint32_t)KSY; ++st) {
				const uint32_t *w = my + ((qy * 2u + st) * 2u) * 128u + 2u * lane;
				kyl[st] = *reinterpret_cast<const long *>(w);
				kyh[st] = *reinterpret_cast<const long *>(w + 128u);
			}
			const uint4 *bp = reinterpret_cast<const uint4 *>(my + 1088u + (qy * 2u + g) * 16u);  // the biases in accumulator order
			uint32_t pix[16];
#pragma unroll
			for (int r = 0; r < 16; ++r) pix[r] = 0;
#pragma unroll
			for (uint32_t c = 0; c < (uint32_t)C; ++c) {
				v16i32 lo, hi = zero;
#pragma unroll
				for (int q = 0; q < 4; ++q) {
					const uint4 bb = bp[q];
					lo[4 * q] = (int)bb.x; lo[4 * q + 1] = (int)bb.y; lo[4 * q + 2] = (int)bb.z; lo[4 * q + 3] = (int)bb.w;
				}
#pragma unroll
				for (uint32_t st = 0; st < (uint32_t)KSY; ++st) {
					const long tv = (long)(((unsigned long long)T[c][st][1] << 32) | (unsigned long long)T[c][st][0]);
					lo = __builtin_amdgcn_mfma_i32_32x32x16_i8(kyl[st], tv, lo, 0, 0, 0);
					hi = __builtin_amdgcn_mfma_i32_32x32x16_i8(kyh[st], tv, hi, 0, 0, 0);
				}
#pragma unroll
				for (uint32_t r = 0; r < 16; ++r) put_byte(pix[r], c, clamp_fixed(hi[r], lo[r], top_y), py);
			}
			if constexpr (C == 4) {
				// un-premultiplying is the identity at alpha 255: skipped when no lane of the wave holds anything else
				uint32_t alpha_and = 0xffffffffu;
#pragma unroll
				for (int r = 0; r < 16; ++r) alpha_and &= pix[r];
				if (__builtin_amdgcn_ballot_w64((alpha_and >> 24) != 255u) != 0ull) {
#pragma unroll
					for (int r = 0; r < 16; ++r) pix[r] = unpremultiply(pix[r]);
				}
			}
			x64_store_quadrant<C>(a, pix, lane, 32u * qx, 32u * qy, dst, s_out);
		}
	}
}

// Height kept (stored tw x 64, tw <= 32): the horizontal product alone, planes [c][64][P]; a block of rows is 32 output rows of
// one channel, the four channels of a (column half, row half) make its pixels.
template <int C>
__device__ __forceinline__ void expand64_honly(const ExpandArgs &a, const uint32_t *s_xmf, const uint8_t *s_pl, uint32_t lane, uint32_t lw,
                                               uint32_t P, uint8_t *dst, uint32_t *s_out)
{
	const uint32_t n = lane & 31u, g = lane >> 5;
	const uint32_t *mx = s_xmf + lw * kXmf64Dw;
	const uint32_t ksx = lw == 5u ? 2u : 1u;
	const uint32_t plane = P * 64u;
	const uint32_t px_ = __builtin_amdgcn_readfirstlane(mx[1152]);
	const int32_t top_x = (int32_t)((256u << px_) - 1u);
	auto put_byte = [&](uint32_t &d, uint32_t j, uint32_t v, uint32_t sh) __attribute__((always_inline)) {
		if (j == 0) put_byte_shr<0>(d, v, sh);
		else if (j == 1) put_byte_shr<1>(d, v, sh);
		else if (j == 2) put_byte_shr<2>(d, v, sh);
		else put_byte_shr<3>(d, v, sh);
	};
	v16i32 zero;
#pragma unroll
	for (int r = 0; r < 16; ++r) zero[r] = 0;
#pragma unroll 1
	for (uint32_t qx = 0; qx < 2; ++qx) {
		long kxl[2], kxh[2];
#pragma unroll
		for (uint32_t st = 0; st < 2; ++st) {
			const uint32_t *w = mx + ((qx * 2u + (st < ksx ? st : 0u)) * 2u) * 128u + 2u * lane;
			kxl[st] = *reinterpret_cast<const long *>(w);
			kxh[st] = *reinterpret_cast<const long *>(w + 128u);
		}
		const int32_t bx = (int32_t)mx[1024u + 32u * qx + n];
#pragma unroll 1
		for (uint32_t yb = 0; yb < 2; ++yb) {
			uint32_t pix[16];
#pragma unroll
			for (int r = 0; r < 16; ++r) pix[r] = 0;
#pragma unroll
			for (uint32_t c = 0; c < (uint32_t)C; ++c) {
				const uint32_t *row = reinterpret_cast<const uint32_t *>(s_pl + c * plane + (32u * yb + n) * P + 4u * g);
				v16i32 lo, hi = zero;
#pragma unroll
				for (int r = 0; r < 16; ++r) lo[r] = bx;
#pragma unroll
				for (uint32_t st = 0; st < 2; ++st) {
					if (st < ksx) {
						const uint32_t a0 = row[4u * st] ^ 0x80808080u, a1 = row[4u * st + 2u] ^ 0x80808080u;
						const long av = (long)(((unsigned long long)a1 << 32) | (unsigned long long)a0);
						lo = __builtin_amdgcn_mfma_i32_32x32x16_i8(av, kxl[st], lo, 0, 0, 0);
						hi = __builtin_amdgcn_mfma_i32_32x32x16_i8(av, kxh[st], hi, 0, 0, 0);
					}
				}
#pragma unroll
				for (uint32_t r = 0; r < 16; ++r) put_byte(pix[r], c, clamp_fixed(hi[r], lo[r], top_x), px_);
			}
			if constexpr (C == 4) {
				// un-premultiplying is the identity at alpha 255: skipped when no lane of the wave holds anything else
				uint32_t alpha_and = 0xffffffffu;
#pragma unroll
				for (int r = 0; r < 16; ++r) alpha_and &= pix[r];
				if (__builtin_amdgcn_ballot_w64((alpha_and >> 24) != 255u) != 0ull) {
#pragma unroll
					for (int r = 0; r < 16; ++r) pix[r] = unpremultiply(pix[r]);
				}
			}
			x64_store_quadrant<C>(a, pix, lane, 32u * qx, 32u * yb, dst, s_out);
		}
	}
}

// C = 3 (round 4): RGB tiles into RGB frames -- 3-byte pixels in (a lane's four as twelve bytes), three planes, no premultiplication
// (fir's U8x3), the quadrants written through LDS as twelve bytes per lane (x64_store_quadrant).
template <int C>
__global__ void __launch_bounds__(1024) expand64_kernel(const ExpandArgs a)
{
	extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
	constexpr uint32_t kWave = C == 4 ? kX64Wave : kX64Wave3;
	const uint32_t wpb = blockDim.x / 64u, sub = threadIdx.x / 64u, lane = threadIdx.x % 64u;
	const uint32_t xmf_dw = a.xmf64 ? kXmf64Levels * kXmf64Dw : 0u;
	for (uint32_t i = threadIdx.x; i < xmf_dw / 4u; i += blockDim.x)
		reinterpret_cast<uint4 *>(lds)[i] = reinterpret_cast<const uint4 *>(a.xmf64)[i];
	const uint32_t *s_xmf = lds;
	uint32_t *s_ticket = lds + xmf_dw + wpb * kWave;
	if (threadIdx.x == 0) *s_ticket = wpb;
	__syncthreads();
	uint32_t *s_wave = lds + xmf_dw + sub * kWave;
	uint32_t *s_out = s_wave + 1536;  // (C = 3 only)
	auto tile_of = [&](uint32_t tk) -> uint32_t {
		const unsigned long long tl = (unsigned long long)blockIdx.x + (unsigned long long)tk * gridDim.x;
		return tl < (unsigned long long)a.n_tiles ? (uint32_t)tl : 0xffffffffu;
	};
	typedef uint32_t u32_a1 __attribute__((aligned(1)));
	typedef uint32_t u32q __attribute__((ext_vector_type(4), aligned(4)));   // (frame rows and slots: dword aligned)
	typedef uint32_t u32q_a1 __attribute__((ext_vector_type(4), aligned(1)));  // (RGB frame rows: any byte address)
	// pixel i of a slot as a dword (RGB: alpha 255; the slot is 12 288 bytes, i < 2048 here: the dword at 3 i lies inside it)
	auto slot_px = [&](const uint8_t *src, uint32_t i) -> uint32_t {
		if constexpr (C == 4) return reinterpret_cast<const uint32_t *>(src)[i];
		else return (*reinterpret_cast<const u32_a1 *>(src + 3u * i) & 0x00ffffffu) | 0xff000000u;
	};
	uint32_t p_tw = 0, p_th = 0, p_px = 0;
	auto prefetch = [&](uint32_t tn) {
		if (tn == 0xffffffffu) return;
		p_tw = a.tile_w[tn];
		p_th = a.tile_h[tn];
		p_px = slot_px(a.slots + (size_t)tn * a.slot_bytes, lane);
	};
	uint32_t t = tile_of(sub);
	prefetch(t);
	while (t != 0xffffffffu) {
		uint32_t nt = 0;
		if (lane == 0) nt = atomicAdd(s_ticket, 1u);
		const uint32_t t_next = tile_of(__builtin_amdgcn_readfirstlane(nt));
		const uint32_t tw = __builtin_amdgcn_readfirstlane(p_tw), th = __builtin_amdgcn_readfirstlane(p_th);
		const uint32_t first_px = p_px;
		const uint32_t frame = fastdiv(t, a.div_gpf), tf = t - frame * a.tiles_per_frame;
		const uint32_t ty = fastdiv(tf, a.div_gcols), tx = tf - ty * a.cols;
		const bool full = (tx + 1u < a.cols || a.edge_w == 64u) && (ty + 1u < a.rows || a.edge_h == 64u);
		const bool pow2 = tw >= 1u && tw <= 64u && th >= 1u && th <= 64u && (tw & (tw - 1u)) == 0u && (th & (th - 1u)) == 0u;
		const bool clone = full && tw == 64u && th == 64u;
		const bool near = full && pow2 && !clone && a.filter == 0u;  // (one axis may be 64: at most 64 x 32 stored pixels)
		const bool conv = full && pow2 && !clone && a.filter != 0u && xmf_dw != 0u;  // (two passes, or one when an axis is stored at 64)
		const uint8_t *src = a.slots + (size_t)t * a.slot_bytes;
		uint8_t *dst = a.dst + (size_t)frame * a.frame_stride + (size_t)(ty * 64u) * a.pitch + (size_t)(tx * 64u) * (uint32_t)C;
		if (clone) {
			// the slot's own bytes (block.rs:279-281) in 16-byte pieces, 16 (RGB: 12) per row; requested before the next tile's prefetch
			// (loads come back in order), four moves in flight
			constexpr uint32_t kPerRow = 4u * (uint32_t)C, kRounds = kPerRow;  // 64 rows x kPerRow pieces = 64 lanes x kRounds
			auto piece_dst = [&](uint32_t c) -> uint8_t * {
				const uint32_t row = C == 4 ? c >> 4 : small_div(c, 12u);
				return dst + (size_t)row * a.pitch + 16u * (c - row * kPerRow);
			};
			u32q v[4];
#pragma unroll
			for (uint32_t k = 0; k < 4; ++k) v[k] = __builtin_nontemporal_load(reinterpret_cast<const u32q *>(src) + 64u * k + lane);
			prefetch(t_next);
#pragma unroll 1
			for (uint32_t k0 = 0; k0 < kRounds; k0 += 4u) {
				u32q w[4] = {v[0], v[1], v[2], v[3]};
				if (k0 + 4u < kRounds) {
#pragma unroll
					for (uint32_t k = 0; k < 4; ++k) w[k] = __builtin_nontemporal_load(reinterpret_cast<const u32q *>(src) + 64u * (k0 + 4u + k) + lane);
				}
#pragma unroll
				for (uint32_t k = 0; k < 4; ++k) {
					const u32q_a1 o = {v[k].x, v[k].y, v[k].z, v[k].w};
					__builtin_nontemporal_store(o, reinterpret_cast<u32q_a1 *>(piece_dst(64u * (k0 + k) + lane)));
				}
#pragma unroll
				for (uint32_t k = 0; k < 4; ++k) v[k] = w[k];
			}
		} else if (near) {
			prefetch(t_next);
			const uint32_t lw = 31u - (uint32_t)__builtin_clz(tw), sx = 6u - lw, sy = 6u - (31u - (uint32_t)__builtin_clz(th));
			const uint32_t npx = tw * th;
			for (uint32_t i = lane; i < npx; i += 64u) s_wave[i] = i == lane ? first_px : slot_px(src, i);
			tile_sync<1>();
			const uint32_t q = lane & 15u, x = 4u * q;
#pragma unroll 4
			for (uint32_t k = 0; k < 16; ++k) {
				const uint32_t oy = 4u * k + (lane >> 4);
				const uint32_t *row = s_wave + ((oy >> sy) << lw);
				const uint32_t v0 = row[x >> sx], v1 = row[(x + 1u) >> sx], v2 = row[(x + 2u) >> sx], v3 = row[(x + 3u) >> sx];
				if constexpr (C == 4) {
					const u32q_a1 w = {v0, v1, v2, v3};
					__builtin_nontemporal_store(w, reinterpret_cast<u32q_a1 *>(dst + (size_t)oy * a.pitch + 16u * q));
				} else {
					u32_a1 *o = reinterpret_cast<u32_a1 *>(dst + (size_t)oy * a.pitch + 12u * q);
					o[0] = (v0 & 0xffffffu) | (v1 << 24);
					o[1] = ((v1 >> 8) & 0xffffu) | (v2 << 16);
					o[2] = ((v2 >> 16) & 0xffu) | (v3 << 8);
				}
			}
			tile_sync<1>();
		} else if (conv) {
			// ---- stored pixels -> (RGBA: premultiplied) byte planes [c][max(th, 8)][P], P = max(tw, 16)
			const uint32_t lw = 31u - (uint32_t)__builtin_clz(tw), lh = 31u - (uint32_t)__builtin_clz(th);
			const uint32_t P = tw < 16u ? 16u : tw, plane = P * (th < 8u ? 8u : th);  // (<= 2 KB: at most 64 x 32 or 32 x 64 stored pixels)
			uint8_t *s_pl = reinterpret_cast<uint8_t *>(s_wave);
			const uint32_t npx = tw * th;
			if (tw < 4u || npx <= 64u) {
				for (uint32_t i = lane; i < npx; i += 64u) {
					uint32_t px = i == lane ? first_px : slot_px(src, i);
					if constexpr (C == 4) {
						if (__builtin_amdgcn_ballot_w64((px >> 24) != 255u) != 0ull) px = premultiply(px);  // fir: U8x4 is alpha-premultiplied before a convolution
					}
					uint8_t *d = s_pl + (i >> lw) * P + (i & (tw - 1u));
					d[0] = (uint8_t)px;
					d[plane] = (uint8_t)(px >> 8);
					d[2u * plane] = (uint8_t)(px >> 16);
					if constexpr (C == 4) d[3u * plane] = (uint8_t)(px >> 24);
				}
			} else {
				// four adjacent pixels (one row: tw >= 4) per lane and round, a dword per channel
				for (uint32_t i4 = lane; i4 < (npx >> 2); i4 += 64u) {
					uint32_t p0, p1, p2, p3;
					if constexpr (C == 4) {
						const u32q v = __builtin_nontemporal_load(reinterpret_cast<const u32q *>(src) + i4);
						p0 = v.x; p1 = v.y; p2 = v.z; p3 = v.w;
						if (__builtin_amdgcn_ballot_w64(((p0 & p1 & p2 & p3) >> 24) != 255u) != 0ull) {
							p0 = premultiply(p0); p1 = premultiply(p1); p2 = premultiply(p2); p3 = premultiply(p3);
						}
					} else {
						// (twelve bytes at 12 i4 -- a three-element vector type is sixteen bytes wide: no pointer arithmetic on it)
						const uint32_t *q3 = reinterpret_cast<const uint32_t *>(src + 12u * i4);
						const uint32_t vx = q3[0], vy = q3[1], vz = q3[2];  // R0 G0 B0 R1 | G1 B1 R2 G2 | B2 R3 G3 B3
						p0 = vx;
						p1 = __builtin_amdgcn_alignbit(vy, vx, 24);
						p2 = __builtin_amdgcn_alignbit(vz, vy, 16);
						p3 = vz >> 8;
					}
					uint8_t *d = s_pl + ((4u * i4) >> lw) * P + ((4u * i4) & (tw - 1u));
#pragma unroll
					for (uint32_t c = 0; c < (uint32_t)C; ++c) {
						const uint32_t sel = c | ((4u + c) << 8) | 0x0c0c0000u;
						const uint32_t lo2 = __builtin_amdgcn_perm(p1, p0, sel), hi2 = __builtin_amdgcn_perm(p3, p2, sel);
						*reinterpret_cast<uint32_t *>(d + c * plane) = lo2 | (hi2 << 16);
					}
				}
			}
			prefetch(t_next);
			tile_sync<1>();
			if (tw == 64u) {
				if (th <= 16u) expand64_vonly<1, C>(a, s_xmf, s_pl, lane, lh, plane, dst, s_out);
				else expand64_vonly<2, C>(a, s_xmf, s_pl, lane, lh, plane, dst, s_out);
			} else if (th == 64u) {
				expand64_honly<C>(a, s_xmf, s_pl, lane, lw, P, dst, s_out);
			} else if (th <= 8u) {
				expand64_conv<1, C>(a, s_xmf, s_pl, lane, lw, lh, P, dst, s_out);
			} else if (th == 16u) {
				expand64_conv<2, C>(a, s_xmf, s_pl, lane, lw, lh, P, dst, s_out);
			} else {
				expand64_conv<4, C>(a, s_xmf, s_pl, lane, lw, lh, P, dst, s_out);
			}
			tile_sync<1>();  // the next tile restages the planes
		} else {
			prefetch(t_next);
			if (lane == 0) a.list[atomicAdd(a.status + 1, 1u)] = t;  // expand_kernel's second launch takes it (and flags what is invalid)
		}
		t = t_next;
	}
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
// F32 (RGBA tiles of 32x32 in RGBA frames): blocks of up to 16 waves that share the matrix-core tables; full tiles stored
// at power-of-two sizes take expand_tile_mfma / expand_tile_nearest_pow2, everything else the general forms below.
// BIG (round 4): tiles whose image (stored pixels + horizontal pass + windows) does not fit LDS keep it in HBM, one image per wave
// of the grid (a.big_scratch): the same code on a pointer that is not LDS, with a fence that waits for the wave's own stores where
// the LDS form only stops the compiler.  Any block size the reference accepts expands; not a fast path.
// LIST (round 4): the second launch of the 16x16 / 64x64 flows -- the tiles expand16_kernel / expand64_kernel left (a.list, status[1]
// of them, complete when this launch starts) instead of every tile.  A template flag, not a run-time one: the 32x32 instance sits
// at its register limit, and the two extra values in its tile loop cost it thirteen more spills (0.31 -> 0.35 ms).
template <int C, bool F32 = false, bool BIG = false, bool LIST = false>
__global__ void __launch_bounds__(F32 ? 1024 : 256) expand_kernel(const ExpandArgs a)
{
	static_assert(!(F32 && BIG), "the 32x32 instance has its image in LDS");
	extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
	const uint32_t wpb = blockDim.x / 64u, sub = threadIdx.x / 64u, lane = threadIdx.x % 64u;
	// (between a phase that writes the wave's image and one that reads it)
	auto wsync = [&]() __attribute__((always_inline)) {
		if constexpr (BIG) {
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // s_waitcnt vmcnt(0): the wave's stores to its image have landed
			__builtin_amdgcn_wave_barrier();
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
		} else {
			tile_sync<1>();
		}
	};
	// the matrix-core tables of the block (32x32 tiles, convolutions), then the waves' tile images
	const uint32_t xmf_dw = F32 && a.xmf ? kXmfLevels * kXmfDw : 0u;
	for (uint32_t i = threadIdx.x; i < xmf_dw / 4u; i += blockDim.x)
		reinterpret_cast<uint4 *>(lds)[i] = reinterpret_cast<const uint4 *>(a.xmf)[i];
	const uint32_t *s_xmf = lds;
	uint32_t *s_ticket = lds + xmf_dw + (BIG ? 0u : wpb * a.tile_dw);
	if (threadIdx.x == 0) *s_ticket = wpb;
	__syncthreads();
	uint32_t *s_src = BIG ? a.big_scratch + (size_t)(blockIdx.x * wpb + sub) * a.tile_dw : lds + xmf_dw + sub * a.tile_dw;
	uint32_t *s_tmp = s_src + a.bw * a.bh;
	// A tile's stored size and its first 64 pixels (all of them for most tiles) are requested one tile ahead: the
	// size -> pixels -> windows chain of dependent memory round trips was most of a tile's time.
	uint32_t n_items = a.n_tiles;
	if constexpr (LIST) n_items = __builtin_amdgcn_readfirstlane(a.status[1]);
	auto tile_of = [&](uint32_t tk) -> uint32_t {
		const unsigned long long tl = (unsigned long long)blockIdx.x + (unsigned long long)tk * gridDim.x;
		if (tl >= (unsigned long long)n_items) return 0xffffffffu;
		if constexpr (LIST) return __builtin_amdgcn_readfirstlane(a.list[(uint32_t)tl]);
		return (uint32_t)tl;
	};
	uint32_t p_tw = 0, p_th = 0, p_px = 0;
	auto prefetch = [&](uint32_t tn) {
		if (tn == 0xffffffffu) return;
		p_tw = a.tile_w[tn];
		p_th = a.tile_h[tn];
		if constexpr (C == 4) {
			if (lane < a.bw * a.bh) p_px = reinterpret_cast<const uint32_t *>(a.slots + (size_t)tn * a.slot_bytes)[lane];
		}
	};
	uint32_t t = tile_of(sub);
	prefetch(t);
#ifdef PXZ_STAMPS
	unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	unsigned long long st_last = stamp_now();
#endif
	while (t != 0xffffffffu) {
		uint32_t nt = 0;
		if (lane == 0) nt = atomicAdd(s_ticket, 1u);
		const uint32_t t_next = tile_of(__builtin_amdgcn_readfirstlane(nt));
		PXZ_STAMP(0);  // ticket
		const uint32_t tw = __builtin_amdgcn_readfirstlane(p_tw), th = __builtin_amdgcn_readfirstlane(p_th);
		uint32_t first_px = p_px;
#ifdef PXZ_STAMPS
		asm volatile("" : "+v"(first_px));
#endif
		PXZ_STAMP(1);  // wait for what was requested a tile ago
		prefetch(t_next);  // in flight while this tile is expanded
		const uint32_t frame = t / a.tiles_per_frame, tf = t - frame * a.tiles_per_frame;
		const uint32_t ty = tf / a.cols, tx = tf - ty * a.cols;
		const uint32_t fw = (tx == a.cols - 1) ? a.edge_w : a.bw, fh = (ty == a.rows - 1) ? a.edge_h : a.bh;
		const bool widen = C == 3 && a.out_channels == 4;  // RGB tiles into an RGBA frame (process())
		const uint32_t opx = widen ? 4u : (uint32_t)C;
		uint8_t *dst = a.dst + (size_t)frame * a.frame_stride + (size_t)(ty * a.bh) * a.pitch + (size_t)(tx * a.bw) * opx;
		auto put = [&](uint32_t ox, uint32_t oy, uint32_t px) {
			uint8_t *p = dst + (size_t)oy * a.pitch + ox * opx;
			if (C == 4 || widen) {
				__builtin_nontemporal_store(px, reinterpret_cast<uint32_t *>(p));  // C == 3: alpha was set to 255 when the tile was staged
			} else {
				p[0] = (uint8_t)px;
				p[1] = (uint8_t)(px >> 8);
				p[2] = (uint8_t)(px >> 16);
			}
		};
		if (tw == 0 || th == 0 || tw > fw || th > fh) {
			if (lane == 0 && !(a.quiet_empty && tw == 0 && th == 0)) atomicOr(a.status, 1u);
		} else if (F32 && fw == 32u && fh == 32u && tw == 32u && th == 32u) {
			if constexpr (F32) expand_tile_clone32(a, lane, a.slots + (size_t)t * a.slot_bytes, dst);
		} else if (F32 && fw == 32u && fh == 32u && (tw & (tw - 1u)) == 0u && (th & (th - 1u)) == 0u &&
		           (a.filter == 0 ? tw * th < 1024u : (xmf_dw != 0u && tw <= 16u && th <= 16u))) {
			if constexpr (F32) {
				const uint8_t *src = a.slots + (size_t)t * a.slot_bytes;
				if (a.filter == 0) expand_tile_nearest_pow2(a, s_src, lane, tw, th, src, first_px, dst);
				else expand_tile_mfma(a, s_xmf, s_src, lane, tw, th, src, first_px, dst
#ifdef PXZ_STAMPS
				                      , st_acc, st_last
#endif
				);
			}
		} else if (!F32 && C == 4 && !widen && tw == fw && th == fh && (fw & 3u) == 0u) {
			// a tile stored at full size (block.rs:279-281: clone), any block size: its slot straight into the frame rows, 16 bytes per
			// lane and move, four moves requested before the first is stored -- not through the LDS image (a 64x64 tile: 64 load ->
			// LDS rounds, a barrier, then 16 rounds out again; these tiles are half of the pixels of a typical frame)
			if constexpr (!F32 && C == 4) {
				typedef uint32_t u32q __attribute__((ext_vector_type(4), aligned(4)));  // (slots and rows of any block size: dword aligned)
				const u32q *src = reinterpret_cast<const u32q *>(a.slots + (size_t)t * a.slot_bytes);
				const uint32_t q4 = fw >> 2, total = q4 * fh;
				for (uint32_t g0 = lane; g0 < total; g0 += 256u) {
					u32q v[4];
#pragma unroll
					for (uint32_t k = 0; k < 4; ++k) {
						const uint32_t g = g0 + 64u * k;
						v[k] = __builtin_nontemporal_load(src + (g < total ? g : g0));
					}
#pragma unroll
					for (uint32_t k = 0; k < 4; ++k) {
						const uint32_t g = g0 + 64u * k;
						if (g < total) {
							const uint32_t oy = small_div(g, q4), q = g - oy * q4;
							__builtin_nontemporal_store(v[k], reinterpret_cast<u32q *>(dst + (size_t)oy * a.pitch + q * 16u));
						}
					}
				}
			}
		} else if (!F32 && C == 3 && !widen && tw == fw && th == fh && (fw & 3u) == 0u) {
			// the same for RGB tiles in RGB frames (round 4): twelve bytes -- four pixels -- per lane and move, at whatever byte address
			// the slot and the row have (through the image these tiles, an eighth of a frame, were a third of the RGB launch's time)
			if constexpr (!F32 && C == 3) {
				// (three dwords, not a three-element vector: that type is sixteen bytes wide, addresses are made in bytes)
				const uint8_t *src = a.slots + (size_t)t * a.slot_bytes;
				const uint32_t q4 = fw >> 2, total = q4 * fh;
				for (uint32_t g0 = lane; g0 < total; g0 += 256u) {
					uint32_t v[4][3];
#pragma unroll
					for (uint32_t k = 0; k < 4; ++k) {
						const uint32_t g = g0 + 64u * k;
						const uint32_t *q = reinterpret_cast<const uint32_t *>(src + 12u * (size_t)(g < total ? g : g0));
						typedef uint32_t u32_a1 __attribute__((aligned(1)));
						const u32_a1 *qa = reinterpret_cast<const u32_a1 *>(q);
						v[k][0] = qa[0]; v[k][1] = qa[1]; v[k][2] = qa[2];
					}
#pragma unroll
					for (uint32_t k = 0; k < 4; ++k) {
						const uint32_t g = g0 + 64u * k;
						if (g < total) {
							const uint32_t oy = small_div(g, q4), q = g - oy * q4;
							typedef uint32_t u32_a1 __attribute__((aligned(1)));
							u32_a1 *o = reinterpret_cast<u32_a1 *>(dst + (size_t)oy * a.pitch + q * 12u);
							o[0] = v[k][0]; o[1] = v[k][1]; o[2] = v[k][2];
						}
					}
				}
			}
		} else {
			// ---- stored pixels -> one dword per pixel
			const uint8_t *src = a.slots + (size_t)t * a.slot_bytes;
			const uint32_t n = tw * th;
			const bool conv = a.filter != 0 && (tw != fw || th != fh);
			for (uint32_t i = lane; i < n; i += 64u) {
				uint32_t px;
				if constexpr (C == 4) {
					px = i == lane ? first_px : reinterpret_cast<const uint32_t *>(src)[i];
					if (conv) px = premultiply(px);  // fir: U8x4 is alpha-premultiplied before a convolution
				} else {
					// one dword from the pixel's byte address (the slot's last pixel: from a byte earlier, shifted)
					typedef uint32_t u32_a1 __attribute__((aligned(1)));
					if (a.slot_bytes >= 4u) {
						const uint32_t at = 3u * i + 4u <= a.slot_bytes ? 3u * i : a.slot_bytes - 4u;
						px = (*reinterpret_cast<const u32_a1 *>(src + at) >> (8u * (3u * i - at))) | 0xff000000u;
					} else {  // (1x1 blocks)
						px = (uint32_t)src[3 * i] | ((uint32_t)src[3 * i + 1] << 8) | ((uint32_t)src[3 * i + 2] << 16) | 0xff000000u;
					}
				}
				s_src[i] = px;
			}
			wsync();
			const uint32_t cls_x = fw == a.bw ? 0u : 1u, cls_y = fh == a.bh ? 0u : 1u;
			const ExpandTab tab_x = a.tabs[(0u * 2u + cls_x) * a.dir_stride + tw];
			const ExpandTab tab_y = a.tabs[(1u * 2u + cls_y) * a.dir_stride + th];
			// Vector form (RGBA tiles in RGBA frames whose full width is a multiple of 4): the windows of the tile
			// are staged into LDS once, a lane then makes 4 rows (horizontal pass) or 4 adjacent columns (vertical
			// pass, nearest, clone) per item, so weights are fetched once per 16 multiply-adds and the frame is
			// written 16 bytes per lane.  Same arithmetic as the scalar form below.
			if ((fw & 3u) == 0 && tab_x.window <= 8 && tab_y.window <= 8) {
				const uint32_t q4 = fw >> 2;
				// RGB instance: the clamp spelled as an instruction.  Left to the compiler, clip8(a) | clip8(b) << 8 of an RGB pixel became
				// v_ashr_pk_u8_i32, whose result it then ORs as if bits 31:16 were zero -- on gfx950 they keep what the destination
				// register held (the previous pixel's blue: every third pixel of a quad came out with a wrong blue, ROCm 7.2)
				auto clipv = [](int32_t acc, int prec) __attribute__((always_inline)) -> uint32_t {
					if constexpr (C == 4) {
						return clip8(acc, prec);
					} else {
						const int32_t v = acc >> prec;
						int32_t r;
						asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(v), "v"(255));
						return (uint32_t)r;
					}
				};
				auto put4 = [&](uint32_t q, uint32_t oy, uint4 px) {
					// (the frame is written once and not read by this launch: streaming stores, 13 % off the kernel)
					if (C == 4 || widen) {
						typedef uint32_t u32q __attribute__((ext_vector_type(4)));
						const uint32_t opaque = C == 3 ? 0xff000000u : 0u;  // RGB tiles in an RGBA frame: the fourth lane of the arithmetic is not an alpha
						const u32q w = {px.x | opaque, px.y | opaque, px.z | opaque, px.w | opaque};
						__builtin_nontemporal_store(w, reinterpret_cast<u32q *>(dst + (size_t)oy * a.pitch + q * 16u));
					} else {
						// RGB rows: the four pixels as twelve bytes, three dwords at whatever byte address they have
						typedef uint32_t u32_a1 __attribute__((aligned(1)));
						u32_a1 *o = reinterpret_cast<u32_a1 *>(dst + (size_t)oy * a.pitch + q * 12u);
						o[0] = (px.x & 0xffffffu) | (px.y << 24);
						o[1] = ((px.y >> 8) & 0xffffu) | (px.z << 16);
						o[2] = ((px.z >> 16) & 0xffu) | (px.w << 8);
					}
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
				wsync();
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
							// two taps per v_dot2_i32_i16: the weights are staged as i16 pairs (an odd count has a zero in the
							// spare half, so the pixel read past the window -- still inside this wave's LDS -- counts for nothing)
							for (uint32_t j = 0; j < cnt; j += 2u) {
								const uint32_t w2 = kk[j >> 1];
#pragma unroll
								for (int r = 0; r < 4; ++r) {
									const uint32_t *pp = s_src + yr[r] * tw + first + j;
									const uint32_t p0 = pp[0], p1 = pp[1];
#pragma unroll
									for (uint32_t c = 0; c < 4; ++c)
										acc[r][c] = dot2(__builtin_amdgcn_perm(p1, p0, c | 0x0c000c00u | ((4u + c) << 16)), w2, acc[r][c]);
								}
							}
#pragma unroll
							for (uint32_t r = 0; r < 4; ++r) {
								const uint32_t y = 4u * yq + r;
								if (y < th) {
									uint32_t px = clipv(acc[r][0], prec) | (clipv(acc[r][1], prec) << 8) | (clipv(acc[r][2], prec) << 16) |
									              (clipv(acc[r][3], prec) << 24);
									if (need_v) {
										s_tmp[y * fw + ox] = px;
									} else {
										put(ox, y, C == 4 ? unpremultiply(px) : (px | 0xff000000u));  // (RGB: the fourth lane of the arithmetic is not an alpha)
									}
								}
							}
						}
						wsync();
					}
					// A stored tile of ONE row (40 % of the tiles of a typical frame are 2x1): every window of the way up is that
					// row with the single weight 2^precision, and clip8((2^(p-1) + v 2^p) >> p) = v -- the vertical pass is the
					// identity, row for row.  The horizontal result is un-premultiplied once and written to every row.
					const bool one_row = need_v && th == 1u && (s_wy[0] >> 16) == 1u && (s_wy[1] & 0xffffu) == (1u << tab_y.precision) &&
					                     tab_y.precision < 15u;
					if (one_row) {
						const uint32_t *cur = need_h ? s_tmp : s_src;
						for (uint32_t i = lane; i < q4 * fh; i += 64u) {
							const uint32_t oy = small_div(i, q4), q = i - oy * q4;
							uint4 v = *reinterpret_cast<const uint4 *>(cur + 4u * q);
							const uint32_t alpha_and = (v.x & v.y & v.z & v.w) >> 24;
							if (C == 4 && __builtin_amdgcn_ballot_w64(alpha_and != 255u) != 0ull) {
								v.x = unpremultiply(v.x); v.y = unpremultiply(v.y); v.z = unpremultiply(v.z); v.w = unpremultiply(v.w);
							}
							put4(q, oy, v);
						}
					} else if (need_v) {
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
							for (uint32_t j = 0; j < cnt; j += 2u) {  // two taps (two source rows) per v_dot2_i32_i16, as above
								const uint32_t w2 = kk[j >> 1];
								const uint4 v0 = *reinterpret_cast<const uint4 *>(cur + (first + j) * fw + 4u * q);
								const uint4 v1 = *reinterpret_cast<const uint4 *>(cur + (first + j + 1u) * fw + 4u * q);
								const uint32_t a4[4] = {v0.x, v0.y, v0.z, v0.w}, b4[4] = {v1.x, v1.y, v1.z, v1.w};
#pragma unroll
								for (int r = 0; r < 4; ++r)
#pragma unroll
									for (uint32_t c = 0; c < 4; ++c)
										acc[r][c] = dot2(__builtin_amdgcn_perm(b4[r], a4[r], c | 0x0c000c00u | ((4u + c) << 16)), w2, acc[r][c]);
							}
							uint32_t o4[4];
#pragma unroll
							for (int r = 0; r < 4; ++r)
								o4[r] = clipv(acc[r][0], prec) | (clipv(acc[r][1], prec) << 8) | (clipv(acc[r][2], prec) << 16) |
								        (clipv(acc[r][3], prec) << 24);
							// un-premultiplying is the identity at alpha 255: skipped (table look-up and three divisions-by-
							// multiplication per pixel) when no lane of the wave holds anything else
							const uint32_t alpha_and = (o4[0] & o4[1] & o4[2] & o4[3]) >> 24;
							if (C == 4 && __builtin_amdgcn_ballot_w64(alpha_and != 255u) != 0ull) {
#pragma unroll
								for (int r = 0; r < 4; ++r) o4[r] = unpremultiply(o4[r]);
							}
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
					wsync();
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
		PXZ_STAMP(2);  // the other paths (clone, general, nearest), and what of the matrix-core path is not stamped inside
		t = t_next;
		wsync();  // the next tile reuses this wave's LDS
		PXZ_STAMP(7);  // closing sync
	}
#ifdef PXZ_STAMPS
	if (lane == 0) {
		unsigned long long *out = reinterpret_cast<unsigned long long *>(a.status + 2);
		for (int i = 0; i < 8; ++i) atomicAdd(out + i, st_acc[i]);
	}
#endif
}

hipError_t launch_expand_general(const ExpandArgs &a, uint32_t n_cus, hipStream_t stream, uint32_t max_blocks);

hipError_t launch_expand(const ExpandArgs &a, uint32_t n_cus, hipStream_t stream)
{
	if (a.list != nullptr && a.bw == 64u) {
		// 64x64 RGBA tiles: one wave per full tile stored at powers of two, then what that left through the list
		const uint32_t xmf_bytes = a.xmf64 ? kXmf64Levels * kXmf64Dw * 4u : 0u;
		const uint32_t wave_bytes = (a.channels == 4 ? kX64Wave : kX64Wave3) * 4u;
		uint32_t wpb = (160u * 1024u - 16u - xmf_bytes) / wave_bytes;
		if (wpb > 16u) wpb = 16u;
		const uint32_t lds_bytes = xmf_bytes + wpb * wave_bytes + 16u;
		const uint32_t need = (a.n_tiles + wpb - 1u) / wpb;
		void (*k64)(const ExpandArgs) = a.channels == 4 ? expand64_kernel<4> : expand64_kernel<3>;
		hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k64), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
		if (e != hipSuccess) return e;
		hipLaunchKernelGGL(k64, dim3(need < n_cus ? need : n_cus), dim3(64u * wpb), lds_bytes, stream, a);
		if ((e = hipGetLastError()) != hipSuccess) return e;
		ExpandArgs b = a;
		b.list_mode = 1u;
		return launch_expand_general(b, n_cus, stream, 4u * n_cus);
	}
	if (a.list != nullptr) {
		// 16x16 RGBA tiles: the 2x2 groups first, then what they left (partial groups, one-pass and odd sizes) through the list
		const uint32_t xmf_bytes = a.xmf16 ? kXmf16Levels * kXmf16Dw * 4u : 0u;
		const uint32_t wpb = 16u, lds_bytes = xmf_bytes + wpb * kX16Wave * 4u + 16u;
		const uint32_t gcols = (a.cols + 1u) >> 1, grows = (a.rows + 1u) >> 1, n_groups = (a.n_tiles / a.tiles_per_frame) * gcols * grows;
		const uint32_t need = (n_groups + wpb - 1u) / wpb, resident = 2u * n_cus;
		hipLaunchKernelGGL(expand16_kernel, dim3(need < resident ? need : resident), dim3(64u * wpb), lds_bytes, stream, a);
		hipError_t e = hipGetLastError();
		if (e != hipSuccess) return e;
		ExpandArgs b = a;
		b.list_mode = 1u;
		return launch_expand_general(b, n_cus, stream, 2u * n_cus);  // (the list is short: a grid the size of the chip)
	}
	return launch_expand_general(a, n_cus, stream, 0xffffffffu);
}

hipError_t launch_expand_general(const ExpandArgs &a, uint32_t n_cus, hipStream_t stream, uint32_t max_blocks)
{
	if (a.big_waves != 0u) {
		// tile images in HBM: blocks of 4 waves, as many as the scratch holds images for
		const uint32_t wpb = 4u, blocks_max = a.big_waves / wpb, need = (a.n_tiles + wpb - 1u) / wpb;
		const uint32_t blocks = need < blocks_max ? need : blocks_max;
		if (a.channels == 4) hipLaunchKernelGGL((expand_kernel<4, false, true>), dim3(blocks), dim3(64u * wpb), 16u, stream, a);
		else hipLaunchKernelGGL((expand_kernel<3, false, true>), dim3(blocks), dim3(64u * wpb), 16u, stream, a);
		return hipGetLastError();
	}
	constexpr uint32_t kLds = 160u * 1024u;
	const uint32_t tile_bytes = a.tile_dw * 4u;
	const bool f32 = a.fast32 && a.channels == 4 && a.out_channels == 4 && a.bw == 32 && a.bh == 32 && (a.filter == 0 || a.xmf);
	const uint32_t xmf_bytes = f32 && a.xmf ? kXmfLevels * kXmfDw * 4u : 0u;
	// 32x32 RGBA tiles: one block of up to 16 waves per CU (they share the matrix-core tables); else blocks of up to 4 waves
	uint32_t wpb = (kLds - 16u - xmf_bytes) / tile_bytes;
	const uint32_t wpb_max = f32 ? 16u : 4u;
	if (wpb > wpb_max) wpb = wpb_max;
	if (wpb < 1u) return hipErrorInvalidValue;
	const uint32_t lds_bytes = xmf_bytes + wpb * tile_bytes + 16u;
	uint32_t per_cu = kLds / lds_bytes;
	if (per_cu > 4u) per_cu = 4u;
	if (per_cu < 1u) per_cu = 1u;
	const uint32_t need = (a.n_tiles + wpb - 1u) / wpb, resident = n_cus * per_cu;
	uint32_t blocks = need < resident ? need : resident;
	if (blocks > max_blocks) blocks = max_blocks;
	hipError_t e;
	if (f32) {
		auto k = expand_kernel<4, true>;
		if (lds_bytes > 64u * 1024u && (e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes)) != hipSuccess) return e;
		hipLaunchKernelGGL(k, dim3(blocks), dim3(64u * wpb), lds_bytes, stream, a);
	} else if (a.channels == 4 && a.list_mode) {
		auto k = expand_kernel<4, false, false, true>;
		if (lds_bytes > 64u * 1024u && (e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes)) != hipSuccess) return e;
		hipLaunchKernelGGL(k, dim3(blocks), dim3(64u * wpb), lds_bytes, stream, a);
	} else if (a.channels == 4) {
		auto k = expand_kernel<4>;
		if (lds_bytes > 64u * 1024u && (e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes)) != hipSuccess) return e;
		hipLaunchKernelGGL(k, dim3(blocks), dim3(64u * wpb), lds_bytes, stream, a);
	} else if (a.list_mode) {
		auto k = expand_kernel<3, false, false, true>;
		if (lds_bytes > 64u * 1024u && (e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes)) != hipSuccess) return e;
		hipLaunchKernelGGL(k, dim3(blocks), dim3(64u * wpb), lds_bytes, stream, a);
	} else {
		auto k = expand_kernel<3>;
		if (lds_bytes > 64u * 1024u && (e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes)) != hipSuccess) return e;
		hipLaunchKernelGGL(k, dim3(blocks), dim3(64u * wpb), lds_bytes, stream, a);
	}
	return hipGetLastError();
}

}  // namespace pxz
