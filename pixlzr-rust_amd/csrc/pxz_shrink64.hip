// pxz_shrink64.hip -- the fast kernel for 64x64 RGBA tiles (four waves per tile, 16x16x64 matrix-core passes) and its
// four-plane instance for tiles with transparency, and the first part of their launch.
//
// Compiled with -ffp-contract=off: the f32 results of the Oklab detector are
// written into the bitstream, and the reference (Rust) never fuses a*b+c.
#include "pxz_device.h"

namespace pxz {


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
constexpr uint32_t kFin64 = 256;                 // tiles a block completed, for their finish at its end (a block takes ~64 of 8 x 8K frames)
constexpr uint32_t kTail64 = 72, kTailLevels = 8;  // per level: bias[32], weight sums[32], "opaque stays opaque" + pad; levels 0..7 (7 = every level whose output is 1 px)
constexpr uint32_t lds64_dwords(uint32_t nch) { return nch * kPD64 + nch * 32u * kTS64 + kRed64 + kTailLevels * kTail64 + kFin64; }  // planes + [channel][ox < 32][kTS64] + s_red + tails + finish list

template <int C = 4, class Args>
__device__ __forceinline__ bool fast64_tile_src(const Args &a, uint32_t tile_g, const uint8_t *&src)
{
	if (tile_g >= a.n_tiles) return false;
	const uint32_t frame = fastdiv(tile_g, a.div_tpf);
	const uint32_t t = tile_g - frame * a.tiles_per_frame;
	const uint32_t ty = fastdiv(t, a.div_cols), tx = t - ty * a.cols;
	src = a.src + (size_t)frame * a.frame_stride + (size_t)(ty * 64u) * a.pitch + (size_t)(tx * 64u) * (uint32_t)C;
	return tx < a.full_cols && ty < a.full_rows;
}

// ALPHA: the full tiles WITH transparency the opaque kernel put on list A -- a fourth plane keeps the alpha
// channel, the colours are premultiplied in place once the detector is done with them (fir's U8x4 path), all
// four planes go through the passes and every output pixel is un-premultiplied.
// FULL: as in shrink32_kernel (out_px, out_w, out_h all there: no run-time tests of them in the tile loop).
// C = 3 (round 2): RGB frames read (12-byte pixel quads) and RGB slots written directly; no opacity test, no ALPHA instance.
template <int MODE, bool ALPHA, bool FULL, int C = 4>
__global__ void __launch_bounds__(256) shrink64_kernel(const Fast64Args a)
{
	static_assert(C == 4 || !ALPHA, "RGB tiles have no alpha plane");
	constexpr uint32_t NCH = ALPHA ? 4u : 3u;
	// one output pixel (a dword R G B A) into a slot of C-byte pixels
	auto store_px = [](uint8_t *slot, uint32_t index, uint32_t px) {
		if constexpr (C == 4) {
			reinterpret_cast<uint32_t *>(slot)[index] = px;
		} else {
			uint8_t *p = slot + 3u * index;
			p[0] = (uint8_t)px;
			p[1] = (uint8_t)(px >> 8);
			p[2] = (uint8_t)(px >> 16);
		}
	};
	extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
	uint32_t *s_pl = lds;                       // NCH planes of u16 pairs
	uint32_t *s_t = lds + NCH * kPD64;          // horizontal-pass results
	uint32_t *s_red = s_t + NCH * 32u * kTS64;  // [0..7] partial sums, [8..11] alpha, [13] ticket, [16..31] list-B batch, [32..47] list-A batch
	// The small ends of the operand tables (biases, weight sums, flags), copied once per block (round 3): read from global memory
	// where they are used, each was a dependent L2 round trip between a tile's level decision and its last store -- with one
	// tile per block in flight.  (The whole tables in LDS, 16 KB, cost a block per CU and bought nothing.)
	uint32_t *s_tail = s_red + kRed64;
	uint32_t *s_fin = s_tail + kTailLevels * kTail64;  // tiles this block completed (thread 0 appends), finished at the block's end
	uint32_t n_fin = 0;                                  // (thread 0's count)
	for (uint32_t i = threadIdx.x; i < kTailLevels * kTail64; i += blockDim.x) {
		const uint32_t lv = i / kTail64, k = i - lv * kTail64;
		const uint32_t out = (64u >> lv) ? (64u >> lv) : 1u, nblk = out > 16u ? 2u : 1u;
		s_tail[i] = a.mf_off[lv] ? a.mf64[a.mf_off[lv] + nblk * 512u + k] : 0u;
	}
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
	// MODE 0 with clone_ahead (round 4): oklab2_kernel<64> has copied every tile into its slot as if it were stored at full size; a
	// tile whose value says it is (two thirds of them in shrink_by) is finished without being read again.  The value of the
	// tile after this one is requested as soon as that tile is known, an iteration before its pixels would be.
	bool pre_skipped = false;          // the current tile's loads were left out (block-uniform)
	uint32_t vb_cur = 0, vb_next = 0;  // value bits of the current tile / the next one
	uint32_t fl_cur = 0, fl_next = 0;  // and sums[2 t + 1]: 1 = the detector copied the tile (at this size it copies every tile it takes)
	auto value_pair_of = [&](uint32_t t) -> uint2 { return t < a.n_tiles ? reinterpret_cast<const uint2 *>(a.sums)[t] : make_uint2(0u, 0u); };
	auto stored_whole = [&](uint32_t vb, uint32_t copied) -> bool {
		return a.clone_ahead && __builtin_amdgcn_readfirstlane(copied) == 1 &&
		       level_of(__float_as_uint(parse_value(__uint_as_float(__builtin_amdgcn_readfirstlane(vb))))) == 0u;
	};
	auto prefetch = [&](uint32_t tile_g, bool skip_loads = false) {
		const uint8_t *src;
		pre_valid = fast64_tile_src<C>(a, tile_g, src);
		if (pre_valid && !skip_loads) {
			const uint8_t *p = src + (size_t)(16u * wave + (lane >> 4)) * a.pitch + (lane & 15u) * (4u * (uint32_t)C);
#pragma unroll
			for (int k = 0; k < 4; ++k) {
				if constexpr (C == 4) {
					pre[k] = *reinterpret_cast<const uint4 *>(p + (size_t)(4 * k) * a.pitch);
				} else {
					const uint3 v = *reinterpret_cast<const uint3 *>(p + (size_t)(4 * k) * a.pitch);  // rows are 4-byte aligned
					pre[k] = make_uint4(v.x, v.y, v.z, 0u);
				}
			}
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
	// (ALPHA with all_tiles, round 2: every tile of the batch -- the launch that skips the opaque instance because most
	// tiles of the last launch had transparency; an opaque tile comes out of the four-plane arithmetic unchanged)
	// (MODE 0 with a list, round 4: the tiles clone_split64_kernel left -- every tile that is NOT stored at full size)
	const bool listed = !ALPHA && MODE == 0 && a.clone_list != nullptr;
	const uint32_t n_items = (ALPHA && !a.all_tiles) ? a.work[kWorkA + a.work_slot] : (listed ? a.clone_list[0] : a.n_tiles);
	uint32_t n_transparent = 0;  // all_tiles: what the list counter would have said (thread 0 counts)
	auto tile_of = [&](uint32_t k) -> uint32_t {
		const unsigned long long t = (unsigned long long)k * n_ctr + cid;
		if (t >= (unsigned long long)n_items) return 0xffffffffu;
		if constexpr (ALPHA) {
			if (!a.all_tiles) return a.work[kWorkList + a.n_tiles + (uint32_t)t];
		}
		if (listed) return a.clone_list[8u + (uint32_t)t];
		return (uint32_t)t;
	};
	uint32_t tile_g = tile_of(blockIdx.x / n_ctr), tile_next = tile_of(blockIdx.x / n_ctr + nb_c);
	if constexpr (MODE == 0 && !ALPHA) {
		const uint2 p0 = value_pair_of(tile_g), p1 = value_pair_of(tile_next);
		vb_cur = p0.x;
		fl_cur = p0.y;
		vb_next = p1.x;
		fl_next = p1.y;
		pre_skipped = stored_whole(vb_cur, fl_cur);
	}
	prefetch(tile_g, pre_skipped);
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
			if constexpr (MODE == 0 && !ALPHA) {
				vb_cur = vb_next;
				fl_cur = fl_next;
				const uint2 pn = value_pair_of(tile_next);
				vb_next = pn.x;
				fl_next = pn.y;
			}
		};
		auto prefetch_next = [&]() {
			if constexpr (MODE == 0 && !ALPHA) {
				pre_skipped = stored_whole(vb_next, fl_next);
				prefetch(tile_next, pre_skipped);
			} else {
				prefetch(tile_next);
			}
		};
		auto defer = [&]() {
			// (block-uniform; only wave 0's lanes 0..15 ever touch the batch)
			list_push(s_red + 16, n_listb, tile_g, a.work + kWorkList, a.work + a.work_slot, threadIdx.x);
			if (threadIdx.x == 0) {
				bool keep = false;  // MODE 0: a value the block-cooperative detector left is final and stays
				if constexpr (MODE == 0) {
					const uint32_t t = tile_g - fastdiv(tile_g, a.div_tpf) * a.tiles_per_frame;
					const uint32_t ty = fastdiv(t, a.div_cols), tx = t - ty * a.cols;
					keep = oklab_value_given(a, tx, ty);
				}
				if (!keep) reinterpret_cast<uint2 *>(a.sums)[tile_g] = make_uint2(kDeferredKey, kDeferredKey);
			}
		};
		if (!pre_valid) {  // ragged edge / unaligned batch (block-uniform)
			defer();
			prefetch_next();
			publish_ticket();
			__syncthreads();
			advance();
			continue;
		}
		if (MODE == 0 && !ALPHA && pre_skipped) {
			// stored at full size, and the detector has put it there (block.rs:279-281: a clone, whatever its alpha)
			const uint32_t vb = __builtin_amdgcn_readfirstlane(vb_cur);
			if (threadIdx.x == 0) {
				if (FULL || a.out_w) a.out_w[tile_g] = 64u;
				if (FULL || a.out_h) a.out_h[tile_g] = 64u;
				if (a.finish_here) {
					if (n_fin < kFin64) s_fin[n_fin++] = tile_g;
					else finish_tile(make_uint2(vb, vb), 64u, 64u, (uint32_t)MODE, a.factor, a.value, a.lod0, a.lod1, tile_g);
				}
			}
			prefetch_next();
			publish_ticket();
			__syncthreads();
			advance();
			continue;
		}
		// ---- stage: registers -> planar u16 pairs
		// (every alpha byte is 255 iff the smallest of the 16 pixel dwords is >= 0xff000000: three-way minima)
		uint32_t least = 0xffffffffu;
		if constexpr (C == 4) {
			const uint32_t m0 = min(min(pre[0].x, pre[0].y), pre[0].z), m1 = min(min(pre[0].w, pre[1].x), pre[1].y);
			const uint32_t m2 = min(min(pre[1].z, pre[1].w), pre[2].x), m3 = min(min(pre[2].y, pre[2].z), pre[2].w);
			const uint32_t m4 = min(min(pre[3].x, pre[3].y), pre[3].z);
			least = min(min(min(m0, m1), m2), min(min(m3, m4), pre[3].w));
		}
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const uint32_t row = 16u * wave + (lane >> 4) + 4u * (uint32_t)k, col = lane & 15u;
			const uint4 v = pre[k];
			uint32_t *d = s_pl + row * kRS64 + col * 2u;
			if constexpr (C == 4) {
#pragma unroll
				for (uint32_t c = 0; c < NCH; ++c) {
					const uint32_t sel = c | 0x0c000c00u | ((4u + c) << 16);
					uint2 pr;
					pr.x = __builtin_amdgcn_perm(v.y, v.x, sel);
					pr.y = __builtin_amdgcn_perm(v.w, v.z, sel);
					*reinterpret_cast<uint2 *>(d + c * kPD64) = pr;
				}
			} else {
				// bytes R0 G0 B0 R1 | G1 B1 R2 G2 | B2 R3 G3 B3 -> the u16 pairs (c0, c1), (c2, c3) of each plane
				*reinterpret_cast<uint2 *>(d) = make_uint2(__builtin_amdgcn_perm(v.x, v.x, 0x0c070c00u), __builtin_amdgcn_perm(v.z, v.y, 0x0c050c02u));
				*reinterpret_cast<uint2 *>(d + kPD64) = make_uint2(__builtin_amdgcn_perm(v.y, v.x, 0x0c040c01u), __builtin_amdgcn_perm(v.z, v.y, 0x0c060c03u));
				*reinterpret_cast<uint2 *>(d + 2 * kPD64) = make_uint2(__builtin_amdgcn_perm(v.y, v.x, 0x0c050c02u), __builtin_amdgcn_perm(v.z, v.z, 0x0c070c00u));
			}
		}
		const bool wave_transparent = C == 4 && __builtin_amdgcn_ballot_w64(least < 0xff000000u) != 0ull;
		if (lane == 0) s_red[8 + wave] = wave_transparent ? 1u : 0u;
		prefetch_next();      // lands while this tile is processed
		__syncthreads();      // B1: the whole tile is staged
		if constexpr (ALPHA) {
			if (a.all_tiles && threadIdx.x == 0 && (s_red[8] | s_red[9] | s_red[10] | s_red[11]) != 0u) ++n_transparent;
		}
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
		if constexpr (MODE == 0) given_bits = ALPHA ? a.sums[2 * tile_g] : vb_cur;  // MODE 0: the value is there already (oklab_kernel)
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
				rA[c] = pk_mad_u16(__builtin_amdgcn_alignbit(a1, a0, 16), two, add2x16(a0, a1));
				rB[c] = pk_mad_u16(__builtin_amdgcn_alignbit(b1, b0, 16), two, add2x16(b0, b1));
				tP[c] = add2x16(a0, b0);
				dP[c] = b0;
			}
#pragma unroll
			for (int st = 0; st < 4; ++st) {
				if (st < 3 || !short_group) {
#pragma unroll
					for (int c = 0; c < 3; ++c) {
						const uint32_t *pr = pc[c] + (2 + 2 * st) * (int)kRS64;
						const uint32_t n0 = pr[0], n1 = pr[1], o0 = pr[kRS64], o1 = pr[kRS64 + 1];
						const uint32_t rN = pk_mad_u16(__builtin_amdgcn_alignbit(n1, n0, 16), two, add2x16(n0, n1));
						sum_hz = sad16(rN, rA[c], sum_hz);
						const uint32_t tN = add2x16(dP[c], n0);
						const uint32_t c0 = add2x16(tP[c], tN);
						sum_vr = sad16(dpp_mov<0x130>(c0), c0, sum_vr);  // wave_shl:1 = the pair to the right
						const uint32_t rO = pk_mad_u16(__builtin_amdgcn_alignbit(o1, o0, 16), two, add2x16(o0, o1));
						sum_hz = sad16(rO, rB[c], sum_hz);
						const uint32_t tO = add2x16(n0, o0);
						const uint32_t e0 = add2x16(tN, tO);
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
			if (a.finish_here) {
				if (n_fin < kFin64) s_fin[n_fin++] = this_tile;
				else finish_tile(make_uint2(sum_hz, sum_vr), 64u, 64u, (uint32_t)MODE, a.factor, a.value, a.lod0, a.lod1, this_tile);  // (list full: on the spot)
			}
		}
		if (!FULL && a.out_px == nullptr) {
			__syncthreads();  // s_red is rewritten by the next tile
			continue;
		}
		uint8_t *dst = a.out_px + (size_t)this_tile * (64u * 64u * (uint32_t)C);
		if (nw == 64u && nh == 64u) {
			// clone (block.rs:279-281): re-interleave this wave's 16 rows, 16 bytes per lane and step.  (Round 4 tried the copy straight
			// from the frame rows instead -- 8 memory instructions for 12 LDS reads and 32 permutes per lane: no faster, and the rows
			// are NOT in L2 any more by then: 0.68 GB more HBM traffic per 8 x 8K in shrink_by, where two thirds of the tiles are clones.)
#pragma unroll
			for (int k = 0; k < 4; ++k) {
				const uint32_t i = lane + 64u * (uint32_t)k;  // 256 groups of 4 pixels
				const uint32_t row = 16u * wave + (i >> 4), c4 = i & 15u;
				const uint32_t *p = s_pl + row * kRS64 + c4 * 2u;
				const uint2 r = *reinterpret_cast<const uint2 *>(p), g = *reinterpret_cast<const uint2 *>(p + kPD64);
				const uint2 b = *reinterpret_cast<const uint2 *>(p + 2 * kPD64);
				uint2 al = make_uint2(0x00ff00ffu, 0x00ff00ffu);
				if constexpr (ALPHA) al = *reinterpret_cast<const uint2 *>(p + 3 * kPD64);
				const uint32_t rg01 = __builtin_amdgcn_perm(g.x, r.x, 0x06020400u), rg23 = __builtin_amdgcn_perm(g.y, r.y, 0x06020400u);
				if constexpr (C == 4) {
					const uint32_t ba01 = __builtin_amdgcn_perm(al.x, b.x, 0x06020400u), ba23 = __builtin_amdgcn_perm(al.y, b.y, 0x06020400u);
					uint4 o;
					o.x = __builtin_amdgcn_perm(ba01, rg01, 0x05040100u);
					o.y = __builtin_amdgcn_perm(ba01, rg01, 0x07060302u);
					o.z = __builtin_amdgcn_perm(ba23, rg23, 0x05040100u);
					o.w = __builtin_amdgcn_perm(ba23, rg23, 0x07060302u);
					reinterpret_cast<uint4 *>(dst)[row * 16u + c4] = o;
				} else {
					uint3 o;
					o.x = __builtin_amdgcn_perm(b.x, rg01, 0x02040100u);                  // R0 G0 B0 R1
					const uint32_t gb1 = __builtin_amdgcn_perm(b.x, rg01, 0x0c0c0603u);   // G1 B1 . .
					o.y = __builtin_amdgcn_perm(rg23, gb1, 0x05040100u);                  // G1 B1 R2 G2
					o.z = __builtin_amdgcn_perm(b.y, rg23, 0x06030204u);                  // B2 R3 G3 B3
					reinterpret_cast<uint3 *>(dst)[row * 16u + c4] = o;
				}
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
		const uint32_t tlx = need_h ? lx : ly, tly = need_v ? ly : lx;
		const uint32_t *mx_tail = s_tail + kTail64 * (tlx < kTailLevels ? tlx : kTailLevels - 1u);  // bias[32], ksum[32], flag
		const uint32_t *my_tail = s_tail + kTail64 * (tly < kTailLevels ? tly : kTailLevels - 1u);
		const uint32_t px_ = a.precision[need_h ? lx : ly], py = a.precision[need_v ? ly : lx];
		const int32_t top_x = (int32_t)((256u << px_) - 1u), top_y = (int32_t)((256u << py) - 1u);
		const uint32_t o = lane & 15u, g = lane >> 4;
		const v4i32 zero = {0, 0, 0, 0};
		// horizontal pass of this wave's 16 rows into s_t[c][ox][y]; A = pixels of row 16w + o, columns 16g .. 16g+15
		auto hpass = [&](const bool signed_t) {
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
						// rows 16w + 4g .. +3 of column ox = 16nb + o; kept as p - 128 (the vertical product's operand form) when a
						// vertical pass follows: one xor here instead of four per channel on the one or two waves that run that pass
						s_t[(c * 32u + 16u * nb + o) * kTS64 + 4u * wave + g] = signed_t ? packed ^ 0x80808080u : packed;
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
					bv[0] = (int)tv.x;  // (bytes are p - 128 already: hpass(true) / the width-kept staging below)
					bv[1] = (int)tv.y;
					bv[2] = (int)tv.z;
					bv[3] = (int)tv.w;
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
						if (oy < nh) store_px(dst, oy * row_w + ox0 + oxl, pix[r]);
					}
				}
			}
		};
		if (need_h && need_v) {
			hpass(true);
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
						s_t[(c * 32u + xl) * kTS64 + 4u * wave + j] = (b0 | (b1 << 8) | (b2 << 16) | (b3 << 24)) ^ 0x80808080u;
					}
				}
				__syncthreads();
				vpass(32u * half, 64u, 2u, false);
				__syncthreads();  // s_t is refilled (next half, or the next tile's horizontal pass after only B1/B2)
			}
		} else {
			// height kept: the horizontal pass is the result; gather [c][ox][y] bytes into pixels
			hpass(false);
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
					store_px(dst, (4u * yq + r) * nw + ox, px);
				}
			}
			__syncthreads();
		}
	}
	list_flush(s_red + 16, n_listb, a.work + kWorkList, a.work + a.work_slot, threadIdx.x);
	list_flush(s_red + 32, n_lista, a.work + kWorkList + a.n_tiles, a.work + kWorkA + a.work_slot, threadIdx.x);
	if (a.finish_here) {
		// Detector sums -> stored value (f64 normalisation, hypot: finish_tile), one tile per thread, for the tiles this block
		// completed -- instead of a scan over every tile of the batch in the worklist kernel (round 4: that kernel then only walks
		// its lists, with a grid sized for them).  Tiles handed to a list carry the marker and are finished by whoever completes them.
		if (threadIdx.x == 0) s_red[14] = n_fin;
		__syncthreads();  // (s_fin / s_red were written by thread 0; the sums it wrote to memory are re-read by other threads: same CU, L2-coherent stores)
		const uint32_t n = s_red[14];
		for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
			const uint32_t t = s_fin[i];
			const unsigned long long k2 = __builtin_nontemporal_load(reinterpret_cast<const unsigned long long *>(a.sums) + t);  // (past L1)
			const uint2 key = make_uint2((uint32_t)k2, (uint32_t)(k2 >> 32));
			finish_tile(key, 64u, 64u, (uint32_t)MODE, a.factor, a.value, a.lod0, a.lod1, t);
		}
	}
	if constexpr (ALPHA) {
		if (a.all_tiles && threadIdx.x == 0 && n_transparent != 0u) atomicAdd(a.work + kWorkA + a.work_slot, n_transparent);
	}
}

// shrink_by on 64x64 tiles with clone_ahead (round 4): oklab2_kernel<64> has left every tile's value in sums[] and a copy of its
// pixels in its slot.  Two thirds of the tiles are stored at full size: those are finished right here (size, stored value); the rest
// -- and whatever is not a full tile -- is listed for shrink64_kernel<0>, which then never meets a finished tile.  (Skipping them
// inside that kernel saved a tenth of its time only: a block has one tile in flight, and what a skipped tile still cost -- its
// ticket's round trip, a block barrier -- is most of what a tile costs.)  One tile per thread; a block appends its tiles with one atomic.
// list[0] = the count (zeroed by the host before the launch), list[8 ..] = the tiles.
__global__ void __launch_bounds__(1024) clone_split64_kernel(const Fast64Args a)
{
	__shared__ uint32_t s_wave[16], s_base;
	const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
	bool listed = false;
	if (t < a.n_tiles) {
		const uint32_t r = t - fastdiv(t, a.div_tpf) * a.tiles_per_frame;
		const uint32_t ty = fastdiv(r, a.div_cols), tx = r - ty * a.cols;
		const uint2 pair = reinterpret_cast<const uint2 *>(a.sums)[t];  // (value bits, 1 = copied by the detector)
		const uint32_t vb = pair.x;
		const uint32_t key = __float_as_uint(parse_value(__uint_as_float(vb)));
		uint32_t m = 0;  // the level, as the kernels' ballot form counts it
#pragma unroll
		for (int j = 0; j < kMaxLevel; ++j) m += ((key < a.breaks[j]) != (a.breaks_asc != 0u)) ? 1u : 0u;
		if (tx < a.full_cols && ty < a.full_rows && m == 0u && pair.y == 1u) {
			a.out_w[t] = 64u;
			a.out_h[t] = 64u;
			finish_tile(make_uint2(vb, vb), 64u, 64u, 0u, a.factor, a.value, a.lod0, a.lod1, t);
		} else {
			listed = true;
		}
	}
	const unsigned long long mask = __builtin_amdgcn_ballot_w64(listed);
	if (lane == 0) s_wave[wave] = (uint32_t)__builtin_popcountll(mask);
	__syncthreads();
	if (threadIdx.x == 0) {
		uint32_t total = 0;
		for (uint32_t w = 0; w < blockDim.x / 64u; ++w) {
			const uint32_t n = s_wave[w];
			s_wave[w] = total;
			total += n;
		}
		s_base = total ? atomicAdd(a.clone_list, total) : 0u;
	}
	__syncthreads();
	if (listed) a.clone_list[8u + s_base + s_wave[wave] + (uint32_t)__builtin_popcountll(mask & ((1ull << lane) - 1ull))] = t;
}

// 64x64 flow, first part: the four-wave kernel (and its four-plane instance); ga = the arguments of the worklist
// kernel that follows (pxz_shrink_generic.hip: launch_shrink)
hipError_t launch_fast64(const ShrinkArgs &a, ShrinkArgs &ga, uint32_t channels, uint32_t n_cus, hipStream_t stream)
{
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
	f.ok_edges = a.ok_edges;
	f.clone_ahead = a.mode == 0 ? a.clone_ahead : 0u;
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
	f.factor = a.factor;
	f.value = a.value;
	f.lod0 = a.lod0;
	f.lod1 = a.lod1;
	const uint32_t lds_bytes = lds64_dwords(3) * 4u;
	constexpr uint32_t kLds = 160u * 1024u;
	const uint32_t per_cu = kLds / lds_bytes;
	const uint32_t resident = n_cus * per_cu;
	const uint32_t blocks = a.n_tiles < resident ? a.n_tiles : resident;
	hipError_t e = hipSuccess;
	// full tiles with transparency are on list A: the four-plane instance takes it when transparency was
	// announced or seen before, else the generic kernel walks it after list B.  When most tiles of the last launch had
	// transparency the four-plane instance takes EVERY tile and the opaque instance is not launched.
	const bool run_alpha = channels == 4 && a.out_px != nullptr && a.out_w != nullptr && a.out_h != nullptr && a.alpha_kernel != 0;
	const bool alpha_first = run_alpha && a.alpha_first != 0;
	// the opaque instance finishes the tiles it completes itself; the worklist kernel then finishes only what the lists hold (and
	// the tiles the four-plane instance completed).  With the four-plane instance first nobody finishes in passing: that kernel scans.
	f.finish_here = alpha_first ? 0u : 1u;
	ga.finish_scan = alpha_first ? 1u : 0u;
	f.clone_list = nullptr;
	if (!alpha_first && f.clone_ahead != 0u && a.clone_list != nullptr && a.out_w != nullptr && a.out_h != nullptr) {
		f.clone_list = a.clone_list;
		if ((e = hipMemsetAsync(a.clone_list, 0, 4, stream)) != hipSuccess) return e;
		hipLaunchKernelGGL(clone_split64_kernel, dim3((a.n_tiles + 1023u) / 1024u), dim3(1024), 0, stream, f);
		if ((e = hipGetLastError()) != hipSuccess) return e;
	}
	if (!alpha_first) {
		const bool full = a.out_px != nullptr && a.out_w != nullptr && a.out_h != nullptr;
		void (*k)(const Fast64Args) = a.mode == 1 ? (full ? shrink64_kernel<1, false, true> : shrink64_kernel<1, false, false>)
		                                          : (full ? shrink64_kernel<0, false, true> : shrink64_kernel<0, false, false>);
		if (channels == 3)  // RGB frames: 12-byte pixel quads in, RGB slots out
			k = a.mode == 1 ? (full ? shrink64_kernel<1, false, true, 3> : shrink64_kernel<1, false, false, 3>)
			                : (full ? shrink64_kernel<0, false, true, 3> : shrink64_kernel<0, false, false, 3>);
		if (lds_bytes > 64u * 1024u && (e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes)) != hipSuccess) return e;
		hipLaunchKernelGGL(k, dim3(blocks), dim3(256), lds_bytes, stream, f);
	}
	if ((e = hipGetLastError()) != hipSuccess) return e;
	if (!alpha_first && a.mid_event && (e = hipEventRecord(static_cast<hipEvent_t>(a.mid_event), stream)) != hipSuccess) return e;
	ga.mid_event = nullptr;
	ga.list_a_too = channels == 4 && a.out_px != nullptr && !run_alpha ? 1u : 0u;
	if (run_alpha) {
		f.all_tiles = alpha_first ? 1u : 0u;
		f.finish_here = 0u;  // (its tiles are finished by the worklist kernel: list A, or the scan)
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
		if (alpha_first && a.mid_event && (e = hipEventRecord(static_cast<hipEvent_t>(a.mid_event), stream)) != hipSuccess) return e;
	}
	return hipSuccess;
}

}  // namespace pxz
