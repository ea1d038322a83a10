// pxz_shrink32.hip -- the fast kernels for 32x32 RGBA tiles (shrink32_kernel, shrink32a_kernel for tiles with
// transparency) and for 2x2 groups of 16x16 tiles (shrink16_kernel), and the first part of their launch.
//
// Compiled with -ffp-contract=off: the f32 results of the Oklab detector are
// written into the bitstream, and the reference (Rust) never fuses a*b+c.
#include "pxz_device.h"

namespace pxz {


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
// C = 3: RGB frames read and RGB slots written directly (round 2) -- 12-byte pixel quads in, the same three LDS planes,
// no opacity test (there is no alpha), outputs packed to 3 bytes per pixel by the flush.
template <int MODE, bool FULL, int C = 4>
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
	// MODE 0 (round 4): the values travel TWO tiles ahead of the pixels.  With clone_ahead the detector has already copied every
	// tile into its slot as if it were stored at full size (oklab2_kernel: the pixels pass through its producers' registers
	// anyway, and it runs under an idle memory system); a tile whose value says so is finished -- it is not read a second time.
	bool pre_skipped = false;       // the current tile's loads were left out: it is one of those
	uint32_t vb_cur = 0, vb_next = 0;  // value bits of the current / the next tile
	uint32_t fl_next = 0;              // and whether the detector copied the next tile (sums[2 t + 1] == 1: it copies the tiles it guesses to be stored whole)
	auto value_pair_of = [&](uint32_t t) -> uint2 { return t < a.n_tiles ? reinterpret_cast<const uint2 *>(a.sums)[t] : make_uint2(0u, 0u); };
	auto stored_whole = [&](uint32_t vb, uint32_t copied) -> bool {
		return a.clone_ahead && copied == 1u && level_of(__float_as_uint(parse_value(__uint_as_float(vb)))) == 0u;
	};
	uint32_t second = 0xffffffffu;
	if constexpr (MODE == 0) {
		second = next_ticket();
		const uint2 p0 = value_pair_of(first), p1 = value_pair_of(second);
		vb_cur = __builtin_amdgcn_readfirstlane(p0.x);  // (scalars from here on)
		vb_next = __builtin_amdgcn_readfirstlane(p1.x);
		fl_next = __builtin_amdgcn_readfirstlane(p1.y);
		pre_skipped = stored_whole(vb_cur, __builtin_amdgcn_readfirstlane(p0.y));
		fast32_prefetch<C>(a, first, tid, pre, pre_valid, pre_skipped);
	} else {
		fast32_prefetch<C>(a, first, tid, pre, pre_valid);
	}
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
			if constexpr (C == 4) {
				uint32_t *d = reinterpret_cast<uint32_t *>(pend_dst);
				if (pend_px >= 4u) {  // whole 16-byte groups (pixel counts are powers of two)
					for (uint32_t i = tid; i < (pend_px >> 2); i += 64u)
						reinterpret_cast<uint4 *>(d)[i] = reinterpret_cast<const uint4 *>(pend_src)[i];
				} else if (tid < pend_px) {
					d[tid] = pend_src[tid];
				}
			} else {
				// RGB slots: four parked pixels (dwords, alpha 255) -> 12 bytes
				if (pend_px >= 4u) {
					for (uint32_t i = tid; i < (pend_px >> 2); i += 64u) {
						const uint4 v = reinterpret_cast<const uint4 *>(pend_src)[i];
						uint3 o;
						o.x = __builtin_amdgcn_perm(v.y, v.x, 0x04020100u);  // R0 G0 B0 R1
						o.y = __builtin_amdgcn_perm(v.z, v.y, 0x05040201u);  // G1 B1 R2 G2
						o.z = __builtin_amdgcn_perm(v.w, v.z, 0x06050402u);  // B2 R3 G3 B3
						reinterpret_cast<uint3 *>(pend_dst)[i] = o;
					}
				} else if (tid < pend_px) {
					const uint32_t v = pend_src[tid];
					pend_dst[3u * tid] = (uint8_t)v;
					pend_dst[3u * tid + 1u] = (uint8_t)(v >> 8);
					pend_dst[3u * tid + 2u] = (uint8_t)(v >> 16);
				}
			}
		} else if (pend_kind == 2) {
			// clone (block.rs:279-281): re-interleave the planes, one 16-byte (RGB: 12-byte) store per 4 pixels
#pragma unroll
			for (int k = 0; k < 4; ++k) {
				const uint32_t i = tid + 64u * (uint32_t)k;
				const uint32_t *p = s_pl + (i >> 3) * kRS32 + (i & 7u) * 2u;
				const uint2 r = *reinterpret_cast<const uint2 *>(p), g = *reinterpret_cast<const uint2 *>(p + kPD32);
				const uint2 b = *reinterpret_cast<const uint2 *>(p + 2 * kPD32);
				const uint32_t rg01 = __builtin_amdgcn_perm(g.x, r.x, 0x06020400u), rg23 = __builtin_amdgcn_perm(g.y, r.y, 0x06020400u);
				if constexpr (C == 4) {
					const uint32_t opq = 0x00ff00ffu;  // the tile is opaque: alpha pair (255, 255)
					const uint32_t ba01 = __builtin_amdgcn_perm(opq, b.x, 0x06020400u), ba23 = __builtin_amdgcn_perm(opq, b.y, 0x06020400u);
					uint4 o;
					o.x = __builtin_amdgcn_perm(ba01, rg01, 0x05040100u);
					o.y = __builtin_amdgcn_perm(ba01, rg01, 0x07060302u);
					o.z = __builtin_amdgcn_perm(ba23, rg23, 0x05040100u);
					o.w = __builtin_amdgcn_perm(ba23, rg23, 0x07060302u);
					reinterpret_cast<uint4 *>(pend_dst)[i] = o;
				} else {
					uint3 o;
					o.x = __builtin_amdgcn_perm(b.x, rg01, 0x02040100u);                        // R0 G0 B0 R1
					const uint32_t gb1 = __builtin_amdgcn_perm(b.x, rg01, 0x0c0c0603u);         // G1 B1 . .
					o.y = __builtin_amdgcn_perm(rg23, gb1, 0x05040100u);                        // G1 B1 R2 G2
					o.z = __builtin_amdgcn_perm(b.y, rg23, 0x06030204u);                        // B2 R3 G3 B3
					reinterpret_cast<uint3 *>(pend_dst)[i] = o;
				}
			}
		}
		pend_kind = 0;
	};
	auto one_tile = [&](const uint32_t tile_g, const uint32_t tile_next) {
		auto prefetch_next = [&]() __attribute__((always_inline)) {
			if constexpr (MODE == 0) {
				pre_skipped = stored_whole(vb_next, fl_next);
				fast32_prefetch<C>(a, tile_next, tid, pre, pre_valid, pre_skipped);
			} else {
				fast32_prefetch<C>(a, tile_next, tid, pre, pre_valid);
			}
		};
		auto defer = [&]() {
			list_push(s_batch, n_listb, tile_g, a.work + kWorkList, a.work + a.work_slot, tid);
			if (tid == 0) {
				bool keep = false;  // MODE 0: a value the block-cooperative detector left is final and stays
				if constexpr (MODE == 0) {
					const uint32_t t = tile_g - fastdiv(tile_g, a.div_tpf) * a.tiles_per_frame;
					const uint32_t ty = fastdiv(t, a.div_cols), tx = t - ty * a.cols;
					keep = oklab_value_given(a, tx, ty);
				}
				if (!keep) reinterpret_cast<uint2 *>(a.sums)[tile_g] = make_uint2(kDeferredKey, kDeferredKey);  // not finished here
			}
		};
		if (!pre_valid) {  // ragged edge / unaligned rows: generic kernel
			flush();
			defer();
			prefetch_next();
			return;
		}
		uint32_t given_bits = 0;
		if constexpr (MODE == 0) given_bits = vb_cur;
		// (MODE 0) stored at full size, and the detector has put it there (block.rs:279-281: a clone, whatever its alpha): nothing was
		// requested, nothing is staged; the tile only passes the two points where the pipeline moves on
		const bool skipped = MODE == 0 && pre_skipped;
		// ---- wait for the prefetched registers (the opacity test is their first use), then emit the
		// previous tile's parked pixels, then stage: registers -> planar u16 pairs
		// every alpha byte is 255 iff the smallest of the 16 pixel dwords is >= 0xff000000 (alpha is the top byte): eight
		// three-way minima instead of sixteen ANDs and a shift
		bool transparent = false;
		if (C == 4 && !skipped) {
			uint32_t least;
			const uint32_t m0 = min(min(pre[0].x, pre[0].y), pre[0].z), m1 = min(min(pre[0].w, pre[1].x), pre[1].y);
			const uint32_t m2 = min(min(pre[1].z, pre[1].w), pre[2].x), m3 = min(min(pre[2].y, pre[2].z), pre[2].w);
			const uint32_t m4 = min(min(pre[3].x, pre[3].y), pre[3].z);
			least = min(min(min(m0, m1), m2), min(min(m3, m4), pre[3].w));
			transparent = __builtin_amdgcn_ballot_w64(least < 0xff000000u) != 0ull;
		}
		__builtin_amdgcn_sched_barrier(0);
		flush();
		__builtin_amdgcn_sched_barrier(0);
		if (!skipped) {
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const uint32_t row = (tid >> 3) + 8u * (uint32_t)k, col = tid & 7u;
			const uint4 v = pre[k];
			uint32_t *d = s_pl + row * kRS32 + col * 2u;
			if constexpr (C == 4) {
#pragma unroll
				for (uint32_t c = 0; c < 3; ++c) {
					const uint32_t sel = c | 0x0c000c00u | ((4u + c) << 16);
					uint2 pr;
					pr.x = __builtin_amdgcn_perm(v.y, v.x, sel);
					pr.y = __builtin_amdgcn_perm(v.w, v.z, sel);
					lds_store2(d + c * kPD32, pr);
				}
			} else {
				// bytes R0 G0 B0 R1 | G1 B1 R2 G2 | B2 R3 G3 B3 -> the u16 pairs (c0, c1), (c2, c3) of each plane
				lds_store2(d, make_uint2(__builtin_amdgcn_perm(v.x, v.x, 0x0c070c00u), __builtin_amdgcn_perm(v.z, v.y, 0x0c050c02u)));
				lds_store2(d + kPD32, make_uint2(__builtin_amdgcn_perm(v.y, v.x, 0x0c040c01u), __builtin_amdgcn_perm(v.z, v.y, 0x0c060c03u)));
				lds_store2(d + 2 * kPD32, make_uint2(__builtin_amdgcn_perm(v.y, v.x, 0x0c050c02u), __builtin_amdgcn_perm(v.z, v.z, 0x0c070c00u)));
			}
		}
		}
		PXZ_STAMP(0);  // wait for the prefetched pixels + staging
		prefetch_next();  // lands while this tile is processed
		if (skipped) {
			uint32_t lane = tid;
			asm volatile("" : "+v"(lane));
			if (lane == 0) {
				if (FULL || a.out_w) a.out_w[tile_g] = 32u;
				if (FULL || a.out_h) a.out_h[tile_g] = 32u;
			}
			return;
		}
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
				const uint32_t a0 = lds_dword(pc[0] + c * kPD32), a1 = lds_dword(pc[0] + c * kPD32 + 1);
				const uint32_t b0 = lds_dword(pc[0] + c * kPD32 + kRS32), b1 = lds_dword(pc[0] + c * kPD32 + kRS32 + 1);
				rA[c] = pk_mad_u16(__builtin_amdgcn_alignbit(a1, a0, 16), two, add2x16(a0, a1));
				rB[c] = pk_mad_u16(__builtin_amdgcn_alignbit(b1, b0, 16), two, add2x16(b0, b1));
				tP[c] = add2x16(a0, b0);
				dP[c] = b0;
			}
#pragma unroll
			for (int st = 0; st < 4; ++st) {
				if (st < 3 || last_rows(g)) {  // the last group has 6 window rows = 3 steps
#pragma unroll
					for (int c = 0; c < 3; ++c) {
						const uint32_t *pr = pc[0] + c * kPD32 + (2 + 2 * st) * (int)kRS32;
						const uint32_t n0 = lds_dword(pr), n1 = lds_dword(pr + 1), o0 = lds_dword(pr + kRS32), o1 = lds_dword(pr + kRS32 + 1);
						const uint32_t rN = pk_mad_u16(__builtin_amdgcn_alignbit(n1, n0, 16), two, add2x16(n0, n1));
						sum_hz = sad16(rN, rA[c], sum_hz);
						const uint32_t tN = add2x16(dP[c], n0);
						const uint32_t c0 = add2x16(tP[c], tN);
						sum_vr = sad16(dpp_mov<0x101>(c0), c0, sum_vr);  // row_shl:1 = the pair to the right
						const uint32_t rO = pk_mad_u16(__builtin_amdgcn_alignbit(o1, o0, 16), two, add2x16(o0, o1));
						sum_hz = sad16(rO, rB[c], sum_hz);
						const uint32_t tO = add2x16(n0, o0);
						const uint32_t e0 = add2x16(tN, tO);
						sum_vr = sad16(dpp_mov<0x101>(e0), e0, sum_vr);
						rA[c] = rN;
						rB[c] = rO;
						tP[c] = tO;
						dP[c] = o0;
					}
				}
			}
			{
				uint32_t qq = q;
				asm volatile("" : "+v"(qq));  // (a fresh compare, not a hoisted and spilled lane mask)
				if (qq == 15u) sum_hz = sum_vr = 0;  // pair 15 starts no window (x = 30, 31)
			}
			sum_hz = wave_sum_sgpr(sum_hz);
			sum_vr = wave_sum_sgpr(sum_vr);
			m0 = level_of(sum_hz);
			m1 = level_of(sum_vr);
			key0 = sum_hz;
			key1 = sum_vr;
		} else {
			m0 = m1 = level_of(__float_as_uint(parse_value(__uint_as_float(given_bits))));
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
			pend_dst = a.out_px + (size_t)tile_g * (1024u * (uint32_t)C);
			pend_px = nw * nh;
			if (nw == 32u && nh == 32u) {
				pend_kind = 2;  // the planes themselves, re-interleaved by the flush
			} else if (nw != 32u && nh != 32u && filt != 0) {
				const uint32_t lx = m0 < (uint32_t)kMaxLevel ? m0 : (uint32_t)kMaxLevel - 1;
				const uint32_t ly = m1 < (uint32_t)kMaxLevel ? m1 : (uint32_t)kMaxLevel - 1;
				const bool both_mf = a.tabs[lx].mf_off != 0 && a.tabs[ly].mf_off != 0;
				if (a.narrow && both_mf && nw <= 4u && nh <= 16u && (s_tab[a.tabs[lx].mf_off + 288u] & s_tab[a.tabs[ly].mf_off + 288u])) {
					// narrow outputs (4, 2, 1 px wide): the three channels through one accumulator
					resample_mfma32_narrow(s_tab, a.tabs[lx], a.tabs[ly], s_pl, tid, nw, nh, s_tmp);
					pend_src = s_tmp;
					pend_kind = 1;
				} else if (both_mf && nw >= 4u && nh >= 4u) {
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
	for (uint32_t tile_g = first, tile_next = second; tile_g < a.n_tiles;) {
#ifdef PXZ_STAMPS
		++st_tiles;
#endif
		if constexpr (MODE == 0) {
			const uint32_t tile_after = next_ticket();
			const uint2 pair_after = value_pair_of(tile_after);  // needed at the next tile's prefetch: a whole tile's time away
			one_tile(tile_g, tile_next);
			tile_g = tile_next;
			tile_next = tile_after;
			vb_cur = vb_next;
			vb_next = __builtin_amdgcn_readfirstlane(pair_after.x);
			fl_next = __builtin_amdgcn_readfirstlane(pair_after.y);
		} else {
			tile_next = next_ticket();
			one_tile(tile_g, tile_next);
			tile_g = tile_next;
		}
	}
#ifdef PXZ_STAMPS
	unsigned long long st_loop_end;
	asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_loop_end)::"memory");
#endif
	flush();  // the last tile's pixels
	list_flush(s_batch, n_listb, a.work + kWorkList, a.work + a.work_slot, tid);
	list_flush(s_batch + kListBatch, n_lista, a.work + kWorkList + a.n_tiles, a.work + kWorkA + a.work_slot, tid);
	if (a.finish_here) {
		// Detector sums -> stored value (f64 normalisation, hypot: finish_tile), one tile per thread, for the tiles this
		// block dealt to its waves -- instead of a scan over every tile of the batch in the worklist kernel.  Tiles handed
		// to a list carry the marker and are finished by whoever completes them.
		__threadfence_block();  // the sums were written by other waves of this block
		__syncthreads();
		for (uint32_t i = threadIdx.x;; i += blockDim.x) {
			const unsigned long long run = (unsigned long long)(i >> a.chunk_lg) * gridDim.x + blockIdx.x;
			const unsigned long long g = (run << a.chunk_lg) + (i & ((1u << a.chunk_lg) - 1u));
			if (g >= (unsigned long long)a.n_tiles) break;  // (g grows with i)
			const uint2 key = reinterpret_cast<const uint2 *>(a.sums)[g];
			if (key.x == kDeferredKey && key.y == kDeferredKey) continue;
			finish_tile(key, 32u, 32u, (uint32_t)MODE, a.factor, a.value, a.lod0, a.lod1, (uint32_t)g);
		}
	}
#ifdef PXZ_STAMPS
	if (tid == 0) {
		unsigned long long *out = reinterpret_cast<unsigned long long *>(a.work + ((2u * a.n_tiles + kWorkList + 1u + 1u) & ~1u));
		for (int i = 0; i < 8; ++i) atomicAdd(out + i, st_acc[i]);
		// per-wave run time (100 MHz ticks) | tiles processed << 48; last launch wins
		out[8 + blockIdx.x * 17u + sub] = ((st_loop_end - st_begin) & 0xffffffffffffull) | ((unsigned long long)st_tiles << 48);  // when the wave ran out of tiles
		if (sub == 0) out[8 + blockIdx.x * 17u + 16u] = st_begin;
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
	// all_tiles (round 2): every tile of the batch comes here first -- the host has seen that most tiles of the last
	// launch had transparency, so shrink32_kernel would only read, test and list them.  An opaque tile comes out of this
	// kernel as it does out of the other one (premultiplying by 255 and dividing by it again are identities, and a
	// constant-255 plane convolves to the weight sums the opaque path uses).
	const uint32_t count = a.all_tiles ? a.n_tiles : __builtin_amdgcn_readfirstlane(a.work[kWorkA + a.work_slot]);
	const uint32_t *list = a.work + kWorkList + a.n_tiles;
	uint32_t n_transparent = 0;  // all_tiles: what the list counter would have said (the host's statistic)
	auto tile_of_ticket = [&](uint32_t t) -> uint32_t {
		const unsigned long long i = (unsigned long long)blockIdx.x + (unsigned long long)t * gridDim.x;
		return i < (unsigned long long)count ? (a.all_tiles ? (uint32_t)i : list[(uint32_t)i]) : 0xffffffffu;
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
		if (!pre_valid) {  // (all_tiles only: a ragged-edge tile, or unaligned rows) generic kernel
			if (tid == 0) {
				bool keep = false;  // MODE 0: a value the block-cooperative detector left is final and stays
				if constexpr (MODE == 0) {
					const uint32_t t = tile_g - fastdiv(tile_g, a.div_tpf) * a.tiles_per_frame;
					const uint32_t ty = fastdiv(t, a.div_cols), tx = t - ty * a.cols;
					keep = oklab_value_given(a, tx, ty);
				}
				if (!keep) reinterpret_cast<uint2 *>(a.sums)[tile_g] = make_uint2(kDeferredKey, kDeferredKey);
			}
			defer();
			fast32_prefetch(a, tile_next, tid, pre, pre_valid);
			tile_g = tile_next;
			continue;
		}
		uint32_t given_bits = 0;
		if constexpr (MODE == 0) given_bits = a.sums[2 * tile_g];
		if (a.all_tiles) {
			const uint32_t m0 = min(min(pre[0].x, pre[0].y), pre[0].z), m1 = min(min(pre[0].w, pre[1].x), pre[1].y);
			const uint32_t m2 = min(min(pre[1].z, pre[1].w), pre[2].x), m3 = min(min(pre[2].y, pre[2].z), pre[2].w);
			const uint32_t m4 = min(min(pre[3].x, pre[3].y), pre[3].z);
			const uint32_t least = min(min(min(m0, m1), m2), min(min(m3, m4), pre[3].w));
			n_transparent += __builtin_amdgcn_ballot_w64(least < 0xff000000u) != 0ull ? 1u : 0u;
		}
		// ---- stage: registers -> four planes of u16 pairs (list mode: only full, aligned tiles are ever listed)
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
				lds_store2(d + c * kPD32, pr);
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
				rA[c] = pk_mad_u16(__builtin_amdgcn_alignbit(a1, a0, 16), two, add2x16(a0, a1));
				rB[c] = pk_mad_u16(__builtin_amdgcn_alignbit(b1, b0, 16), two, add2x16(b0, b1));
				tP[c] = add2x16(a0, b0);
				dP[c] = b0;
			}
#pragma unroll
			for (int st = 0; st < 4; ++st) {
				if (st < 3 || g != 3u) {
#pragma unroll
					for (int c = 0; c < 3; ++c) {
						const uint32_t *pr = pc[c] + (2 + 2 * st) * (int)kRS32;
						const uint32_t n0 = pr[0], n1 = pr[1], o0 = pr[kRS32], o1 = pr[kRS32 + 1];
						const uint32_t rN = pk_mad_u16(__builtin_amdgcn_alignbit(n1, n0, 16), two, add2x16(n0, n1));
						sum_hz = sad16(rN, rA[c], sum_hz);
						const uint32_t tN = add2x16(dP[c], n0);
						const uint32_t c0 = add2x16(tP[c], tN);
						sum_vr = sad16(dpp_mov<0x101>(c0), c0, sum_vr);
						const uint32_t rO = pk_mad_u16(__builtin_amdgcn_alignbit(o1, o0, 16), two, add2x16(o0, o1));
						sum_hz = sad16(rO, rB[c], sum_hz);
						const uint32_t tO = add2x16(n0, o0);
						const uint32_t e0 = add2x16(tN, tO);
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
			m0 = m1 = level_of(__float_as_uint(parse_value(__uint_as_float(given_bits))));
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
	if (a.all_tiles && n_transparent != 0u && tid == 0) atomicAdd(a.work + kWorkA + a.work_slot, n_transparent);
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
// C = 3 (round 2): RGB frames read as 12-byte pixel quads, RGB slots written; no opacity test.
template <int MODE, bool FULL, int C = 4>
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
		p.src = a.src + (size_t)p.frame * a.frame_stride + (size_t)(p.gy * 32u) * a.pitch + (size_t)(p.gx * 32u) * (uint32_t)C;
		p.full = 2u * p.gx + 1u < a.full_cols && 2u * p.gy + 1u < a.full_rows;
		return p;
	};
	uint4 pre[4];
	bool pre_valid = false;
	// MODE 0 with clone_ahead (round 4): oklab2_kernel<16> has copied every tile into its slot as if it were stored at full size.  A
	// tile whose value says it is keeps that copy (no clone here); a group of four such tiles is not even read.  The values of the
	// NEXT group's tiles are requested as soon as the group is known (the top of an iteration) and decide its prefetch in mid-iteration;
	// they then are that group's `given` values.
	bool pre_skipped = false;
	uint32_t pre_mask = 0u;  // which tiles of the group in flight were left out of its loads
	uint32_t gv[4] = {0, 0, 0, 0}, gvn[4] = {0, 0, 0, 0};
	uint32_t gf[4] = {0, 0, 0, 0}, gfn[4] = {0, 0, 0, 0};  // sums[2 t + 1] of the same tiles: 1 = the detector copied the tile
	auto stored_whole = [&](uint32_t vb, uint32_t copied) -> bool {
		return __builtin_amdgcn_readfirstlane(copied) == 1 && level_of(__float_as_uint(parse_value(__uint_as_float(__builtin_amdgcn_readfirstlane(vb))))) == 0u;
	};
	auto request_values = [&](uint32_t grp, uint32_t (&v)[4], uint32_t (&f)[4]) {
		const Place p = place_of(grp);
		if (p.full) {
			const uint32_t t00 = p.frame * a.tiles_per_frame + (2u * p.gy) * a.cols + 2u * p.gx;
#pragma unroll
			for (uint32_t k = 0; k < 4; ++k) {
				const uint2 pr = reinterpret_cast<const uint2 *>(a.sums)[t00 + (k & 1u) + (k >> 1) * a.cols];
				v[k] = pr.x;
				f[k] = pr.y;
			}
		}
	};
	// whole_mask (MODE 0 with clone_ahead): bit k = tile k of the group is stored at full size and has the detector's copy in its
	// slot -- its quarter of the region is not requested (a tile row is one 64-byte sector: what a lane leaves out is never
	// fetched); the image then holds stale bytes there, which only reach outputs nobody stores.  All four: nothing is requested.
	auto prefetch = [&](uint32_t grp, uint32_t whole_mask = 0u) {
		const Place p = place_of(grp);
		pre_valid = p.full;
		if (pre_valid && whole_mask != 15u) {
			const uint8_t *q = p.src + (size_t)(tid >> 3) * a.pitch + (tid & 7u) * (4u * (uint32_t)C);
			const bool right = (tid & 4u) != 0u;  // this lane's pixel quad lies in the right-hand tiles
			const bool want_top = C == 3 || ((right ? whole_mask >> 1 : whole_mask) & 1u) == 0u;
			const bool want_bottom = C == 3 || ((right ? whole_mask >> 3 : whole_mask >> 2) & 1u) == 0u;
#pragma unroll
			for (int k = 0; k < 4; ++k) {
				if constexpr (C == 4) {
					if (k < 2 ? want_top : want_bottom) pre[k] = *reinterpret_cast<const uint4 *>(q + (size_t)(8 * k) * a.pitch);
				} else {
					const uint3 v = *reinterpret_cast<const uint3 *>(q + (size_t)(8 * k) * a.pitch);  // rows are 4-byte aligned
					pre[k] = make_uint4(v.x, v.y, v.z, 0u);
				}
			}
		}
	};
	auto whole_mask_of = [&](const uint32_t (&v)[4], const uint32_t (&f)[4]) -> uint32_t {
		return (stored_whole(v[0], f[0]) ? 1u : 0u) | (stored_whole(v[1], f[1]) ? 2u : 0u) | (stored_whole(v[2], f[2]) ? 4u : 0u) | (stored_whole(v[3], f[3]) ? 8u : 0u);
	};
	const uint32_t first = group_of_ticket(__builtin_amdgcn_readfirstlane(sub));
	if constexpr (MODE == 0) {
		request_values(first, gv, gf);
		if (a.clone_ahead && place_of(first).full) pre_mask = whole_mask_of(gv, gf);
		pre_skipped = pre_mask == 15u;
	}
	prefetch(first, pre_mask);
	for (uint32_t grp = first; grp < a.n_groups;) {
		const uint32_t grp_next = next_ticket();
		if constexpr (MODE == 0) request_values(grp_next, gvn, gfn);
		auto prefetch_next = [&]() __attribute__((always_inline)) {
			if constexpr (MODE == 0) {
				pre_mask = a.clone_ahead && place_of(grp_next).full ? whole_mask_of(gvn, gfn) : 0u;
				pre_skipped = pre_mask == 15u;
				prefetch(grp_next, pre_mask);
			} else {
				prefetch(grp_next);
			}
		};
		auto advance = [&]() __attribute__((always_inline)) {
			grp = grp_next;
			if constexpr (MODE == 0) {
#pragma unroll
				for (int k = 0; k < 4; ++k) gv[k] = gvn[k];
			}
		};  // (the flags are only looked at in prefetch_next(): gfn, then cur_mask)
		const bool skipped = MODE == 0 && pre_skipped;  // (this group's loads were left out)
		const uint32_t cur_mask = pre_mask;              // (and which of its tiles': prefetch_next() moves pre_mask on)
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
					keep = oklab_value_given(a, tx, ty);
				}
				if (!keep) reinterpret_cast<uint2 *>(a.sums)[t] = make_uint2(kDeferredKey, kDeferredKey);
			}
		};
		if (!pre_valid) {
			// a ragged or incomplete group (or an unaligned batch): its tiles one by one to the generic kernel
#pragma unroll
			for (uint32_t k = 0; k < 4; ++k)
				if (2u * pl.gx + (k & 1u) < a.cols && 2u * pl.gy + (k >> 1) < a.rows) defer_tile(tile_id(k));
			prefetch_next();
			advance();
			continue;
		}
		if (skipped) {
			// all four tiles are stored at full size, and the detector has put them there (block.rs:279-281: clones, whatever their alpha)
			prefetch_next();
			uint32_t lane = tid;
			asm volatile("" : "+v"(lane));
			if (lane < 4u) {
				const uint32_t t = t00 + (lane & 1u) + (lane >> 1) * a.cols;
				if (FULL || a.out_w) a.out_w[t] = 16u;
				if (FULL || a.out_h) a.out_h[t] = 16u;
			}
			advance();
			continue;
		}
		uint32_t given[4] = {0, 0, 0, 0};
		if constexpr (MODE == 0) {
#pragma unroll
			for (uint32_t k = 0; k < 4; ++k) given[k] = gv[k];
		}
		// ---- stage: registers -> planar u16 pairs (as shrink32_kernel)
		// every alpha byte is 255 iff the smallest of the 16 pixel dwords is >= 0xff000000 (alpha is the top byte): eight
		// three-way minima instead of sixteen ANDs and a shift
		uint32_t least = 0xffffffffu;
		if constexpr (C == 4 && MODE == 0) {
			// (the quarters that were not requested hold an earlier group's pixels: they do not vote)
			const bool right = (tid & 4u) != 0u;
			const bool have_top = ((right ? cur_mask >> 1 : cur_mask) & 1u) == 0u, have_bottom = ((right ? cur_mask >> 3 : cur_mask >> 2) & 1u) == 0u;
			const uint32_t t0 = min(min(pre[0].x, pre[0].y), pre[0].z), t1 = min(min(pre[0].w, pre[1].x), pre[1].y);
			const uint32_t top = min(min(t0, t1), min(pre[1].z, pre[1].w));
			const uint32_t b0 = min(min(pre[2].x, pre[2].y), pre[2].z), b1 = min(min(pre[2].w, pre[3].x), pre[3].y);
			const uint32_t bottom = min(min(b0, b1), min(pre[3].z, pre[3].w));
			least = min(have_top ? top : 0xffffffffu, have_bottom ? bottom : 0xffffffffu);
		} else if constexpr (C == 4) {
			const uint32_t m0 = min(min(pre[0].x, pre[0].y), pre[0].z), m1 = min(min(pre[0].w, pre[1].x), pre[1].y);
			const uint32_t m2 = min(min(pre[1].z, pre[1].w), pre[2].x), m3 = min(min(pre[2].y, pre[2].z), pre[2].w);
			const uint32_t m4 = min(min(pre[3].x, pre[3].y), pre[3].z);
			least = min(min(min(m0, m1), m2), min(min(m3, m4), pre[3].w));
		}
		const bool transparent = C == 4 && __builtin_amdgcn_ballot_w64(least < 0xff000000u) != 0ull;
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const uint32_t row = (tid >> 3) + 8u * (uint32_t)k, col = tid & 7u;
			const uint4 v = pre[k];
			uint32_t *d = s_pl + row * kRS32 + col * 2u;
			if constexpr (C == 4) {
#pragma unroll
				for (uint32_t c = 0; c < 3; ++c) {
					const uint32_t sel = c | 0x0c000c00u | ((4u + c) << 16);
					uint2 pr;
					pr.x = __builtin_amdgcn_perm(v.y, v.x, sel);
					pr.y = __builtin_amdgcn_perm(v.w, v.z, sel);
					lds_store2(d + c * kPD32, pr);
				}
			} else {
				// bytes R0 G0 B0 R1 | G1 B1 R2 G2 | B2 R3 G3 B3 (as shrink32_kernel<.., 3>)
				lds_store2(d, make_uint2(__builtin_amdgcn_perm(v.x, v.x, 0x0c070c00u), __builtin_amdgcn_perm(v.z, v.y, 0x0c050c02u)));
				lds_store2(d + kPD32, make_uint2(__builtin_amdgcn_perm(v.y, v.x, 0x0c040c01u), __builtin_amdgcn_perm(v.z, v.y, 0x0c060c03u)));
				lds_store2(d + 2 * kPD32, make_uint2(__builtin_amdgcn_perm(v.y, v.x, 0x0c050c02u), __builtin_amdgcn_perm(v.z, v.z, 0x0c070c00u)));
			}
		}
		prefetch_next();
		if (transparent) {
			// (one transparent tile sends the whole group: the generic kernel has the alpha plane)
#pragma unroll
			for (uint32_t k = 0; k < 4; ++k) defer_tile(tile_id(k));
			advance();
			continue;
		}
		tile_sync<1>();
		// ---- detector: as shrink32_kernel, minus the windows that would straddle two tiles
		uint32_t sum_hz = 0, sum_vr = 0;
		if constexpr (MODE == 1) {
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
				rA[c] = pk_mad_u16(__builtin_amdgcn_alignbit(a1, a0, 16), two, add2x16(a0, a1));
				rB[c] = pk_mad_u16(__builtin_amdgcn_alignbit(b1, b0, 16), two, add2x16(b0, b1));
				tP[c] = add2x16(a0, b0);
				dP[c] = b0;
			}
#pragma unroll
			for (int st = 0; st < 4; ++st) {
				if (st < 3 || (g & 1u) == 0u) {  // groups 1 and 3 hold window rows 8 .. 13 of their tiles: 3 steps
#pragma unroll
					for (int c = 0; c < 3; ++c) {
						const uint32_t *pr = pc[c] + (2 + 2 * st) * (int)kRS32;
						const uint32_t n0 = pr[0], n1 = pr[1], o0 = pr[kRS32], o1 = pr[kRS32 + 1];
						const uint32_t rN = pk_mad_u16(__builtin_amdgcn_alignbit(n1, n0, 16), two, add2x16(n0, n1));
						sum_hz = sad16(rN, rA[c], sum_hz);
						const uint32_t tN = add2x16(dP[c], n0);
						const uint32_t c0 = add2x16(tP[c], tN);
						sum_vr = sad16(dpp_mov<0x101>(c0), c0, sum_vr);
						const uint32_t rO = pk_mad_u16(__builtin_amdgcn_alignbit(o1, o0, 16), two, add2x16(o0, o1));
						sum_hz = sad16(rO, rB[c], sum_hz);
						const uint32_t tO = add2x16(n0, o0);
						const uint32_t e0 = add2x16(tN, tO);
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
		}
		// ---- per tile: level decision, metadata, then clone / nearest straight into its slot; the two-pass tiles of the group
		// go through the matrix cores together (resample_group16_mfma) when they have their operand tables (built only where a
		// constant-255 alpha stays 255), else each through its own dot2 form.  Everything the group form needs is kept per LANE
		// (its three views of the 2x2 tiles), so that no per-tile scalar outlives its tile: the kernel sits at its scalar-register
		// limit, and a spilled scalar costs vector instructions.
		uint32_t filt = a.filter;
		asm volatile("" : "+s"(filt));  // scalar compares per use instead of a hoisted mask
		const bool want_px = FULL || a.out_px != nullptr;
		const bool group16 = a.group16 != 0u && filt != 0u && want_px;
		const bool second = (tid & 8u) != 0u, low = (tid & 32u) != 0u;  // lane & 15 >= 8; lane >> 4 >= 2
		uint32_t v_tx[2] = {0u, 0u}, v_ty[2] = {0u, 0u}, v_tyo = 0u, v_nw = 0u, v_nh = 0u;
		bool v_ok = false;
		uint8_t *v_dst = nullptr;
		uint32_t any_x = 0u, any_y = 0u;  // (scalar: a table of one of the two-pass tiles, 0 = there is none)
#pragma unroll
		for (uint32_t k = 0; k < 4; ++k) {
			uint32_t key0, key1, m0, m1;
			if constexpr (MODE == 1) {
				const uint32_t l0 = 32u * (k >> 1) + 8u * (k & 1u);  // first lane of tile k's first row; its second row is 16 on
				key0 = (uint32_t)__builtin_amdgcn_readlane((int)sum_hz, l0) + (uint32_t)__builtin_amdgcn_readlane((int)sum_hz, l0 + 16u);
				key1 = (uint32_t)__builtin_amdgcn_readlane((int)sum_vr, l0) + (uint32_t)__builtin_amdgcn_readlane((int)sum_vr, l0 + 16u);
				m0 = level_of(key0);
				m1 = level_of(key1);
			} else {
				const uint32_t vb = __builtin_amdgcn_readfirstlane(given[k]);
				key0 = key1 = vb;
				m0 = m1 = level_of(__float_as_uint(parse_value(__uint_as_float(vb))));
			}
			const uint32_t nw = reduced_size(16u, m0), nh = reduced_size(16u, m1);
			const uint32_t t = tile_id(k);
			const bool one_pass = want_px && (nw == 16u) != (nh == 16u) && filt != 0;
			{
				uint32_t lane = tid;
				asm volatile("" : "+v"(lane));  // a fresh compare, not a hoisted (and spilled) lane mask
				if (lane == 0) {
					reinterpret_cast<uint2 *>(a.sums)[t] = make_uint2(key0, key1);
					if (FULL || a.out_w) a.out_w[t] = nw;
					if (FULL || a.out_h) a.out_h[t] = nh;
				}
			}
			if (!want_px) continue;
			uint32_t *dst = reinterpret_cast<uint32_t *>(a.out_px + (size_t)t * (256u * (uint32_t)C));
			const uint32_t *tile_pl = s_pl + (16u * (k >> 1)) * kRS32 + 8u * (k & 1u);  // first pixel pair of the tile
			if (nw == 16u && nh == 16u) {
				if (MODE == 0 && ((cur_mask >> k) & 1u)) continue;  // (the detector's copy is in the slot)
				// clone (block.rs:279-281): 64 groups of 4 pixels, one per lane
				const uint32_t row = tid >> 2, c4 = tid & 3u;
				const uint32_t *p = tile_pl + row * kRS32 + c4 * 2u;
				const uint2 r = *reinterpret_cast<const uint2 *>(p), gch = *reinterpret_cast<const uint2 *>(p + kPD32);
				const uint2 b = *reinterpret_cast<const uint2 *>(p + 2 * kPD32);
				const uint32_t opq = 0x00ff00ffu;
				const uint32_t rg01 = __builtin_amdgcn_perm(gch.x, r.x, 0x06020400u), rg23 = __builtin_amdgcn_perm(gch.y, r.y, 0x06020400u);
				if constexpr (C == 4) {
					const uint32_t ba01 = __builtin_amdgcn_perm(opq, b.x, 0x06020400u), ba23 = __builtin_amdgcn_perm(opq, b.y, 0x06020400u);
					uint4 o;
					o.x = __builtin_amdgcn_perm(ba01, rg01, 0x05040100u);
					o.y = __builtin_amdgcn_perm(ba01, rg01, 0x07060302u);
					o.z = __builtin_amdgcn_perm(ba23, rg23, 0x05040100u);
					o.w = __builtin_amdgcn_perm(ba23, rg23, 0x07060302u);
					reinterpret_cast<uint4 *>(dst)[tid] = o;
				} else {
					uint3 o;
					o.x = __builtin_amdgcn_perm(b.x, rg01, 0x02040100u);                  // R0 G0 B0 R1
					const uint32_t gb1 = __builtin_amdgcn_perm(b.x, rg01, 0x0c0c0603u);   // G1 B1 . .
					o.y = __builtin_amdgcn_perm(rg23, gb1, 0x05040100u);                  // G1 B1 R2 G2
					o.z = __builtin_amdgcn_perm(b.y, rg23, 0x06030204u);                  // B2 R3 G3 B3
					reinterpret_cast<uint3 *>(dst)[tid] = o;
				}
			} else if (filt == 0) {
				// ResizeAlg::Nearest: source index = floor((o + 0.5) * 2^m); any (nw, nh)
				const uint32_t lgx = 31u - (uint32_t)__builtin_clz(nw);
				const uint16_t *pl16 = reinterpret_cast<const uint16_t *>(tile_pl);
				for (uint32_t i = tid; i < nw * nh; i += 64u) {
					const uint32_t ox = i & (nw - 1u), oy = i >> lgx;
					const uint32_t x = m0 == 0 ? ox : (m0 < 5u ? (2u * ox + 1u) << (m0 - 1u) : 8u);
					const uint32_t y = m1 == 0 ? oy : (m1 < 5u ? (2u * oy + 1u) << (m1 - 1u) : 8u);
					const uint32_t idx = y * (2u * kRS32) + x;
					if constexpr (C == 4) {
						dst[i] = (uint32_t)pl16[idx] | ((uint32_t)pl16[idx + 2u * kPD32] << 8) | ((uint32_t)pl16[idx + 4u * kPD32] << 16) | 0xff000000u;
					} else {
						uint8_t *o3 = reinterpret_cast<uint8_t *>(dst) + 3u * i;
						o3[0] = (uint8_t)pl16[idx];
						o3[1] = (uint8_t)pl16[idx + 2u * kPD32];
						o3[2] = (uint8_t)pl16[idx + 4u * kPD32];
					}
				}
			} else if (one_pass) {
				// 16 x n, n x 16 (round 4: here, not in the worklist kernel -- 0.9 % of the tiles cost it 36 us of a 0.41-ms step)
				const uint32_t lx = m0 < (uint32_t)kMaxLevel ? m0 : (uint32_t)kMaxLevel - 1;
				const uint32_t ly = m1 < (uint32_t)kMaxLevel ? m1 : (uint32_t)kMaxLevel - 1;
				if (nw == 16u) fast32_v_only<16, C>(s_tab, a.tabs[ly], tile_pl, tid, nh, dst);
				else fast32_h_only<16, C>(s_tab, a.tabs[lx], tile_pl, tid, nw, dst);
			} else {
				const uint32_t lx = m0 < (uint32_t)kMaxLevel ? m0 : (uint32_t)kMaxLevel - 1;
				const uint32_t ly = m1 < (uint32_t)kMaxLevel ? m1 : (uint32_t)kMaxLevel - 1;
				const uint32_t txk = a.tabs[lx].mf_off, tyk = a.tabs[ly].mf_off;
				if (group16 && txk != 0u && tyk != 0u) {
					// tile k = (dx, dy) in this lane's three views
					const bool dx = (k & 1u) != 0u, dy = (k >> 1) != 0u;
					v_tx[k >> 1] = second == dx ? txk : v_tx[k >> 1];
					v_ty[k & 1u] = second == dy ? tyk : v_ty[k & 1u];
					if (second == dx && low == dy) {
						v_tyo = tyk;
						v_nw = nw;
						v_nh = nh;
						v_dst = reinterpret_cast<uint8_t *>(dst);
						v_ok = true;
					}
					any_x = txk;
					any_y = tyk;
				} else {
					resample_fast16_hv<C>(s_tab, a.tabs[lx], a.tabs[ly], tile_pl, s_tmp, tid, nw, nh, dst);
				}
			}
		}
		if (any_x != 0u) {
			// (views without a two-pass tile ride along with any valid table: their columns and rows are not stored)
#pragma unroll
			for (int i = 0; i < 2; ++i) {
				v_tx[i] = v_tx[i] != 0u ? v_tx[i] : any_x;
				v_ty[i] = v_ty[i] != 0u ? v_ty[i] : any_y;
			}
			v_tyo = v_tyo != 0u ? v_tyo : any_y;
			resample_group16_mfma<C>(s_tab, v_tx, v_ty, v_tyo, s_pl, tid, v_nw, v_nh, v_ok, v_dst);
		}
		tile_sync<1>();  // the next group reuses this wave's LDS image
		advance();
	}
	list_flush(s_batch, n_listb, a.work + kWorkList, a.work + a.work_slot, tid);
	if (a.finish_here) {
		// as shrink32_kernel: detector sums -> stored value for the tiles of the groups this block dealt to its waves, one tile per
		// thread (round 4; the worklist kernel scanned all 1 036 800 tiles of 8 x 8K frames for it)
		__threadfence_block();  // the sums were written by other waves of this block
		__syncthreads();
		for (uint32_t i = threadIdx.x;; i += blockDim.x) {
			const uint32_t gi = i >> 2, k = i & 3u;
			const unsigned long long run = (unsigned long long)(gi >> a.chunk_lg) * gridDim.x + blockIdx.x;
			const unsigned long long g = (run << a.chunk_lg) + (gi & ((1u << a.chunk_lg) - 1u));
			if (g >= (unsigned long long)a.n_groups) break;  // (g grows with i)
			const uint32_t frame = fastdiv((uint32_t)g, a.div_gpf), r = (uint32_t)g - frame * gpf;
			const uint32_t gy = fastdiv(r, a.div_gcols), gx = r - gy * gcols;
			const uint32_t tx = 2u * gx + (k & 1u), ty = 2u * gy + (k >> 1);
			if (tx >= a.cols || ty >= a.rows) continue;
			const uint32_t t = frame * a.tiles_per_frame + ty * a.cols + tx;
			const uint2 key = reinterpret_cast<const uint2 *>(a.sums)[t];
			if (key.x == kDeferredKey && key.y == kDeferredKey) continue;
			finish_tile(key, 16u, 16u, (uint32_t)MODE, a.factor, a.value, a.lod0, a.lod1, t);
		}
	}
}

// 32x32 / 16x16 flow, first part: shrink32_kernel (+ shrink32a_kernel) or shrink16_kernel; ga = the arguments of
// the worklist kernel that follows (pxz_shrink_generic.hip: launch_shrink)
hipError_t launch_fast32_16(const ShrinkArgs &a, ShrinkArgs &ga, uint32_t channels, uint32_t n_cus, hipStream_t stream)
{
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
	f.ok_edges = a.ok_edges;
	f.filter = a.filter;
	f.sums = a.sums;
	f.out_w = a.out_w;
	f.out_h = a.out_h;
	f.out_px = a.out_px;
	f.work = a.work;
	f.work_slot = a.work_slot;
	f.finish_here = 1u;
	f.narrow = knobs().no_narrow ? 0u : 1u;
	f.group16 = knobs().no_group16 ? 0u : 1u;
	f.clone_ahead = a.mode == 0 ? a.clone_ahead : 0u;
	f.factor = a.factor;
	f.value = a.value;
	f.lod0 = a.lod0;
	f.lod1 = a.lod1;
	ga.finish_scan = 0u;
	// full tiles with transparency always go to list A; shrink32a_kernel takes it when transparency was announced
	// or seen before, else the worklist kernel walks it after list B
	f.alpha_list = (!groups16 && channels == 4 && a.out_px != nullptr && (a.filter == 0 || a.tab_dw != 0)) ? 1u : 0u;
	const bool run_alpha = f.alpha_list != 0 && a.alpha_kernel != 0;
	// most tiles of the last launch had transparency: shrink32a_kernel takes every tile, shrink32_kernel is not launched
	const bool alpha_first = run_alpha && a.alpha_first != 0;
	ga.list_a_too = f.alpha_list != 0 && !run_alpha ? 1u : 0u;
	if (alpha_first) ga.finish_scan = 1u;  // (nobody finishes tiles in passing: the worklist kernel scans)
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
	if (const uint32_t v = (uint32_t)knobs().wpb; v >= 1 && v < wpb) wpb = v;
	const uint32_t lds_bytes = f.tab_dw * 4u + wpb * f.tile_dw * 4u + kTail;
	const uint32_t per_cu = kLds / lds_bytes > 0 ? kLds / lds_bytes : 1u;
	const uint32_t resident = n_cus * (per_cu > 2u ? 2u : per_cu);
	const uint32_t units = groups16 ? f.n_groups : a.n_tiles;
	const uint32_t need = (units + wpb - 1u) / wpb;
	const uint32_t blocks = need < resident ? need : resident;
	f.chunk_lg = 3;
	if (knobs().chunk_lg >= 0) f.chunk_lg = (uint32_t)knobs().chunk_lg;
	hipError_t e = hipSuccess;  // the worklist counter of this launch was zeroed by the previous one (or at allocation)
	if (alpha_first) {
		// (nothing here: the four-plane kernel below is the first kernel of the step)
	} else if (groups16) {
		const bool full = a.out_px != nullptr && a.out_w != nullptr && a.out_h != nullptr;
		void (*k)(const Fast32Args) = a.mode == 1 ? (full ? shrink16_kernel<1, true> : shrink16_kernel<1, false>)
		                                          : (full ? shrink16_kernel<0, true> : shrink16_kernel<0, false>);
		if (channels == 3)
			k = a.mode == 1 ? (full ? shrink16_kernel<1, true, 3> : shrink16_kernel<1, false, 3>)
			                : (full ? shrink16_kernel<0, true, 3> : shrink16_kernel<0, false, 3>);
		if (lds_bytes > 64u * 1024u && (e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes)) != hipSuccess) return e;
		hipLaunchKernelGGL(k, dim3(blocks), dim3(64u * wpb), lds_bytes, stream, f);
	} else {
		const bool full = a.out_px != nullptr && a.out_w != nullptr && a.out_h != nullptr;
		void (*k)(const Fast32Args) = a.mode == 1 ? (full ? shrink32_kernel<1, true> : shrink32_kernel<1, false>)
		                                          : (full ? shrink32_kernel<0, true> : shrink32_kernel<0, false>);
		if (channels == 3)  // RGB frames: 12-byte pixel quads in, RGB slots out
			k = a.mode == 1 ? (full ? shrink32_kernel<1, true, 3> : shrink32_kernel<1, false, 3>)
			                : (full ? shrink32_kernel<0, true, 3> : shrink32_kernel<0, false, 3>);
		if (lds_bytes > 64u * 1024u && (e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes)) != hipSuccess) return e;
		hipLaunchKernelGGL(k, dim3(blocks), dim3(64u * wpb), lds_bytes, stream, f);
	}
	if ((e = hipGetLastError()) != hipSuccess) return e;
	if (!alpha_first && a.mid_event && (e = hipEventRecord(static_cast<hipEvent_t>(a.mid_event), stream)) != hipSuccess) return e;
	ga.mid_event = nullptr;
	if (run_alpha) {
		// 1b) the full tiles with transparency that shrink32_kernel listed: four planes, no output region
		Fast32Args fa = f;
		fa.all_tiles = alpha_first ? 1u : 0u;
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
		if (alpha_first && a.mid_event && (e = hipEventRecord(static_cast<hipEvent_t>(a.mid_event), stream)) != hipSuccess) return e;
	}
	// 2) the generic kernel walks the worklist (usually empty or a few percent of the tiles)
	return hipSuccess;
}

}  // namespace pxz
