// pxz_shrink_generic.hip -- the generic tile kernel (any tile size, RGB/RGBA, ragged edges, transparency; it also walks
// the worklists the fast kernels leave and finishes their tiles), finish_kernel, and launch_shrink, the dispatcher
// of the whole shrink step.
//
// Compiled with -ffp-contract=off: the f32 results of the Oklab detector are
// written into the bitstream, and the reference (Rust) never fuses a*b+c.
#include "pxz_device.h"

namespace pxz {

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
	} else if (a.oklab_given && oklab_value_given(a, tx, ty)) {
		// full tile of a batch the block-cooperative detector (oklab_kernel) has already been over
		const float value = __uint_as_float(a.sums[2u * tile_g]);
		key0 = key1 = __float_as_uint(value);
		m0 = m1 = level_count(__float_as_uint(parse_value(value)), a.breaks[cls], a.breaks_asc[cls]);
	} else if (a.lab_dw == 0) {
		// (never reached: the host sizes the LDS image without detector planes only when oklab_kernel covers every tile,
		// and refuses the launch otherwise -- but no tile may write planes that are not there)
		key0 = key1 = 0u;
		m0 = m1 = (uint32_t)kMaxLevel;
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


// ---------------------------------------------------------------------------
// generic kernel: persistent over tiles (or over the worklist left by shrink32_kernel).  NW == 1:
// every wave of the block owns one LDS tile image and walks tiles wave_id, wave_id + total_waves, ...;
// NW > 1: one tile per block iteration.
// ---------------------------------------------------------------------------
// BIG (NW = 16 only, round 4): a tile whose image does not fit the 160 KB of LDS (sides above ~141 px: any -b the reference's CLI
// accepts, src/bin/main.rs:19-24) keeps it in HBM instead -- one image per block of the grid (a.big_scratch), the same code on
// a pointer that is not LDS; the block barriers of tile_sync<NW> order the accesses (workgroup-scope fences: one CU, one L1).
// Not a fast path: every access of the image is an L2 round trip.
template <int NW, int C, int MODE, bool BIG = false>
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
			if (a.finish_scan) {
				for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < a.n_tiles; t += gridDim.x * blockDim.x) {
					const uint2 key = reinterpret_cast<const uint2 *>(a.sums)[t];
					if (key.x == kDeferredKey && key.y == kDeferredKey) continue;
					const uint32_t tf = t % a.tiles_per_frame;
					const uint32_t ty = tf / a.cols, tx = tf - ty * a.cols;
					finish_tile(key, (tx == a.cols - 1) ? a.edge_w : a.bw, (ty == a.rows - 1) ? a.edge_h : a.bh, (uint32_t)MODE, a.factor,
					            a.value, a.lod0, a.lod1, t);
				}
			} else if (!a.list_a_too) {
				// shrink32_kernel finished its own tiles; the ones shrink32a_kernel completed (list A: full tiles) are left
				for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count_a; i += gridDim.x * blockDim.x) {
					const uint32_t t = a.work[kWorkList + a.n_tiles + i];
					const uint2 key = reinterpret_cast<const uint2 *>(a.sums)[t];
					if (key.x == kDeferredKey && key.y == kDeferredKey) continue;  // (went on to list B: finished above)
					finish_tile(key, a.bw, a.bh, (uint32_t)MODE, a.factor, a.value, a.lod0, a.lod1, t);
				}
			}
			if (blockIdx.x == 0) {  // the other set of counters is the next launch's
				if (threadIdx.x == 0) {
					a.work[a.work_slot ^ 1u] = 0u;
					a.work[kWorkA + (a.work_slot ^ 1u)] = 0u;
					if (a.stats) {  // steers the next launches' kernel choice and the size of this kernel's grid (pxz_api.cpp)
						a.stats[0] = count_a;
						a.stats[1] = count_b + count_a;
						a.stats[2] = a.stats_sig;
					}
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
			if constexpr (BIG) process_tile<NW, C, MODE>(a, tile_g, a.big_scratch + (size_t)blockIdx.x * a.tile_dw, lds, threadIdx.x);
			else process_tile<NW, C, MODE>(a, tile_g, lds, lds + a.tile_dw, threadIdx.x);
			__syncthreads();
		}
		if (a.work) {
			// as in the single-wave form: finish the tiles the fast kernel completed (a scan, or -- when that kernel finished its
			// own -- only the full tiles its four-plane instance took from list A), zero the next launch's counter
			if (a.finish_scan) {
				for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < a.n_tiles; t += gridDim.x * blockDim.x) {
					const uint2 key = reinterpret_cast<const uint2 *>(a.sums)[t];
					if (key.x == kDeferredKey && key.y == kDeferredKey) continue;
					const uint32_t tf = t % a.tiles_per_frame;
					const uint32_t ty = tf / a.cols, tx = tf - ty * a.cols;
					finish_tile(key, (tx == a.cols - 1) ? a.edge_w : a.bw, (ty == a.rows - 1) ? a.edge_h : a.bh, (uint32_t)MODE, a.factor,
					            a.value, a.lod0, a.lod1, t);
				}
			} else if (!a.list_a_too) {
				for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count_a; i += gridDim.x * blockDim.x) {
					const uint32_t t = a.work[kWorkList + a.n_tiles + i];
					const uint2 key = reinterpret_cast<const uint2 *>(a.sums)[t];
					if (key.x == kDeferredKey && key.y == kDeferredKey) continue;  // (went on to list B: finished above)
					finish_tile(key, a.bw, a.bh, (uint32_t)MODE, a.factor, a.value, a.lod0, a.lod1, t);
				}
			}
			if (blockIdx.x == 0) {  // the other set of counters is the next launch's
				if (threadIdx.x == 0) {
					a.work[a.work_slot ^ 1u] = 0u;
					a.work[kWorkA + (a.work_slot ^ 1u)] = 0u;
					if (a.stats) {
						a.stats[0] = count_a;
						a.stats[1] = count_b + count_a;
						a.stats[2] = a.stats_sig;
					}
				}
				if (threadIdx.x < kTicketCounters) a.work[2u + kTicketCounters * (a.work_slot ^ 1u) + threadIdx.x] = 0u;
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
	const uint32_t t = tile_g % f.tiles_per_frame;
	const uint32_t ty = t / f.cols, tx = t - ty * f.cols;
	finish_tile(key, (tx == f.cols - 1) ? f.edge_w : f.bw, (ty == f.rows - 1) ? f.edge_h : f.bh, f.mode, f.factor, f.value, f.lod0,
	            f.lod1, tile_g);
}
// ---------------------------------------------------------------------------
// launchers (called from pxz_api.cpp)
// ---------------------------------------------------------------------------
template <int NW, int C, int MODE>
static hipError_t launch_one(const ShrinkArgs &a, const LaunchGeom &g, hipStream_t stream)
{
	if constexpr (NW == 16) {
		if (a.big_blocks != 0u) {
			hipLaunchKernelGGL((shrink_kernel<NW, C, MODE, true>), dim3(g.blocks), dim3(g.threads), g.lds_bytes, stream, a);
			return hipGetLastError();
		}
	}
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
		if (const uint32_t v = (uint32_t)knobs().wpb; v >= 1 && v < wpb) wpb = v;  // tuning knob: waves per block
		if (wpb < 1u) wpb = 1u;
		g.threads = 64u * wpb;
		g.lds_bytes = wpb * tile_bytes + 16u;
		const uint32_t per_cu = kLds / g.lds_bytes > 0 ? kLds / g.lds_bytes : 1u;
		uint32_t resident = n_cus * (per_cu > 2u ? 2u : per_cu);
		uint32_t need = (a.n_tiles + wpb - 1u) / wpb;
		// Behind shrink32_kernel (which finishes its own tiles) this kernel only walks the lists, and on most batches they are
		// empty: 512 blocks of 12 waves and 110 KB of LDS then take 6 us to start and leave.  The grid follows what the last
		// finished launch of the handle listed (twice that, at least 8 blocks); any grid walks any list, so only this
		// kernel's duration depends on it.
		if (a.work && !a.finish_scan && a.expect_listed != 0xffffffffu) {
			const uint32_t expect = a.expect_listed < a.n_tiles ? a.expect_listed : a.n_tiles;
			const uint32_t few = (2u * expect + wpb - 1u) / wpb + 8u;
			if (few < need) need = few;
		}
		g.blocks = need < resident ? need : resident;
	} else if (a.big_blocks != 0u) {
		g.threads = 64u * nw;   // (nw = 16: a big tile has more than 8192 pixels)
		g.lds_bytes = 16u * nw;  // the partial sums only: the image is in HBM
		g.blocks = a.big_blocks;
	} else {
		g.threads = 64u * nw;
		g.lds_bytes = tile_bytes + 16u * nw;
		const uint32_t per_cu = kLds / g.lds_bytes > 0 ? kLds / g.lds_bytes : 1u;
		const uint32_t resident = n_cus * per_cu * 2u;
		g.blocks = a.n_tiles < resident ? a.n_tiles : resident;
		// (behind a fast kernel that finished its own tiles: a block per listed tile of the last finished launch, twice that and 8 more)
		if (a.work && !a.finish_scan && a.expect_listed != 0xffffffffu) {
			const uint32_t expect = a.expect_listed < a.n_tiles ? a.expect_listed : a.n_tiles;
			const uint32_t few = 2u * expect + 8u;
			if (few < g.blocks) g.blocks = few;
		}
	}
	return g;
}

bool fast64_applicable(const ShrinkArgs &a, uint32_t channels)
{
	return (channels == 4 || channels == 3) && a.bw == 64 && a.bh == 64 && (a.mode == 1 || a.oklab_given) && a.work != nullptr &&
	       (a.out_px == nullptr || (a.filter != 0 && a.mf64 != nullptr));
}

bool fast16_applicable(const ShrinkArgs &a, uint32_t channels)
{
	return (channels == 4 || channels == 3) && a.bw == 16 && a.bh == 16 && a.work != nullptr &&
	       (a.out_px == nullptr || a.filter == 0 || a.tab_dw != 0) && !(a.mode == 0 && !a.oklab_given);
}

bool fast32_applicable(const ShrinkArgs &a, uint32_t channels)
{
	return (channels == 4 || channels == 3) && a.bw == 32 && a.bh == 32 && a.work != nullptr &&
	       (a.out_px == nullptr || a.filter == 0 || a.tab_dw != 0) && !(a.mode == 0 && !a.oklab_given);
}

hipError_t launch_fast64(const ShrinkArgs &a, ShrinkArgs &ga, uint32_t channels, uint32_t n_cus, hipStream_t stream);     // pxz_shrink64.hip
hipError_t launch_fast32_16(const ShrinkArgs &a, ShrinkArgs &ga, uint32_t channels, uint32_t n_cus, hipStream_t stream);  // pxz_shrink32.hip

hipError_t launch_shrink(const ShrinkArgs &a, uint32_t channels, uint32_t n_cus, hipStream_t stream)
{
	ShrinkArgs ga = a;
	hipError_t e = hipSuccess;
	if (fast64_applicable(a, channels)) e = launch_fast64(a, ga, channels, n_cus, stream);
	else if (fast32_applicable(a, channels) || fast16_applicable(a, channels)) e = launch_fast32_16(a, ga, channels, n_cus, stream);
	else ga.work = nullptr;
	if (e != hipSuccess) return e;
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


hipError_t launch_finish(const FinishArgs &f, hipStream_t stream)
{
	hipLaunchKernelGGL(finish_kernel, dim3((f.n_tiles + 255) / 256), dim3(256), 0, stream, f);
	return hipGetLastError();
}


}  // namespace pxz
