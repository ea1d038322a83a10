// pxz_expand.hip -- decode side: Pixlzr::expand + to_image on the device (expand_kernel) and its launcher.
//
// Compiled with -ffp-contract=off: the f32 results of the Oklab detector are
// written into the bitstream, and the reference (Rust) never fuses a*b+c.
#include "pxz_device.h"

namespace pxz {


// ---------------------------------------------------------------------------
// Decode side: Pixlzr::expand (reference pixlzr.rs:77-122) + to_image
// (pixlzr_image.rs:24-74) in one pass: every stored tile is resized back to its full size with
// PixlzrBlock::resize (block.rs:273-334: clone, ResizeAlg::Nearest, or the two-pass convolution with
// u8 intermediate and alpha pre-/un-multiplication for RGBA) and written to its place in the frame.
// One wave per tile, persistent, tiles dealt by an LDS ticket counter.  LDS per wave: the source tile as
// one dword per pixel (premultiplied) and the horizontal pass's result [y][ox].  First version: scalar
// multiply-adds straight from the global tables; correctness and coalesced frame writes first.
// ---------------------------------------------------------------------------
template <int C>
__global__ void __launch_bounds__(256) expand_kernel(const ExpandArgs a)
{
	extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
	const uint32_t wpb = blockDim.x / 64u, sub = threadIdx.x / 64u, lane = threadIdx.x % 64u;
	uint32_t *s_ticket = lds + wpb * a.tile_dw;
	if (threadIdx.x == 0) *s_ticket = wpb;
	__syncthreads();
	uint32_t *s_src = lds + sub * a.tile_dw;
	uint32_t *s_tmp = s_src + a.bw * a.bh;
	// A tile's stored size and its first 64 pixels (all of them for most tiles) are requested one tile ahead: the
	// size -> pixels -> windows chain of dependent memory round trips was most of a tile's time.
	auto tile_of = [&](uint32_t tk) -> uint32_t {
		const unsigned long long tl = (unsigned long long)blockIdx.x + (unsigned long long)tk * gridDim.x;
		return tl < (unsigned long long)a.n_tiles ? (uint32_t)tl : 0xffffffffu;
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
	while (t != 0xffffffffu) {
		uint32_t nt = 0;
		if (lane == 0) nt = atomicAdd(s_ticket, 1u);
		const uint32_t t_next = tile_of(__builtin_amdgcn_readfirstlane(nt));
		const uint32_t tw = __builtin_amdgcn_readfirstlane(p_tw), th = __builtin_amdgcn_readfirstlane(p_th);
		const uint32_t first_px = p_px;
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
					px = (uint32_t)src[3 * i] | ((uint32_t)src[3 * i + 1] << 8) | ((uint32_t)src[3 * i + 2] << 16) | 0xff000000u;
				}
				s_src[i] = px;
			}
			tile_sync<1>();
			const uint32_t cls_x = fw == a.bw ? 0u : 1u, cls_y = fh == a.bh ? 0u : 1u;
			const ExpandTab tab_x = a.tabs[(0u * 2u + cls_x) * a.dir_stride + tw];
			const ExpandTab tab_y = a.tabs[(1u * 2u + cls_y) * a.dir_stride + th];
			// Vector form (RGBA tiles in RGBA frames whose full width is a multiple of 4): the windows of the tile
			// are staged into LDS once, a lane then makes 4 rows (horizontal pass) or 4 adjacent columns (vertical
			// pass, nearest, clone) per item, so weights are fetched once per 16 multiply-adds and the frame is
			// written 16 bytes per lane.  Same arithmetic as the scalar form below.
			if (C == 4 && (fw & 3u) == 0 && tab_x.window <= 8 && tab_y.window <= 8) {
				const uint32_t q4 = fw >> 2;
				auto put4 = [&](uint32_t q, uint32_t oy, uint4 px) {
					// (the frame is written once and not read by this launch: streaming stores, 13 % off the kernel)
					typedef uint32_t u32q __attribute__((ext_vector_type(4)));
					const u32q w = {px.x, px.y, px.z, px.w};
					__builtin_nontemporal_store(w, reinterpret_cast<u32q *>(dst + (size_t)oy * a.pitch + q * 16u));
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
				tile_sync<1>();
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
									uint32_t px = clip8(acc[r][0], prec) | (clip8(acc[r][1], prec) << 8) | (clip8(acc[r][2], prec) << 16) |
									              (clip8(acc[r][3], prec) << 24);
									if (need_v) {
										s_tmp[y * fw + ox] = px;
									} else {
										put(ox, y, unpremultiply(px));
									}
								}
							}
						}
						tile_sync<1>();
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
							if (__builtin_amdgcn_ballot_w64(alpha_and != 255u) != 0ull) {
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
								o4[r] = clip8(acc[r][0], prec) | (clip8(acc[r][1], prec) << 8) | (clip8(acc[r][2], prec) << 16) |
								        (clip8(acc[r][3], prec) << 24);
							// un-premultiplying is the identity at alpha 255: skipped (table look-up and three divisions-by-
							// multiplication per pixel) when no lane of the wave holds anything else
							const uint32_t alpha_and = (o4[0] & o4[1] & o4[2] & o4[3]) >> 24;
							if (__builtin_amdgcn_ballot_w64(alpha_and != 255u) != 0ull) {
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
					tile_sync<1>();
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
		t = t_next;
		tile_sync<1>();  // the next tile reuses this wave's LDS
	}
}

hipError_t launch_expand(const ExpandArgs &a, uint32_t n_cus, hipStream_t stream)
{
	constexpr uint32_t kLds = 160u * 1024u;
	const uint32_t tile_bytes = a.tile_dw * 4u;
	uint32_t wpb = (kLds - 16u) / tile_bytes;
	if (wpb > 4u) wpb = 4u;
	if (wpb < 1u) return hipErrorInvalidValue;
	const uint32_t lds_bytes = wpb * tile_bytes + 16u;
	uint32_t per_cu = kLds / lds_bytes;
	if (per_cu > 4u) per_cu = 4u;
	if (per_cu < 1u) per_cu = 1u;
	const uint32_t need = (a.n_tiles + wpb - 1u) / wpb, resident = n_cus * per_cu;
	const uint32_t blocks = need < resident ? need : resident;
	hipError_t e;
	if (a.channels == 4) {
		auto k = expand_kernel<4>;
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
