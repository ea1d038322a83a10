// pxz_oklab.hip -- the block-cooperative Oklab-MAD detector (oklab_kernel<16|32|64>), the per-pixel form of its
// conversion (oklab_pixels_kernel) and their launchers.
//
// Compiled with -ffp-contract=off: the f32 results of the Oklab detector are
// written into the bitstream, and the reference (Rust) never fuses a*b+c.
#include "pxz_device.h"
#include "pxz_oklab_math.h"
#include <cstdlib>

namespace pxz {


// ---------------------------------------------------------------------------
// Oklab-MAD detector, block-cooperative (full 32x32 RGBA tiles): get_block_variance
// (reference src/operations.rs:26-126) with shrink_by's closures (pixlzr.rs:160-162).
//
// The two f32 accumulations of the reference are sequential over the tile's pixels and
// end up in the bitstream, so they are replayed in exactly that order.  To keep the chip
// busy anyway, a block of 16 waves works on 15 tiles at once: waves 0..14 ("producers")
// each convert one tile to Oklab (f32 + the glibc-cbrtf double-precision steps) and keep
// the 16 pixels x 3 values of every lane in registers; wave 15 ("chain") walks 15 x 4 chains
// (channels a, b, l, alpha of every tile) in lock-step, 60 lanes wide, reading the values
// from LDS bands of 8 tile rows.
//
// Software pipeline over the batches b_0, b_1, ... of a block, one "period" per batch, four
// "intervals" (bands) per period, two barriers per interval:
//   convert phase   producers convert band k of batch p into spare registers and request the
//                   same band of batch p+1 from HBM; meanwhile the chain adds up the two bands
//                   written one interval earlier: pass 1 (sum) of batch p and pass 2 (sum of
//                   |x - mean|) of batch p-1, whose means it published when its pass 1 ended
//   -- barrier A --
//   write phase     producers write band k of pass 2 (old registers minus the means), move the
//                   spare registers in, write band k of pass 1
//   -- barrier B --
// so the conversion (the expensive part) and the dependent-add chains run side by side, and
// one band buffer per pass is enough.
// ---------------------------------------------------------------------------
constexpr uint32_t kOkTiles = 15;           // tiles per block and batch (one per producer wave)
constexpr uint32_t kOkPlane = 256 + 4;      // floats per (tile, channel) band: 8 rows x 32 px + bank skew
constexpr uint32_t kOkBand = kOkTiles * 4 * kOkPlane;  // floats per band buffer

// Per-pixel form of the same conversion (pxz_oklab_pixels_device): out[i] = {l, a, b, alpha} of RGBA pixel i.
__global__ void __launch_bounds__(256) oklab_pixels_kernel(const uint32_t *px, uint32_t n, float4 *out)
{
	__shared__ float4 s_lms[768];
	__shared__ float s_alpha[256];
	__shared__ double s_scale[128];
	oklab_fill_tables(s_lms, s_alpha, s_scale, threadIdx.x);
	__syncthreads();
	for (uint32_t i = 2u * (blockIdx.x * blockDim.x + threadIdx.x); i < n; i += 2u * gridDim.x * blockDim.x) {
		const uint32_t v0 = px[i], v1 = i + 1u < n ? px[i + 1u] : 0u;
		float o0[3], o1[3];
		oklab_pair(v0, v1, s_lms, s_scale, o0, o1);
		out[i] = make_float4(o0[2], o0[0], o0[1], s_alpha[v0 >> 24]);
		if (i + 1u < n) out[i + 1u] = make_float4(o1[2], o1[0], o1[1], s_alpha[v1 >> 24]);
	}
}

// Tile geometry of the block-cooperative Oklab detector.  A band is 256 pixels of a tile in row-major order = 4
// consecutive pixels per lane.  T = 16 | 32 | 64: square tiles, a band is 64 / (T/4) whole rows of T pixels.
// T = 0: any tile whose width is a multiple of 4 (so that a lane's 4 pixels sit in one row, 16-byte aligned), sizes
// at run time; NBR = 1..4: that many bands, kept in registers between the passes; NBR = 0: any number, parked.
template <int T>
struct OkGeom {
	static constexpr uint32_t kBands = T > 0 ? T * T / 256 : 1;   // 1 | 4 | 16
	static constexpr uint32_t kLanesPerRow = T > 0 ? T / 4 : 1;   // 4 | 8 | 16
	static constexpr uint32_t kRowsPerBand = T > 0 ? 256 / T : 1; // 16 | 8 | 4
	// up to 4 bands the converted tile stays in registers between the passes; a 64x64 tile (192 values per
	// lane) parks it in a scratch buffer in HBM instead (13 dwords per lane and band: 12 values + the alpha bytes)
	static constexpr bool kInRegs = kBands <= 4;
};

// Tiles the block-cooperative detector takes: full width, and a height of whole bands (every full tile; the
// ragged last row of the grid when its height happens to be one).  tile_h = 0 when it does not.
// item = the launch's running index: the tile number (region 0: every tile is asked, the full ones are taken), or the
// index inside an edge region (T = 0 only: all of its tiles are taken).  tile_g = the tile's number in the batch.
template <int T, int C = 4, class Args>
__device__ __forceinline__ uint32_t oklab_tile_src(const Args &a, uint32_t item, const uint8_t *&src, uint32_t &tile_g)
{
	tile_g = item;
	if (item >= a.ok_count) return 0u;
	if (T == 0 && a.ok_region != 0u) {
		// right column: full_rows tiles per frame; bottom row: full_cols; corner: one
		const uint32_t per_frame = a.ok_region == 1u ? a.full_rows : (a.ok_region == 2u ? a.full_cols : 1u);
		const uint32_t frame = item / per_frame, k = item - frame * per_frame;
		const uint32_t tx = a.ok_region == 2u ? k : a.cols - 1u, ty = a.ok_region == 1u ? k : a.rows - 1u;
		tile_g = frame * a.tiles_per_frame + ty * a.cols + tx;
		src = a.src + (size_t)frame * a.frame_stride + (size_t)(ty * a.bh) * a.pitch + (size_t)(tx * a.bw) * 4u;
		return a.ok_region == 1u ? a.bh : a.edge_h;
	}
	const uint32_t frame = fastdiv(item, a.div_tpf);
	const uint32_t t = item - frame * a.tiles_per_frame;
	const uint32_t ty = fastdiv(t, a.div_cols), tx = t - ty * a.cols;
	const uint32_t tw = T > 0 ? (uint32_t)T : a.bw, th = T > 0 ? (uint32_t)T : a.bh;
	src = a.src + (size_t)frame * a.frame_stride + (size_t)(ty * th) * a.pitch + (size_t)(tx * tw) * (uint32_t)C;
	if (tx >= a.full_cols || ty >= a.ok_rows) return 0u;
	return ty == a.rows - 1u ? a.edge_h : th;  // (T = 0: only full tiles are eligible, so this is th)
}

// C = 3 (T = 64 only): RGB frames, a lane's 4 pixels are 12 bytes (rows 4-byte aligned), alpha 255.
template <int T, int NBR = 0, int C = 4>
__global__ void __launch_bounds__(1024) oklab_kernel(const ShrinkArgs a)
{
	static_assert(C == 4 || T > 0, "the run-time geometry reads RGBA");
	using G = OkGeom<T>;
	constexpr bool kGeneral = T == 0;
	constexpr bool kInRegs = kGeneral ? NBR > 0 : G::kInRegs;
	constexpr uint32_t NBC = kGeneral ? (NBR > 0 ? (uint32_t)NBR : 1u) : G::kBands;  // band count where it is static
	const uint32_t NB = (kGeneral && NBR == 0) ? a.ok_bands : NBC;
	// (T = 0: the tiles of this launch are a.bw x a.bh, or an edge region's smaller ones)
	const uint32_t tile_w = kGeneral ? ((a.ok_region & 1u) ? a.edge_w : a.bw) : (uint32_t)T;
	// (a width that is not a multiple of 4 -- the ragged right column of any image -- is walked with rows padded to whole
	// quads: the padding pixels are exact zeros in both passes, which leave the f32 sums of the real ones as they are)
	const uint32_t tile_wp = kGeneral ? (tile_w + 3u) & ~3u : (uint32_t)T;
	const uint32_t tile_px = kGeneral ? tile_wp * ((a.ok_region & 2u) ? a.edge_h : a.bh) : (uint32_t)(T * T);
	extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
	float4 *s_lms = reinterpret_cast<float4 *>(lds);         // 3 x 256: sRGB u8 -> linear, times the first matrix's columns
	float *s_alpha = reinterpret_cast<float *>(s_lms + 768); // 256: a / 255
	double *s_scale = reinterpret_cast<double *>(s_alpha + 256);  // 128: 2^(xe/3) * 2^((xe%3)/3) by exponent field
	float *s_mean = reinterpret_cast<float *>(s_scale + 128);     // 64: per (tile, channel) means of the batch in pass 2
	float *s_p1 = s_mean + 64;                               // pass-1 band: values
	float *s_p2 = s_p1 + kOkBand;                            // pass-2 band: values minus means
	const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
	oklab_fill_tables(s_lms, s_alpha, s_scale, threadIdx.x);
	__syncthreads();

	const uint32_t n_batches = (a.ok_count + kOkTiles - 1) / kOkTiles;
	// batches of this block: blockIdx.x + j * gridDim.x, j < own; periods 0 .. own + 1 drain the pipeline
	const uint32_t own = n_batches > blockIdx.x ? (n_batches - blockIdx.x + gridDim.x - 1) / gridDim.x : 0u;
	const uint32_t periods = own + 2u;

	if (wave < kOkTiles) {
		// ---------------- producers ----------------
		const uint32_t row_off = lane / G::kLanesPerRow, col_off = (lane % G::kLanesPerRow) * (4u * (uint32_t)C);
		const size_t band_step = (size_t)G::kRowsPerBand * a.pitch;
		// this lane's 4 pixels of a band: how many exist (T = 0: short last band, padded rows), and where in the tile
		// real pixels among the lane's four of a band (4, or fewer in the last quad of a padded row; 0: none)
		auto lane_px = [&](uint32_t band) -> uint32_t {
			if constexpr (!kGeneral) return 4u;
			const uint32_t first = 256u * band + 4u * lane;
			if (first >= tile_px) return 0u;
			const uint32_t col = first - small_div(first, tile_wp) * tile_wp;  // (first < 2^20)
			return tile_w - col < 4u ? tile_w - col : 4u;
		};
		auto lane_off = [&](uint32_t band) -> size_t {
			if constexpr (kGeneral) {
				const uint32_t first = 256u * band + 4u * lane, row = small_div(first, tile_wp);
				return (size_t)row * a.pitch + (size_t)(first - row * tile_wp) * 4u;
			} else {
				return (size_t)row_off * a.pitch + col_off + (size_t)band * band_step;
			}
		};
		auto load_band = [&](const uint8_t *tile, uint32_t band) -> uint4 {
			const uint32_t have = lane_px(band);
			uint4 v = make_uint4(0, 0, 0, 0);  // black, alpha 0: converts to exact zeros
			if constexpr (C == 3) {
				const uint3 t = *reinterpret_cast<const uint3 *>(tile + lane_off(band));
				v = make_uint4(t.x | 0xff000000u, __builtin_amdgcn_alignbit(t.y, t.x, 24) | 0xff000000u,
				               __builtin_amdgcn_alignbit(t.z, t.y, 16) | 0xff000000u, (t.z >> 8) | 0xff000000u);
			} else if (have == 4u) {
				v = *reinterpret_cast<const uint4 *>(tile + lane_off(band));
			} else if (have != 0u) {  // the last quad of a padded row: nothing is read past the row's real pixels
				const uint32_t *p = reinterpret_cast<const uint32_t *>(tile + lane_off(band));
				v.x = p[0];
				if (have > 1u) v.y = p[1];
				if (have > 2u) v.z = p[2];
			}
			return v;
		};
		// source pointers (first byte) of this wave's tiles in batches p, p+1, p+2; null: nothing there
		auto batch_src = [&](uint32_t j, uint32_t &bands) -> const uint8_t * {
			const uint8_t *src;
			bands = 0;
			if (j >= own) return nullptr;
			uint32_t unused_tile;
			const uint32_t th = oklab_tile_src<T, C>(a, (blockIdx.x + j * gridDim.x) * kOkTiles + wave, src, unused_tile);
			if (th == 0) return nullptr;
			bands = kGeneral ? NB : th / G::kRowsPerBand;  // (a ragged tile is only taken with a whole number of bands)
			return src;
		};
		uint32_t nb0 = 0, nb1 = 0, nb2 = 0, nb_prev = 0;  // bands of this wave's tile in batches p, p+1, p+2, p-1
		const uint8_t *src0 = nullptr, *src1 = batch_src(0, nb1), *src2 = batch_src(1, nb2);
		// raw pixels: the band being converted and the one after it (requested one interval ahead)
		uint4 px_cur = make_uint4(0, 0, 0, 0), px_nxt = make_uint4(0, 0, 0, 0);
		if (src1) px_cur = load_band(src1, 0);
		if (NB > 1) {
			if (src1 && nb1 > 1u) px_nxt = load_band(src1, 1);
		} else if (src2) {
			px_nxt = load_band(src2, 0);
		}
		float lab[kInRegs ? NBC : 1][4][3];   // [band][pixel][a, b, l] of the batch whose pass 2 is being staged
		uint32_t alpha_px[kInRegs ? NBC : 1];  // its 4 alpha bytes per band
		bool have_prev = false, elig_cur = false;
		uint32_t tile_prev = 0, tile_cur = 0;
		for (uint32_t p = 0; p < periods; ++p) {
			tile_prev = tile_cur;
			tile_cur = (blockIdx.x + p * gridDim.x) * kOkTiles + wave;
			src0 = src1;  // batch p
			src1 = src2;  // batch p + 1
			nb_prev = nb0;
			nb0 = nb1;
			nb1 = nb2;
			src2 = batch_src(p + 2u, nb2);
			elig_cur = src0 != nullptr;  // (false past the last batch and for tiles the detector does not take)
			constexpr int kUnroll = kInRegs ? (int)NBC : 1;  // register form: lab[k] must be a static index
#pragma unroll kUnroll
			for (uint32_t k = 0; k < NB; ++k) {
				// ---- convert phase
				float4 old[4];  // scratch form only: band k of the previous batch, back from HBM for pass 2
				if constexpr (!kInRegs) {
					if (have_prev && k < nb_prev) {
						// per (tile, band): three arrays of 64 float4 (the 12 values of a lane) + 64 alpha words = 3328 bytes
						const float *sb = a.ok_scratch + ((size_t)tile_prev * NB + k) * 832u;
						const float4 *sp = reinterpret_cast<const float4 *>(sb) + lane;
#pragma unroll
						for (int q = 0; q < 3; ++q) old[q] = sp[64 * q];
						old[3].x = sb[768u + lane];
					}
				}
				float fresh[4][3];
				uint32_t fresh_alpha = 0;
				if (elig_cur && k < nb0) {
					const uint32_t v[4] = {px_cur.x, px_cur.y, px_cur.z, px_cur.w};
					fresh_alpha = (v[0] >> 24) | ((v[1] >> 24) << 8) | ((v[2] >> 24) << 16) | ((v[3] >> 24) << 24);
#pragma unroll
					for (int j = 0; j < 4; j += 2) {
#if defined(PXZ_EXP) && (PXZ_EXP == 1 || PXZ_EXP == 3)
						for (int c = 0; c < 3; ++c) { fresh[j][c] = __uint_as_float(v[j] >> (8 * c)); fresh[j + 1][c] = __uint_as_float(v[j + 1] >> (8 * c)); }
#else
						oklab_pair(v[j], v[j + 1], s_lms, s_scale, fresh[j], fresh[j + 1]);
#endif
						// two pixels (six cube-root chains) at a time: the register file also holds a whole tile of results
						__builtin_amdgcn_sched_barrier(0);
					}
				}
				// px_cur is consumed: move the window on by one step of the sequence
				px_cur = px_nxt;
				{
					// two steps ahead: band k+2 of this batch, or an early band of the next one / the one after
					const uint32_t ahead = k + 2u, bo = ahead / NB, band = ahead % NB;
					const uint8_t *base = bo == 0 ? src0 : (bo == 1 ? src1 : src2);
					const uint32_t nbb = bo == 0 ? nb0 : (bo == 1 ? nb1 : nb2);
					if (base && band < nbb) px_nxt = load_band(base, band);
				}
				__syncthreads();  // A: the chain has consumed the bands of the previous interval
				// ---- write phase
				const uint32_t slot = (wave * 4u) * kOkPlane + lane * 4u;  // 4 consecutive pixels of this lane
				if (have_prev && k < nb_prev) {
					// operations.rs:75-84: the chain only has to add |x| of these.  The means were published during
					// the first convert phase of this period.
					const float4 mean = *reinterpret_cast<const float4 *>(s_mean + wave * 4u);
					const float mean4[4] = {mean.x, mean.y, mean.z, mean.w};
					float x[4][3];
					uint32_t al4;
					if constexpr (kInRegs) {
#pragma unroll
						for (int j = 0; j < 4; ++j)
#pragma unroll
							for (int c = 0; c < 3; ++c) x[j][c] = lab[k][j][c];
						al4 = alpha_px[k];
					} else {
						const float o12[12] = {old[0].x, old[0].y, old[0].z, old[0].w, old[1].x, old[1].y,
						                       old[1].z, old[1].w, old[2].x, old[2].y, old[2].z, old[2].w};
#pragma unroll
						for (int j = 0; j < 4; ++j)
#pragma unroll
							for (int c = 0; c < 3; ++c) x[j][c] = o12[3 * j + c];
						al4 = __float_as_uint(old[3].x);
					}
					float *d = s_p2 + slot;
					// (pixels that do not exist -- beyond a short last band, or the padding of a row -- are exact zeros here too)
					const uint32_t have = lane_px(k);
#pragma unroll
					for (int c = 0; c < 3; ++c) {
						float4 v = make_float4(x[0][c] - mean4[c], x[1][c] - mean4[c], x[2][c] - mean4[c], x[3][c] - mean4[c]);
						if constexpr (kGeneral) v = make_float4(have > 0u ? v.x : 0.f, have > 1u ? v.y : 0.f, have > 2u ? v.z : 0.f, have > 3u ? v.w : 0.f);
						*reinterpret_cast<float4 *>(d + c * kOkPlane) = v;
					}
					float4 va = make_float4(s_alpha[al4 & 255u] - mean4[3], s_alpha[(al4 >> 8) & 255u] - mean4[3],
					                        s_alpha[(al4 >> 16) & 255u] - mean4[3], s_alpha[al4 >> 24] - mean4[3]);
					if constexpr (kGeneral) va = make_float4(have > 0u ? va.x : 0.f, have > 1u ? va.y : 0.f, have > 2u ? va.z : 0.f, have > 3u ? va.w : 0.f);
					*reinterpret_cast<float4 *>(d + 3 * kOkPlane) = va;
				}
				if (elig_cur && k < nb0) {
					float *d = s_p1 + slot;
#pragma unroll
					for (int c = 0; c < 3; ++c)
						*reinterpret_cast<float4 *>(d + c * kOkPlane) = make_float4(fresh[0][c], fresh[1][c], fresh[2][c], fresh[3][c]);
					*reinterpret_cast<float4 *>(d + 3 * kOkPlane) =
					    make_float4(s_alpha[fresh_alpha & 255u], s_alpha[(fresh_alpha >> 8) & 255u], s_alpha[(fresh_alpha >> 16) & 255u],
					                s_alpha[fresh_alpha >> 24]);
					if constexpr (kInRegs) {
#pragma unroll
						for (int j = 0; j < 4; ++j)
#pragma unroll
							for (int c = 0; c < 3; ++c) lab[k][j][c] = fresh[j][c];
						alpha_px[k] = fresh_alpha;
					} else {
						float *sb = a.ok_scratch + ((size_t)tile_cur * NB + k) * 832u;
						float4 *sp = reinterpret_cast<float4 *>(sb) + lane;
						sp[0] = make_float4(fresh[0][0], fresh[0][1], fresh[0][2], fresh[1][0]);
						sp[64] = make_float4(fresh[1][1], fresh[1][2], fresh[2][0], fresh[2][1]);
						sp[128] = make_float4(fresh[2][2], fresh[3][0], fresh[3][1], fresh[3][2]);
						sb[768u + lane] = __uint_as_float(fresh_alpha);
					}
				}
				__syncthreads();  // B: the bands of this interval are complete
			}
			have_prev = elig_cur;
		}
	} else {
		// ---------------- chain wave: lane = tile*4 + channel (a, b, l, alpha) ----------------
		const uint32_t ct = lane >> 2, cc = lane & 3u;
		const bool live = ct < kOkTiles;
		float acc1 = 0.0f, acc2 = 0.0f;
		__builtin_amdgcn_s_setprio(3);  // the serial part of every interval: first pick of its SIMD's issue slots
		// one dependent add chain per lane and pass; two register sets take turns so that 16 values are
		// in flight from LDS while 16 are added
		auto walk = [&](const float *band, float acc, const bool magnitude) -> float {
			const float4 *x = reinterpret_cast<const float4 *>(band + (ct * 4u + cc) * kOkPlane);
			auto add16 = [&](const float4 (&v)[4]) {
#pragma unroll
				for (int q = 0; q < 4; ++q) {
					if (magnitude) {
						acc += fabsf(v[q].x);  // operations.rs:80-83
						acc += fabsf(v[q].y);
						acc += fabsf(v[q].z);
						acc += fabsf(v[q].w);
					} else {
						acc += v[q].x;  // operations.rs:60-63, row-major pixel order
						acc += v[q].y;
						acc += v[q].z;
						acc += v[q].w;
					}
				}
			};
			float4 va[4] = {x[0], x[1], x[2], x[3]}, vb[4];
#pragma unroll 1
			for (uint32_t i = 0; i < 64; i += 8) {
#pragma unroll
				for (int q = 0; q < 4; ++q) vb[q] = x[i + 4u + (uint32_t)q];
				__builtin_amdgcn_sched_barrier(0);
				add16(va);
				__builtin_amdgcn_sched_barrier(0);
				const uint32_t n = i + 8u < 64u ? i + 8u : 0u;  // the last round re-reads the first (unused)
#pragma unroll
				for (int q = 0; q < 4; ++q) va[q] = x[n + (uint32_t)q];
				__builtin_amdgcn_sched_barrier(0);
				add16(vb);
				__builtin_amdgcn_sched_barrier(0);
			}
			return acc;
		};
		uint32_t h0 = 0, hm1 = 0, hm2 = 0;  // height of this lane's tile in batches p, p-1, p-2 (0: not taken)
		uint32_t g0 = 0, gm1 = 0, gm2 = 0;  // and its number in the batch
		for (uint32_t p = 0; p < periods; ++p) {
			{
				const uint8_t *unused;
				hm2 = hm1;
				hm1 = h0;
				gm2 = gm1;
				gm1 = g0;
				h0 = (live && p < own) ? oklab_tile_src<T, C>(a, (blockIdx.x + p * gridDim.x) * kOkTiles + ct, unused, g0) : 0u;
			}
#pragma unroll 1
			for (uint32_t k = 0; k < NB; ++k) {
				// the bands written one interval ago: band kk of period pp
				const uint32_t pp = k > 0 ? p : p - 1u, kk = k > 0 ? k - 1u : NB - 1u;
				const bool any = k > 0 || p > 0;
				const bool p1_valid = any && pp < own;                   // pass 1 of batch pp
				const bool p2_valid = any && pp >= 1u && pp - 1u < own;  // pass 2 of batch pp - 1
				const uint32_t h1 = k > 0 ? h0 : hm1, h2 = k > 0 ? hm1 : hm2;  // tile heights of those two batches
				const uint32_t tg2 = k > 0 ? gm1 : gm2;                       // the tile whose pass 2 ends here
				// (T = 0: a tile that is taken has all its bands; a short last band is padded with exact zeros)
#if !(defined(PXZ_EXP) && (PXZ_EXP == 2 || PXZ_EXP == 3))
				if (p1_valid && (kGeneral ? h1 != 0u : kk * G::kRowsPerBand < h1)) acc1 = walk(s_p1, acc1, false);
				if (p2_valid && (kGeneral ? h2 != 0u : kk * G::kRowsPerBand < h2)) acc2 = walk(s_p2, acc2, true);
#endif
				if (kk == NB - 1u) {
					if (p1_valid) {
						s_mean[lane] = __fdiv_rn(acc1, (float)(tile_w * h1));  // operations.rs:65-68; read after barrier A
						acc1 = 0.0f;
					}
					if (p2_valid) {
						const float d0 = __shfl(acc2, (int)(lane & ~3u) + 0, 64), d1 = __shfl(acc2, (int)(lane & ~3u) + 1, 64);
						const float d2 = __shfl(acc2, (int)(lane & ~3u) + 2, 64), d3 = __shfl(acc2, (int)(lane & ~3u) + 3, 64);
						const float total = d0 + d1 + d2 + d3;  // :89
						const float value = __fdiv_rn(total, (float)(tile_w * h2)) * a.factor * a.scale2;  // pixlzr.rs:162
						if (live && cc == 0 && h2 != 0u)
							reinterpret_cast<uint2 *>(a.sums)[tg2] = make_uint2(__float_as_uint(value), __float_as_uint(value));
						acc2 = 0.0f;
					}
				}
				__syncthreads();  // A
				__syncthreads();  // B
			}
		}
	}
}

// ---------------------------------------------------------------------------
// oklab2_kernel<T>, T = 16 | 32: the detector for square tiles whose converted pixels fit one wave's registers,
// re-cut so that nothing waits for anything else (round 2):
//   * 14 producer waves (one tile each per batch) and TWO chain waves: wave 14 walks pass 1 (the sums,
//     operations.rs:55-64) of the batch being converted, wave 15 pass 2 (the sums of |x - mean|, :75-84) of the batch
//     before it -- each chain has half the dependent adds of the single chain wave of oklab_kernel;
//   * bands of 128 pixels (2 per producer lane) and ONE barrier per interval.  A producer issues its LDS stores at
//     the START of an interval -- the band it converted in the interval before (pass 1) and the band of the previous
//     batch that this interval's conversion is about to overwrite (pass 2) -- so they drain under the conversion
//     instead of in front of the barrier (oklab_kernel: 17 % of its time in a store phase with the vector ALUs idle);
//     the chains read a band two intervals after its conversion / re-delivery (pass-1 buffers x2, pass-2 buffers x3);
//   * pass 2 is handed over as the raw values again and the chain wave subtracts its lane's mean itself (one
//     independent v_sub per element beside the dependent v_add: that wave has the issue slots to spare) -- so a
//     producer re-delivers band k of the previous batch just before it converts band k of the next one into the same
//     registers: no spare register set, no moves, and the mean is only needed by the chain, two intervals later.
// With g = p * NB + k the running interval number and s = q * NB + band the running number of a converted band:
//   producers at g   store band s = g - 1 -> p1[g & 1]; store raw band g - NB (batch p - 1) -> p2[g % 3]; convert band g
//   chain A at g     adds band s = g - 2 (p1[(g - 1) & 1]); after a batch's last band it publishes the means
//   chain B at g     adds raw band s = g - 2 - NB (p2[(g - 2) % 3]); it picks the means up with a batch's first band
//                    (published one barrier earlier) and writes the values after its last
// ---------------------------------------------------------------------------
constexpr uint32_t kOk2Tables = 3072u + 256u + 2u * 128u + 64u; // dwords: matrix-column products, alpha, scale (doubles), means

// Geometry of oklab2_kernel<T>.  T = 16 | 32: one producer wave per tile (G = 1), 14 tiles per batch.
// T = 64 (round 3): a tile's 4096 converted pixels are 48 KB, so FOUR producer waves share a tile (G = 4: wave j of the
// group converts the 128-px bands 4k + j, k = 0..7, i.e. rows 8k + 2j, 8k + 2j + 1) and three tiles are in flight per
// block; an interval hands the chains a "super-band" of 4 x 128 = 512 consecutive pixels per tile.  Nothing is parked
// in HBM (oklab_kernel<64> wrote and re-read 13 bytes per pixel: 6.9 GB per 8 x 8K frames).
template <int T>
struct Ok2Geom {
	static constexpr uint32_t G = T == 64 ? 4u : 1u;                  // producer waves per tile
#if defined(PXZ_EXP) && (PXZ_EXP == 21 || PXZ_EXP == 22)
	// (experiment builds of round 4: what the detector alone takes with 12 / 10 producer waves and the other waves idle -- the
	// lower bound of any fused form that gives waves of this block to the shrink step; results are unchanged)
	static constexpr uint32_t kProd = T == 64 ? 12u : (PXZ_EXP == 21 ? 12u : 10u);
#else
	static constexpr uint32_t kProd = T == 64 ? 12u : 14u;            // producer waves per block
#endif
	static constexpr uint32_t kTiles = kProd / G;                     // tiles per batch: 14 | 3
	static constexpr uint32_t NB = T == 64 ? 8u : T * T / 128u;       // bands per producer and tile = intervals per period: 2 | 8 | 8
	static constexpr uint32_t kLanesPerRow = T / 2;                   // 8 | 16 | 32
	static constexpr uint32_t kRowsPerBand = 128 / T;                 // 8 | 4 | 2
	// floats per (tile, channel) and interval + bank skew.  G = 4: the chains read 16 consecutive floats per quad of
	// lanes with ds_read_b128 (16 lanes per LDS cycle); a plane stride of 16 mod 64 dwords puts the four chains of
	// every such group on four different quarters of the banks
	static constexpr uint32_t kPlane = G * 128u + (G == 4u ? 16u : 4u);
	static constexpr uint32_t kBand = kTiles * 4u * kPlane;           // floats per band buffer
	static constexpr uint32_t kLdsBytes = (kOk2Tables + 5u * kBand) * 4u;
};
constexpr uint32_t kOk2Prod = Ok2Geom<32>::kProd;

// The chain step of oklab2_kernel<64>: sum = (..((sum + e0(lane 0)) + e1(lane 0)) .. + e3(lane 3)), sixteen dependent adds
// whose operands come from the four lanes of the quad in turn (DPP quad_perm broadcast on the first source; the running
// sum is the plain second source, so consecutive adds forward it as usual).  One asm statement: left to the compiler
// the sixteen DPP moves are not folded into the adds and, being independent of the sum, are hoisted and spilled.
// ABS: add |e| (pass 2).  The operands of that variant were just written by vector instructions of the same wave: a DPP
// read of such a register needs two wait states, which the compiler does not count for an asm statement -- s_nop 1.
#define PXZ_QADD(Q, E, M) "v_add_f32_dpp %0, " M "%" #E M ", %0 quad_perm:[" #Q "," #Q "," #Q "," #Q "] row_mask:0xf bank_mask:0xf\n\t"
#define PXZ_QADD4(Q, M) PXZ_QADD(Q, 1, M) PXZ_QADD(Q, 2, M) PXZ_QADD(Q, 3, M) PXZ_QADD(Q, 4, M)
// (the quad's first lane is the chain: its own four values come in as plain adds -- 2 cycles of the SIMD each instead of the
// 4.3 of a DPP add, profiles/r03_issue_probe.txt -- and only the other twelve through the broadcast; the other three lanes
// add THEIR own values first and hold nothing anyone reads.  The four plain adds also are the wait states a DPP read needs
// behind the vector instructions that have just written e0..e3.)
template <bool ABS>
__device__ __forceinline__ void quad_chain_add16(float &sum, float e0, float e1, float e2, float e3)
{
	if constexpr (ABS)
		asm volatile("v_add_f32_e64 %0, |%1|, %0\n\tv_add_f32_e64 %0, |%2|, %0\n\tv_add_f32_e64 %0, |%3|, %0\n\tv_add_f32_e64 %0, |%4|, %0\n\t"
		             PXZ_QADD4(1, "|") PXZ_QADD4(2, "|") PXZ_QADD4(3, "|") : "+v"(sum) : "v"(e0), "v"(e1), "v"(e2), "v"(e3));
	else
		asm volatile("v_add_f32_e32 %0, %1, %0\n\tv_add_f32_e32 %0, %2, %0\n\tv_add_f32_e32 %0, %3, %0\n\tv_add_f32_e32 %0, %4, %0\n\t"
		             PXZ_QADD4(1, "") PXZ_QADD4(2, "") PXZ_QADD4(3, "") : "+v"(sum) : "v"(e0), "v"(e1), "v"(e2), "v"(e3));
}

// C = 3 (round 2): RGB frames read directly -- a lane's two pixels are six bytes inside an aligned eight (rows are 4-byte
// aligned), cut out with two funnel shifts; there is no alpha (the plane is the constant 1).
// AHEAD (round 4, RGBA): every band is also stored where the tile's pixels go if the tile is stored at full size -- see the producers.
template <int T, int C = 4, bool AHEAD = false>
__global__ void __launch_bounds__(1024) oklab2_kernel(const ShrinkArgs a)
{

	using Geo = Ok2Geom<T>;
	constexpr uint32_t G = Geo::G, NB = Geo::NB, kProd = Geo::kProd, kTiles = Geo::kTiles;
	constexpr uint32_t kLanesPerRow = Geo::kLanesPerRow, kRowsPerBand = Geo::kRowsPerBand;
	constexpr uint32_t kPlane = Geo::kPlane, kBand = Geo::kBand;
	static_assert(NB % 2 == 0, "the pass-1 buffers alternate with the band index");
	// (a static array: its address is a compile-time constant, so table and buffer offsets fold into the instructions'
	// offset fields -- with `extern __shared__` every LDS access of the loop paid a v_add of the relocated base)
	__shared__ __attribute__((aligned(16))) uint32_t lds[Geo::kLdsBytes / 4u];
	float4 *s_lms = reinterpret_cast<float4 *>(lds);
	float *s_alpha = reinterpret_cast<float *>(s_lms + 768);
	double *s_scale = reinterpret_cast<double *>(s_alpha + 256);
	float *s_mean = reinterpret_cast<float *>(s_scale + 128);
	float *s_p1 = s_mean + 64;                  // [2][kBand]: converted values, pass 1
	float *s_p2 = s_p1 + 2 * kBand;             // [3][kBand]: the same values of the batch before, pass 2
	// (the wave number as a scalar: roles, tiles, source and slot addresses and every condition on them are then the scalar unit's
	// business -- left as a function of threadIdx the compiler takes them to differ between the lanes of a wave)
	// (T = 64 keeps the plain form: its detector came out 9 % SLOWER with the scalar one, 1.10 against 1.01 ms -- its chain waves set
	// the pace, and their code is laid out differently behind scalar role branches)
	const uint32_t wave = G == 1u ? (uint32_t)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : threadIdx.x >> 6, lane = threadIdx.x & 63u;
	oklab_fill_tables(s_lms, s_alpha, s_scale, threadIdx.x);
	__syncthreads();

	// Roles.  G = 1: waves 0..13 produce, 14 walks pass 1, 15 pass 2.  G = 4: a block's waves go to the four SIMDs in
	// turn (waves w and w + 4 share one), and a chain wave here issues four times the adds of the G = 1 one, so it gets a
	// SIMD with two producers beside it instead of three or four: wave 0 = pass 1, wave 1 = pass 2, waves 12 and 13
	// have no work (they only keep the barriers), the other twelve produce: tile slot 0 = waves 2, 6, 10, 14; slot 1 = 3, 7, 11,
	// 15; slot 2 = 4, 8, 5, 9.  (The placement is a matter of speed only.)
	uint32_t pi;           // producer index: tile slot pi / G, part pi % G; >= kProd: not a producer
	bool chain_a, chain_b;
	if constexpr (G == 1u) {
		pi = wave;
		chain_a = wave == kProd;
		chain_b = wave == kProd + 1u;
	} else {
		chain_a = wave == 0u;
		chain_b = wave == 1u;
		const uint32_t cls = wave & 3u, q = wave >> 2;
		pi = cls == 2u ? q : (cls == 3u ? 4u + q : (q == 1u || q == 2u ? 8u + (cls << 1) + (q - 1u) : kProd));
	}

	const uint32_t n_batches = (a.ok_count + kTiles - 1) / kTiles;
	const uint32_t own = n_batches > blockIdx.x ? (n_batches - blockIdx.x + gridDim.x - 1) / gridDim.x : 0u;
	// batch q is converted in period q, re-delivered in period q + 1, and chain B adds its last band in interval 1 of
	// period q + 2: the last period ends after two intervals
	const uint32_t periods = own + 2u;
	constexpr uint32_t kLastIntervals = 2u;

	if (pi < kProd) {
		// ---------------- producers ----------------
		const uint32_t pslot = pi / G, part = pi % G;
		// RGBA: the lane's two pixels are 8 aligned bytes.  RGB: bytes 6l .. 6l+5 of the row (l = lane in row) lie inside the
		// aligned 8 bytes at 6l & ~3, shifted by 0 or 2 bytes
		const uint32_t rgb_off = (lane % kLanesPerRow) * 6u, rgb_shift = (rgb_off & 3u) * 8u;
		const size_t lane_off = (size_t)(lane / kLanesPerRow + part * kRowsPerBand) * a.pitch + (C == 4 ? (lane % kLanesPerRow) * 8u : (rgb_off & ~3u));
		const size_t band_step = (size_t)(G * kRowsPerBand) * a.pitch;  // from a producer's band k to its band k + 1
		// clone_ahead (round 4, RGBA): every band is also stored, as it is, where the tile's pixels go if the tile is stored at full
		// size (block.rs:279-281) -- the slot's rows are the tile's rows back to back, a band is 512 consecutive bytes of it, a lane's
		// two pixels 8 of those.  Whether the tile IS stored whole is known two periods later; the shrink kernel behind this one then
		// leaves such tiles alone instead of reading them a second time.  This kernel is bound by its arithmetic: the stores ride
		// under it.
		// (RGB slots: a band is 128 x 3 = 384 bytes, a lane's two pixels six of them)
		const uint32_t dst_lane_off = C == 4 ? part * 512u + lane * 8u : part * 384u + lane * 6u;
		constexpr uint32_t kBandBytes = C == 4 ? 512u : 384u;
		// (the slots as SCALAR addresses -- a tile is a matter of the wave -- so that a copy is one store with a scalar base, the lane's
		// offset and an immediate: as per-lane pointers every store cost a 64-bit add and two selects)
		unsigned long long dst0 = 0, dst1 = 0;  // slot of the tile in conversion / the next one
		// Which tiles are copied (T = 16 | 32): the ones whose first band looks busy -- two neighbouring pixels of it 24 or more apart
		// (|dr| + |dg| + |db| + |da|).  Whether a tile is stored whole is not known before its second pass; on the bench frames this
		// guess copies 45 % of the tiles instead of all (2-7 % of the tiles stored whole are missed, a tenth of the tiles copied in
		// vain), which is what the copies cost in HBM traffic.  The guess never decides a result: sums[2 t + 1] says whether tile t
		// was copied, and the shrink kernel leaves a tile alone only if it is stored whole AND was copied -- the others it reads, as
		// before round 4.
		constexpr uint32_t kCopyContrast = 24u;
		bool copy0 = true, copy1 = true;  // the tile in conversion / the next one (T = 64: every tile, two thirds are stored whole)
		unsigned long long flag1 = 0;     // where the next tile's flag goes
		auto batch_src = [&](uint32_t j, uint32_t &bands) -> const uint8_t * {
			const uint8_t *src;
			bands = 0;
			if constexpr (AHEAD) {
				const unsigned long long fa = reinterpret_cast<unsigned long long>(a.sums) + 8ull * ((blockIdx.x + j * gridDim.x) * kTiles + pslot) + 4ull;
				flag1 = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(fa >> 32)) << 32) |
				        (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)fa);
				// (in front of the conditions below, which the compiler takes to differ between lanes: set behind them the address
				// would live in vector registers; the item number IS the tile's number in this launch, whether the tile is taken or not)
				const unsigned long long d = reinterpret_cast<unsigned long long>(a.out_px) +
				                             (unsigned long long)((blockIdx.x + j * gridDim.x) * kTiles + pslot) * a.slot_bytes;
				// (the builtin returns a signed int: the low half must not be sign-extended into the high one)
				dst1 = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(d >> 32)) << 32) |
				       (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)d);
			}
			if (j >= own) return nullptr;
			uint32_t tile_g;
			const uint32_t th = oklab_tile_src<T, C>(a, (blockIdx.x + j * gridDim.x) * kTiles + pslot, src, tile_g);
			if (th == 0) return nullptr;
			bands = th / (G * kRowsPerBand);  // (a ragged tile is only taken when its height is a multiple of 8 rows)
			return src;  // (the tile's first byte, a scalar; a lane adds lane_off where it loads)
		};
		uint32_t nb0 = 0, nb1 = 0, nb_prev = 0;
		const uint8_t *src0 = nullptr, *src1 = batch_src(0, nb1);
		// the pixel pairs of the next two bands in sequence, requested two intervals before their conversion; the band
		// index picks the register pair (NB is even), so nothing is moved
		uint2 q[2] = {make_uint2(0, 0), make_uint2(0, 0)};
		auto request = [&](uint32_t ahead_of_k0, const uint8_t *s_a, uint32_t n_a, const uint8_t *s_b, uint32_t n_b, uint2 &dst) {
			const uint32_t bo = ahead_of_k0 / NB, band = ahead_of_k0 % NB;
			const uint8_t *base = bo == 0 ? s_a : s_b;
			const uint32_t nbb = bo == 0 ? n_a : n_b;
			if constexpr (AHEAD) {
				// (no branch around a memory instruction in this loop: behind one the compiler waits for EVERYTHING in flight, and the
				// copies below must stay in flight across two intervals.  A band nobody converts reads the frame's first rows.)
				const uint8_t *from = base && band < nbb ? base + (size_t)band * band_step : a.src;
				dst = *reinterpret_cast<const uint2 *>(from + lane_off);
			} else {
				if (base && band < nbb) dst = *reinterpret_cast<const uint2 *>(base + (size_t)band * band_step + lane_off);
			}
		};
		request(0, src1, nb1, nullptr, 0, q[0]);
		request(1, src1, nb1, nullptr, 0, q[1]);
		float lab[NB][2][3];  // [band][pixel][a, b, l]: this wave's share of the tile in conversion / re-delivery
		// Alpha.  An opaque band's alpha plane is the constant 1 (255 / 255): no bytes are kept and nothing is looked up.
		// Only tiles with transparency keep their alpha bytes (two per band): al_cur for the tile in conversion, al_old for
		// the one in re-delivery; the flags are wave-uniform.
		uint32_t al_cur[NB / 2], al_old[NB / 2];
#pragma unroll
		for (uint32_t k = 0; k < NB / 2; ++k) al_cur[k] = al_old[k] = 0xffffffffu;
		bool cur_opaque = true, old_opaque = true, band_opaque = true;  // tile in conversion / in re-delivery / band awaiting its store
		const uint32_t slot = (pslot * 4u) * kPlane + part * 128u + lane * 2u;
		// one band (two pixels per lane, three planes + alpha) into a band buffer
		float2 ones;  // (kept in a register pair: as a constant it is re-made with two moves at each of its uses)
		asm volatile("v_mov_b32 %0, 1.0\n\tv_mov_b32 %1, 1.0" : "=v"(ones.x), "=v"(ones.y));
		auto store_band = [&](float *d, const float (&x)[2][3], bool opaque, uint32_t ab) {
#pragma unroll
			for (int c = 0; c < 3; ++c) *reinterpret_cast<float2 *>(d + c * kPlane) = make_float2(x[0][c], x[1][c]);
			// (a store in each branch: a value merged after the branch makes the compiler wait for every LDS operation in flight)
			if (opaque) {
				*reinterpret_cast<float2 *>(d + 3 * kPlane) = ones;
			} else {
				*reinterpret_cast<float2 *>(d + 3 * kPlane) = make_float2(s_alpha[ab & 255u], s_alpha[(ab >> 8) & 255u]);
			}
		};
		bool have_prev = false, elig_cur = false, pending = false;  // pending: the band converted last interval awaits its store
		uint32_t g3 = 0;  // interval number mod 3
		// The conversion is cut in two (oklab_pair_head: table reads, first matrix, mantissas and scale factors;
		// oklab_pair_tail: the cube roots' arithmetic and the second matrix) and the head of the NEXT band runs at the end of
		// an interval, in front of the barrier: the waves of a block move in lock-step, so a table read at the start of an
		// interval would be waited for by all of them at once, behind 112 stores, with nothing to issue meanwhile.
		Cbrt6State st;          // head of the band about to be converted
		bool st_opaque = true;  // and its alpha: all 255 (wave-uniform)?
		uint32_t st_ab = 0;     //   the two alpha bytes of this lane
		bool st_valid = false;
		// (the copy of a band is issued BEHIND the request of the band after next: loads and stores share one in-order counter, a store
		// takes longer than an interval to be acknowledged, and in front of that load it made every wait for the load a wait for the
		// store -- detector 0.87 -> 1.04-1.10 ms; behind it the wait leaves one operation in flight: s_waitcnt vmcnt(1))
		auto copy_band = [&](const uint2 &px, unsigned long long band_of_slot, bool copied) {
			if constexpr (AHEAD) {
				// (a band that is not converted -- past the last batch, a tile the detector leaves to the generic kernel -- and the bands of
				// a tile that is not copied go to spare bytes: no branch around the store)
				// (G = 4: the flag is a lane mask there -- one lane's copy of it)
				const bool v = (G == 1u ? st_valid : __builtin_amdgcn_readfirstlane((uint32_t)st_valid) != 0u) && copied;
				const unsigned long long base = v ? band_of_slot : reinterpret_cast<unsigned long long>(a.ahead_spare);
				// (an address the compiler knows to be global: made from an integer it is generic, and the store a flat_store.  A plain
				// store: marked non-temporal it cost the kernel 0.03 ms more -- experiment builds, detector 0.924 against 0.897 ms)
				if constexpr (C == 4) {
					typedef __attribute__((address_space(1))) unsigned long long *global_qword;
					*(global_qword)(base + dst_lane_off) = *reinterpret_cast<const unsigned long long *>(&px);
				} else {
					// the six bytes R0 G0 B0 R1 G1 B1 out of the aligned eight they were loaded in, at an even address: four + two
					typedef __attribute__((address_space(1))) uint32_t global_u32_a2 __attribute__((aligned(2)));
					typedef __attribute__((address_space(1))) uint16_t global_u16;
					*(global_u32_a2 *)(base + dst_lane_off) = __builtin_amdgcn_alignbit(px.y, px.x, rgb_shift);
					*(global_u16 *)(base + dst_lane_off + 4u) = (uint16_t)(px.y >> rgb_shift);
				}
			}
		};
		// band 0 of the next tile is in px: decide whether the tile is copied, say so in sums[2 t + 1] (every lane the same dword: no
		// branch), copy the band
		auto first_band = [&](const uint2 &px) {
			if constexpr (AHEAD) {
				if constexpr (G == 1u) {
					// |dr| + |dg| + |db| (+ |da|) between the lane's two neighbouring pixels: one v_sad_u8
					uint32_t d;
					if constexpr (C == 4) {
						d = __builtin_amdgcn_sad_u8(px.x, px.y, 0u);
					} else {
						const uint32_t w0 = __builtin_amdgcn_alignbit(px.y, px.x, rgb_shift);       // R0 G0 B0 R1
						const uint32_t p1 = __builtin_amdgcn_alignbit(px.y >> rgb_shift, w0, 24);  // R1 G1 B1 .
						d = __builtin_amdgcn_sad_u8(w0 & 0x00ffffffu, p1 & 0x00ffffffu, 0u);
					}
					copy1 = __builtin_amdgcn_ballot_w64(d >= kCopyContrast) != 0ull;
				}
				const bool v = G == 1u ? st_valid : __builtin_amdgcn_readfirstlane((uint32_t)st_valid) != 0u;
				const unsigned long long fa = v ? flag1 : reinterpret_cast<unsigned long long>(a.ahead_spare);
				typedef __attribute__((address_space(1))) uint32_t *global_dword;
				*(global_dword)fa = copy1 ? 1u : 0u;
				copy_band(px, dst1, copy1);
			}
		};
		auto head = [&](bool valid, const uint2 &px) {
			st_valid = valid;
			if (valid) {
				if constexpr (C == 4) {
					oklab_pair_head(px.x, px.y, s_lms, st);
					st_opaque = __all((px.x & px.y) >= 0xff000000u);
					st_ab = (px.x >> 24) | ((px.y >> 24) << 8);
				} else {
					const uint32_t w0 = __builtin_amdgcn_alignbit(px.y, px.x, rgb_shift);  // R0 G0 B0 R1
					const uint32_t w1 = px.y >> rgb_shift;                                  // G1 B1 . .
					oklab_pair_head(w0, __builtin_amdgcn_alignbit(w1, w0, 24), s_lms, st);  // (byte 3 of either is not looked at)
					st_opaque = true;
				}
			}
		};
		head(src1 != nullptr && nb1 > 0u, q[0]);
		first_band(q[0]);
		for (uint32_t p = 0; p < periods; ++p) {
			src0 = src1;
			dst0 = dst1;
			copy0 = copy1;
			nb_prev = nb0;
			nb0 = nb1;
			src1 = batch_src(p + 1u, nb1);
			elig_cur = src0 != nullptr;
#pragma unroll
			for (uint32_t k = 0; k < NB; ++k) {
				if (p + 1u == periods && k == kLastIntervals) break;  // (block-uniform)
				const uint32_t kp = (k + NB - 1u) % NB;  // the band converted in the interval before (k = 0: of the batch before)
				const bool convert = st_valid;  // (= elig_cur && k < nb0)
				if (pending) store_band(s_p1 + (k & 1u) * kBand + slot, lab[kp], band_opaque, al_cur[kp >> 1] >> (16u * (kp & 1u)));
				if (k == 0u) {
					// the tile converted last period is the one in re-delivery now
					old_opaque = cur_opaque;
					if (!cur_opaque) {
#pragma unroll
						for (uint32_t i = 0; i < NB / 2; ++i) { al_old[i] = al_cur[i]; al_cur[i] = 0xffffffffu; }
					}
					cur_opaque = true;
				}
				if (have_prev && k < nb_prev) store_band(s_p2 + g3 * kBand + slot, lab[k], old_opaque, al_old[k >> 1] >> (16u * (k & 1u)));
				__builtin_amdgcn_sched_barrier(0);
				pending = convert;
				if (convert) {
#if defined(PXZ_EXP) && (PXZ_EXP == 1 || PXZ_EXP == 3)
					for (int c = 0; c < 3; ++c) { lab[k][0][c] = st.x[c]; lab[k][1][c] = st.x[3 + c]; }
#else
					oklab_pair_tail(st, s_scale, lab[k][0], lab[k][1]);
#endif
					band_opaque = st_opaque;
					if (!st_opaque) {
						al_cur[k >> 1] = (k & 1u) == 0u ? (al_cur[k >> 1] & 0xffff0000u) | st_ab : (al_cur[k >> 1] & 0x0000ffffu) | (st_ab << 16);
						cur_opaque = false;
					}
				}
				__builtin_amdgcn_sched_barrier(0);
				// the head of the next band in sequence: band k + 1 of this batch, or band 0 of the next one
				if (k + 1u < NB) head(elig_cur && k + 1u < nb0, q[(k + 1u) & 1u]);
				else head(src1 != nullptr && nb1 > 0u, q[0]);
				// (after the head: the wait for its pixels is a wait for every load in flight)
				request(k + 2u, src0, nb0, src1, nb1, q[k & 1u]);
				if (k + 1u < NB) copy_band(q[(k + 1u) & 1u], dst0 + (k + 1u) * (G * kBandBytes), copy0);
				else first_band(q[0]);
				g3 = g3 == 2u ? 0u : g3 + 1u;
				__syncthreads();
			}
			have_prev = elig_cur;
		}
	} else if (chain_a || chain_b) {
		// ---------------- chain waves ----------------
		// G = 1: lane = tile * 4 + channel (a, b, l, alpha), 56 lanes live.  G = 4: a QUAD of lanes per chain (12 chains
		// = 3 tiles x 4 channels): a ds_read_b128 hands the quad 16 consecutive values, and the adds take them from
		// the quad's lanes in turn through DPP -- one LDS instruction per 16 dependent adds, and every lane of a quad
		// carries the same running sum
		const bool second = chain_b;  // pass 2
		const uint32_t chain = G == 1u ? lane : lane >> 2, sub = G == 1u ? 0u : lane & 3u;
		const uint32_t ct = chain >> 2, cc = chain & 3u;
		const bool live = ct < kTiles;
		const float *plane = (second ? s_p2 : s_p1) + (live ? chain : chain - kTiles * 4u) * kPlane + sub * 4u;
		float acc = 0.0f, mean = 0.0f;
		__builtin_amdgcn_s_setprio(3);  // the serial part of every interval: first pick of its SIMD's issue slots
		// G = 1: one dependent add per element; 32 values in flight from LDS while 32 are added
		auto walk = [&](const float *band, float sum, const bool deviation) __attribute__((always_inline)) -> float {
			const float4 *x = reinterpret_cast<const float4 *>(band);
			if constexpr (G == 1u) {
				auto add32 = [&](const float4 (&v)[8]) __attribute__((always_inline)) {
#pragma unroll
					for (int q = 0; q < 8; ++q) {
						if (deviation) {
							// operations.rs:80-83 with `before` = |x - avg| (pixlzr.rs:160-161); the differences two per packed
							// instruction (independent of the running sum), |.| as the add's source modifier
							const f32x2 m2 = {mean, mean};
							const f32x2 d0 = f32x2{v[q].x, v[q].y} - m2, d1 = f32x2{v[q].z, v[q].w} - m2;
							sum += fabsf(d0.x);
							sum += fabsf(d0.y);
							sum += fabsf(d1.x);
							sum += fabsf(d1.y);
						} else {
							sum += v[q].x;  // operations.rs:60-63, row-major pixel order
							sum += v[q].y;
							sum += v[q].z;
							sum += v[q].w;
						}
					}
				};
				float4 va[8], vb[8];
#pragma unroll
				for (int q = 0; q < 8; ++q) va[q] = x[q];
#pragma unroll
				for (int q = 0; q < 8; ++q) vb[q] = x[8 + q];
				__builtin_amdgcn_sched_barrier(0);
				add32(va);
				__builtin_amdgcn_sched_barrier(0);
#pragma unroll
				for (int q = 0; q < 8; ++q) va[q] = x[16 + q];
				__builtin_amdgcn_sched_barrier(0);
				add32(vb);
				__builtin_amdgcn_sched_barrier(0);
#pragma unroll
				for (int q = 0; q < 8; ++q) vb[q] = x[24 + q];
				__builtin_amdgcn_sched_barrier(0);
				add32(va);
				__builtin_amdgcn_sched_barrier(0);
				add32(vb);
			} else {
				// 512 values per chain: 32 reads of 16 (x[4 i] is this lane's float4 of read i), four reads per register set
				auto add64 = [&](const float4 (&v)[4]) __attribute__((always_inline)) {
#pragma unroll
					for (int r = 0; r < 4; ++r) {
						if (deviation) {
							// operations.rs:80-83: every lane takes the mean off its own four values (two packed subtractions,
							// independent of the running sum); the chain adds |x - avg| in pixel order
							const f32x2 m2 = {mean, mean};
							const f32x2 d0 = f32x2{v[r].x, v[r].y} - m2, d1 = f32x2{v[r].z, v[r].w} - m2;
							quad_chain_add16<true>(sum, d0.x, d0.y, d1.x, d1.y);
						} else {
							quad_chain_add16<false>(sum, v[r].x, v[r].y, v[r].z, v[r].w);  // operations.rs:60-63, row-major pixel order
						}
					}
				};
				float4 va[4], vb[4];
#pragma unroll
				for (int q = 0; q < 4; ++q) va[q] = x[4 * q];
#pragma unroll
				for (int q = 0; q < 4; ++q) vb[q] = x[4 * (4 + q)];
#pragma unroll
				for (int it = 0; it < 4; ++it) {
					__builtin_amdgcn_sched_barrier(0);
					add64(va);
					__builtin_amdgcn_sched_barrier(0);
					if (it < 3) {
#pragma unroll
						for (int q = 0; q < 4; ++q) va[q] = x[4 * (8 * it + 8 + q)];
					}
					__builtin_amdgcn_sched_barrier(0);
					add64(vb);
					__builtin_amdgcn_sched_barrier(0);
					if (it < 3) {
#pragma unroll
						for (int q = 0; q < 4; ++q) vb[q] = x[4 * (8 * it + 12 + q)];
					}
				}
			}
			return sum;
		};
		uint32_t h0 = 0, hm1 = 0, hm2 = 0;  // height of this lane's tile in batches p, p-1, p-2 (0: not taken)
		uint32_t g0 = 0, gm1 = 0, gm2 = 0;  // and its number in the batch of frames
		uint32_t g3 = 0;                    // interval number mod 3
		const int lane_ch0 = G == 1u ? (int)(lane & ~3u) : (int)(lane & ~15u);  // the lane that holds channel 0 of this lane's tile
		constexpr int kChStep = G == 1u ? 1 : 4;                               // and the distance to the next channel's
		for (uint32_t p = 0; p < periods; ++p) {
			{
				const uint8_t *unused;
				hm2 = hm1;
				hm1 = h0;
				gm2 = gm1;
				gm1 = g0;
				h0 = (live && p < own) ? oklab_tile_src<T, C>(a, (blockIdx.x + p * gridDim.x) * kTiles + ct, unused, g0) : 0u;
			}
#pragma unroll 1
			for (uint32_t k = 0; k < NB; ++k) {
				if (p + 1u == periods && k == kLastIntervals) break;
				// the band converted (chain A) / re-delivered (chain B) two intervals ago
				const bool same = k >= 2u;                           // it belongs to the period's own batch (A: p, B: p - 1)
				const uint32_t kk = same ? k - 2u : k + NB - 2u;
				if (!second) {
					const uint32_t h = same ? h0 : hm1;
					const float *band = plane + ((k + 1u) & 1u) * kBand;  // stored in interval g - 1
#if !(defined(PXZ_EXP) && (PXZ_EXP == 2 || PXZ_EXP == 3))
					if (kk * (G * kRowsPerBand) < h) acc = walk(band, acc, false);
#endif
					if (kk == NB - 1u) {
						if (sub == 0u) s_mean[chain] = __fdiv_rn(acc, (float)((uint32_t)T * h));  // operations.rs:65-68 (h = 0: never read)
						acc = 0.0f;
					}
				} else {
					const uint32_t h = same ? hm1 : hm2;
					const float *band = plane + (g3 == 0u ? 1u : (g3 == 1u ? 2u : 0u)) * kBand;  // stored in interval g - 2
					if (kk == 0u) mean = s_mean[chain];  // published by chain A during the interval before
#if !(defined(PXZ_EXP) && (PXZ_EXP == 2 || PXZ_EXP == 3))
					if (kk * (G * kRowsPerBand) < h) acc = walk(band, acc, true);
#endif
					if (kk == NB - 1u) {
						const float d0 = __shfl(acc, lane_ch0, 64), d1 = __shfl(acc, lane_ch0 + kChStep, 64);
						const float d2 = __shfl(acc, lane_ch0 + 2 * kChStep, 64), d3 = __shfl(acc, lane_ch0 + 3 * kChStep, 64);
						const float total = d0 + d1 + d2 + d3;  // :89
						const float value = __fdiv_rn(total, (float)((uint32_t)T * h)) * a.factor * a.scale2;  // pixlzr.rs:162
						const uint32_t tg = same ? gm1 : gm2;
						if (live && cc == 0 && sub == 0u && h != 0u) {
							// (AHEAD: sums[2 t + 1] is the producers' "this tile was copied" flag)
							if constexpr (AHEAD) a.sums[2u * tg] = __float_as_uint(value);
							else reinterpret_cast<uint2 *>(a.sums)[tg] = make_uint2(__float_as_uint(value), __float_as_uint(value));
						}
						acc = 0.0f;
					}
				}
				g3 = g3 == 2u ? 0u : g3 + 1u;
				__syncthreads();
			}
		}
	} else {
		// (G = 4: the two waves without a role keep the barriers)
		for (uint32_t p = 0; p < periods; ++p) {
#pragma unroll 1
			for (uint32_t k = 0; k < NB; ++k) {
				if (p + 1u == periods && k == kLastIntervals) break;
				__syncthreads();
			}
		}
	}
}

hipError_t launch_oklab(const ShrinkArgs &a, uint32_t n_cus, hipStream_t stream, uint32_t channels)
{
	const uint32_t lds_bytes = (3072u + 256u + 2u * 128u + 64u) * 4u + 2u * kOkBand * 4u;
	const uint32_t n_batches = (a.ok_count + kOkTiles - 1) / kOkTiles;
	const uint32_t blocks = n_batches < n_cus ? n_batches : n_cus;
	hipError_t e;
	auto go = [&](auto kernel) -> hipError_t {
		if ((e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes)) != hipSuccess) return e;
		hipLaunchKernelGGL(kernel, dim3(blocks), dim3(1024), lds_bytes, stream, a);
		return hipGetLastError();
	};
	if (a.bw == a.bh && a.ok_region == 0u && (a.bw == 16u || a.bw == 32u || a.bw == 64u) && !knobs().oklab_v1) {
		const uint32_t per_batch = a.bw == 64u ? Ok2Geom<64>::kTiles : kOk2Prod;
		const uint32_t nb2 = (a.ok_count + per_batch - 1) / per_batch, blocks2 = nb2 < n_cus ? nb2 : n_cus;
		auto go2 = [&](auto kernel) -> hipError_t {
			hipLaunchKernelGGL(kernel, dim3(blocks2), dim3(1024), 0, stream, a);
			return hipGetLastError();
		};
		if (channels == 3 && a.clone_ahead) return a.bw == 16u ? go2(oklab2_kernel<16, 3, true>) : (a.bw == 32u ? go2(oklab2_kernel<32, 3, true>) : go2(oklab2_kernel<64, 3, true>));
		if (channels == 3) return a.bw == 16u ? go2(oklab2_kernel<16, 3>) : (a.bw == 32u ? go2(oklab2_kernel<32, 3>) : go2(oklab2_kernel<64, 3>));
		if (a.clone_ahead) return a.bw == 16u ? go2(oklab2_kernel<16, 4, true>) : (a.bw == 32u ? go2(oklab2_kernel<32, 4, true>) : go2(oklab2_kernel<64, 4, true>));
		return a.bw == 16u ? go2(oklab2_kernel<16>) : (a.bw == 32u ? go2(oklab2_kernel<32>) : go2(oklab2_kernel<64>));
	}
	if (channels == 3 && a.bw == 64u && a.bh == 64u && a.ok_region == 0u) return go(oklab_kernel<64, 0, 3>);
	if (channels != 4) return hipErrorInvalidValue;  // (the other round-1 kernels read RGBA)
	if (a.bw == a.bh && a.ok_region == 0u) {
		switch (a.bw) {
		case 16: return go(oklab_kernel<16>);
		case 32: return go(oklab_kernel<32>);
		case 64: return go(oklab_kernel<64>);
		default: break;
		}
	}
	// any other tile with a width of whole pixel quads: run-time geometry
	if (a.bw % 4u != 0u || a.ok_bands == 0u || a.ok_count == 0u) return hipErrorInvalidValue;  // (tile origins are 16-byte aligned)
	switch (a.ok_bands) {
	case 1: return go(oklab_kernel<0, 1>);
	case 2: return go(oklab_kernel<0, 2>);
	case 3: return go(oklab_kernel<0, 3>);
	case 4: return go(oklab_kernel<0, 4>);
	default: return go(oklab_kernel<0, 0>);
	}
}

hipError_t launch_oklab_pixels(const uint32_t *px, uint32_t n, float *out, uint32_t n_cus, hipStream_t stream)
{
	const uint32_t need = (n / 2u + 255u) / 256u + 1u, blocks = need < 8u * n_cus ? need : 8u * n_cus;
	hipLaunchKernelGGL(oklab_pixels_kernel, dim3(blocks), dim3(256), 0, stream, px, n, reinterpret_cast<float4 *>(out));
	return hipGetLastError();
}


}  // namespace pxz
