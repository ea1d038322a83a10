// pxz_oklab.hip -- the block-cooperative Oklab-MAD detector (oklab_kernel<16|32|64>), the per-pixel form of its
// conversion (oklab_pixels_kernel) and their launchers.
//
// Compiled with -ffp-contract=off: the f32 results of the Oklab detector are
// written into the bitstream, and the reference (Rust) never fuses a*b+c.
#include "pxz_device.h"

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

// The quotient of the Halley step inside cbrt_f32_lut: v_rcp_f64 (2^-23 or better) and ONE Newton step, i.e. a
// quotient good to ~2^-46 instead of the correctly rounded one glibc's `/` produces.  That is enough here, and
// provably so: the result is rounded to f32 right after, the inputs of this path are the l, m, s of the 2^24
// possible RGB triples and nothing else, and tests/test_gpu_parity.py::test_oklab_conversion_of_every_colour runs
// every one of them through this very function against the oracle's exact division: no bit differs.  (The
// generic kernel's cbrt_f32, which sees the same inputs, keeps the exact division.)  4 instructions instead of 8.
__device__ __forceinline__ double div_f64_oklab_domain(double n, double d)
{
	double r = __builtin_amdgcn_rcp(d);
	const double e = __builtin_fma(-d, r, 1.0);
	r = __builtin_fma(r, e, r);
	return n * r;
}

// glibc 2.35 cbrtf for the inputs of the Oklab detector: same arithmetic as cbrt_f32 above with the quotient below.
// The tail `(float)(q * third[2 + xe % 3])` followed by `ldexpf(.., xe / 3)` is folded into ONE
// multiplication by 2^(xe/3) * third[..]: scaling a double by a power of two is exact and commutes
// with the rounding to float (no underflow in this range), so the bits are unchanged.  `scale` is the
// LDS table of those 132 doubles indexed by xe + 130 (xe in [-130, 1]).
template <bool ZERO_CHECK = true>
__device__ __forceinline__ float cbrt_f32_lut(float x, const double *scale)
{
	int xe;
	const float xm = frexpf(x, &xe);
	// glibc: (float)(0.4926.. + (0.6975.. - 0.1915.. * xm) * xm) with separate double operations.  The fused form
	// differs from it by at most a few 2^-53 before the rounding to float, and for none of this path's inputs
	// does that cross a rounding boundary (same exhaustive test as for the quotient below).
	const float u = (float)__builtin_fma(__builtin_fma(-0.191502161678719066, (double)xm, 0.697570460207922770), (double)xm, 0.492659620528969547);
	const float t2 = u * u * u;
	// t2 + 2 xm and 2 t2 + xm are exact in double (24-bit operands a few binades apart), so the fused forms give
	// the same values as glibc's separate multiplications and additions
	const double num = (double)u * __builtin_fma(2.0, (double)xm, (double)t2);
	const double den = __builtin_fma(2.0, (double)t2, (double)xm);
	const float y = (float)(div_f64_oklab_domain(num, den) * scale[xe + 130]);
	if constexpr (!ZERO_CHECK) return y;  // (x == 0 is the caller's business)
	return x == 0.0f ? 0.0f : y;
}

// byte BYTE of v, times 4: the byte offset of a 256-entry f32 table row, in one SDWA shift
template <int BYTE>
__device__ __forceinline__ uint32_t byte_times4(uint32_t v)
{
	uint32_t r;
	const uint32_t two = 2u;
	if constexpr (BYTE == 0)
		asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(two), "v"(v));
	else if constexpr (BYTE == 1)
		asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(two), "v"(v));
	else
		asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(two), "v"(v));
	return r;
}
__device__ __forceinline__ float table_at(const float *table, uint32_t byte_offset)
{
	return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(table) + byte_offset);
}

// Srgba<u8> -> linear -> Oklab of two pixels (operations.rs:56-59; palette 0.7.6): LUT, then Ottosson's matrices
// with left-to-right f32 sums; two pixels per packed-f32 instruction (same IEEE results per component).
// out[k] = {a, b, l} of pixel k, the order the reference sums them in.
__device__ __forceinline__ void oklab_pair(uint32_t v0, uint32_t v1, const float *s_srgb, const double *s_scale,
                                           float (&out0)[3], float (&out1)[3])
{
	const f32x2 r = {table_at(s_srgb, byte_times4<0>(v0)), table_at(s_srgb, byte_times4<0>(v1))};
	const f32x2 g = {table_at(s_srgb, byte_times4<1>(v0)), table_at(s_srgb, byte_times4<1>(v1))};
	const f32x2 b = {table_at(s_srgb, byte_times4<2>(v0)), table_at(s_srgb, byte_times4<2>(v1))};
	const f32x2 l = 0.4122214708f * r + 0.5363325363f * g + 0.0514459929f * b;
	const f32x2 m = 0.2119034982f * r + 0.6806995451f * g + 0.1073969566f * b;
	const f32x2 s3 = 0.0883024619f * r + 0.2817188376f * g + 0.6299787005f * b;
	// l, m, s are zero only for black (every coefficient is positive, the table is zero at 0 only), and then all
	// three are: one test per pixel on the colour bytes instead of one per cube root
	const f32x2 l_ = {cbrt_f32_lut<false>(l.x, s_scale), cbrt_f32_lut<false>(l.y, s_scale)};
	const f32x2 m_ = {cbrt_f32_lut<false>(m.x, s_scale), cbrt_f32_lut<false>(m.y, s_scale)};
	const f32x2 s_ = {cbrt_f32_lut<false>(s3.x, s_scale), cbrt_f32_lut<false>(s3.y, s_scale)};
	const f32x2 L = 0.2104542553f * l_ + 0.7936177850f * m_ - 0.0040720468f * s_;
	const f32x2 A = 1.9779984951f * l_ - 2.4285922050f * m_ + 0.4505937099f * s_;
	const f32x2 B = 0.0259040371f * l_ + 0.7827717662f * m_ - 0.8086757660f * s_;
	const bool black0 = (v0 & 0x00ffffffu) == 0u, black1 = (v1 & 0x00ffffffu) == 0u;  // cbrt(0) = 0 -> L = a = b = +0
	out0[2] = black0 ? 0.0f : L.x; out1[2] = black1 ? 0.0f : L.y;
	out0[0] = black0 ? 0.0f : A.x; out1[0] = black1 ? 0.0f : A.y;
	out0[1] = black0 ? 0.0f : B.x; out1[1] = black1 ? 0.0f : B.y;
}

// The conversion tables of the Oklab kernels in LDS: sRGB u8 -> linear (256), a / 255 (256), and the 132 doubles
// 2^(xe/3) * cbrt(2)^(xe%3), xe = i - 130.  Call from the first 256 threads of a block, then a block barrier.
__device__ __forceinline__ void oklab_fill_tables(float *s_srgb, float *s_alpha, double *s_scale, uint32_t t)
{
	if (t < 256) {
		s_srgb[t] = __uint_as_float(kSrgbToLinearBits[t]);
		s_alpha[t] = __fdiv_rn((float)t, 255.0f);
	}
	if (t < 132) {
		const int xe = (int)t - 130;
		const int q3 = xe / 3, r3 = xe - 3 * q3;  // C semantics: the remainder carries the sign of xe
		const double third = r3 == 0 ? 1.0
		                   : r3 == 1 ? 1.2599210498948731648
		                   : r3 == 2 ? 1.5874010519681994748
		                   : r3 == -1 ? 1.0 / 1.2599210498948731648
		                              : 1.0 / 1.5874010519681994748;
		s_scale[t] = ldexp(third, q3);  // exact
	}
}

// Per-pixel form of the same conversion (pxz_oklab_pixels_device): out[i] = {l, a, b, alpha} of RGBA pixel i.
__global__ void __launch_bounds__(256) oklab_pixels_kernel(const uint32_t *px, uint32_t n, float4 *out)
{
	__shared__ float s_srgb[256], s_alpha[256];
	__shared__ double s_scale[132];
	oklab_fill_tables(s_srgb, s_alpha, s_scale, threadIdx.x);
	__syncthreads();
	for (uint32_t i = 2u * (blockIdx.x * blockDim.x + threadIdx.x); i < n; i += 2u * gridDim.x * blockDim.x) {
		const uint32_t v0 = px[i], v1 = i + 1u < n ? px[i + 1u] : 0u;
		float o0[3], o1[3];
		oklab_pair(v0, v1, s_srgb, s_scale, o0, o1);
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
template <int T, class Args>
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
	src = a.src + (size_t)frame * a.frame_stride + (size_t)(ty * th) * a.pitch + (size_t)(tx * tw) * 4u;
	if (tx >= a.full_cols || ty >= a.ok_rows) return 0u;
	return ty == a.rows - 1u ? a.edge_h : th;  // (T = 0: only full tiles are eligible, so this is th)
}

template <int T, int NBR = 0>
__global__ void __launch_bounds__(1024) oklab_kernel(const ShrinkArgs a)
{
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
	float *s_srgb = reinterpret_cast<float *>(lds);          // 256: sRGB u8 -> linear
	float *s_alpha = s_srgb + 256;                           // 256: a / 255
	double *s_scale = reinterpret_cast<double *>(s_alpha + 256);  // 132: 2^(xe/3) * 2^((xe%3)/3), xe = i - 130
	float *s_mean = reinterpret_cast<float *>(s_scale + 132);     // 64: per (tile, channel) means of the batch in pass 2
	float *s_p1 = s_mean + 64;                               // pass-1 band: values
	float *s_p2 = s_p1 + kOkBand;                            // pass-2 band: values minus means
	const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
	oklab_fill_tables(s_srgb, s_alpha, s_scale, threadIdx.x);
	__syncthreads();

	const uint32_t n_batches = (a.ok_count + kOkTiles - 1) / kOkTiles;
	// batches of this block: blockIdx.x + j * gridDim.x, j < own; periods 0 .. own + 1 drain the pipeline
	const uint32_t own = n_batches > blockIdx.x ? (n_batches - blockIdx.x + gridDim.x - 1) / gridDim.x : 0u;
	const uint32_t periods = own + 2u;

	if (wave < kOkTiles) {
		// ---------------- producers ----------------
		const uint32_t row_off = lane / G::kLanesPerRow, col_off = (lane % G::kLanesPerRow) * 16u;
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
			if (have == 4u) {
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
			const uint32_t th = oklab_tile_src<T>(a, (blockIdx.x + j * gridDim.x) * kOkTiles + wave, src, unused_tile);
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
						oklab_pair(v[j], v[j + 1], s_srgb, s_scale, fresh[j], fresh[j + 1]);
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
				h0 = (live && p < own) ? oklab_tile_src<T>(a, (blockIdx.x + p * gridDim.x) * kOkTiles + ct, unused, g0) : 0u;
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
				if (p1_valid && (kGeneral ? h1 != 0u : kk * G::kRowsPerBand < h1)) acc1 = walk(s_p1, acc1, false);
				if (p2_valid && (kGeneral ? h2 != 0u : kk * G::kRowsPerBand < h2)) acc2 = walk(s_p2, acc2, true);
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

hipError_t launch_oklab(const ShrinkArgs &a, uint32_t n_cus, hipStream_t stream)
{
	const uint32_t lds_bytes = (512u + 2u * 132u + 64u) * 4u + 2u * kOkBand * 4u;
	const uint32_t n_batches = (a.ok_count + kOkTiles - 1) / kOkTiles;
	const uint32_t blocks = n_batches < n_cus ? n_batches : n_cus;
	hipError_t e;
	auto go = [&](auto kernel) -> hipError_t {
		if ((e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes)) != hipSuccess) return e;
		hipLaunchKernelGGL(kernel, dim3(blocks), dim3(1024), lds_bytes, stream, a);
		return hipGetLastError();
	};
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
