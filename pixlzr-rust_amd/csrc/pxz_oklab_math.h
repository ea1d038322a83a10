// pxz_oklab_math.h -- the colour conversion of the Oklab-MAD detector (Srgba<u8> -> linear -> Oklab, reference
// src/operations.rs:56-59 with palette 0.7.6), shared by the block-cooperative detector kernels (pxz_oklab.hip) and the
// rectangle kernel of tree::process (pxz_tree.hip).  Every translation unit that includes it is compiled with
// -ffp-contract=off.
#pragma once
#include "pxz_device.h"

namespace pxz {

// glibc 2.35 cbrtf for the inputs of the Oklab detector, six at a time (l, m, s of two pixels).  Same arithmetic as
// cbrt_f32 (pxz_device.h), in two parts (cbrt6_head, cbrt6_tail: LDS traffic of the caller goes between them), with these liberties, every one of them checked over the whole domain -- the conversion
// is a pure function of the colour bytes, 2^24 inputs, and tests/test_gpu_parity.py::test_oklab_conversion_of_every_colour
// runs all of them through this very function against the oracle's exact arithmetic, bit for bit:
//  * frexpf by bit fields (the inputs are zero or normal, < 2): mantissa = fraction bits under the exponent of 0.5,
//    exponent = the biased exponent field;
//  * the tail `(float)(q * third[2 + xe % 3])` followed by `ldexpf(.., xe / 3)` is ONE multiplication by
//    2^(xe/3) * third[..]: scaling a double by a power of two is exact and commutes with the rounding to float (no
//    underflow in this range).  `scale` is the LDS table of those doubles indexed by the exponent field; entry 0 is
//    0.0, which makes cbrt(0) = +0 fall out of the same instructions (no zero test, no select);
//  * the seed polynomial and the sums t2 + 2 xm, 2 t2 + xm as FMAs (the sums are exact in double either way; the
//    polynomial differs from glibc's separate operations by a few 2^-53 before the rounding to float);
//  * the quotient of the Halley step is v_rcp_f64 + ONE Newton step times the numerator (4 instructions instead of the
//    8 of a correctly rounded division; a raw v_rcp_f64 fails on 4.2 M colours, so the test bites).
// Every liberty is a function of ONE input value, so the 2^24-colour test covers it completely.  (Sharing one
// reciprocal among the six quotients -- Montgomery's trick, 2 instructions per root cheaper -- was tried and dropped
// for that reason: the products make a root's last double bits depend on its neighbours, the exhaustive test then
// samples 2^24 of 2^48 pairs, and with one partner per colour it already found a colour whose float flipped.)
struct Cbrt6State {
	float x[6];  // the six inputs (l, m, s of two pixels): everything else is derived in cbrt6_tail
};
// first part: nothing but the hand-over (round 3: mantissas and scale factors used to be extracted here, 24 registers that
// then lived across the interval's barrier; as six floats the state leaves room for the shared reciprocal below)
__device__ __forceinline__ void cbrt6_head(const float (&x)[6], Cbrt6State &st)
{
#pragma unroll
	for (int i = 0; i < 6; ++i) st.x[i] = x[i];
}
// mantissa and scale factor of one input (bit fields, one LDS read whose result is not needed before the root's last step)
__device__ __forceinline__ void cbrt_split(float x, const double *scale, double &xmd, double &sc)
{
	const uint32_t bits = __float_as_uint(x);
	uint32_t xmb;  // frexpf mantissa, [0.5, 1): one v_and_or_b32 (the compiler splits it in two)
	asm("v_and_or_b32 %0, %1, %2, 0.5" : "=v"(xmb) : "v"(bits), "s"(0x007fffffu));
	sc = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(scale) + ((bits >> 20) & 0x7f8u));
	xmd = (double)__uint_as_float(xmb);
}
// second part: mantissas, seed, Halley step, scaling
#if !defined(PXZ_CBRT_OWN_RCP)
// ONE reciprocal for the three roots of a pixel (l, m, s: indices j, j + 2, j + 4): 1 / (d0 d1 d2), then each root's own
// reciprocal as that times the other two denominators -- six multiplications instead of two v_rcp_f64 (16 issue cycles
// each) and two Newton steps.  The three roots of a pixel are functions of the same colour, so the conversion stays a pure
// function of one colour and the 2^24-colour test still covers it completely (it passes; PXZ_CBRT_OWN_RCP builds the
// round-2 form with a reciprocal per root).
__device__ __forceinline__ void cbrt6_tail(const Cbrt6State &st, const double *scale, float (&y)[6])
{
	float x[6];
#pragma unroll
	for (int i = 0; i < 6; ++i) x[i] = st.x[i];
	// one pixel after the other (its three roots side by side): both at once want 8 registers more than the kernels have
#pragma unroll
	for (int j = 0; j < 2; ++j) {
		double den[3], num[3], sc[3];
#pragma unroll
		for (int k = 0; k < 3; ++k) {
			double xmd;
			cbrt_split(x[j + 2 * k], scale, xmd, sc[k]);
			const float u = (float)__builtin_fma(__builtin_fma(-0.191502161678719066, xmd, 0.697570460207922770), xmd, 0.492659620528969547);
			const float t2 = u * u * u;  // two f32 roundings, as glibc's float t2 = u * u * u
			const double ud = (double)u, t2d = (double)t2;
			den[k] = __builtin_fma(2.0, t2d, xmd);
			num[k] = ud * __builtin_fma(2.0, xmd, t2d);
		}
		const double p01 = den[0] * den[1], d012 = p01 * den[2];
		double r = __builtin_amdgcn_rcp(d012);
		r = __builtin_fma(r, __builtin_fma(-d012, r, 1.0), r);
		const double r2 = r * p01, t = r * den[2], r1 = t * den[0], r0 = t * den[1];
		y[j] = (float)(num[0] * r0 * sc[0]);  // (u * N / D) * factor, glibc's order
		y[j + 2] = (float)(num[1] * r1 * sc[1]);
		y[j + 4] = (float)(num[2] * r2 * sc[2]);
		// (an anchor: the second pixel's inputs pass through a statement that needs the first pixel's results)
		if (j == 0) asm volatile("" : "+v"(y[0]), "+v"(y[2]), "+v"(y[4]), "+v"(x[1]), "+v"(x[3]), "+v"(x[5]));
	}
}
#else
__device__ __forceinline__ void cbrt6_tail(const Cbrt6State &st, const double *scale, float (&y)[6])
{
	double xmd[6], sc[6], den[6], num[6];
#pragma unroll
	for (int i = 0; i < 6; ++i) cbrt_split(st.x[i], scale, xmd[i], sc[i]);
	f32x2 u[3];
#pragma unroll
	for (int i = 0; i < 6; ++i) {
		const float ui = (float)__builtin_fma(__builtin_fma(-0.191502161678719066, xmd[i], 0.697570460207922770), xmd[i], 0.492659620528969547);
		if (i & 1) u[i >> 1].y = ui; else u[i >> 1].x = ui;
	}
#pragma unroll
	for (int k = 0; k < 3; ++k) {
		const f32x2 t2 = u[k] * u[k] * u[k];  // two f32 roundings, as glibc's float t2 = u * u * u
#pragma unroll
		for (int h = 0; h < 2; ++h) {
			const int i = 2 * k + h;
			const double ud = (double)(h ? u[k].y : u[k].x), t2d = (double)(h ? t2.y : t2.x);
			den[i] = __builtin_fma(2.0, t2d, xmd[i]);
			num[i] = ud * __builtin_fma(2.0, xmd[i], t2d);
		}
	}
#pragma unroll
	for (int i = 0; i < 6; ++i) {
		double r = __builtin_amdgcn_rcp(den[i]);
		r = __builtin_fma(r, __builtin_fma(-den[i], r, 1.0), r);
		y[i] = (float)(num[i] * r * sc[i]);  // (u * N / D) * factor, glibc's order
	}
}
#endif

// byte BYTE of v, times 16: the byte offset of a 256-entry float4 table row, in one SDWA shift
template <int BYTE>
__device__ __forceinline__ uint32_t byte_times16(uint32_t v)
{
	uint32_t r;
	const uint32_t four = 4u;
	if constexpr (BYTE == 0)
		asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(four), "v"(v));
	else if constexpr (BYTE == 1)
		asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(four), "v"(v));
	else
		asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(four), "v"(v));
	return r;
}
__device__ __forceinline__ float4 row_at(const float4 *table, uint32_t byte_offset)
{
	return *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(table) + byte_offset);
}

__device__ __forceinline__ f32x2 pk_add_f32_asm(f32x2 a, f32x2 b)
{
	f32x2 r;
	asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
	return r;
}

// Srgba<u8> -> linear -> Oklab of two pixels (operations.rs:56-59; palette 0.7.6): LUT, then Ottosson's matrices
// with left-to-right f32 sums.  The first matrix's nine products per pixel come out of the table (s_lms[256 c + v] =
// the three products of channel c's linear value: the same single-rounded f32 multiplications, done once per
// block), which leaves its six additions; the second matrix runs two pixels per packed-f32 instruction (same IEEE
// results per component).  out[k] = {a, b, l} of pixel k, the order the reference sums them in.
__device__ __forceinline__ void oklab_pair_head(uint32_t v0, uint32_t v1, const float4 *s_lms, Cbrt6State &st)
{
	const float4 r0 = row_at(s_lms, byte_times16<0>(v0)), g0 = row_at(s_lms + 256, byte_times16<1>(v0)), b0 = row_at(s_lms + 512, byte_times16<2>(v0));
	const float4 r1 = row_at(s_lms, byte_times16<0>(v1)), g1 = row_at(s_lms + 256, byte_times16<1>(v1)), b1 = row_at(s_lms + 512, byte_times16<2>(v1));
	// l, m (packed) and s of each pixel: (r + g) + b, as 0.41.. * r + 0.53.. * g + 0.05.. * b evaluates
	const f32x2 lm0 = f32x2{r0.x, r0.y} + f32x2{g0.x, g0.y} + f32x2{b0.x, b0.y};
	const f32x2 lm1 = f32x2{r1.x, r1.y} + f32x2{g1.x, g1.y} + f32x2{b1.x, b1.y};
	// s on the (z, w) halves of the rows with spelled-out packed adds (w is zero padding).  Two reasons: left alone, the
	// compiler gathers the two pixels' z terms into register pairs with six moves in order to pack the adds; and with
	// all four components in use the rows are fetched by ds_read_b128 (4 LDS cycles, 16 lanes each) instead of
	// ds_read_b96 (8 cycles, 8 lanes each).
	const float s0 = pk_add_f32_asm(pk_add_f32_asm(f32x2{r0.z, r0.w}, f32x2{g0.z, g0.w}), f32x2{b0.z, b0.w}).x;
	const float s1 = pk_add_f32_asm(pk_add_f32_asm(f32x2{r1.z, r1.w}, f32x2{g1.z, g1.w}), f32x2{b1.z, b1.w}).x;
	// l, m, s are zero only for black (every coefficient is positive, the table is zero at 0 only); their cube roots
	// are then +0 (entry 0 of the scale table) and so are L = (+0 + +0) - +0, a and b
	const float x[6] = {lm0.x, lm1.x, lm0.y, lm1.y, s0, s1};
	cbrt6_head(x, st);
}
__device__ __forceinline__ void oklab_pair_tail(const Cbrt6State &st, const double *s_scale, float (&out0)[3], float (&out1)[3])
{
	float c[6];
	cbrt6_tail(st, s_scale, c);
	const f32x2 l_ = {c[0], c[1]}, m_ = {c[2], c[3]}, s_ = {c[4], c[5]};
	const f32x2 L = 0.2104542553f * l_ + 0.7936177850f * m_ - 0.0040720468f * s_;
	const f32x2 A = 1.9779984951f * l_ - 2.4285922050f * m_ + 0.4505937099f * s_;
	const f32x2 B = 0.0259040371f * l_ + 0.7827717662f * m_ - 0.8086757660f * s_;
	out0[2] = L.x; out1[2] = L.y;
	out0[0] = A.x; out1[0] = A.y;
	out0[1] = B.x; out1[1] = B.y;
}
__device__ __forceinline__ void oklab_pair(uint32_t v0, uint32_t v1, const float4 *s_lms, const double *s_scale,
                                           float (&out0)[3], float (&out1)[3])
{
	Cbrt6State st;
	oklab_pair_head(v0, v1, s_lms, st);
	oklab_pair_tail(st, s_scale, out0, out1);
}

// The conversion tables of the Oklab kernels in LDS: the products of the sRGB u8 -> linear values with the columns of
// the first matrix (3 x 256 float4), a / 255 (256), and 128 doubles
// 2^(xe/3) * cbrt(2)^(xe%3) indexed by the biased exponent field (xe = i - 126; entry 0 is 0.0).  Call from the first 256 threads of a block, then a block barrier.
__device__ __forceinline__ void oklab_fill_tables(float4 *s_lms, float *s_alpha, double *s_scale, uint32_t t)
{
	if (t < 256) {
		const float x = __uint_as_float(kSrgbToLinearBits[t]);  // sRGB u8 -> linear
		s_lms[t] = make_float4(0.4122214708f * x, 0.2119034982f * x, 0.0883024619f * x, 0.0f);        // red's share of l, m, s
		s_lms[256 + t] = make_float4(0.5363325363f * x, 0.6806995451f * x, 0.2817188376f * x, 0.0f);  // green's
		s_lms[512 + t] = make_float4(0.0514459929f * x, 0.1073969566f * x, 0.6299787005f * x, 0.0f);  // blue's
		s_alpha[t] = __fdiv_rn((float)t, 255.0f);
	}
	if (t < 128) {
		// entry f: x = xm * 2^xe with xe = f - 126 (frexpf's convention); 0 for f = 0 (x = 0: its cube root is +0)
		const int xe = (int)t - 126;
		const int q3 = xe / 3, r3 = xe - 3 * q3;  // C semantics: the remainder carries the sign of xe
		const double third = r3 == 0 ? 1.0
		                   : r3 == 1 ? 1.2599210498948731648
		                   : r3 == 2 ? 1.5874010519681994748
		                   : r3 == -1 ? 1.0 / 1.2599210498948731648
		                              : 1.0 / 1.5874010519681994748;
		s_scale[t] = t == 0 ? 0.0 : ldexp(third, q3);  // exact
	}
}


}  // namespace pxz
