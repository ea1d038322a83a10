/*
 * pixlzr_oracle.c — CPU restatement of the pixlzr encode hot path (test oracle).
 * See pixlzr_oracle.h for what this is, who may use it and how it is pinned.
 *
 * Build: gcc -O3 -march=x86-64-v3 -ffp-contract=off -fno-fast-math (see Makefile).
 * -ffp-contract=off matters: Rust never contracts a*b+c into an FMA, and the
 * f32 results of the Oklab path end up in the bitstream.
 */
#define _GNU_SOURCE
#include "pixlzr_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------ */
/* math primitives                                                          */
/* ------------------------------------------------------------------------ */

/* f32::cbrt -> platform libm cbrtf (Rust std, called at src/operations.rs:56-59
 * through palette 0.7.6's linear-sRGB -> Oklab).  This is the glibc 2.35
 * algorithm (frexp, quadratic seed in double, one Halley step in double),
 * restated so that the HIP kernel can run the very same arithmetic; it is
 * bit-identical to this image's cbrtf on every float in [0,4]
 * (orc_selftest_cbrtf, tests/test_oracle_math.py). */
float orc_cbrtf(float x)
{
	static const double third_factor[5] = {
		1.0 / 1.5874010519681994748, /* 2^(-2/3) */
		1.0 / 1.2599210498948731648, /* 2^(-1/3) */
		1.0,
		1.2599210498948731648, /* 2^(1/3) */
		1.5874010519681994748, /* 2^(2/3) */
	};
	int xe;
	float xm = frexpf(fabsf(x), &xe);
	if (xe == 0 && (x == 0.0f || x != x || isinf(x)))
		return x + x;
	float u = (float)(0.492659620528969547 +
	                  (0.697570460207922770 - 0.191502161678719066 * (double)xm) * (double)xm);
	float t2 = u * u * u;
	float ym = (float)((double)u * ((double)t2 + 2.0 * (double)xm) / (2.0 * (double)t2 + (double)xm) *
	                   third_factor[2 + xe % 3]);
	return ldexpf(x > 0.0f ? ym : -ym, xe / 3);
}

/* f32::hypot -> libm hypotf (src/operations.rs:154).  glibc 2.35 computes it in
 * double; bit-identical on 2e8 random pairs (orc_selftest_hypotf). */
float orc_hypotf(float x, float y)
{
	return (float)sqrt((double)x * (double)x + (double)y * (double)y);
}

/* sRGB u8 -> linear f32: palette 0.7.6 `Srgb::into_linear` for u8 -> f32 goes
 * through fast-srgb8 1.0.0's 256-entry table (Cargo.lock:427-428), whose
 * entries are the correctly rounded f32 of the sRGB EOTF.  Table generated
 * with 80-digit decimal arithmetic (identical to rounding the f64 formula). */
static const uint32_t SRGB_U8_TO_LINEAR_BITS[256] = {
#include "srgb_lut.inc"
};

float orc_srgb_u8_to_linear(uint8_t v)
{
	float f;
	memcpy(&f, &SRGB_U8_TO_LINEAR_BITS[v], 4);
	return f;
}

uint64_t orc_selftest_cbrtf(uint32_t lo_bits, uint32_t hi_bits, uint32_t step)
{
	uint64_t bad = 0;
	if (step == 0)
		step = 1;
	for (uint64_t b = lo_bits; b <= hi_bits; b += step) {
		uint32_t bb = (uint32_t)b;
		float x, a, m;
		memcpy(&x, &bb, 4);
		a = cbrtf(x);
		m = orc_cbrtf(x);
		if (memcmp(&a, &m, 4) != 0)
			bad++;
	}
	return bad;
}

static inline uint32_t fmix32(uint32_t h)
{
	h ^= h >> 16;
	h *= 0x85ebca6bu;
	h ^= h >> 13;
	h *= 0xc2b2ae35u;
	h ^= h >> 16;
	return h;
}

uint64_t orc_selftest_hypotf(uint64_t n, uint32_t seed)
{
	uint64_t bad = 0;
	for (uint64_t i = 0; i < n; i++) {
		uint32_t b1 = fmix32((uint32_t)(2 * i) ^ seed) % 0x42000000u; /* [0,32) */
		uint32_t b2 = (i & 1) ? b1 : fmix32((uint32_t)(2 * i + 1) ^ seed) % 0x42000000u;
		float x, y, a, m;
		memcpy(&x, &b1, 4);
		memcpy(&y, &b2, 4);
		a = hypotf(x, y);
		m = orc_hypotf(x, y);
		if (memcmp(&a, &m, 4) != 0)
			bad++;
	}
	return bad;
}

/* ------------------------------------------------------------------------ */
/* tiling                                                                   */
/* ------------------------------------------------------------------------ */

/* src/data_types/iter.rs:38-41, src/split.rs:45-46: ceil in f64 */
void orc_grid(uint32_t iw, uint32_t ih, uint32_t bw, uint32_t bh, uint32_t *cols, uint32_t *rows)
{
	*cols = (uint32_t)ceil((double)iw / (double)bw);
	*rows = (uint32_t)ceil((double)ih / (double)bh);
}

/* src/split.rs:10-27 (clamp at the right/bottom edge), iter.rs:64-76 (row-major order) */
void orc_tile_rect(uint32_t iw, uint32_t ih, uint32_t bw, uint32_t bh, uint32_t tile,
                   uint32_t *x, uint32_t *y, uint32_t *w, uint32_t *h)
{
	uint32_t cols, rows;
	orc_grid(iw, ih, bw, bh, &cols, &rows);
	uint32_t tx = tile % cols, ty = tile / cols;
	*x = tx * bw;
	*y = ty * bh;
	*w = bw < iw - *x ? bw : iw - *x;
	*h = bh < ih - *y ? bh : ih - *y;
}

/* ------------------------------------------------------------------------ */
/* directional detector: src/operations.rs:192-259                          */
/* ------------------------------------------------------------------------ */

static float x86_default_nan(void)
{
	/* 0.0/0.0 evaluated at run time by SSE yields the "real indefinite"
	 * QNaN, sign bit SET; spelled out so the compiler cannot fold it to +NaN. */
	uint32_t bits = 0xFFC00000u;
	float f;
	memcpy(&f, &bits, 4);
	return f;
}

void orc_lod_directional(const uint8_t *tile, uint32_t w, uint32_t h, uint32_t c, uint32_t pitch,
                         float *hz, float *vr, uint64_t *out_sum_hz, uint64_t *out_sum_vr)
{
	uint64_t sum_hz = 0, sum_vr = 0;
	/* operations.rs:220-221: `0..height-2`, `0..width-2` (callers guarantee w,h >= 2) */
	for (uint32_t y = 0; y + 2 < h; y++) {
		const uint8_t *r0 = tile + (size_t)y * pitch;
		const uint8_t *r1 = r0 + pitch;
		const uint8_t *r2 = r1 + pitch;
		for (uint32_t x = 0; x + 2 < w; x++) {
			for (uint32_t k = 0; k < 3; k++) { /* alpha ignored, :218 */
				int v00 = r0[(x + 0) * c + k], v01 = r0[(x + 1) * c + k], v02 = r0[(x + 2) * c + k];
				int v10 = r1[(x + 0) * c + k], v12 = r1[(x + 2) * c + k];
				int v20 = r2[(x + 0) * c + k], v21 = r2[(x + 1) * c + k], v22 = r2[(x + 2) * c + k];
				int16_t px_hz = (int16_t)(-v00 - 2 * v01 - v02 + v20 + 2 * v21 + v22); /* :240-241 */
				int16_t px_vr = (int16_t)(-v00 - 2 * v10 - v20 + v02 + 2 * v12 + v22); /* :244-245 */
				sum_hz += (uint64_t)(px_hz < 0 ? -px_hz : px_hz);                       /* :247 */
				sum_vr += (uint64_t)(px_vr < 0 ? -px_vr : px_vr);                       /* :248 */
			}
		}
	}
	/* :253-258, BASE_FACTOR = 2<<11 = 4096 (:158) */
	uint64_t fac = (uint64_t)(w - 2) * (uint64_t)(h - 2) * 4096u;
	if (fac == 0) {
		*hz = x86_default_nan();
		*vr = x86_default_nan();
	} else {
		double factor = (double)fac;
		*hz = (float)((double)sum_hz / factor);
		*vr = (float)((double)sum_vr / factor);
	}
	if (out_sum_hz)
		*out_sum_hz = sum_hz;
	if (out_sum_vr)
		*out_sum_vr = sum_vr;
}

/* ------------------------------------------------------------------------ */
/* Oklab MAD detector: src/operations.rs:26-126                             */
/* ------------------------------------------------------------------------ */

/* palette 0.7.6: Srgb(a)<u8>::into_linear() (LUT; alpha = a/255) then
 * Oklab::from_color_unclamped(LinSrgb) = Ottosson's direct matrices in f32,
 * sums evaluated left to right, no FMA. */
void orc_oklab_pixel(const uint8_t *px, uint32_t c, float out[4])
{
	float r = orc_srgb_u8_to_linear(px[0]);
	float g = orc_srgb_u8_to_linear(px[1]);
	float b = orc_srgb_u8_to_linear(px[2]);
	float l = 0.4122214708f * r + 0.5363325363f * g + 0.0514459929f * b;
	float m = 0.2119034982f * r + 0.6806995451f * g + 0.1073969566f * b;
	float s = 0.0883024619f * r + 0.2817188376f * g + 0.6299787005f * b;
	float l_ = orc_cbrtf(l), m_ = orc_cbrtf(m), s_ = orc_cbrtf(s);
	out[0] = 0.2104542553f * l_ + 0.7936177850f * m_ - 0.0040720468f * s_; /* L */
	out[1] = 1.9779984951f * l_ - 2.4285922050f * m_ + 0.4505937099f * s_; /* a */
	out[2] = 0.0259040371f * l_ + 0.7827717662f * m_ - 0.8086757660f * s_; /* b */
	out[3] = c == 4 ? (float)px[3] / 255.0f : 1.0f;
}

/* the same conversion over n packed pixels of c channels: out[4i..4i+3] = {L, a, b, alpha} */
void orc_oklab_pixels(const uint8_t *px, uint32_t c, uint64_t n, float *out)
{
	for (uint64_t i = 0; i < n; i++)
		orc_oklab_pixel(px + i * c, c, out + 4 * i);
}

static float lod_oklab_scaled(const uint8_t *tile, uint32_t w, uint32_t h, uint32_t c, uint32_t pitch, float factor,
                              float scale2);

float orc_lod_oklab(const uint8_t *tile, uint32_t w, uint32_t h, uint32_t c, uint32_t pitch, float factor)
{
	/* after = x * factor * BASE_FACTOR(10.0) (pixlzr.rs:15,162) */
	return lod_oklab_scaled(tile, w, h, c, pitch, factor, 10.0f);
}

/* get_block_variance (operations.rs:26-126) with before = |x - avg| and after = (x * factor) * scale2;
 * factor = scale2 = 1 is the identity closure of process() (process/mod.rs:108-111): x * 1 * 1 == x */
static float lod_oklab_scaled(const uint8_t *tile, uint32_t w, uint32_t h, uint32_t c, uint32_t pitch, float factor,
                              float scale2)
{
	float count = (float)(w * h); /* :51 */
	/* sums in the reference's order [a, b, l, alpha] (:60-63 / :98-100) */
	float sum[4] = {0.f, 0.f, 0.f, 0.f};
	for (uint32_t y = 0; y < h; y++) {
		const uint8_t *row = tile + (size_t)y * pitch;
		for (uint32_t x = 0; x < w; x++) {
			float laba[4];
			orc_oklab_pixel(row + x * c, c, laba);
			sum[0] += laba[1];
			sum[1] += laba[2];
			sum[2] += laba[0];
			if (c == 4)
				sum[3] += laba[3];
		}
	}
	float avg[4];
	for (int k = 0; k < 4; k++)
		avg[k] = sum[k] / count; /* :65-68 */
	float delta[4] = {0.f, 0.f, 0.f, 0.f};
	for (uint32_t y = 0; y < h; y++) {
		const uint8_t *row = tile + (size_t)y * pitch;
		for (uint32_t x = 0; x < w; x++) {
			float laba[4];
			orc_oklab_pixel(row + x * c, c, laba); /* recomputed, :76-79 */
			/* before = |x - avg| (pixlzr.rs:160-161) */
			delta[0] += fabsf(laba[1] - avg[0]);
			delta[1] += fabsf(laba[2] - avg[1]);
			delta[2] += fabsf(laba[0] - avg[2]);
			if (c == 4)
				delta[3] += fabsf(laba[3] - avg[3]);
		}
	}
	float total = c == 4 ? (delta[0] + delta[1] + delta[2] + delta[3]) /* :89 */
	                     : (delta[0] + delta[1] + delta[2]);            /* :124 */
	float x = total / count;
	return x * factor * scale2;
}

/* ------------------------------------------------------------------------ */
/* level decision: src/operations.rs:128-156                                */
/* ------------------------------------------------------------------------ */

static int sign_positive(float v)
{
	uint32_t b;
	memcpy(&b, &v, 4);
	return (b >> 31) == 0;
}

/* Rust f32::max: returns the non-NaN operand */
static float rust_maxf(float a, float b)
{
	if (a != a)
		return b;
	if (b != b)
		return a;
	return a > b ? a : b;
}
static float rust_minf(float a, float b)
{
	if (a != a)
		return b;
	if (b != b)
		return a;
	return a < b ? a : b;
}

static float parse_value(float value) /* :128-138 */
{
	if (sign_positive(value))
		return value;
	float v = rust_maxf(1.0f + value, 0.0f);
	return sign_positive(v) ? v : 1.0f;
}

void orc_reduce_dims(float v0, float v1, uint32_t w, uint32_t h,
                     uint32_t *nw, uint32_t *nh, float *stored_value)
{
	float a = parse_value(v0), b = parse_value(v1);         /* :145 */
	float level_hz = exp2f(rust_minf(roundf(log2f(a)), 0.0f)); /* :147 */
	float level_vr = exp2f(rust_minf(roundf(log2f(b)), 0.0f)); /* :148 */
	double dw = (double)w * (double)level_hz;               /* :150 */
	double dh = (double)h * (double)level_vr;               /* :151 */
	if (!(dw > 1.0)) /* f64::max(1.0): NaN -> 1.0 */
		dw = 1.0;
	if (!(dh > 1.0))
		dh = 1.0;
	*nw = (uint32_t)ceil(dw);
	*nh = (uint32_t)ceil(dh);
	*stored_value = orc_hypotf(a, b); /* :154 */
}

/* ------------------------------------------------------------------------ */
/* resample: fast_image_resize 4.2.1 (restated from its published algorithm) */
/* ------------------------------------------------------------------------ */

#define FIR_PI 3.14159265358979323846

static double fir_box(double x) { return (x > -0.5 && x <= 0.5) ? 1.0 : 0.0; }
static double fir_hamming(double x)
{
	x = fabs(x);
	if (x == 0.0)
		return 1.0;
	if (x >= 1.0)
		return 0.0;
	x *= FIR_PI;
	return (0.54 + 0.46 * cos(x)) * sin(x) / x;
}
static double fir_catmullrom(double x)
{
	const double a = -0.5;
	x = fabs(x);
	if (x < 1.0)
		return ((a + 2.0) * x - (a + 3.0)) * x * x + 1.0;
	if (x < 2.0)
		return (((x - 5.0) * x + 8.0) * x - 4.0) * a;
	return 0.0;
}
static double fir_gaussian(double x)
{
	/* gaussian, sigma = 0.5, support 3 (image-rs compatible) */
	if (x <= -3.0 || x >= 3.0)
		return 0.0;
	return exp(-(x * x) / 0.5) / sqrt(2.0 * FIR_PI * 0.25);
}
static double fir_sinc(double x)
{
	if (x == 0.0)
		return 1.0;
	x *= FIR_PI;
	return sin(x) / x;
}
static double fir_lanczos3(double x)
{
	if (x >= -3.0 && x < 3.0)
		return fir_sinc(x) * fir_sinc(x / 3.0);
	return 0.0;
}

static double fir_bilinear(double x)
{
	x = fabs(x);
	return x < 1.0 ? 1.0 - x : 0.0;
}

typedef double (*fir_filter_fn)(double);

/* src/data_types/mod.rs:65-107: the downscale branch maps Triangle -> Hamming (!), the upscale branch
 * (`SuperSampling(filter, 2)`, which is a plain convolution when nothing shrinks) Triangle -> Bilinear */
static int fir_filter_for(uint32_t filter, int upscale, fir_filter_fn *fn, double *support)
{
	switch (filter) {
	case ORC_TRIANGLE:
		*fn = upscale ? fir_bilinear : fir_hamming;
		*support = 1.0;
		return 0;
	case ORC_CATMULLROM:
		*fn = fir_catmullrom;
		*support = 2.0;
		return 0;
	case ORC_GAUSSIAN:
		*fn = fir_gaussian;
		*support = 3.0;
		return 0;
	case ORC_LANCZOS3:
		*fn = fir_lanczos3;
		*support = 3.0;
		return 0;
	default:
		(void)fir_box;
		return -1;
	}
}

typedef struct {
	int out_size, window, precision;
	int32_t *start, *size;
	int16_t *k; /* out_size * window */
} fir_axis;

static void fir_axis_free(fir_axis *a)
{
	free(a->start);
	free(a->size);
	free(a->k);
}

/* fir `precompute_coefficients` + `Normalizer16::new` (Pillow-SIMD lineage) */
static int fir_axis_build(fir_axis *a, uint32_t in_size, uint32_t out_size, uint32_t filter, int upscale)
{
	fir_filter_fn fn;
	double support;
	if (fir_filter_for(filter, upscale, &fn, &support) != 0)
		return -1;
	double scale = (double)in_size / (double)out_size;
	double filter_scale = scale > 1.0 ? scale : 1.0;
	double radius = support * filter_scale;
	int window = (int)ceil(radius) * 2 + 1;
	double recip = 1.0 / filter_scale;
	a->out_size = (int)out_size;
	a->window = window;
	a->start = (int32_t *)calloc(out_size, sizeof(int32_t));
	a->size = (int32_t *)calloc(out_size, sizeof(int32_t));
	a->k = (int16_t *)calloc((size_t)out_size * window, sizeof(int16_t));
	double *wf = (double *)calloc((size_t)out_size * window, sizeof(double));
	double max_w = 0.0;
	for (uint32_t o = 0; o < out_size; o++) {
		double in_center = ((double)o + 0.5) * scale;
		double lo = floor(in_center - radius);
		double hi = ceil(in_center + radius);
		int x_min = lo < 0.0 ? 0 : (int)lo;
		int x_max = hi > (double)in_size ? (int)in_size : (int)hi;
		double center = in_center - 0.5;
		double ww = 0.0;
		double *wrow = wf + (size_t)o * window;
		int n = x_max - x_min;
		for (int i = 0; i < n; i++) {
			double w = fn(((double)(x_min + i) - center) * recip);
			wrow[i] = w;
			ww += w;
		}
		if (ww != 0.0)
			for (int i = 0; i < n; i++)
				wrow[i] /= ww;
		a->start[o] = x_min;
		a->size[o] = n;
		for (int i = 0; i < n; i++)
			if (wrow[i] > max_w)
				max_w = wrow[i];
	}
	/* Normalizer16: the largest precision that keeps max coefficient < 2^15 */
	int precision = 0;
	for (int cur = 0; cur < 22; cur++) {
		precision = cur;
		int next_value = (int)round(max_w * (double)(1 << (precision + 1)));
		if (next_value >= (1 << 15))
			break;
	}
	a->precision = precision;
	double sc = (double)(1 << precision);
	for (size_t i = 0; i < (size_t)out_size * window; i++) {
		double v = round(wf[i] * sc); /* f64::round = half away from zero */
		if (v > 32767.0)
			v = 32767.0;
		if (v < -32768.0)
			v = -32768.0;
		a->k[i] = (int16_t)v;
	}
	free(wf);
	return 0;
}

int orc_fir_coeffs(uint32_t in_size, uint32_t out_size, uint32_t filter,
                   int32_t *starts, int32_t *sizes, int16_t *coeffs, int32_t *window, int32_t *precision)
{
	fir_axis a;
	if (fir_axis_build(&a, in_size, out_size, filter, out_size > in_size) != 0)
		return -1;
	*window = a.window;
	*precision = a.precision;
	if (starts)
		memcpy(starts, a.start, sizeof(int32_t) * out_size);
	if (sizes)
		memcpy(sizes, a.size, sizeof(int32_t) * out_size);
	if (coeffs)
		memcpy(coeffs, a.k, sizeof(int16_t) * (size_t)out_size * a.window);
	fir_axis_free(&a);
	return 0;
}

static inline uint8_t fir_clip8(int32_t v, int precision)
{
	v >>= precision; /* arithmetic */
	return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

/* fir alpha/u8x4: mul_div_255 and the reciprocal-table un-premultiply */
static inline uint8_t fir_mul_div_255(uint8_t a, uint8_t b)
{
	uint32_t t = (uint32_t)a * b + 128u;
	return (uint8_t)(((t >> 8) + t) >> 8);
}
static inline uint32_t fir_recip_alpha(uint32_t alpha)
{
	if (alpha == 0)
		return 0;
	return ((255u * 512u) / alpha + 1u) >> 1; /* precision 8 */
}
static inline uint8_t fir_div_and_clip(uint8_t v, uint32_t recip)
{
	uint32_t r = ((uint32_t)v * recip + 128u) >> 8;
	return (uint8_t)(r > 255u ? 255u : r);
}

/* ResizeAlg::Nearest */
static void fir_nearest(const uint8_t *src, uint32_t w, uint32_t h, uint32_t c, uint32_t pitch,
                        uint8_t *dst, uint32_t nw, uint32_t nh)
{
	double x_scale = (double)w / (double)nw, y_scale = (double)h / (double)nh;
	double x0 = x_scale * 0.5, y0 = y_scale * 0.5;
	for (uint32_t oy = 0; oy < nh; oy++) {
		uint32_t sy = (uint32_t)(y0 + y_scale * (double)oy);
		if (sy >= h)
			sy = h - 1;
		for (uint32_t ox = 0; ox < nw; ox++) {
			uint32_t sx = (uint32_t)(x0 + x_scale * (double)ox);
			if (sx >= w)
				sx = w - 1;
			memcpy(dst + ((size_t)oy * nw + ox) * c, src + (size_t)sy * pitch + (size_t)sx * c, c);
		}
	}
}

int orc_resize(const uint8_t *src, uint32_t w, uint32_t h, uint32_t c, uint32_t pitch,
               uint8_t *dst, uint32_t nw, uint32_t nh, uint32_t filter)
{
	if (nw == w && nh == h) { /* block.rs:279-281: clone */
		for (uint32_t y = 0; y < h; y++)
			memcpy(dst + (size_t)y * w * c, src + (size_t)y * pitch, (size_t)w * c);
		return 0;
	}
	if (filter == ORC_NEAREST) {
		fir_nearest(src, w, h, c, pitch, dst, nw, nh);
		return 0;
	}
	if (filter > ORC_LANCZOS3)
		return -1;
	/* working copy (block.rs:309), premultiplied when U8x4 (fir default mul_div_alpha) */
	uint8_t *work = (uint8_t *)malloc((size_t)w * h * c);
	for (uint32_t y = 0; y < h; y++) {
		const uint8_t *s = src + (size_t)y * pitch;
		uint8_t *d = work + (size_t)y * w * c;
		if (c == 4) {
			for (uint32_t x = 0; x < w; x++) {
				uint8_t al = s[x * 4 + 3];
				d[x * 4 + 0] = fir_mul_div_255(s[x * 4 + 0], al);
				d[x * 4 + 1] = fir_mul_div_255(s[x * 4 + 1], al);
				d[x * 4 + 2] = fir_mul_div_255(s[x * 4 + 2], al);
				d[x * 4 + 3] = al;
			}
		} else {
			memcpy(d, s, (size_t)w * c);
		}
	}
	int need_h = nw != w, need_v = nh != h;
	const int upscale = nw > w || nh > h; /* block.rs:301-304: one flag for both axes */
	const uint8_t *cur = work;
	uint32_t cur_w = w;
	uint8_t *tmp = NULL;
	if (need_h) {
		fir_axis ax;
		fir_axis_build(&ax, w, nw, filter, upscale);
		uint8_t *out = need_v ? (tmp = (uint8_t *)malloc((size_t)nw * h * c)) : dst;
		int32_t init = 1 << (ax.precision - 1);
		for (uint32_t y = 0; y < h; y++) {
			const uint8_t *row = cur + (size_t)y * w * c;
			for (uint32_t ox = 0; ox < nw; ox++) {
				const int16_t *k = ax.k + (size_t)ox * ax.window;
				int32_t ss[4] = {init, init, init, init};
				for (int i = 0; i < ax.size[ox]; i++) {
					const uint8_t *p = row + (size_t)(ax.start[ox] + i) * c;
					for (uint32_t ch = 0; ch < c; ch++)
						ss[ch] += (int32_t)p[ch] * (int32_t)k[i];
				}
				for (uint32_t ch = 0; ch < c; ch++)
					out[((size_t)y * nw + ox) * c + ch] = fir_clip8(ss[ch], ax.precision);
			}
		}
		fir_axis_free(&ax);
		cur = out;
		cur_w = nw;
	}
	if (need_v) {
		fir_axis ay;
		fir_axis_build(&ay, h, nh, filter, upscale);
		int32_t init = 1 << (ay.precision - 1);
		for (uint32_t oy = 0; oy < nh; oy++) {
			const int16_t *k = ay.k + (size_t)oy * ay.window;
			for (uint32_t xb = 0; xb < cur_w * c; xb++) {
				int32_t ss = init;
				for (int i = 0; i < ay.size[oy]; i++)
					ss += (int32_t)cur[(size_t)(ay.start[oy] + i) * cur_w * c + xb] * (int32_t)k[i];
				dst[(size_t)oy * cur_w * c + xb] = fir_clip8(ss, ay.precision);
			}
		}
		fir_axis_free(&ay);
	}
	if (c == 4) {
		for (size_t i = 0; i < (size_t)nw * nh; i++) {
			uint8_t *p = dst + i * 4;
			uint32_t rc = fir_recip_alpha(p[3]);
			p[0] = fir_div_and_clip(p[0], rc);
			p[1] = fir_div_and_clip(p[1], rc);
			p[2] = fir_div_and_clip(p[2], rc);
		}
	}
	free(tmp);
	free(work);
	return 0;
}

/* ------------------------------------------------------------------------ */
/* whole image: from_image + shrink_by | shrink_directionally               */
/* ------------------------------------------------------------------------ */

typedef struct {
	const uint8_t *pixels;
	uint32_t width, height, channels, pitch, bw, bh, mode, filter;
	float factor;
	float *block_value;
	uint32_t *out_w, *out_h;
	uint8_t *out_pixels;
	uint32_t cols, rows, row_begin, row_end;
	int err;
} shrink_job;

static void *shrink_rows(void *arg)
{
	shrink_job *j = (shrink_job *)arg;
	size_t slot = (size_t)j->bw * j->bh * j->channels;
	for (uint32_t ty = j->row_begin; ty < j->row_end; ty++) {
		for (uint32_t tx = 0; tx < j->cols; tx++) {
			uint32_t t = ty * j->cols + tx, x, y, w, h;
			orc_tile_rect(j->width, j->height, j->bw, j->bh, t, &x, &y, &w, &h);
			const uint8_t *tile = j->pixels + (size_t)y * j->pitch + (size_t)x * j->channels;
			float v0, v1;
			if (j->mode == ORC_MODE_SHRINK_DIRECTIONALLY) {
				if (w < 2 || h < 2) { /* reference underflows usize and panics */
					j->err = -2;
					return NULL;
				}
				float hz, vr;
				orc_lod_directional(tile, w, h, j->channels, j->pitch, &hz, &vr, NULL, NULL);
				v0 = hz * j->factor; /* pixlzr.rs:199 */
				v1 = vr * j->factor;
			} else {
				v0 = v1 = orc_lod_oklab(tile, w, h, j->channels, j->pitch, j->factor); /* pixlzr.rs:177-178 */
			}
			uint32_t nw, nh;
			float stored;
			orc_reduce_dims(v0, v1, w, h, &nw, &nh, &stored);
			j->block_value[t] = stored;
			j->out_w[t] = nw;
			j->out_h[t] = nh;
			if (j->out_pixels) {
				if (orc_resize(tile, w, h, j->channels, j->pitch, j->out_pixels + slot * t, nw, nh, j->filter) != 0) {
					j->err = -3;
					return NULL;
				}
			}
		}
	}
	return NULL;
}

int orc_shrink_image(const uint8_t *pixels, uint32_t width, uint32_t height, uint32_t channels,
                     uint32_t pitch, uint32_t bw, uint32_t bh, uint32_t mode, uint32_t filter,
                     float factor, float *block_value, uint32_t *out_w, uint32_t *out_h,
                     uint8_t *out_pixels, int nthreads)
{
	if (!pixels || (channels != 3 && channels != 4) || bw == 0 || bh == 0 || width == 0 || height == 0)
		return -1;
	uint32_t cols, rows;
	orc_grid(width, height, bw, bh, &cols, &rows);
	if (nthreads < 1)
		nthreads = 1;
	if ((uint32_t)nthreads > rows)
		nthreads = (int)rows;
	shrink_job *jobs = (shrink_job *)calloc((size_t)nthreads, sizeof(shrink_job));
	pthread_t *th = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
	for (int i = 0; i < nthreads; i++) {
		shrink_job *j = &jobs[i];
		j->pixels = pixels;
		j->width = width;
		j->height = height;
		j->channels = channels;
		j->pitch = pitch;
		j->bw = bw;
		j->bh = bh;
		j->mode = mode;
		j->filter = filter;
		j->factor = factor;
		j->block_value = block_value;
		j->out_w = out_w;
		j->out_h = out_h;
		j->out_pixels = out_pixels;
		j->cols = cols;
		j->rows = rows;
		j->row_begin = (uint32_t)((uint64_t)rows * i / nthreads);
		j->row_end = (uint32_t)((uint64_t)rows * (i + 1) / nthreads);
	}
	if (nthreads == 1) {
		shrink_rows(&jobs[0]);
	} else {
		for (int i = 0; i < nthreads; i++)
			pthread_create(&th[i], NULL, shrink_rows, &jobs[i]);
		for (int i = 0; i < nthreads; i++)
			pthread_join(th[i], NULL);
	}
	int err = 0;
	for (int i = 0; i < nthreads; i++)
		if (jobs[i].err)
			err = jobs[i].err;
	free(jobs);
	free(th);
	return err;
}

/* ------------------------------------------------------------------------ */
/* synthetic frames                                          */
/* ------------------------------------------------------------------------ */

void orc_synth_frame(uint8_t *pixels, uint32_t width, uint32_t height, uint32_t channels,
                     uint32_t pitch, uint32_t frame_index, uint32_t dist)
{
	static const uint32_t AMP[8] = {0, 0, 1, 2, 4, 16, 64, 255};
	uint32_t seed = 0x5049584Cu + frame_index;
	for (uint32_t y = 0; y < height; y++) {
		uint8_t *row = pixels + (size_t)y * pitch;
		for (uint32_t x = 0; x < width; x++) {
			uint32_t amp = AMP[((x >> 5) * 7u + (y >> 5) * 13u + seed) & 7u];
			if (dist == 2)
				amp = 0;
			if (dist == 3)
				amp = 255;
			uint32_t idx = (y * width + x) * 4u;
			for (uint32_t c = 0; c < 3; c++) {
				int base = (int)(((3u * x + 5u * y + 85u * c) >> 3) & 255u);
				int n = (int)(fmix32((idx + c) ^ seed) % (amp + 1u));
				int v = base + n - (int)(amp / 2u);
				row[x * channels + c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
			}
			if (channels == 4)
				row[x * 4 + 3] = dist == 1 ? (uint8_t)(128u + fmix32((idx + 3u) ^ seed) % 128u) : 255u;
		}
	}
}

/* ------------------------------------------------------------------------ */
/* decode side: Pixlzr::expand (pixlzr.rs:77-122) + to_image (pixlzr_image.rs:24-74)              */
/* every tile is resized back to its full size (edge tiles: the trailing size) with             */
/* PixlzrBlock::resize (block.rs:273-334: upscale flag -> mod.rs:65-107) and copied to its place */
/* ------------------------------------------------------------------------ */
int orc_expand_image(uint32_t width, uint32_t height, uint32_t bw, uint32_t bh, uint32_t channels, uint32_t filter,
                     const uint32_t *tile_w, const uint32_t *tile_h, const uint8_t *slots, size_t slot_stride,
                     uint8_t *out_pixels, uint32_t out_pitch)
{
	if (!tile_w || !tile_h || !slots || !out_pixels || (channels != 3 && channels != 4) || bw == 0 || bh == 0)
		return -1;
	uint32_t cols, rows;
	orc_grid(width, height, bw, bh, &cols, &rows);
	uint8_t *full = (uint8_t *)malloc((size_t)bw * bh * channels);
	if (!full)
		return -2;
	for (uint32_t t = 0; t < cols * rows; t++) {
		uint32_t x0, y0, fw, fh;
		orc_tile_rect(width, height, bw, bh, t, &x0, &y0, &fw, &fh);
		if (tile_w[t] == 0 || tile_h[t] == 0 || tile_w[t] > fw || tile_h[t] > fh) {
			free(full);
			return -3;
		}
		const uint8_t *src = slots + slot_stride * t;
		if (orc_resize(src, tile_w[t], tile_h[t], channels, tile_w[t] * channels, full, fw, fh, filter) != 0) {
			free(full);
			return -4;
		}
		for (uint32_t y = 0; y < fh; y++)
			memcpy(out_pixels + (size_t)(y0 + y) * out_pitch + (size_t)x0 * channels, full + (size_t)y * fw * channels,
			       (size_t)fw * channels);
	}
	free(full);
	return 0;
}

/* ------------------------------------------------------------------------ */
/* legacy image -> image filter: process() (process/mod.rs:107-121) = process_custom (:71-102) with         */
/* |x - avg|, identity, Lanczos3 down, Nearest up; generalised to any down/up filter (process_custom's own  */
/* parameters).  Output is RGBA8 (DynamicImage::new_rgba8, :80-81; RGB tiles gain alpha 255 in copy_from).  */
/* ------------------------------------------------------------------------ */
int orc_process_image(const uint8_t *pixels, uint32_t width, uint32_t height, uint32_t channels, uint32_t pitch,
                      uint32_t bw, uint32_t bh, uint32_t filter_down, uint32_t filter_up, uint8_t *out_rgba,
                      uint32_t out_pitch)
{
	if (!pixels || !out_rgba || (channels != 3 && channels != 4) || bw == 0 || bh == 0)
		return -1;
	uint32_t cols, rows;
	orc_grid(width, height, bw, bh, &cols, &rows);
	uint8_t *small = (uint8_t *)malloc((size_t)bw * bh * channels);
	uint8_t *full = (uint8_t *)malloc((size_t)bw * bh * channels);
	if (!small || !full) {
		free(small);
		free(full);
		return -2;
	}
	int err = 0;
	for (uint32_t t = 0; t < cols * rows && !err; t++) {
		uint32_t x0, y0, w, h;
		orc_tile_rect(width, height, bw, bh, t, &x0, &y0, &w, &h);
		const uint8_t *tile = pixels + (size_t)y0 * pitch + (size_t)x0 * channels;
		float value = lod_oklab_scaled(tile, w, h, channels, pitch, 1.0f, 1.0f);
		uint32_t nw, nh;
		float stored;
		orc_reduce_dims(value, value, w, h, &nw, &nh, &stored);
		if (orc_resize(tile, w, h, channels, pitch, small, nw, nh, filter_down) != 0 ||
		    orc_resize(small, nw, nh, channels, nw * channels, full, w, h, filter_up) != 0) {
			err = -3;
			break;
		}
		for (uint32_t y = 0; y < h; y++) {
			uint8_t *d = out_rgba + (size_t)(y0 + y) * out_pitch + (size_t)x0 * 4;
			const uint8_t *sp = full + (size_t)y * w * channels;
			for (uint32_t x = 0; x < w; x++) {
				d[4 * x + 0] = sp[channels * x + 0];
				d[4 * x + 1] = sp[channels * x + 1];
				d[4 * x + 2] = sp[channels * x + 2];
				d[4 * x + 3] = channels == 4 ? sp[4 * x + 3] : 255;
			}
		}
	}
	free(small);
	free(full);
	return err;
}


/* tree::process_custom, src/process/tree.rs:23-83 (TEST INFRASTRUCTURE, as everything here).  `px` is the (sub-)image the
 * call receives, `out` the RGBA image it returns (same size), both addressed from their own origin.
 *   :32-36  min sizes are at least 4; a block no larger than the minimum on either axis returns image.clone()
 *           (written here as RGBA: an RGB image gains alpha 255, which is what the caller's copy_from at :79 makes of it)
 *   :37-38  is_positive from the sign of the threshold; the recursion passes |threshold| on, so only the outermost call
 *           can be "inverted"
 *   :47-58  per tile of the (bw, bh) grid: get_block_variance with |x - avg| and the identity
 *   :60-69  (value >= threshold) ^ is_positive -> the tile is pixelised exactly as process() does it
 *   :70-79  else -> the same function on the tile with halved block sizes */
static int tree_rec(const uint8_t *px, uint32_t w, uint32_t h, uint32_t c, uint32_t pitch, uint32_t bw, uint32_t bh,
                    uint32_t min_w, uint32_t min_h, float threshold, uint32_t filter_down, uint32_t filter_up,
                    uint8_t *out, uint32_t out_pitch, uint8_t *small, uint8_t *full)
{
	const uint32_t mbw = min_w > 4 ? min_w : 4, mbh = min_h > 4 ? min_h : 4;
	if (bw <= mbw || bh <= mbh) {
		for (uint32_t y = 0; y < h; y++)
			for (uint32_t x = 0; x < w; x++) {
				const uint8_t *sp = px + (size_t)y * pitch + (size_t)x * c;
				uint8_t *d = out + (size_t)y * out_pitch + (size_t)x * 4;
				d[0] = sp[0];
				d[1] = sp[1];
				d[2] = sp[2];
				d[3] = c == 4 ? sp[3] : 255;
			}
		return 0;
	}
	const int is_positive = threshold >= 0.0f;
	const float thr = fabsf(threshold);
	uint32_t cols, rows;
	orc_grid(w, h, bw, bh, &cols, &rows);
	for (uint32_t t = 0; t < cols * rows; t++) {
		uint32_t x0, y0, tw, th;
		orc_tile_rect(w, h, bw, bh, t, &x0, &y0, &tw, &th);
		const uint8_t *tile = px + (size_t)y0 * pitch + (size_t)x0 * c;
		uint8_t *dst = out + (size_t)y0 * out_pitch + (size_t)x0 * 4;
		const float value = lod_oklab_scaled(tile, tw, th, c, pitch, 1.0f, 1.0f);
		if ((value >= thr) ^ is_positive) {
			uint32_t nw, nh;
			float stored;
			orc_reduce_dims(value, value, tw, th, &nw, &nh, &stored);
			if (orc_resize(tile, tw, th, c, pitch, small, nw, nh, filter_down) != 0 ||
			    orc_resize(small, nw, nh, c, nw * c, full, tw, th, filter_up) != 0)
				return -3;
			for (uint32_t y = 0; y < th; y++)
				for (uint32_t x = 0; x < tw; x++) {
					const uint8_t *sp = full + ((size_t)y * tw + x) * c;
					uint8_t *d = dst + (size_t)y * out_pitch + (size_t)x * 4;
					d[0] = sp[0];
					d[1] = sp[1];
					d[2] = sp[2];
					d[3] = c == 4 ? sp[3] : 255;
				}
		} else {
			const int rc = tree_rec(tile, tw, th, c, pitch, bw >> 1, bh >> 1, mbw, mbh, thr, filter_down, filter_up, dst,
			                        out_pitch, small, full);
			if (rc)
				return rc;
		}
	}
	return 0;
}

int orc_tree_process_image(const uint8_t *pixels, uint32_t width, uint32_t height, uint32_t channels, uint32_t pitch,
                           uint32_t bw, uint32_t bh, uint32_t min_bw, uint32_t min_bh, float threshold,
                           uint32_t filter_down, uint32_t filter_up, uint8_t *out_rgba, uint32_t out_pitch)
{
	if (!pixels || !out_rgba || (channels != 3 && channels != 4) || bw == 0 || bh == 0)
		return -1;
	uint8_t *small = (uint8_t *)malloc((size_t)bw * bh * channels);
	uint8_t *full = (uint8_t *)malloc((size_t)bw * bh * channels);
	if (!small || !full) {
		free(small);
		free(full);
		return -2;
	}
	const int rc = tree_rec(pixels, width, height, channels, pitch, bw, bh, min_bw, min_bh, threshold, filter_down, filter_up,
	                        out_rgba, out_pitch, small, full);
	free(small);
	free(full);
	return rc;
}
