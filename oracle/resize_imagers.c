/*
 * resize_imagers.c — oracle restatement of image 0.25 `imageops::resize`
 * (vertical pass into f32, then horizontal pass, f32 weights, round at the end).
 *
 * NOT on the product path.  pixlzr's default build resamples with
 * fast_image_resize (oracle: orc_resize).  This variant exists only because the
 * reference fixture Big-Ruscher.pix was produced by the crate's older
 * `image-rs` resize feature (Cargo.toml:47; block.rs:282-290), so its payload
 * pixels can only be checked against this model.
 */
#include "pixlzr_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

static float sincf_(float t)
{
	float a = t * 3.14159265358979323846f;
	return t == 0.0f ? 1.0f : sinf(a) / a;
}
static float k_lanczos3(float x) { return fabsf(x) < 3.0f ? sincf_(x) * sincf_(x / 3.0f) : 0.0f; }
static float k_triangle(float x) { return fabsf(x) < 1.0f ? 1.0f - fabsf(x) : 0.0f; }
static float k_catmullrom(float x)
{
	/* bc_cubic_spline(x, 0, 0.5) */
	float a = fabsf(x), b = 0.0f, c = 0.5f, k;
	if (a < 1.0f)
		k = (12.0f - 9.0f * b - 6.0f * c) * a * a * a + (-18.0f + 12.0f * b + 6.0f * c) * a * a + (6.0f - 2.0f * b);
	else if (a < 2.0f)
		k = (-b - 6.0f * c) * a * a * a + (6.0f * b + 30.0f * c) * a * a + (-12.0f * b - 48.0f * c) * a + (8.0f * b + 24.0f * c);
	else
		k = 0.0f;
	return k / 6.0f;
}
static float k_gaussian(float x)
{
	float r = 0.5f;
	return 1.0f / (sqrtf(2.0f * 3.14159265358979323846f) * r) * expf(-(x * x) / (2.0f * r * r));
}

typedef float (*kern_fn)(float);

static int pick(uint32_t filter, kern_fn *k, float *support)
{
	switch (filter) {
	case ORC_TRIANGLE: *k = k_triangle; *support = 1.0f; return 0;
	case ORC_CATMULLROM: *k = k_catmullrom; *support = 2.0f; return 0;
	case ORC_GAUSSIAN: *k = k_gaussian; *support = 3.0f; return 0;
	case ORC_LANCZOS3: *k = k_lanczos3; *support = 3.0f; return 0;
	default: return -1;
	}
}

static long clampl(long v, long lo, long hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* weights of one output index along an axis of `in` samples */
static int axis_weights(uint32_t in, uint32_t out, uint32_t o, kern_fn k, float support, float *ws, uint32_t *left_out)
{
	float ratio = (float)in / (float)out;
	float sratio = ratio < 1.0f ? 1.0f : ratio;
	float src_support = support * sratio;
	float input = ((float)o + 0.5f) * ratio;
	long left = clampl((long)floorf(input - src_support), 0, (long)in - 1);
	long right = clampl((long)ceilf(input + src_support), left + 1, (long)in);
	input -= 0.5f;
	float sum = 0.0f;
	int n = 0;
	for (long i = left; i < right; i++) {
		float w = k(((float)i - input) / sratio);
		ws[n++] = w;
		sum += w;
	}
	for (int i = 0; i < n; i++)
		ws[i] /= sum;
	*left_out = (uint32_t)left;
	return n;
}

int orc_resize_imagers(const uint8_t *src, uint32_t w, uint32_t h, uint32_t c, uint32_t pitch,
                       uint8_t *dst, uint32_t nw, uint32_t nh, uint32_t filter)
{
	if (nw == w && nh == h) {
		for (uint32_t y = 0; y < h; y++)
			memcpy(dst + (size_t)y * w * c, src + (size_t)y * pitch, (size_t)w * c);
		return 0;
	}
	kern_fn k;
	float support;
	if (pick(filter, &k, &support) != 0)
		return -1;
	float *ws = (float *)malloc(sizeof(float) * ((size_t)(w > h ? w : h) + 2));
	/* vertical_sample -> Rgba32F (w x nh) */
	float *tmp = (float *)malloc(sizeof(float) * (size_t)w * nh * c);
	for (uint32_t oy = 0; oy < nh; oy++) {
		uint32_t left;
		int n = axis_weights(h, nh, oy, k, support, ws, &left);
		for (uint32_t x = 0; x < w; x++) {
			float t[4] = {0, 0, 0, 0};
			for (int i = 0; i < n; i++) {
				const uint8_t *p = src + (size_t)(left + i) * pitch + (size_t)x * c;
				for (uint32_t ch = 0; ch < c; ch++)
					t[ch] += (float)p[ch] * ws[i];
			}
			for (uint32_t ch = 0; ch < c; ch++)
				tmp[((size_t)oy * w + x) * c + ch] = t[ch];
		}
	}
	/* horizontal_sample -> u8, clamp + round-half-away */
	for (uint32_t ox = 0; ox < nw; ox++) {
		uint32_t left;
		int n = axis_weights(w, nw, ox, k, support, ws, &left);
		for (uint32_t y = 0; y < nh; y++) {
			float t[4] = {0, 0, 0, 0};
			for (int i = 0; i < n; i++) {
				const float *p = tmp + ((size_t)y * w + left + i) * c;
				for (uint32_t ch = 0; ch < c; ch++)
					t[ch] += p[ch] * ws[i];
			}
			for (uint32_t ch = 0; ch < c; ch++) {
				float v = t[ch] < 0.0f ? 0.0f : (t[ch] > 255.0f ? 255.0f : t[ch]);
				dst[((size_t)y * nw + ox) * c + ch] = (uint8_t)roundf(v);
			}
		}
	}
	free(tmp);
	free(ws);
	return 0;
}
