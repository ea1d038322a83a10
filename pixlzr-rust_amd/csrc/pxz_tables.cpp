// pxz_tables.cpp — host-side tables for the shrink kernels.
//
//  * down-scaling windows: what `PixlzrBlock::resize` (reference
//    src/data_types/block.rs:273-334) gets from fast_image_resize 4.2.1 for
//    ResizeAlg::Convolution(filter) / ResizeAlg::Nearest with default
//    ResizeOptions, per (source size, target size) along one axis.  The crate
//    is Pillow-lineage: f64 window weights normalised to 1, converted to i16
//    at the largest precision that keeps the biggest weight below 2^15.
//  * level thresholds: `value.log2().round().min(0).exp2()` (reference
//    src/operations.rs:147-148) only depends on n = round(log2f(v)); the
//    kernel compares v against the smallest float reaching each n, found here
//    with the platform's own log2f (the one Rust's f32::log2 calls).
#include "pxz_tables.h"
#include "pxz_internal.h"
#include <cstdlib>

#include <cmath>
#include <cstring>

namespace pxz {

namespace {

constexpr double kPi = 3.14159265358979323846;

struct Kernel1D {
	double support;
	double (*eval)(double);
};

double hamming(double x)
{
	x = std::fabs(x);
	if (x == 0.0) return 1.0;
	if (x >= 1.0) return 0.0;
	x *= kPi;
	return (0.54 + 0.46 * std::cos(x)) * std::sin(x) / x;
}

double catmull_rom(double x)
{
	constexpr double a = -0.5;
	x = std::fabs(x);
	if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1.0;
	if (x < 2.0) return (((x - 5.0) * x + 8.0) * x - 4.0) * a;
	return 0.0;
}

double gaussian(double x)
{
	if (x <= -3.0 || x >= 3.0) return 0.0;
	return std::exp(-(x * x) / 0.5) / std::sqrt(2.0 * kPi * 0.25);
}

double sinc(double x)
{
	if (x == 0.0) return 1.0;
	x *= kPi;
	return std::sin(x) / x;
}

double lanczos3(double x)
{
	return (x >= -3.0 && x < 3.0) ? sinc(x) * sinc(x / 3.0) : 0.0;
}

double bilinear(double x)
{
	x = std::fabs(x);
	return x < 1.0 ? 1.0 - x : 0.0;
}

// FilterType -> fir filter (reference src/data_types/mod.rs:65-107): the down-scale branch maps Triangle to
// Convolution(Hamming), the up-scale branch (SuperSampling(filter, 2): a plain convolution when nothing
// shrinks) maps it to Bilinear; the other three keep their kernel
bool kernel_for(uint32_t filter, bool upscale, Kernel1D *k)
{
	switch (filter) {
	case 1:
		if (upscale) *k = {1.0, bilinear};
		else *k = {1.0, hamming};
		return true;
	case 2: *k = {2.0, catmull_rom}; return true;  // CatmullRom
	case 3: *k = {3.0, gaussian}; return true;     // Gaussian
	case 4: *k = {3.0, lanczos3}; return true;     // Lanczos3
	default: return false;
	}
}

}  // namespace

bool build_axis(uint32_t in_size, uint32_t out_size, uint32_t filter, AxisWindows *out, bool upscale)
{
	out->in_size = in_size;
	out->out_size = out_size;
	out->starts.assign(out_size, 0);
	out->sizes.assign(out_size, 0);
	if (filter == 0) {  // ResizeAlg::Nearest: source index per output index
		out->window = 1;
		out->precision = 0;
		out->coeffs.clear();
		const double scale = static_cast<double>(in_size) / static_cast<double>(out_size);
		const double first = scale * 0.5;
		for (uint32_t o = 0; o < out_size; ++o) {
			uint32_t s = static_cast<uint32_t>(first + scale * static_cast<double>(o));
			out->starts[o] = static_cast<int32_t>(s < in_size ? s : in_size - 1);
			out->sizes[o] = 1;
		}
		return true;
	}
	Kernel1D k;
	if (!kernel_for(filter, upscale, &k)) return false;

	const double scale = static_cast<double>(in_size) / static_cast<double>(out_size);
	const double stretch = scale > 1.0 ? scale : 1.0;
	const double radius = k.support * stretch;
	const double inv_stretch = 1.0 / stretch;
	const int window = static_cast<int>(std::ceil(radius)) * 2 + 1;
	out->window = window;

	std::vector<double> weights(static_cast<size_t>(out_size) * window, 0.0);
	double biggest = 0.0;
	for (uint32_t o = 0; o < out_size; ++o) {
		const double centre = (static_cast<double>(o) + 0.5) * scale;
		const double lo = std::floor(centre - radius);
		const double hi = std::ceil(centre + radius);
		const int first = lo < 0.0 ? 0 : static_cast<int>(lo);
		const int last = hi > static_cast<double>(in_size) ? static_cast<int>(in_size) : static_cast<int>(hi);
		const int n = last - first;
		double *w = &weights[static_cast<size_t>(o) * window];
		const double shifted = centre - 0.5;
		double total = 0.0;
		for (int i = 0; i < n; ++i) {
			w[i] = k.eval((static_cast<double>(first + i) - shifted) * inv_stretch);
			total += w[i];
		}
		if (total != 0.0)
			for (int i = 0; i < n; ++i) w[i] /= total;
		for (int i = 0; i < n; ++i)
			if (w[i] > biggest) biggest = w[i];
		out->starts[o] = first;
		out->sizes[o] = n;
	}

	int precision = 0;
	for (int p = 0; p < 22; ++p) {
		precision = p;
		const int next = static_cast<int>(std::round(biggest * static_cast<double>(1 << (p + 1))));
		if (next >= (1 << 15)) break;
	}
	out->precision = precision;
	out->coeffs.assign(weights.size(), 0);
	const double unit = static_cast<double>(1 << precision);
	for (size_t i = 0; i < weights.size(); ++i) {
		double v = std::round(weights[i] * unit);
		if (v > 32767.0) v = 32767.0;
		if (v < -32768.0) v = -32768.0;
		out->coeffs[i] = static_cast<int16_t>(v);
	}
	return true;
}

static inline float from_bits(uint32_t b)
{
	float f;
	std::memcpy(&f, &b, 4);
	return f;
}
static inline uint32_t to_bits(float f)
{
	uint32_t b;
	std::memcpy(&b, &f, 4);
	return b;
}

bool build_level_thresholds(float *thresholds, int count)
{
	for (int k = 0; k < count; ++k) {
		auto reaches = [k](float v) { return std::roundf(::log2f(v)) >= static_cast<float>(-k); };
		uint32_t lo = to_bits(std::ldexp(1.0f, -k - 1));  // log2 = -k-1: does not reach
		uint32_t hi = to_bits(std::ldexp(1.0f, -k));      // log2 = -k: reaches
		if (reaches(from_bits(lo)) || !reaches(from_bits(hi))) return false;
		while (hi - lo > 1) {
			const uint32_t mid = lo + (hi - lo) / 2;
			if (reaches(from_bits(mid))) hi = mid; else lo = mid;
		}
		// the decision must be a clean step around the threshold
		for (uint32_t d = 1; d <= 4096; ++d) {
			if (!reaches(from_bits(hi + d - 1)) || reaches(from_bits(hi - d))) return false;
		}
		thresholds[k] = from_bits(hi);
	}
	return true;
}


const Knobs &knobs()
{
	static const Knobs k = [] {
		auto on = [](const char *name) { return getenv(name) != nullptr; };
		Knobs v;
		v.no_alpha_kernel = on("PXZ_NO_ALPHA_KERNEL");
		v.no_oklab_general = on("PXZ_NO_OKLAB_GENERAL");
		v.no_oklab32 = on("PXZ_NO_OKLAB32");
		v.no_oklab_edges = on("PXZ_NO_OKLAB_EDGES");
		v.no_repitch = on("PXZ_NO_REPITCH");
		v.no_widen = on("PXZ_NO_WIDEN");
		v.no_native_rgb = on("PXZ_NO_NATIVE_RGB");
		v.no_narrow = on("PXZ_NO_NARROW");
		v.no_group16 = on("PXZ_NO_GROUP16");
		v.no_clone_ahead = on("PXZ_NO_CLONE_AHEAD");
		v.no_big_tiles = on("PXZ_NO_BIG_TILES");
		v.no_alpha_first = on("PXZ_NO_ALPHA_FIRST");
		v.oklab_v1 = on("PXZ_OKLAB_V1");
		v.tree_rects = on("PXZ_TREE_RECTS");
		v.no_expand_fast32 = on("PXZ_NO_EXPAND_FAST32");
		const char *e = getenv("PXZ_WPB");
		v.wpb = e ? atoi(e) : 0;
		e = getenv("PXZ_CHUNK_LG");
		v.chunk_lg = e ? (atoi(e) & 15) : -1;
		return v;
	}();
	return k;
}

}  // namespace pxz
