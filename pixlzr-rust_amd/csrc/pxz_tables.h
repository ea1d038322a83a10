#pragma once
#include <stdint.h>

#include <vector>

namespace pxz {

struct AxisWindows {
	uint32_t in_size = 0, out_size = 0;
	int window = 0, precision = 0;
	std::vector<int32_t> starts, sizes;
	std::vector<int16_t> coeffs;  // out_size * window (empty for Nearest)
};

// filter: FilterType repr(u8) 0..4.  false for an unknown filter.  upscale: the flag PixlzrBlock::resize
// passes to to_fir_resizing_algorithm (block.rs:301-304) -- it only changes the kernel of Triangle.
bool build_axis(uint32_t in_size, uint32_t out_size, uint32_t filter, AxisWindows *out, bool upscale = false);

// thresholds[k] = smallest positive float v with round(log2f(v)) >= -k
bool build_level_thresholds(float *thresholds, int count);

}  // namespace pxz
