// pixlzr.hpp — C++ host-side mirror of the reference's encode operator surface (the reference is Rust;
// this image has no Rust toolchain, so the host side above the C ABI is C++).  Same names, argument
// meaning and error behaviour as the reference:
//   Pixlzr::from_image            src/data_types/pixlzr_image.rs:6-22
//   Pixlzr::shrink_by             src/data_types/pixlzr.rs:155-185   (runs on the MI355X)
//   Pixlzr::shrink_directionally  src/data_types/pixlzr.rs:187-205   (runs on the MI355X)
//   Pixlzr::encode_to_vec / save  src/encoding/mod.rs:40-89, src/io.rs:88-95
//   Pixlzr::decode_from_vec / open  src/encoding/mod.rs:95-165, src/io.rs:80-87   (runs on the MI355X)
//   Pixlzr::expand / to_image     src/data_types/pixlzr.rs:77-122, pixlzr_image.rs:24-74 (runs on the MI355X)
//   PixlzrBlock                   src/data_types/block.rs:56-230
// Errors: the reference panics inside this path (unwrap / usize underflow); here they are
// std::runtime_error carrying the pxz_status text.  There is no CPU fallback.
#pragma once
#include <cstdint>
#include <optional>
#include <string>
#include <utility>
#include <vector>

namespace pixlzr {

// src/data_types/mod.rs:10-30, repr(u8)
enum class FilterType : uint8_t { Nearest = 0, Triangle = 1, CatmullRom = 2, Gaussian = 3, Lanczos3 = 4 };

// Stand-in for `&image::DynamicImage` (ImageRgba8 | ImageRgb8): interleaved 8-bit, pitch-linear.
struct ImageView {
	const uint8_t *data;
	uint32_t width, height;
	uint32_t channels;     // 3 | 4
	uint32_t pitch_bytes;  // >= width*channels
};

// PixlzrBlockRaw (block.rs:76-81) + RawImage (:56-61)
struct PixlzrBlock {
	uint32_t width = 0, height = 0;
	std::optional<float> block_value;
	bool alpha = false;
	std::vector<uint8_t> data;

	std::pair<uint32_t, uint32_t> dimensions() const { return {width, height}; }  // block.rs:196-198
	bool has_alpha() const { return alpha; }                                      // :206-212
	bool block_value_was_calculated() const { return block_value.has_value(); }   // :213-215
	const std::vector<uint8_t> &as_slice() const { return data; }                 // :216-222
	void set_block_value(float v) { block_value = v; }                            // :223-229
	size_t pixel_size() const { return alpha ? 4 : 3; }                           // pixels(): chunks of 3+alpha, :260-271
};

class Pixlzr {
public:
	uint32_t width = 0, height = 0, block_width = 0, block_height = 0;  // pixlzr.rs:17-25
	std::optional<FilterType> filter;
	std::vector<PixlzrBlock> blocks;

	std::pair<uint32_t, uint32_t> dimensions() const { return {width, height}; }
	std::pair<uint32_t, uint32_t> block_dimensions() const { return {block_width, block_height}; }
	uint32_t block_grid_width() const;   // pixlzr.rs:36-39 (ceil in f32)
	uint32_t block_grid_height() const;  // :40-43

	// Splits the image into the regular grid of owned tiles, row-major (split.rs:10-27, iter.rs:28-87).
	static Pixlzr from_image(const ImageView &image, uint32_t block_width, uint32_t block_height);

	// Per tile: Oklab mean-absolute-deviation value -> power-of-two reduction -> resample -> Some(value).
	// Tiles that already carry a value are kept (pixlzr.rs:168-170).  device_id selects the GPU.
	void shrink_by(FilterType filter_downscale, float factor, int device_id = 0);
	// Per tile: directional gradient sums -> per-axis reduction (no skip, pixlzr.rs:192-204).
	void shrink_directionally(FilterType filter_downscale, float factor, int device_id = 0);

	std::vector<uint8_t> encode_to_vec() const;   // encoding/mod.rs:40-89
	void save(const std::string &path) const;     // io.rs:88-95

	// decode side
	static Pixlzr decode_from_vec(const std::vector<uint8_t> &bytes, int device_id = 0);  // encoding/mod.rs:95-165
	static Pixlzr open(const std::string &path, int device_id = 0);                       // io.rs:80-87
	// every tile resized back to its full size (pixlzr.rs:77-122); filter = Some(filter_upscale)
	Pixlzr expand(FilterType filter_upscale, int device_id = 0) const;
	// expand + reassembly (pixlzr_image.rs:24-74): interleaved pixels, RGBA if any tile has alpha, else RGB
	struct Image {
		uint32_t width = 0, height = 0, channels = 0;
		std::vector<uint8_t> data;
	};
	Image to_image(FilterType filter_upscale, int device_id = 0) const;

private:
	void shrink_on_device(uint32_t mode, FilterType f, float factor, int device_id);
};

}  // namespace pixlzr
