// pixlzr_host.cpp — implementation of include/pixlzr.hpp on top of the C ABI (include/pixlzr_hip.h).
#include "../../include/pixlzr.hpp"

#include <cmath>
#include <cstring>
#include <fstream>
#include <iterator>
#include <map>
#include <stdexcept>

#include "../../include/pixlzr_hip.h"

namespace pixlzr {

namespace {

// one handle per (thread, device): concurrent calls on different Pixlzr objects are legal in the
// reference (no shared state), distinct handles are independent here
struct HandleCache {
	std::map<int, pxz_handle *> by_device;
	~HandleCache()
	{
		for (auto &kv : by_device) pxz_destroy(kv.second);
	}
	pxz_handle *get(int device)
	{
		auto it = by_device.find(device);
		if (it != by_device.end()) return it->second;
		pxz_handle *h = nullptr;
		const int rc = pxz_create(device, &h);
		if (rc != PXZ_OK) throw std::runtime_error("pxz_create failed (" + std::to_string(rc) + "): no usable gfx950 device");
		by_device[device] = h;
		return h;
	}
};
thread_local HandleCache g_handles;

}  // namespace

// (pixlzr.rs:36-46 rounds up in f32; pxz_grid is the integer ceiling, equal to it up to 2^24 and refusing larger sides)
static void grid_of(const Pixlzr &p, uint32_t &cols, uint32_t &rows)
{
	if (pxz_grid(p.width, p.height, p.block_width, p.block_height, &cols, &rows) != PXZ_OK)
		throw std::runtime_error("Pixlzr: zero block size or image side above 2^24");
}
uint32_t Pixlzr::block_grid_width() const { uint32_t c, r; grid_of(*this, c, r); return c; }
uint32_t Pixlzr::block_grid_height() const { uint32_t c, r; grid_of(*this, c, r); return r; }

Pixlzr Pixlzr::from_image(const ImageView &image, uint32_t bw, uint32_t bh)
{
	if (!image.data || (image.channels != 3 && image.channels != 4) || bw == 0 || bh == 0)
		throw std::runtime_error("from_image: bad image or block size");
	Pixlzr p;
	p.width = image.width;
	p.height = image.height;
	p.block_width = bw;
	p.block_height = bh;
	uint32_t cols = 0, rows = 0;
	pxz_grid(image.width, image.height, bw, bh, &cols, &rows);  // iter.rs:38-41
	p.blocks.reserve((size_t)cols * rows);
	for (uint32_t ty = 0; ty < rows; ++ty) {
		for (uint32_t tx = 0; tx < cols; ++tx) {
			const uint32_t x = tx * bw, y = ty * bh;
			PixlzrBlock b;
			b.width = bw < image.width - x ? bw : image.width - x;    // split.rs:18
			b.height = bh < image.height - y ? bh : image.height - y;  // :19
			b.alpha = image.channels == 4;
			b.data.resize((size_t)b.width * b.height * image.channels);
			for (uint32_t r = 0; r < b.height; ++r)  // crop_imm: one owned copy per tile (:24)
				std::memcpy(b.data.data() + (size_t)r * b.width * image.channels,
				            image.data + (size_t)(y + r) * image.pitch_bytes + (size_t)x * image.channels,
				            (size_t)b.width * image.channels);
			p.blocks.push_back(std::move(b));
		}
	}
	return p;
}

void Pixlzr::shrink_on_device(uint32_t mode, FilterType f, float factor, int device_id)
{
	if (blocks.empty()) return;
	const uint32_t channels = blocks[0].has_alpha() ? 4u : 3u;
	const uint32_t cols = block_grid_width(), rows = block_grid_height();
	const size_t tiles = (size_t)cols * rows;
	if (tiles != blocks.size()) throw std::runtime_error("shrink: block list does not match the grid");
	// stitch the owned tiles back into one pitch-linear image: ONE device call covers all tiles
	std::vector<uint8_t> image((size_t)width * height * channels);
	for (uint32_t ty = 0; ty < rows; ++ty)
		for (uint32_t tx = 0; tx < cols; ++tx) {
			const PixlzrBlock &b = blocks[(size_t)ty * cols + tx];
			if (b.width == 0 || b.data.size() != (size_t)b.width * b.height * channels) throw std::runtime_error("shrink: malformed block");
			// a block that already went through shrink is smaller than its grid cell: it is kept (below)
			if (b.block_value && mode == PXZ_MODE_SHRINK_BY) continue;
			const uint32_t cw = block_width < width - tx * block_width ? block_width : width - tx * block_width;
			const uint32_t ch = block_height < height - ty * block_height ? block_height : height - ty * block_height;
			if (b.width != cw || b.height != ch) throw std::runtime_error("shrink: block is not full-size (already shrunk?)");
			for (uint32_t r = 0; r < b.height; ++r)
				std::memcpy(image.data() + ((size_t)(ty * block_height + r) * width + (size_t)tx * block_width) * channels,
				            b.data.data() + (size_t)r * b.width * channels, (size_t)b.width * channels);
		}
	std::vector<float> value(tiles);
	std::vector<uint32_t> ow(tiles), oh(tiles);
	pxz_handle *h = g_handles.get(device_id);
	// the tiles' pixels come back as one tightly packed stream: only what the blocks keep crosses PCIe
	uint64_t packed_len = 0;
	int rc = pxz_shrink_image_packed(h, image.data(), width, height, channels, width * channels, block_width, block_height,
	                                 mode, (uint32_t)f, factor, value.data(), ow.data(), oh.data(), &packed_len);
	if (rc != PXZ_OK) throw std::runtime_error(std::string("pxz_shrink_image_packed: ") + pxz_last_error(h));
	std::vector<uint8_t> px(packed_len);
	rc = pxz_fetch_packed(h, px.data(), packed_len);
	if (rc != PXZ_OK) throw std::runtime_error(std::string("pxz_fetch_packed: ") + pxz_last_error(h));
	size_t at = 0;
	for (size_t t = 0; t < tiles; ++t) {
		const size_t n = (size_t)ow[t] * oh[t] * channels;
		PixlzrBlock &b = blocks[t];
		if (!(mode == PXZ_MODE_SHRINK_BY && b.block_value)) {  // pixlzr.rs:168-170: a block with a value is kept
			b.width = ow[t];
			b.height = oh[t];
			b.block_value = value[t];  // operations.rs:154
			b.data.assign(px.begin() + at, px.begin() + at + n);
		}
		at += n;
	}
}

void Pixlzr::shrink_by(FilterType f, float factor, int device_id) { shrink_on_device(PXZ_MODE_SHRINK_BY, f, factor, device_id); }
void Pixlzr::shrink_directionally(FilterType f, float factor, int device_id)
{
	shrink_on_device(PXZ_MODE_SHRINK_DIRECTIONALLY, f, factor, device_id);
}

std::vector<uint8_t> Pixlzr::encode_to_vec() const
{
	const uint32_t channels = !blocks.empty() && blocks[0].has_alpha() ? 4u : 3u;
	const size_t tiles = blocks.size();
	const size_t slot = (size_t)block_width * block_height * channels;
	std::vector<float> value(tiles);
	std::vector<uint8_t> has(tiles);
	std::vector<uint32_t> tw(tiles), th(tiles);
	std::vector<uint8_t> slots(tiles * slot);
	for (size_t t = 0; t < tiles; ++t) {
		const PixlzrBlock &b = blocks[t];
		value[t] = b.block_value.value_or(0.0f);  // mod.rs:173-178
		has[t] = b.block_value ? 1 : 0;
		tw[t] = b.width;
		th[t] = b.height;
		if (b.data.size() > slot) throw std::runtime_error("encode_to_vec: block larger than its grid cell");
		std::memcpy(slots.data() + t * slot, b.data.data(), b.data.size());
	}
	const uint32_t fb = (uint32_t)filter.value_or(FilterType::Nearest);  // unwrap_or_default, mod.rs:53
	const int64_t bound = pxz_encode_container(width, height, block_width, block_height, channels, fb, value.data(), has.data(),
	                                           tw.data(), th.data(), nullptr, nullptr, 0);
	if (bound < 0) throw std::runtime_error("pxz_encode_container(bound) failed: " + std::to_string(bound));
	std::vector<uint8_t> out((size_t)bound);
	const int64_t n = pxz_encode_container(width, height, block_width, block_height, channels, fb, value.data(), has.data(),
	                                       tw.data(), th.data(), slots.data(), out.data(), out.size());
	if (n < 0) throw std::runtime_error("pxz_encode_container failed: " + std::to_string(n));
	out.resize((size_t)n);
	return out;
}

Pixlzr Pixlzr::decode_from_vec(const std::vector<uint8_t> &bytes, int device_id)
{
	pxz_handle *h = g_handles.get(device_id);
	uint32_t w, hh, bw, bh, c, fb;
	int rc = pxz_decode_file(h, bytes.data(), bytes.size(), &w, &hh, &bw, &bh, &c, &fb, nullptr, nullptr, nullptr, nullptr);
	if (rc != PXZ_OK) throw std::runtime_error(std::string("decode_from_vec: ") + pxz_last_error(h));
	uint32_t cols = 0, rows = 0;
	pxz_grid(w, hh, bw, bh, &cols, &rows);
	const size_t tiles = (size_t)cols * rows, slot = (size_t)bw * bh * c;
	std::vector<float> value(tiles);
	std::vector<uint32_t> tw(tiles), th(tiles);
	std::vector<uint8_t> slots(tiles * slot);
	rc = pxz_decode_file(h, bytes.data(), bytes.size(), &w, &hh, &bw, &bh, &c, &fb, value.data(), tw.data(), th.data(), slots.data());
	if (rc != PXZ_OK) throw std::runtime_error(std::string("decode_from_vec: ") + pxz_last_error(h));
	Pixlzr p;
	p.width = w;
	p.height = hh;
	p.block_width = bw;
	p.block_height = bh;
	p.filter = (FilterType)(fb <= 4 ? fb : 0);  // From<u8> for FilterType, mod.rs:110-121
	p.blocks.resize(tiles);
	for (size_t t = 0; t < tiles; ++t) {
		PixlzrBlock &b = p.blocks[t];
		b.width = tw[t];
		b.height = th[t];
		b.block_value = value[t];  // decode_block always yields Some(value), mod.rs:210-215
		b.alpha = c == 4;
		b.data.assign(slots.begin() + t * slot, slots.begin() + t * slot + (size_t)tw[t] * th[t] * c);
	}
	return p;
}

Pixlzr Pixlzr::open(const std::string &path, int device_id)
{
	std::ifstream f(path, std::ios::binary);
	if (!f) throw std::runtime_error("open: cannot open " + path);
	std::vector<uint8_t> bytes((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
	return decode_from_vec(bytes, device_id);
}

Pixlzr::Image Pixlzr::to_image(FilterType f, int device_id) const
{
	Image img;
	if (blocks.empty()) return img;
	bool any_alpha = false;
	for (const PixlzrBlock &b : blocks) any_alpha = any_alpha || b.has_alpha();  // pixlzr_image.rs:28-33
	const uint32_t channels = any_alpha ? 4u : 3u;
	const size_t tiles = blocks.size(), slot = (size_t)block_width * block_height * channels;
	if (tiles != (size_t)block_grid_width() * block_grid_height()) throw std::runtime_error("to_image: block list does not match the grid");
	std::vector<uint32_t> tw(tiles), th(tiles);
	std::vector<uint8_t> slots(tiles * slot);
	for (size_t t = 0; t < tiles; ++t) {
		const PixlzrBlock &b = blocks[t];
		tw[t] = b.width;
		th[t] = b.height;
		const size_t n = (size_t)b.width * b.height;
		if (n * channels > slot || b.data.size() != n * b.pixel_size()) throw std::runtime_error("to_image: malformed block");
		uint8_t *d = slots.data() + t * slot;
		if (b.pixel_size() == channels) {
			std::memcpy(d, b.data.data(), b.data.size());
		} else {  // an RGB tile in an RGBA image: alpha 255 (copy_from's pixel conversion)
			for (size_t i = 0; i < n; ++i) {
				d[4 * i] = b.data[3 * i];
				d[4 * i + 1] = b.data[3 * i + 1];
				d[4 * i + 2] = b.data[3 * i + 2];
				d[4 * i + 3] = 255;
			}
		}
	}
	img.width = width;
	img.height = height;
	img.channels = channels;
	img.data.resize((size_t)width * height * channels);
	pxz_handle *h = g_handles.get(device_id);
	const int rc = pxz_expand_image(h, width, height, channels, width * channels, block_width, block_height, (uint32_t)f, tw.data(),
	                                th.data(), slots.data(), img.data.data());
	if (rc != PXZ_OK) throw std::runtime_error(std::string("pxz_expand_image: ") + pxz_last_error(h));
	return img;
}

Pixlzr Pixlzr::expand(FilterType f, int device_id) const
{
	// expand = the tiles of to_image before reassembly (pixlzr.rs:77-122): cut them back out of the image
	const Image img = to_image(f, device_id);
	ImageView view{img.data.data(), img.width, img.height, img.channels, img.width * img.channels};
	Pixlzr out = from_image(view, block_width, block_height);
	out.filter = f;
	return out;
}

void Pixlzr::save(const std::string &path) const
{
	const std::vector<uint8_t> bytes = encode_to_vec();
	std::ofstream f(path, std::ios::binary);
	if (!f) throw std::runtime_error("save: cannot open " + path);
	f.write(reinterpret_cast<const char *>(bytes.data()), (std::streamsize)bytes.size());
}

}  // namespace pixlzr
