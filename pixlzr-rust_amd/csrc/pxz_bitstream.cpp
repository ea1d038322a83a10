// pxz_bitstream.cpp — host-side .pixlzr writer: the replacement for
// `Pixlzr::encode_to_vec` (reference src/encoding/mod.rs:40-89) and
// `encode_block` (:168-200), including the tile payload codec the reference
// delegates to the `qoi` crate 0.4.1 (`qoi::Encoder::new(..).encode_to_vec()`,
// mod.rs:181-189).  Tile rows are encoded in parallel like the reference's
// rayon `par_lines` (one task per row of tiles, order preserved).
#include <cmath>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/pixlzr_hip.h"

namespace {

// reference src/constants.rs
constexpr char kFileMagic[6] = {'P', 'I', 'X', 'L', 'Z', 'R'};
constexpr uint8_t kFileVersion[3] = {0, 0, 2};
constexpr size_t kFileHeader = 6 + 3 + 1 + 4 * 4;  // PIXLZR_HEADER_SIZE = 26
constexpr char kTileMagic[5] = {'b', 'l', 'o', 'c', 'k'};
constexpr size_t kTileHeader = 5 + 4 + 4;  // PIXLZR_BLOCK_HEADER_BASE_SIZE = 13
constexpr size_t kQoiMagic = 4;

struct ByteSink {
	uint8_t *p;
	void u8(uint8_t v) { *p++ = v; }
	void be32(uint32_t v)
	{
		p[0] = (uint8_t)(v >> 24);
		p[1] = (uint8_t)(v >> 16);
		p[2] = (uint8_t)(v >> 8);
		p[3] = (uint8_t)v;
		p += 4;
	}
	void bytes(const void *src, size_t n)
	{
		std::memcpy(p, src, n);
		p += n;
	}
};

// pixel packed as r | g<<8 | b<<16 | a<<24
inline uint32_t load_px(const uint8_t *d, uint32_t channels)
{
	return (uint32_t)d[0] | ((uint32_t)d[1] << 8) | ((uint32_t)d[2] << 16) |
	       (channels == 4 ? (uint32_t)d[3] << 24 : 0xff000000u);
}
inline uint8_t qoi_slot(uint32_t px)
{
	return (uint8_t)(((px & 255u) * 3u + ((px >> 8) & 255u) * 5u + ((px >> 16) & 255u) * 7u + (px >> 24) * 11u) & 63u);
}

// The qoi crate's encoder: the QOI spec's op selection, plus one deviation — a
// pending run of exactly one pixel that ends because the NEXT pixel differs is
// emitted as OP_INDEX of the repeated pixel (whose slot is known to hold it)
// instead of OP_RUN(1), once at least one literal/diff/index op has been written.
size_t qoi_stream(const uint8_t *data, uint32_t w, uint32_t h, uint32_t channels, uint8_t *out)
{
	ByteSink s{out};
	s.bytes("qoif", 4);
	s.be32(w);
	s.be32(h);
	s.u8((uint8_t)channels);
	s.u8(0);  // colourspace byte: ColorSpace::Srgb = 0
	uint32_t table[64] = {0};
	uint32_t last = 0xff000000u;  // (0,0,0,255)
	uint8_t last_slot = qoi_slot(last);
	bool seen_op = false;
	uint32_t pending = 0;
	const size_t total = (size_t)w * h;
	const uint8_t *d = data;
	for (size_t i = 0; i < total; ++i, d += channels) {
		const uint32_t px = load_px(d, channels);
		if (px == last) {
			if (++pending == 62 || i + 1 == total) {
				s.u8((uint8_t)(0xc0u | (pending - 1)));
				pending = 0;
			}
			continue;
		}
		if (pending) {
			s.u8(pending == 1 && seen_op ? (uint8_t)last_slot : (uint8_t)(0xc0u | (pending - 1)));
			pending = 0;
		}
		seen_op = true;
		last_slot = qoi_slot(px);
		if (table[last_slot] == px) {
			s.u8(last_slot);  // OP_INDEX
		} else {
			table[last_slot] = px;
			const uint8_t dr = (uint8_t)((px & 255u) - (last & 255u));
			const uint8_t dg = (uint8_t)(((px >> 8) & 255u) - ((last >> 8) & 255u));
			const uint8_t db = (uint8_t)(((px >> 16) & 255u) - ((last >> 16) & 255u));
			if (channels == 4 && (px >> 24) != (last >> 24)) {
				s.u8(0xff);  // OP_RGBA
				s.u8((uint8_t)px);
				s.u8((uint8_t)(px >> 8));
				s.u8((uint8_t)(px >> 16));
				s.u8((uint8_t)(px >> 24));
			} else if ((uint8_t)(dr + 2) < 4 && (uint8_t)(dg + 2) < 4 && (uint8_t)(db + 2) < 4) {
				s.u8((uint8_t)(0x40u | (((dr + 2) & 3) << 4) | (((dg + 2) & 3) << 2) | ((db + 2) & 3)));  // OP_DIFF
			} else if ((uint8_t)(dg + 32) < 64 && (uint8_t)(dr - dg + 8) < 16 && (uint8_t)(db - dg + 8) < 16) {
				s.u8((uint8_t)(0x80u | (uint8_t)(dg + 32)));  // OP_LUMA
				s.u8((uint8_t)(((uint8_t)(dr - dg + 8) << 4) | (uint8_t)(db - dg + 8)));
			} else {
				s.u8(0xfe);  // OP_RGB
				s.u8((uint8_t)px);
				s.u8((uint8_t)(px >> 8));
				s.u8((uint8_t)(px >> 16));
			}
		}
		last = px;
	}
	static const uint8_t tail[8] = {0, 0, 0, 0, 0, 0, 0, 1};
	s.bytes(tail, 8);
	return (size_t)(s.p - out);
}

}  // namespace

extern "C" {

size_t pxz_qoi_bound(uint32_t w, uint32_t h, uint32_t channels)
{
	return 14 + (size_t)w * h * ((size_t)channels + 1) + 8;
}

int64_t pxz_qoi_encode(const uint8_t *data, uint32_t w, uint32_t h, uint32_t channels, uint8_t *out,
                       size_t out_capacity)
{
	if (!data || !out || (channels != 3 && channels != 4) || w == 0 || h == 0) return PXZ_ERR_INVALID_ARG;
	if (out_capacity < pxz_qoi_bound(w, h, channels)) return PXZ_ERR_BUFFER_TOO_SMALL;
	return (int64_t)qoi_stream(data, w, h, channels, out);
}

int64_t pxz_encode_container(uint32_t width, uint32_t height, uint32_t block_w, uint32_t block_h, uint32_t channels,
                             uint32_t filter_byte, const float *block_value, const uint8_t *has_value,
                             const uint32_t *tile_w, const uint32_t *tile_h, const uint8_t *slots, uint8_t *out,
                             size_t out_capacity)
{
	if (!block_value || !tile_w || !tile_h || (channels != 3 && channels != 4) || block_w == 0 || block_h == 0 ||
	    width == 0 || height == 0)
		return PXZ_ERR_INVALID_ARG;
	// the library's one grid (pxz_grid: integer ceiling = the reference's f32 and f64 forms up to 2^24, refused beyond)
	uint32_t cols, rows;
	if (pxz_grid(width, height, block_w, block_h, &cols, &rows) != PXZ_OK) return PXZ_ERR_UNSUPPORTED;
	const size_t slot = (size_t)block_w * block_h * channels;

	// per-row upper bounds -> each row of tiles is encoded into its own span
	std::vector<size_t> row_bound(rows, 0), row_off(rows + 1, 0);
	for (uint32_t r = 0; r < rows; ++r) {
		for (uint32_t c = 0; c < cols; ++c) {
			const size_t t = (size_t)r * cols + c;
			if (tile_w[t] == 0 || tile_h[t] == 0 || (size_t)tile_w[t] * tile_h[t] * channels > slot) return PXZ_ERR_INVALID_ARG;
			row_bound[r] += kTileHeader + pxz_qoi_bound(tile_w[t], tile_h[t], channels) - kQoiMagic;
		}
		row_off[r + 1] = row_off[r] + row_bound[r];
	}
	const size_t bound = kFileHeader + (size_t)rows * 4 + row_off[rows];
	if (!out) return (int64_t)bound;
	if (!slots) return PXZ_ERR_INVALID_ARG;

	// encode rows in parallel into scratch, then splice (mod.rs:59-87)
	std::vector<uint8_t> scratch(row_off[rows] + 16);
	std::vector<size_t> row_len(rows, 0);
	auto encode_rows = [&](uint32_t r0, uint32_t r1) {
		for (uint32_t r = r0; r < r1; ++r) {
			uint8_t *base = scratch.data() + row_off[r];
			ByteSink s{base};
			for (uint32_t c = 0; c < cols; ++c) {
				const size_t t = (size_t)r * cols + c;
				s.bytes(kTileMagic, 5);
				const float v = (has_value && !has_value[t]) ? 0.0f : block_value[t];  // mod.rs:173-178
				uint32_t vb;
				std::memcpy(&vb, &v, 4);
				s.be32(vb);
				// the stream is produced 4 bytes early so that its "qoif" magic lands where the
				// length goes (mod.rs:191 drops the magic, :195 writes the length)
				const size_t n = qoi_stream(slots + slot * t, tile_w[t], tile_h[t], channels, s.p) - kQoiMagic;
				s.be32((uint32_t)n);
				s.p += n;
			}
			row_len[r] = (size_t)(s.p - base);
		}
	};
	unsigned workers = std::thread::hardware_concurrency();
	if (workers == 0) workers = 1;
	if (workers > rows) workers = rows;
	if (workers > 16) workers = 16;
	if (workers <= 1 || (size_t)cols * rows < 64) {
		encode_rows(0, rows);
	} else {
		std::vector<std::thread> pool;
		for (unsigned i = 0; i < workers; ++i)
			pool.emplace_back(encode_rows, (uint32_t)((uint64_t)rows * i / workers), (uint32_t)((uint64_t)rows * (i + 1) / workers));
		for (auto &th : pool) th.join();
	}

	size_t total = kFileHeader + (size_t)rows * 4;
	for (uint32_t r = 0; r < rows; ++r) total += row_len[r];
	if (total > out_capacity) return PXZ_ERR_BUFFER_TOO_SMALL;

	ByteSink s{out};
	s.bytes(kFileMagic, 6);
	s.bytes(kFileVersion, 3);
	s.u8((uint8_t)filter_byte);  // mod.rs:53
	s.be32(width);
	s.be32(height);
	s.be32(block_w);
	s.be32(block_h);
	for (uint32_t r = 0; r < rows; ++r) s.be32((uint32_t)row_len[r]);  // mod.rs:77-82
	for (uint32_t r = 0; r < rows; ++r) s.bytes(scratch.data() + row_off[r], row_len[r]);
	return (int64_t)(s.p - out);
}

}  // extern "C"
