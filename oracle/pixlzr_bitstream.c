/*
 * pixlzr_bitstream.c — oracle restatement of the .pixlzr writer/reader and of
 * the `qoi` crate 0.4.1 encoder it calls (test oracle; see pixlzr_oracle.h).
 *
 * Pinned byte-for-byte by the reference fixtures benches/base.png ->
 * benches/base.pixlzr (RGBA, 442 tiles) and Big-Ruscher.pix (RGB, 2040 tiles):
 * tests/test_oracle_golden.py.
 */
#include "pixlzr_oracle.h"

#include <math.h>
#include <string.h>

#define QOI_OP_INDEX 0x00
#define QOI_OP_DIFF 0x40
#define QOI_OP_LUMA 0x80
#define QOI_OP_RUN 0xc0
#define QOI_OP_RGB 0xfe
#define QOI_OP_RGBA 0xff

size_t orc_qoi_bound(uint32_t w, uint32_t h, uint32_t c)
{
	return 14 + (size_t)w * h * (c + 1) + 8;
}

static inline void put_be32(uint8_t *p, uint32_t v)
{
	p[0] = (uint8_t)(v >> 24);
	p[1] = (uint8_t)(v >> 16);
	p[2] = (uint8_t)(v >> 8);
	p[3] = (uint8_t)v;
}
static inline uint32_t get_be32(const uint8_t *p)
{
	return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
}

/* qoi 0.4.1 `encode_impl` (non-"reference" feature): canonical QOI except that
 * a pending run of exactly ONE pixel, flushed because the next pixel differs,
 * is written as QOI_OP_INDEX|hash(prev) once any non-run pixel has been seen
 * (`index_allowed`).  Called from src/encoding/mod.rs:181-189 with channels
 * inferred from len/(w*h) and colourspace byte 0. */
size_t orc_qoi_encode(const uint8_t *data, uint32_t w, uint32_t h, uint32_t c, uint8_t *out)
{
	uint8_t *p = out;
	memcpy(p, "qoif", 4);
	put_be32(p + 4, w);
	put_be32(p + 8, h);
	p[12] = (uint8_t)c;
	p[13] = 0;
	p += 14;
	uint8_t index[64][4];
	memset(index, 0, sizeof index);
	uint8_t prev[4] = {0, 0, 0, 255};
	uint8_t hash_prev = (uint8_t)((prev[0] * 3 + prev[1] * 5 + prev[2] * 7 + prev[3] * 11) % 64);
	uint32_t run = 0;
	int index_allowed = 0;
	size_t n = (size_t)w * h;
	for (size_t i = 0; i < n; i++) {
		uint8_t px[4] = {data[i * c], data[i * c + 1], data[i * c + 2], c == 4 ? data[i * c + 3] : (uint8_t)255};
		if (memcmp(px, prev, 4) == 0) {
			run++;
			if (run == 62 || i == n - 1) {
				*p++ = (uint8_t)(QOI_OP_RUN | (run - 1));
				run = 0;
			}
			continue;
		}
		if (run != 0) {
			*p++ = (run == 1 && index_allowed) ? (uint8_t)(QOI_OP_INDEX | hash_prev)
			                                   : (uint8_t)(QOI_OP_RUN | (run - 1));
			run = 0;
		}
		index_allowed = 1;
		hash_prev = (uint8_t)((px[0] * 3 + px[1] * 5 + px[2] * 7 + px[3] * 11) % 64);
		if (memcmp(index[hash_prev], px, 4) == 0) {
			*p++ = (uint8_t)(QOI_OP_INDEX | hash_prev);
		} else {
			memcpy(index[hash_prev], px, 4);
			if (c == 3 || px[3] == prev[3]) {
				uint8_t vg = (uint8_t)(px[1] - prev[1]);
				uint8_t vg_32 = (uint8_t)(vg + 32);
				if ((vg_32 | 63) == 63) {
					uint8_t vr = (uint8_t)(px[0] - prev[0]);
					uint8_t vb = (uint8_t)(px[2] - prev[2]);
					uint8_t vg_r = (uint8_t)(vr - vg), vg_b = (uint8_t)(vb - vg);
					uint8_t vr_2 = (uint8_t)(vr + 2), vg_2 = (uint8_t)(vg + 2), vb_2 = (uint8_t)(vb + 2);
					if ((vr_2 | vg_2 | vb_2 | 3) == 3) {
						*p++ = (uint8_t)(QOI_OP_DIFF | (vr_2 << 4) | (vg_2 << 2) | vb_2);
					} else {
						uint8_t vg_r_8 = (uint8_t)(vg_r + 8), vg_b_8 = (uint8_t)(vg_b + 8);
						if ((vg_r_8 | vg_b_8 | 15) == 15) {
							*p++ = (uint8_t)(QOI_OP_LUMA | vg_32);
							*p++ = (uint8_t)((vg_r_8 << 4) | vg_b_8);
						} else {
							*p++ = QOI_OP_RGB;
							*p++ = px[0];
							*p++ = px[1];
							*p++ = px[2];
						}
					}
				} else {
					*p++ = QOI_OP_RGB;
					*p++ = px[0];
					*p++ = px[1];
					*p++ = px[2];
				}
			} else {
				*p++ = QOI_OP_RGBA;
				*p++ = px[0];
				*p++ = px[1];
				*p++ = px[2];
				*p++ = px[3];
			}
		}
		memcpy(prev, px, 4);
	}
	static const uint8_t padding[8] = {0, 0, 0, 0, 0, 0, 0, 1};
	memcpy(p, padding, 8);
	p += 8;
	return (size_t)(p - out);
}

/* canonical QOI decoder (qoi 0.4.1 `decode_to_vec`, src/encoding/mod.rs:226);
 * `body` starts right after the 4-byte magic (the form stored in a .pixlzr). */
static int qoi_decode_body(const uint8_t *body, size_t len, uint32_t *w, uint32_t *h, uint32_t *c,
                           uint8_t *out, size_t out_cap)
{
	if (len < 10 + 8)
		return -1;
	*w = get_be32(body);
	*h = get_be32(body + 4);
	*c = body[8];
	if (*c != 3 && *c != 4)
		return -1;
	size_t n = (size_t)*w * *h;
	if (n * *c > out_cap)
		return -2;
	const uint8_t *p = body + 10, *end = body + len - 8;
	uint8_t index[64][4];
	memset(index, 0, sizeof index);
	uint8_t px[4] = {0, 0, 0, 255};
	uint32_t run = 0;
	for (size_t i = 0; i < n; i++) {
		if (run > 0) {
			run--;
		} else if (p < end) {
			uint8_t b1 = *p++;
			int is_run = 0;
			if (b1 == QOI_OP_RGB) {
				px[0] = *p++;
				px[1] = *p++;
				px[2] = *p++;
			} else if (b1 == QOI_OP_RGBA) {
				if (*c == 3) {
					/* qoi 0.4.1 decode_impl_slice<3, RGBA = false> [from memory: the crate's source is not in this environment]: the
					 * RGBA arm is guarded by the channel count, so 0xff falls into the catch-all arm, which fails only with
					 * fewer than 8 bytes left (never in front of the end marker) and otherwise consumes nothing: the
					 * unchanged pixel is stored in the index and written, and the same byte is met again by every pixel
					 * that follows -- the rest of the tile repeats the last pixel, the decode succeeds. */
					p--;
				} else {
					px[0] = *p++;
					px[1] = *p++;
					px[2] = *p++;
					px[3] = *p++;
				}
			} else if ((b1 & 0xc0) == QOI_OP_INDEX) {
				memcpy(px, index[b1], 4);
			} else if ((b1 & 0xc0) == QOI_OP_DIFF) {
				px[0] = (uint8_t)(px[0] + ((b1 >> 4) & 3) - 2);
				px[1] = (uint8_t)(px[1] + ((b1 >> 2) & 3) - 2);
				px[2] = (uint8_t)(px[2] + (b1 & 3) - 2);
			} else if ((b1 & 0xc0) == QOI_OP_LUMA) {
				uint8_t b2 = *p++;
				int vg = (b1 & 0x3f) - 32;
				px[0] = (uint8_t)(px[0] + vg - 8 + ((b2 >> 4) & 0x0f));
				px[1] = (uint8_t)(px[1] + vg);
				px[2] = (uint8_t)(px[2] + vg - 8 + (b2 & 0x0f));
			} else {
				run = b1 & 0x3f;
				is_run = 1;
			}
			/* qoi 0.4.1: the RUN (and INDEX) arms continue with the next op before the index store; only a
			 * stream that opens with a run of the implicit opaque black can tell */
			if (!is_run)
				memcpy(index[(px[0] * 3 + px[1] * 5 + px[2] * 7 + px[3] * 11) % 64], px, 4);
		} else {
			return -3;
		}
		memcpy(out + i * *c, px, *c);
	}
	return 0;
}

int orc_qoi_decode(const uint8_t *in, size_t len, uint32_t *w, uint32_t *h, uint32_t *c,
                   uint8_t *out, size_t out_cap)
{
	if (len < 14 + 8 || memcmp(in, "qoif", 4) != 0)
		return -1;
	return qoi_decode_body(in + 4, len - 4, w, h, c, out, out_cap);
}

/* src/encoding/mod.rs:40-89 + encode_block :168-200 + constants.rs.
 * Layout: "PIXLZR" 00 00 02 filter | w h bw bh (u32 BE) | line_len[rows] (u32 BE)
 * | tiles row-major: "block" f32BE(value) u32BE(len) qoi-minus-magic.
 * Grid here is ceil in f32 (pixlzr.rs:36-46). */
size_t orc_encode_container(uint32_t width, uint32_t height, uint32_t bw, uint32_t bh, uint32_t channels,
                            uint32_t filter_byte, const float *block_value, const uint8_t *has_value,
                            const uint32_t *tw, const uint32_t *th, const uint8_t *slots,
                            uint8_t *out, size_t out_cap)
{
	uint32_t cols = (uint32_t)ceilf((float)width / (float)bw);
	uint32_t rows = (uint32_t)ceilf((float)height / (float)bh);
	size_t ntiles = (size_t)cols * rows;
	size_t slot = (size_t)bw * bh * channels;
	if (!out) {
		size_t bound = 26 + (size_t)rows * 4;
		for (size_t t = 0; t < ntiles; t++)
			bound += 13 + orc_qoi_bound(tw[t], th[t], channels) - 4;
		return bound;
	}
	uint8_t *p = out;
	memcpy(p, "PIXLZR", 6);
	p[6] = 0;
	p[7] = 0;
	p[8] = 2;
	p[9] = (uint8_t)filter_byte;
	put_be32(p + 10, width);
	put_be32(p + 14, height);
	put_be32(p + 18, bw);
	put_be32(p + 22, bh);
	uint8_t *line_table = p + 26;
	p = line_table + (size_t)rows * 4;
	for (uint32_t r = 0; r < rows; r++) {
		uint32_t line_len = 0;
		for (uint32_t cx = 0; cx < cols; cx++) {
			size_t t = (size_t)r * cols + cx;
			if ((size_t)(p - out) + 9 + orc_qoi_bound(tw[t], th[t], channels) > out_cap)  /* stream starts 4 B early */
				return 0;
			memcpy(p, "block", 5);
			float v = (has_value && !has_value[t]) ? 0.0f : block_value[t];
			uint32_t vb;
			memcpy(&vb, &v, 4);
			put_be32(p + 5, vb);
			/* encode after a 9-byte gap so that the "qoif" magic lands on the
			 * bytes the length field will overwrite (mod.rs:191 drops it) */
			size_t qlen = orc_qoi_encode(slots + slot * t, tw[t], th[t], channels, p + 9) - 4;
			put_be32(p + 9, (uint32_t)qlen);
			p += 13 + qlen;
			line_len += 13 + (uint32_t)qlen;
		}
		put_be32(line_table + (size_t)r * 4, line_len);
	}
	return (size_t)(p - out);
}

int orc_decode_container(const uint8_t *in, size_t len, uint32_t *width, uint32_t *height,
                         uint32_t *bw, uint32_t *bh, uint32_t *filter_byte,
                         float *block_value, uint32_t *tw, uint32_t *th, uint32_t *tc,
                         uint8_t *slots, size_t slot_stride, uint32_t max_tiles)
{
	if (len < 26 || memcmp(in, "PIXLZR", 6) != 0)
		return -1;
	if (!(in[6] == 0 && in[7] == 0 && in[8] == 2))
		return -2; /* only v0.0.2 (filter byte + line table) */
	*filter_byte = in[9];
	*width = get_be32(in + 10);
	*height = get_be32(in + 14);
	*bw = get_be32(in + 18);
	*bh = get_be32(in + 22);
	uint32_t cols = (uint32_t)ceilf((float)*width / (float)*bw);
	uint32_t rows = (uint32_t)ceilf((float)*height / (float)*bh);
	if ((uint64_t)cols * rows > max_tiles)
		return -3;
	const uint8_t *p = in + 26 + (size_t)rows * 4;
	size_t total = 0;
	for (uint32_t r = 0; r < rows; r++)
		total += get_be32(in + 26 + (size_t)r * 4);
	if ((size_t)(p - in) + total != len)
		return -4; /* mod.rs:141 */
	for (size_t t = 0; t < (size_t)cols * rows; t++) {
		if (memcmp(p, "block", 5) != 0)
			return -5;
		uint32_t vb = get_be32(p + 5);
		memcpy(&block_value[t], &vb, 4);
		uint32_t qlen = get_be32(p + 9);
		if (qoi_decode_body(p + 13, qlen, &tw[t], &th[t], &tc[t], slots + slot_stride * t, slot_stride) != 0)
			return -6;
		p += 13 + qlen;
	}
	return 0;
}
