// pxz_api.cpp — host runtime behind the C ABI of include/pixlzr_hip.h:
// handle, per-configuration table cache, HBM staging for the host-buffer entry
// point, kernel launches on the caller's stream, HIP-event timing.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <condition_variable>
#include <mutex>
#include <thread>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <system_error>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/pixlzr_hip.h"
#include "pxz_internal.h"
#include "pxz_tables.h"

namespace pxz {
hipError_t launch_shrink(const ShrinkArgs &a, uint32_t channels, uint32_t n_cus, hipStream_t stream);
hipError_t launch_synth(const SynthArgs &s, hipStream_t stream);
hipError_t launch_tree_decide(const TreeArgs &a, hipStream_t stream);
hipError_t launch_tree_rects(const TreeRectArgs &a, hipStream_t stream);
hipError_t launch_oklab_pixels(const uint32_t *px, uint32_t n, float *out, uint32_t n_cus, hipStream_t stream);
hipError_t launch_finish(const FinishArgs &f, hipStream_t stream);
bool fast32_applicable(const ShrinkArgs &a, uint32_t channels);
bool fast64_applicable(const ShrinkArgs &a, uint32_t channels);
bool fast16_applicable(const ShrinkArgs &a, uint32_t channels);
hipError_t launch_expand(const ExpandArgs &a, uint32_t n_cus, hipStream_t stream);
hipError_t launch_decode(const DecodeArgs &a, bool bins_clean, hipStream_t stream);
hipError_t launch_widen(const WidenArgs &a, hipStream_t stream);
hipError_t launch_narrow(const NarrowArgs &a, hipStream_t stream);
hipError_t launch_pack(const PackArgs &a, hipStream_t stream);
hipError_t launch_oklab(const ShrinkArgs &a, uint32_t n_cus, hipStream_t stream, uint32_t channels = 4);
hipError_t launch_qoi(const QoiArgs &a, bool bins_clean, uint32_t n_cus, hipStream_t stream);
size_t qoi_scratch_bytes(uint32_t n_tiles, uint32_t slot_px, uint32_t channels);
uint32_t qoi_bins_dwords();
uint32_t waves_per_tile(uint32_t bw, uint32_t bh);
}  // namespace pxz

namespace {

using pxz::AxisTab;
using pxz::kMaxLevel;

// device copy of the down-scaling tables for one (tile geometry, filter)
struct TableSet {
	std::vector<AxisTab> tabs;  // host copy, passed by value in the kernel arguments
	uint16_t *d_bounds = nullptr;
	uint32_t *d_coeffs = nullptr;
	int32_t *d_ksums = nullptr;
	uint32_t *d_rows = nullptr;
	uint32_t rows_dw = 0;
	uint32_t *d_mf64 = nullptr;  // 64x64 fast path: matrix-core operand tables (null: not available)
	bool opaque_stays = true;    // a constant-255 alpha comes back as 255 from every window of every table
};

// decode side: up-scaling tables of every source size to the full tile size (expand_kernel)
struct ExpandTables {
	pxz::ExpandTab *d_dir = nullptr;
	uint16_t *d_starts = nullptr, *d_sizes = nullptr;
	int16_t *d_coeffs = nullptr;
	uint32_t dir_stride = 0;
	uint32_t *d_xmf = nullptr;  // 32x32 tiles, convolutions: matrix-core operand tables (pxz_internal.h: kXmfDw)
	uint32_t *d_xmf16 = nullptr;  // 16x16 tiles, convolutions: the same for expand16_kernel (kXmf16Dw)
	uint32_t *d_xmf64 = nullptr;  // 64x64 tiles, convolutions: the same for expand64_kernel (kXmf64Dw)
};

struct DeviceBuffer {
	void *ptr = nullptr;
	size_t cap = 0;
};

// tree::process on rectangle lists: every axis table (down with the one filter, back up with the other) of every tile
// size the recursion can reach from one (frame, block, minimum) geometry
struct TreeTables {
	pxz::TreeAxisEntry *d_dir = nullptr;
	int32_t *d_starts = nullptr, *d_sizes = nullptr;
	int16_t *d_coeffs = nullptr;
	uint32_t n_dir = 0;
};

}  // namespace

constexpr uint32_t kMaxImageSide = 1u << 24;  // see pxz_grid

struct pxz_handle {
	int device = 0;
	uint32_t n_cus = 256;
	hipStream_t stream = nullptr;
	std::string error;
	float thresholds[pxz::kNumThresholds];
	std::map<std::tuple<uint32_t, uint32_t, uint32_t, uint32_t, uint32_t>, TableSet> tables;
	// level breakpoints per (mode, factor bits, bw, bh, edge_w, edge_h)
	struct Breaks { uint32_t b[4][pxz::kMaxLevel]; uint32_t asc[4]; };
	std::map<std::tuple<uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t>, Breaks> breaks;
	DeviceBuffer in, val, ow, oh, out, sums, chunks, work, qscratch, qmeta, status, dmeta, okscratch, rgba, slots4, pk, pkoff, tree, xlist, bigscratch;
	uint64_t packed_len = 0;   // bytes of the stream pxz_shrink_image_packed left in `pk` (0: none)
	static constexpr int kRing = 3;  // buffer sets of the pipelined host boundary (pxz_shrink_images*)
	DeviceBuffer ring_in[kRing], ring_val[kRing], ring_ow[kRing], ring_oh[kRing], ring_out[kRing], ring_pk[kRing], ring_pkoff[kRing];
	std::map<std::tuple<uint32_t, uint32_t, uint32_t, uint32_t, uint32_t>, ExpandTables> expand_tables;
	std::map<std::tuple<uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t>, TreeTables> tree_tables;
	DeviceBuffer tree_rects[2], tree_count;
	uint32_t *host_stats = nullptr;  // pinned, device-visible: [0] = tiles with transparency the last finished 32x32 launch saw
	uint32_t *dev_stats = nullptr;   //   (its device-side address); read without synchronisation, steers only the kernel choice
	uint32_t last_alpha_kernel = 0, last_alpha_first = 0;  // what the last launch set up through this handle chose (pxz_handle_state)
	bool work_ready = false;   // both worklist counters are zero / consistent with work_slot
	const uint32_t *qbins_clean = nullptr;  // the writer's binning counters at this address were left zeroed by the last launch_qoi
	const uint32_t *dbins_clean = nullptr;  // the same for the reader's (launch_decode)
	uint32_t work_slot = 0;    // the counter the next 32x32 launch uses
	bool timing = false;
	uint32_t timing_stride = 1, timing_count = 0;  // every stride-th step is bracketed by events
	std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
	std::vector<hipEvent_t> mid_events;  // one per pair: behind the first kernel of the step
	size_t events_used = 0;
};

namespace {

int fail(pxz_handle *h, int code, const char *fmt, ...)
{
	if (h) {
		char buf[512];
		va_list ap;
		va_start(ap, fmt);
		vsnprintf(buf, sizeof buf, fmt, ap);
		va_end(ap);
		h->error = buf;
	}
	return code;
}

void drop_tree_tables(pxz_handle *h)
{
	for (auto &kv : h->tree_tables) {
		(void)hipFree(kv.second.d_dir);
		(void)hipFree(kv.second.d_starts);
		(void)hipFree(kv.second.d_sizes);
		(void)hipFree(kv.second.d_coeffs);
	}
	h->tree_tables.clear();
}

#define PXZ_HIP(h, call)                                                                      \
	do {                                                                                      \
		hipError_t e_ = (call);                                                               \
		if (e_ != hipSuccess) return fail((h), PXZ_ERR_HIP, "%s: %s", #call, hipGetErrorString(e_)); \
	} while (0)

int ensure(pxz_handle *h, DeviceBuffer &b, size_t bytes)
{
	if (b.cap >= bytes) return PXZ_OK;
	if (b.ptr) (void)hipFree(b.ptr);
	b.ptr = nullptr;
	b.cap = 0;
	if (hipMalloc(&b.ptr, bytes) != hipSuccess) return fail(h, PXZ_ERR_NOMEM, "hipMalloc(%zu) failed", bytes);
	b.cap = bytes;
	return PXZ_OK;
}

uint32_t ceil_div(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

// magic number for unsigned division by d >= 1 (exact for every 32-bit dividend)
pxz::FastDiv make_fastdiv(uint32_t d)
{
	uint32_t l = 0;
	while ((1ull << l) < d) ++l;  // ceil(log2 d)
	const uint64_t m = ((1ull << 32) * ((1ull << l) - d)) / d + 1;
	return pxz::FastDiv{(uint32_t)m, l < 1 ? l : 1u, l < 1 ? 0u : l - 1};
}

// reduced size for level exponent m (reference operations.rs:150-151)
uint32_t reduced(uint32_t size, uint32_t m)
{
	if (m >= 31) return 1;
	uint64_t r = ((uint64_t)size + ((1ull << m) - 1ull)) >> m;
	return r < 1 ? 1u : (uint32_t)r;
}

int get_tables(pxz_handle *h, uint32_t bw, uint32_t bh, uint32_t edge_w, uint32_t edge_h, uint32_t filter,
               const TableSet **out)
{
	auto key = std::make_tuple(bw, bh, edge_w, edge_h, filter);
	auto it = h->tables.find(key);
	if (it != h->tables.end()) {
		*out = &it->second;
		return PXZ_OK;
	}
	std::vector<AxisTab> tabs(2 * 2 * kMaxLevel);
	std::vector<uint16_t> bounds;
	std::vector<uint32_t> coeffs;
	std::vector<int32_t> ksums;
	std::vector<uint32_t> rows;
	bool all_opaque_stays = true;
	std::vector<uint32_t> mf64(4, 0u);  // offset 0 means "no table"
	bool mf64_complete = bw == 64 && bh == 64 && filter != PXZ_FILTER_NEAREST;
	const uint32_t sizes[2][2] = {{bw, edge_w}, {bh, edge_h}};
	for (int axis = 0; axis < 2; ++axis) {
		for (int cls = 0; cls < 2; ++cls) {
			const uint32_t in = sizes[axis][cls];
			for (int m = 0; m < kMaxLevel; ++m) {
				AxisTab &t = tabs[(axis * 2 + cls) * kMaxLevel + m];
				const uint32_t outsz = reduced(in, (uint32_t)m);
				t = AxisTab{0, 0, 0, 0, 0, (uint16_t)outsz, 0, 0, (uint16_t)in, 0};
				if (outsz == in) continue;  // identity: never looked up
				// identical (in, out) pairs share one table: the edge class of a grid without ragged
				// edge, and every level past the first that reaches 1 px
				if (cls == 1 && sizes[axis][1] == sizes[axis][0]) {
					t = tabs[(axis * 2 + 0) * kMaxLevel + m];
					continue;
				}
				if (m > 0 && tabs[(axis * 2 + cls) * kMaxLevel + m - 1].out_size == outsz &&
				    tabs[(axis * 2 + cls) * kMaxLevel + m - 1].in_size == in && reduced(in, (uint32_t)m - 1) != in) {
					t = tabs[(axis * 2 + cls) * kMaxLevel + m - 1];
					continue;
				}
				pxz::AxisWindows win;
				if (!pxz::build_axis(in, outsz, filter, &win)) return fail(h, PXZ_ERR_INVALID_ARG, "unknown filter %u", filter);
				t.bounds_off = (uint32_t)bounds.size();
				t.coeff_off = (uint32_t)coeffs.size();
				t.ksum_off = (uint32_t)ksums.size();
				t.precision = (uint16_t)win.precision;
				if (filter == PXZ_FILTER_NEAREST) {
					for (uint32_t o = 0; o < outsz; ++o) bounds.push_back((uint16_t)win.starts[o]);
					continue;
				}
				// pad every window to whole quads of 4 source samples (8-byte aligned LDS reads)
				uint32_t wquads = 1;
				for (uint32_t o = 0; o < outsz; ++o) {
					const uint32_t lead = (uint32_t)win.starts[o] & 3u;
					const uint32_t nq = (lead + (uint32_t)win.sizes[o] + 3u) / 4u;
					if (nq > wquads) wquads = nq;
				}
				t.wquads = (uint16_t)wquads;
				t.rows_off = (uint32_t)rows.size();
				// rows of up to 8 quads are padded with zero weights to header + 16 dwords: the fast path fetches
				// whole rows with 16-byte loads and needs no per-quad guard
				t.row_stride = wquads <= 8u ? 20u : (4u + wquads * 2u + 3u) & ~3u;
				for (uint32_t o = 0; o < outsz; ++o) {
					const uint32_t first = (uint32_t)win.starts[o], n = (uint32_t)win.sizes[o];
					const uint32_t lead = first & 3u, nq = (lead + n + 3u) / 4u;
					bounds.push_back((uint16_t)(first / 4u));
					bounds.push_back((uint16_t)nq);
					std::vector<int16_t> k(wquads * 4u, 0);
					int32_t total = 0;
					for (uint32_t i = 0; i < n; ++i) {
						k[lead + i] = win.coeffs[(size_t)o * win.window + i];
						total += k[lead + i];
					}
					const size_t row0 = rows.size();
					rows.resize(row0 + t.row_stride, 0u);
					rows[row0 + 0] = first / 4u;
					rows[row0 + 1] = nq;
					rows[row0 + 2] = (uint32_t)total;
					for (uint32_t d = 0; d < wquads * 2u; ++d) {
						const uint32_t pr = (uint32_t)(uint16_t)k[2 * d] | ((uint32_t)(uint16_t)k[2 * d + 1] << 16);
						coeffs.push_back(pr);
						rows[row0 + 4 + d] = pr;
					}
					ksums.push_back(total);
					if ((((1 << (win.precision - 1)) + 255 * total) >> win.precision) < 255) all_opaque_stays = false;
				}
				// 32x32 tiles: operands for the matrix-core form of the two-pass resample (x axis table,
				// used for both axes of a full tile)
				if (axis == 0 && cls == 0 && bw == 32 && bh == 32 && outsz <= 16) {
					std::vector<uint32_t> mf(pxz::kMfDwords, 0u);
					bool fits = true, opaque_stays = true;
					const int32_t half = 1 << (win.precision - 1);
					for (uint32_t o = 0; o < outsz; ++o) {
						int32_t k[32] = {0};
						int32_t total = 0;
						for (uint32_t i = 0; i < (uint32_t)win.sizes[o]; ++i) {
							k[(uint32_t)win.starts[o] + i] = win.coeffs[(size_t)o * win.window + i];
							total += k[(uint32_t)win.starts[o] + i];
						}
						for (uint32_t g = 0; g < 4; ++g) {
							for (uint32_t j = 0; j < 8; ++j) {
								const uint32_t src = j < 4 ? 4 * g + j : 16 + 4 * g + (j - 4);
								const int32_t lo = ((k[src] + 128) & 255) - 128, hi = (k[src] - lo) / 256;
								if (hi < -128 || hi > 127) fits = false;
								const uint32_t lane = g * 16 + o, dw = 2 * lane + j / 4, sh = 8 * (j & 3);
								mf[dw] |= (uint32_t)(uint8_t)lo << sh;
								mf[128 + dw] |= (uint32_t)(uint8_t)hi << sh;
							}
						}
						mf[256 + o] = (uint32_t)(128 * total + half);
						mf[272 + o] = (uint32_t)total;
						const int32_t al = (half + 255 * total) >> win.precision;
						if (al < 255) opaque_stays = false;  // clip8: above 255 is clamped to 255
					}
					mf[288] = opaque_stays ? 1u : 0u;
					if (fits) {
						t.mf_off = (uint32_t)rows.size();
						rows.insert(rows.end(), mf.begin(), mf.end());
					}
				}
				// 16x16 tiles: operands of the group-of-four matrix-core resample (resample_group16_mfma): 16 -> 8 | 4 | 2 | 1
				if (axis == 0 && cls == 0 && bw == 16 && bh == 16 && outsz <= 8) {
					std::vector<uint32_t> mf(pxz::kMf16Dwords, 0u);
					bool fits = true, opaque_stays = true;
					const int32_t half = 1 << (win.precision - 1);
					for (uint32_t o = 0; o < outsz; ++o) {
						int32_t k[16] = {0};
						int32_t total = 0;
						for (uint32_t i = 0; i < (uint32_t)win.sizes[o]; ++i) {
							k[(uint32_t)win.starts[o] + i] = win.coeffs[(size_t)o * win.window + i];
							total += k[(uint32_t)win.starts[o] + i];
						}
						for (uint32_t i = 0; i < 16; ++i) {
							const int32_t lo = ((k[i] + 128) & 255) - 128, hi = (k[i] - lo) / 256;
							if (hi < -128 || hi > 127) fits = false;
							mf[o * 4 + i / 4] |= (uint32_t)(uint8_t)lo << (8 * (i & 3));
							mf[32 + o * 4 + i / 4] |= (uint32_t)(uint8_t)hi << (8 * (i & 3));
						}
						mf[64 + o] = (uint32_t)(128 * total + half);
						mf[72 + o] = (uint32_t)total;
						if (((half + 255 * total) >> win.precision) < 255) opaque_stays = false;
					}
					mf[80] = opaque_stays ? 1u : 0u;
					mf[81] = (uint32_t)win.precision;
					if (fits && opaque_stays) {  // (the group form writes alpha 255: only where the windows keep it)
						t.mf_off = (uint32_t)rows.size();
						rows.insert(rows.end(), mf.begin(), mf.end());
					}
				}
				// 64x64 tiles: operands of shrink64_kernel (Fast64Args), every level from 32 px down to 1 px
				if (axis == 0 && cls == 0 && bw == 64 && bh == 64 && outsz < 64) {
					const uint32_t nblk = outsz > 16 ? outsz / 16 : 1;
					std::vector<uint32_t> mf((size_t)nblk * 512 + 72, 0u);
					bool fits = true, opaque_stays = true;
					const int32_t half = 1 << (win.precision - 1);
					for (uint32_t o = 0; o < outsz; ++o) {
						int32_t k[64] = {0};
						int32_t total = 0;
						for (uint32_t i = 0; i < (uint32_t)win.sizes[o]; ++i) {
							k[(uint32_t)win.starts[o] + i] = win.coeffs[(size_t)o * win.window + i];
							total += k[(uint32_t)win.starts[o] + i];
						}
						const uint32_t blk = o / 16, ol = o % 16;
						for (uint32_t g = 0; g < 4; ++g) {
							for (uint32_t j = 0; j < 16; ++j) {
								const int32_t v = k[16 * g + j];
								const int32_t lo = ((v + 128) & 255) - 128, hi = (v - lo) / 256;
								if (hi < -128 || hi > 127) fits = false;
								const uint32_t lane = g * 16 + ol, dw = blk * 512 + lane * 4 + j / 4, sh = 8 * (j & 3);
								mf[dw] |= (uint32_t)(uint8_t)lo << sh;
								mf[256 + dw] |= (uint32_t)(uint8_t)hi << sh;
							}
						}
						mf[nblk * 512 + o] = (uint32_t)(128 * total + half);
						mf[nblk * 512 + 32 + o] = (uint32_t)total;
						if (((half + 255 * total) >> win.precision) < 255) opaque_stays = false;
					}
					mf[nblk * 512 + 64] = opaque_stays ? 1u : 0u;
					if (fits && outsz <= 32) {
						t.mf_off = (uint32_t)mf64.size();
						mf64.insert(mf64.end(), mf.begin(), mf.end());
					} else {
						mf64_complete = false;
					}
				}
			}
		}
	}
	if (bounds.empty()) bounds.push_back(0);
	if (coeffs.empty()) coeffs.push_back(0);
	if (ksums.empty()) ksums.push_back(0);
	rows.resize(rows.size() + 32, 0u);  // the fast path always fetches 4+16 dwords per row
	TableSet ts;
	ts.tabs = tabs;
	ts.opaque_stays = all_opaque_stays;
	PXZ_HIP(h, hipMalloc((void **)&ts.d_bounds, bounds.size() * sizeof(uint16_t)));
	PXZ_HIP(h, hipMalloc((void **)&ts.d_coeffs, coeffs.size() * sizeof(uint32_t)));
	PXZ_HIP(h, hipMalloc((void **)&ts.d_ksums, ksums.size() * sizeof(int32_t)));
	PXZ_HIP(h, hipMemcpy(ts.d_bounds, bounds.data(), bounds.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
	PXZ_HIP(h, hipMemcpy(ts.d_coeffs, coeffs.data(), coeffs.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
	PXZ_HIP(h, hipMemcpy(ts.d_ksums, ksums.data(), ksums.size() * sizeof(int32_t), hipMemcpyHostToDevice));
	if (mf64_complete) {
		PXZ_HIP(h, hipMalloc((void **)&ts.d_mf64, mf64.size() * sizeof(uint32_t)));
		PXZ_HIP(h, hipMemcpy(ts.d_mf64, mf64.data(), mf64.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
	}
	ts.rows_dw = (uint32_t)rows.size();
	PXZ_HIP(h, hipMalloc((void **)&ts.d_rows, rows.size() * sizeof(uint32_t)));
	PXZ_HIP(h, hipMemcpy(ts.d_rows, rows.data(), rows.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
	h->tables[key] = ts;
	*out = &h->tables[key];
	return PXZ_OK;
}

// Tables of the decode side: for each axis and size class (full / ragged edge) one up-scaling table per
// source size 1 .. full-1 (PixlzrBlock::resize with the upscale flag set, block.rs:301-304).
int get_expand_tables(pxz_handle *h, uint32_t bw, uint32_t bh, uint32_t edge_w, uint32_t edge_h, uint32_t filter,
                      const ExpandTables **out)
{
	auto key = std::make_tuple(bw, bh, edge_w, edge_h, filter);
	auto it = h->expand_tables.find(key);
	if (it != h->expand_tables.end()) {
		*out = &it->second;
		return PXZ_OK;
	}
	const uint32_t stride = (bw > bh ? bw : bh) + 1u;
	std::vector<pxz::ExpandTab> dir(4u * stride, pxz::ExpandTab{0, 0, 0, 0});
	std::vector<uint16_t> starts, sizes;
	std::vector<int16_t> coeffs;
	const uint32_t full[2][2] = {{bw, edge_w}, {bh, edge_h}};
	for (uint32_t axis = 0; axis < 2; ++axis) {
		for (uint32_t cls = 0; cls < 2; ++cls) {
			const uint32_t outsz = full[axis][cls];
			if (cls == 1 && outsz == full[axis][0]) {
				for (uint32_t in = 0; in < stride; ++in) dir[(axis * 2 + 1) * stride + in] = dir[(axis * 2 + 0) * stride + in];
				continue;
			}
			for (uint32_t in = 1; in < outsz; ++in) {
				pxz::AxisWindows win;
				if (!pxz::build_axis(in, outsz, filter, &win, true)) return fail(h, PXZ_ERR_INVALID_ARG, "unknown filter %u", filter);
				pxz::ExpandTab &t = dir[(axis * 2 + cls) * stride + in];
				t.start_off = (uint32_t)starts.size();
				t.coeff_off = (uint32_t)coeffs.size();
				t.window = (uint16_t)win.window;
				t.precision = (uint16_t)win.precision;
				for (uint32_t o = 0; o < outsz; ++o) {
					starts.push_back((uint16_t)win.starts[o]);
					sizes.push_back((uint16_t)win.sizes[o]);
				}
				coeffs.insert(coeffs.end(), win.coeffs.begin(), win.coeffs.end());
			}
		}
	}
	if (starts.empty()) {
		starts.push_back(0);
		sizes.push_back(0);
	}
	if (coeffs.empty()) coeffs.push_back(0);
	ExpandTables et;
	et.dir_stride = stride;
	// 32x32 tiles: the up-scales 1, 2, 4, 8, 16 -> 32 as matrix-core operands (layout: pxz_internal.h)
	if (bw == 32 && bh == 32 && filter != 0) {
		std::vector<uint32_t> xmf((size_t)pxz::kXmfLevels * pxz::kXmfDw, 0u);
		bool fits = true;
		for (uint32_t li = 0; li < pxz::kXmfLevels; ++li) {
			const uint32_t in = 1u << li;
			pxz::AxisWindows win;
			if (!pxz::build_axis(in, 32, filter, &win, true)) return fail(h, PXZ_ERR_INVALID_ARG, "unknown filter %u", filter);
			uint32_t *mf = xmf.data() + (size_t)li * pxz::kXmfDw;
			const int32_t half = 1 << (win.precision - 1);
			bool copies = in == 1 && win.precision < 15;
			for (uint32_t o = 0; o < 32; ++o) {
				int32_t k[16] = {0};
				int32_t total = 0;
				for (uint32_t i = 0; i < (uint32_t)win.sizes[o]; ++i) {
					k[(uint32_t)win.starts[o] + i] = win.coeffs[(size_t)o * win.window + i];
					total += k[(uint32_t)win.starts[o] + i];
				}
				if (win.sizes[o] != 1 || k[0] != (1 << win.precision)) copies = false;
				for (uint32_t kg = 0; kg < 2; ++kg) {
					for (uint32_t j = 0; j < 8; ++j) {
						const int32_t v = k[pxz::xmf_src(kg, j)];
						const int32_t lo = ((v + 128) & 255) - 128, hi = (v - lo) / 256;
						if (hi < -128 || hi > 127) fits = false;
						const uint32_t lane = kg * 32 + o, dw = 2 * lane + j / 4, sh = 8 * (j & 3);
						mf[dw] |= (uint32_t)(uint8_t)lo << sh;
						mf[128 + dw] |= (uint32_t)(uint8_t)hi << sh;
					}
				}
				mf[256 + o] = (uint32_t)(128 * total + half);
			}
			for (uint32_t g = 0; g < 2; ++g)
				for (uint32_t reg = 0; reg < 16; ++reg) mf[288 + 16 * g + reg] = mf[256 + pxz::xmf_row(g, reg)];
			mf[320] = (uint32_t)win.precision;
			mf[321] = copies ? 1u : 0u;
		}
		if (fits) {
			PXZ_HIP(h, hipMalloc((void **)&et.d_xmf, xmf.size() * sizeof(uint32_t)));
			PXZ_HIP(h, hipMemcpy(et.d_xmf, xmf.data(), xmf.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
		}
	}
	// 16x16 tiles: the up-scales 1, 2, 4, 8 -> 16 as matrix-core operands of expand16_kernel (layout: pxz_internal.h)
	if (bw == 16 && bh == 16 && filter != 0) {
		std::vector<uint32_t> xmf((size_t)pxz::kXmf16Levels * pxz::kXmf16Dw, 0u);
		bool fits = true;
		for (uint32_t li = 0; li < pxz::kXmf16Levels; ++li) {
			const uint32_t in = 1u << li;
			pxz::AxisWindows win;
			if (!pxz::build_axis(in, 16, filter, &win, true)) return fail(h, PXZ_ERR_INVALID_ARG, "unknown filter %u", filter);
			uint32_t *mf = xmf.data() + (size_t)li * pxz::kXmf16Dw;
			const int32_t half = 1 << (win.precision - 1);
			for (uint32_t o = 0; o < 16; ++o) {
				int32_t k[8] = {0};
				int32_t total = 0;
				for (uint32_t i = 0; i < (uint32_t)win.sizes[o]; ++i) {
					k[(uint32_t)win.starts[o] + i] = win.coeffs[(size_t)o * win.window + i];
					total += k[(uint32_t)win.starts[o] + i];
				}
				for (uint32_t i = 0; i < 8; ++i) {
					const int32_t lo = ((k[i] + 128) & 255) - 128, hi = (k[i] - lo) / 256;
					if (hi < -128 || hi > 127) fits = false;
					mf[o * 2 + i / 4] |= (uint32_t)(uint8_t)lo << (8 * (i & 3));
					mf[32 + o * 2 + i / 4] |= (uint32_t)(uint8_t)hi << (8 * (i & 3));
				}
				mf[64 + o] = (uint32_t)(128 * total + half);
			}
			for (uint32_t g = 0; g < 2; ++g)
				for (uint32_t r = 0; r < 8; ++r) mf[80 + 8 * g + r] = mf[64 + (r & 3) + 8 * (r >> 2) + 4 * g];
			mf[96] = (uint32_t)win.precision;
		}
		if (fits) {
			PXZ_HIP(h, hipMalloc((void **)&et.d_xmf16, xmf.size() * sizeof(uint32_t)));
			PXZ_HIP(h, hipMemcpy(et.d_xmf16, xmf.data(), xmf.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
		}
	}
	// 64x64 tiles: the up-scales 1 .. 32 -> 64 as matrix-core operands of expand64_kernel (layout: pxz_internal.h)
	if (bw == 64 && bh == 64 && filter != 0) {
		std::vector<uint32_t> xmf((size_t)pxz::kXmf64Levels * pxz::kXmf64Dw, 0u);
		bool fits = true;
		for (uint32_t li = 0; li < pxz::kXmf64Levels; ++li) {
			const uint32_t in = 1u << li;
			pxz::AxisWindows win;
			if (!pxz::build_axis(in, 64, filter, &win, true)) return fail(h, PXZ_ERR_INVALID_ARG, "unknown filter %u", filter);
			uint32_t *mf = xmf.data() + (size_t)li * pxz::kXmf64Dw;
			const int32_t half = 1 << (win.precision - 1);
			for (uint32_t o = 0; o < 64; ++o) {
				int32_t k[32] = {0};
				int32_t total = 0;
				for (uint32_t i = 0; i < (uint32_t)win.sizes[o]; ++i) {
					k[(uint32_t)win.starts[o] + i] = win.coeffs[(size_t)o * win.window + i];
					total += k[(uint32_t)win.starts[o] + i];
				}
				const uint32_t q = o >> 5, ol = o & 31u;
				for (uint32_t st = 0; st < 2; ++st)
					for (uint32_t kg = 0; kg < 2; ++kg)
						for (uint32_t j = 0; j < 8; ++j) {
							const int32_t v = k[16 * st + pxz::xmf_src(kg, j)];
							const int32_t lo = ((v + 128) & 255) - 128, hi = (v - lo) / 256;
							if (hi < -128 || hi > 127) fits = false;
							const uint32_t lane = kg * 32 + ol, dw = ((q * 2 + st) * 2) * 128 + 2 * lane + j / 4, sh = 8 * (j & 3);
							mf[dw] |= (uint32_t)(uint8_t)lo << sh;
							mf[128 + dw] |= (uint32_t)(uint8_t)hi << sh;
						}
				mf[1024 + o] = (uint32_t)(128 * total + half);
			}
			for (uint32_t q = 0; q < 2; ++q)
				for (uint32_t g = 0; g < 2; ++g)
					for (uint32_t reg = 0; reg < 16; ++reg) mf[1088 + (q * 2 + g) * 16 + reg] = mf[1024 + 32 * q + pxz::xmf_row(g, reg)];
			mf[1152] = (uint32_t)win.precision;
		}
		if (fits) {
			PXZ_HIP(h, hipMalloc((void **)&et.d_xmf64, xmf.size() * sizeof(uint32_t)));
			PXZ_HIP(h, hipMemcpy(et.d_xmf64, xmf.data(), xmf.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
		}
	}
	PXZ_HIP(h, hipMalloc((void **)&et.d_dir, dir.size() * sizeof(pxz::ExpandTab)));
	PXZ_HIP(h, hipMalloc((void **)&et.d_starts, starts.size() * sizeof(uint16_t)));
	PXZ_HIP(h, hipMalloc((void **)&et.d_sizes, sizes.size() * sizeof(uint16_t)));
	PXZ_HIP(h, hipMalloc((void **)&et.d_coeffs, coeffs.size() * sizeof(int16_t)));
	PXZ_HIP(h, hipMemcpy(et.d_dir, dir.data(), dir.size() * sizeof(pxz::ExpandTab), hipMemcpyHostToDevice));
	PXZ_HIP(h, hipMemcpy(et.d_starts, starts.data(), starts.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
	PXZ_HIP(h, hipMemcpy(et.d_sizes, sizes.data(), sizes.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
	PXZ_HIP(h, hipMemcpy(et.d_coeffs, coeffs.data(), coeffs.size() * sizeof(int16_t), hipMemcpyHostToDevice));
	h->expand_tables[key] = et;
	*out = &h->expand_tables[key];
	return PXZ_OK;
}

// ---- level decision tables ---------------------------------------------------
// reference src/operations.rs:128-138
float parse_value_host(float value)
{
	uint32_t bits;
	std::memcpy(&bits, &value, 4);
	if ((bits >> 31) == 0) return value;
	float t = 1.0f + value;
	return (t != t) ? 0.0f : (t > 0.0f ? t : 0.0f);
}

// level exponent of one tile of (w,h) whose directional gradient sum is `sum`
// (operations.rs:253-258 -> pixlzr.rs:199 -> operations.rs:145-148)
uint32_t level_of_sum(const pxz_handle *h, uint64_t sum, uint32_t w, uint32_t hh, float factor)
{
	const uint64_t fac = (uint64_t)(w - 2) * (uint64_t)(hh - 2) * 4096ull;
	const float raw = (float)((double)sum / (double)fac);
	const float v = parse_value_host(raw * factor);
	uint32_t m = 0;
	for (int j = 0; j < kMaxLevel; ++j) m += (v < h->thresholds[j]) ? 1u : 0u;
	return m;
}

// Fills a->breaks / a->breaks_asc.  Oklab mode: the key is the bit pattern of the (non-negative)
// parsed value, compared against the float thresholds' bit patterns.  Directional mode: the key is
// the integer gradient sum; the float pipeline sum -> value -> level is monotone in the sum, so each
// threshold becomes one integer breakpoint per tile class, found by bisection with the exact formula.
void build_breaks(pxz_handle *h, pxz::ShrinkArgs *a)
{
	uint32_t fbits;
	std::memcpy(&fbits, &a->factor, 4);
	const auto key = std::make_tuple(a->mode, fbits, a->bw, a->bh, a->edge_w, a->edge_h);
	auto it = h->breaks.find(key);
	if (it != h->breaks.end()) {
		std::memcpy(a->breaks, it->second.b, sizeof a->breaks);
		std::memcpy(a->breaks_asc, it->second.asc, sizeof a->breaks_asc);
		return;
	}
	struct Saver {
		pxz_handle *h; pxz::ShrinkArgs *a; decltype(key) k;
		~Saver() {
			pxz_handle::Breaks br;
			std::memcpy(br.b, a->breaks, sizeof br.b);
			std::memcpy(br.asc, a->breaks_asc, sizeof br.asc);
			h->breaks[k] = br;
		}
	} saver{h, a, key};
	for (int cls = 0; cls < 4; ++cls) {
		a->breaks_asc[cls] = 0;
		for (int j = 0; j < kMaxLevel; ++j) std::memcpy(&a->breaks[cls][j], &h->thresholds[j], 4);
	}
	if (a->mode != PXZ_MODE_SHRINK_DIRECTIONALLY) return;
	for (int cls = 0; cls < 4; ++cls) {
		const uint32_t w = (cls & 1) ? a->edge_w : a->bw, hh = (cls & 2) ? a->edge_h : a->bh;
		if (w <= 2 || hh <= 2) {  // 0/0 tiles are special-cased in the kernel
			for (int j = 0; j < kMaxLevel; ++j) a->breaks[cls][j] = 0xffffffffu;
			continue;
		}
		const uint64_t max_sum = (uint64_t)(w - 2) * (hh - 2) * 3u * 1020u;
		const uint32_t m_lo = level_of_sum(h, 0, w, hh, a->factor), m_hi = level_of_sum(h, max_sum, w, hh, a->factor);
		const bool asc = m_hi > m_lo;  // level exponent grows with the sum (negative factors)
		a->breaks_asc[cls] = asc ? 1u : 0u;
		for (int j = 0; j < kMaxLevel; ++j) {
			// predicate "v < T[j]"  <=>  level exponent > j
			auto below = [&](uint64_t s) { return level_of_sum(h, s, w, hh, a->factor) > (uint32_t)j; };
			const bool at0 = below(0), atmax = below(max_sum);
			uint32_t brk;
			if (at0 == atmax) {
				// constant: encode always-true / never for the class' comparison direction
				const bool always = at0;
				brk = asc ? (always ? 0u : 0xffffffffu) : (always ? 0xffffffffu : 0u);
			} else {
				uint64_t lo = 0, hi = max_sum;  // below(lo) == at0, below(hi) == atmax
				while (hi - lo > 1) {
					const uint64_t mid = lo + (hi - lo) / 2;
					if (below(mid) == at0) lo = mid; else hi = mid;
				}
				brk = (uint32_t)hi;  // first sum on the far side
			}
			a->breaks[cls][j] = brk;
		}
	}
}

int check_frames(pxz_handle *h, const pxz_frames *f, const pxz_params *p)
{
	if (!h) return PXZ_ERR_INVALID_ARG;
	if (!f || !p) return fail(h, PXZ_ERR_INVALID_ARG, "null descriptor");
	if (f->width == 0 || f->height == 0 || f->n_frames == 0) return fail(h, PXZ_ERR_INVALID_ARG, "empty frame batch");
	if (f->width > kMaxImageSide || f->height > kMaxImageSide)
		return fail(h, PXZ_ERR_UNSUPPORTED, "image sides above 2^24 are not supported (the reference's f32 and f64 tile grids part there)");
	if (f->channels != 3 && f->channels != 4) return fail(h, PXZ_ERR_INVALID_ARG, "channels must be 3 or 4, got %u", f->channels);
	if ((uint64_t)f->pitch_bytes < (uint64_t)f->width * f->channels) return fail(h, PXZ_ERR_INVALID_ARG, "pitch smaller than a row");
	if (f->n_frames > 1 && f->frame_stride_bytes < (uint64_t)f->pitch_bytes * f->height)
		return fail(h, PXZ_ERR_INVALID_ARG, "frame stride smaller than a frame");
	if (p->block_w == 0 || p->block_h == 0) return fail(h, PXZ_ERR_INVALID_ARG, "zero block size");
	if (p->mode > 1) return fail(h, PXZ_ERR_INVALID_ARG, "mode must be 0 or 1");
	if (p->filter > 4) return fail(h, PXZ_ERR_INVALID_ARG, "filter must be 0..4");
	if (!std::isfinite(p->factor)) return fail(h, PXZ_ERR_INVALID_ARG, "factor must be finite");
	return PXZ_OK;
}

// Fills the kernel arguments for a batch; returns the LDS bytes needed per block.
int prepare(pxz_handle *h, const pxz_frames *f, const pxz_params *p, bool want_pixels, pxz::ShrinkArgs *a, bool no_lab = false)
{
	int rc = check_frames(h, f, p);
	if (rc != PXZ_OK) return rc;
	uint32_t cols, rows;
	pxz_grid(f->width, f->height, p->block_w, p->block_h, &cols, &rows);
	const uint32_t bw = p->block_w, bh = p->block_h;
	const uint32_t edge_w = f->width - (cols - 1) * bw, edge_h = f->height - (rows - 1) * bh;
	if (p->mode == PXZ_MODE_SHRINK_DIRECTIONALLY && (edge_w < 2 || edge_h < 2 || bw < 2 || bh < 2))
		return fail(h, PXZ_ERR_TILE_TOO_SMALL,
		            "directional detector needs tiles of at least 2x2 px (edge tile is %ux%u); the reference panics here",
		            edge_w, edge_h);
	if ((uint64_t)cols * rows * f->n_frames > 0xffffffffull) return fail(h, PXZ_ERR_UNSUPPORTED, "too many tiles");
	if (bw > 0xffffu || bh > 0xffffu) return fail(h, PXZ_ERR_UNSUPPORTED, "block side above 65535");

	auto round2 = [](uint32_t v) { return (v + 1u) & ~1u; };
	const bool conv = want_pixels && p->filter != PXZ_FILTER_NEAREST;
	auto skew = [](uint32_t v) { return (v & 15u) == 0 ? v + 2u : v; };  // keep rows off a common LDS bank
	a->rs = skew(round2(ceil_div(bw, 2)));
	a->plane_dw = a->rs * bh;
	a->hps = skew(round2(ceil_div(bh, 2)));
	a->tmp_dw = conv ? ceil_div(bw, 2) * a->hps : 0;
	// (no_lab: every tile's Oklab value will come from oklab_kernel launches -- the generic kernel then needs no f32
	// planes of its own, which is what limits shrink_by to ~7000-pixel tiles otherwise)
	a->lab_dw = p->mode == PXZ_MODE_SHRINK_BY && !no_lab ? 3 * bw * bh : 0;
	// planes | max(transposed planes, Oklab scratch) | slack for zero-weight over-reads past the last row
	const uint32_t scratch = 4 * a->tmp_dw > a->lab_dw ? 4 * a->tmp_dw : a->lab_dw;
	a->tile_dw = (4 * a->plane_dw + scratch + 4 * a->rs + 4 * a->hps + 3u) & ~3u;
	const uint32_t nw = pxz::waves_per_tile(bw, bh);
	const uint64_t lds_bytes = (uint64_t)a->tile_dw * 4u + (nw > 1 ? 16u * nw : 0u);
	a->big_scratch = nullptr;
	a->big_blocks = 0;
	if (lds_bytes > 160u * 1024u) {
		// (round 4) a tile image beyond LDS lives in HBM, one per block of the generic kernel: any block size the reference's CLI
		// accepts runs (src/bin/main.rs:19-24), at the speed of L2 round trips.  The kernel's index arithmetic (small_div, RowWalker)
		// holds below 2^20 pixels per tile; the grid is capped so that the images stay under 2 GB.
		if ((uint64_t)bw * bh >= (1u << 20) || pxz::knobs().no_big_tiles)
			return fail(h, PXZ_ERR_UNSUPPORTED, "a %ux%u tile needs %llu B of LDS (limit 163840) and has too many pixels for the HBM-resident form (limit 2^20 - 1)", bw, bh,
			            (unsigned long long)lds_bytes);
		const uint64_t tile_bytes = (uint64_t)a->tile_dw * 4u;
		uint64_t blocks = (2ull << 30) / tile_bytes;
		if (blocks > 2ull * h->n_cus) blocks = 2ull * h->n_cus;
		if (blocks < 1) blocks = 1;
		a->big_blocks = (uint32_t)blocks;  // (capped by the tile count at launch)
	}

	a->frame_stride = f->n_frames > 1 ? f->frame_stride_bytes : (uint64_t)f->pitch_bytes * f->height;
	a->pitch = f->pitch_bytes;
	a->width = f->width;
	a->height = f->height;
	a->bw = bw;
	a->bh = bh;
	a->cols = cols;
	a->rows = rows;
	a->tiles_per_frame = cols * rows;
	a->div_tpf = make_fastdiv(cols * rows);
	a->div_cols = make_fastdiv(cols);
	a->n_tiles = cols * rows * f->n_frames;
	a->edge_w = edge_w;
	a->edge_h = edge_h;
	a->mode = p->mode;
	a->filter = p->filter;
	a->factor = p->factor;
	a->scale2 = 10.0f;  // BASE_FACTOR, pixlzr.rs:15
	a->alpha_kernel = (p->reserved & PXZ_HINT_TRANSPARENCY) != 0 && !pxz::knobs().no_alpha_kernel;
	a->list_a_too = 0;
	a->clone_ahead = 0;
	a->finish_scan = 1;
	a->stats = nullptr;
	a->slot_bytes = bw * bh * f->channels;
	build_breaks(h, a);
	std::memset(a->tabs, 0, sizeof a->tabs);
	a->bounds = nullptr;
	a->coeffs = nullptr;
	a->ksums = nullptr;
	a->trows = nullptr;
	a->tab_dw = 0;
	if (want_pixels) {
		const TableSet *tsp = nullptr;
		rc = get_tables(h, bw, bh, edge_w, edge_h, p->filter, &tsp);
		if (rc != PXZ_OK) return rc;
		const TableSet &ts = *tsp;
		std::memcpy(a->tabs, ts.tabs.data(), sizeof a->tabs);
		a->bounds = ts.d_bounds;
		a->coeffs = ts.d_coeffs;
		a->ksums = ts.d_ksums;
		a->trows = ts.d_rows;
		a->tab_dw = ts.rows_dw * 4u <= 48u * 1024u ? (ts.rows_dw + 3u) & ~3u : 0u;
		a->mf64 = ts.d_mf64;
	}
	return PXZ_OK;
}

// fused kernel + finishing kernel (stored value / raw detector outputs), timed together
int timed_launch(pxz_handle *h, pxz::ShrinkArgs &a, uint32_t channels, float *value, float *lod0, float *lod1)
{
	int rc = ensure(h, h->sums, (size_t)a.n_tiles * 8u);
	if (rc != PXZ_OK) return rc;
	a.sums = (uint32_t *)h->sums.ptr;
	const void *work_before = h->work.ptr;
	const size_t work_bytes = (2u * (size_t)a.n_tiles + pxz::kWorkList + 4u) * 4u + 64u + 4608u * 8u;  // + diagnostic stamps (8 phase sums + 256 blocks x 17 qwords)
	// + the 2 KB spare bytes of the detector's copies (ahead_spare) + the list of clone_split64_kernel (8 + n_tiles dwords)
	if ((rc = ensure(h, h->work, work_bytes + 2048u + 8u + (8u + (size_t)a.n_tiles) * 4u)) != PXZ_OK) return rc;
	if (h->work.ptr != work_before) h->work_ready = false;
	a.work = (uint32_t *)h->work.ptr;
	a.ahead_spare = reinterpret_cast<uint32_t *>(static_cast<uint8_t *>(h->work.ptr) + ((work_bytes + 7u) & ~(size_t)7u));
	a.clone_list = a.ahead_spare + 512u;
	a.value = value;
	a.lod0 = lod0;
	a.lod1 = lod1;
	if (a.big_blocks != 0u) {
		if (a.big_blocks > a.n_tiles) a.big_blocks = a.n_tiles;
		if ((rc = ensure(h, h->bigscratch, (size_t)a.big_blocks * a.tile_dw * 4u)) != PXZ_OK) return rc;
		a.big_scratch = (uint32_t *)h->bigscratch.ptr;
	}
	// Transparency without the caller's hint: the last finished launch reported how many full tiles had any
	// (one dword in pinned memory, written by the worklist kernel).  Past ~2000 tiles shrink32a_kernel pays for
	// its launch.  Either way the results are the same; only the kernel that produces them differs.
	a.stats = h->dev_stats;
	// (the counts are only trusted when they come from a launch of THIS configuration: the kernel writes the signature beside them)
	uint32_t factor_bits;
	std::memcpy(&factor_bits, &a.factor, 4);
	a.stats_sig = ((a.n_tiles * 2654435761u) ^ (a.bw << 20) ^ (a.bh << 8) ^ (a.mode << 31) ^ (channels << 28) ^ a.width ^ (a.height * 40503u) ^
	               (a.filter * 0x9e3779b1u) ^ (factor_bits * 31u)) | 1u;
	volatile uint32_t *hs = h->host_stats ? const_cast<volatile uint32_t *>(h->host_stats) : nullptr;
	const bool stats_ours = hs != nullptr && hs[2] == a.stats_sig;
	const uint32_t seen_transparent = stats_ours ? hs[0] : 0u;
	a.expect_listed = stats_ours ? hs[1] : 0xffffffffu;
	if (!a.alpha_kernel && seen_transparent >= 2048u && !pxz::knobs().no_alpha_kernel)
		a.alpha_kernel = 1;
	// ... and past half of the tiles the lean kernel would only read, test and list them: the four-plane kernel goes first
	a.alpha_first = a.alpha_kernel && seen_transparent >= a.n_tiles / 2u && seen_transparent >= 2048u && !pxz::knobs().no_alpha_first ? 1u : 0u;
	h->last_alpha_kernel = a.alpha_kernel;
	h->last_alpha_first = a.alpha_first;
	// 32x32 fast path: which tiles are full-size, and whether every tile row of the batch is 16-byte aligned
	a.full_cols = a.full_rows = a.ok_rows = 0;
	const bool aligned16 = channels == 4 &&
	    ((reinterpret_cast<uintptr_t>(a.src) | a.pitch | (a.n_tiles > a.tiles_per_frame ? a.frame_stride : 0)) & 15u) == 0;
	const bool square_fast = a.bw == a.bh && (a.bw == 16 || a.bw == 32 || a.bw == 64);
	// any other tile whose rows are whole pixel quads: the Oklab detector with run-time geometry takes the full tiles
	// (64 .. 16384 pixels); the generic kernel then only stages and resamples them
	const bool general_oklab = !square_fast && a.mode == PXZ_MODE_SHRINK_BY && a.bw % 4u == 0 && a.bw * a.bh >= 64u &&
	                           a.bw * a.bh <= 16384u && !pxz::knobs().no_oklab_general;
	a.ok_bands = (a.bw * a.bh + 255u) / 256u;
	// rows of a ragged last tile row the square detectors can take: whole 256-px bands (oklab_kernel), for 64-px tiles
	// whole 512-px super-bands (oklab2_kernel<64>: four producer waves x two rows)
	const uint32_t ok_row_quantum = a.bw == 64u ? 8u : (a.bw != 0u && a.bw <= 256u ? 256u / a.bw : 1u);
	// 64x64 tiles are converted by oklab2_kernel<64> (round 3), which parks nothing in HBM; PXZ_OKLAB_V1 keeps oklab_kernel<64>
	const bool parks64 = !(a.bw == 64u && a.bh == 64u) || pxz::knobs().oklab_v1;
	a.ok_region = 0;
	a.ok_count = a.n_tiles;
	a.ok_edges = 0;
	// RGB frames on the square fast paths (round 2; both callers): 12-byte pixel quads, rows 4-byte aligned
	const bool rgb_native = channels == 3 && square_fast && !pxz::knobs().no_native_rgb &&
	    ((reinterpret_cast<uintptr_t>(a.src) | a.pitch | (a.n_tiles > a.tiles_per_frame ? a.frame_stride : 0)) & 3u) == 0;
	if (rgb_native) {
		a.full_cols = a.edge_w == a.bw ? a.cols : a.cols - 1;
		a.full_rows = a.edge_h == a.bh ? a.rows : a.rows - 1;
		a.ok_rows = a.edge_h % ok_row_quantum == 0 ? a.rows : a.full_rows;  // (a ragged last row of whole bands, as below)
	}
	if (aligned16 && (square_fast || general_oklab)) {
		a.full_cols = a.edge_w == a.bw ? a.cols : a.cols - 1;
		a.full_rows = a.edge_h == a.bh ? a.rows : a.rows - 1;
		// the Oklab detector also takes a ragged last row of whole bands (256 pixels = 256/bw rows) of the square sizes
		a.ok_rows = square_fast && a.edge_h % ok_row_quantum == 0 ? a.rows : a.full_rows;
	}
	const pxz::FinishArgs fin{a.sums, value, lod0, lod1, a.n_tiles, a.tiles_per_frame, a.cols, a.rows,
	                          a.bw, a.bh, a.edge_w, a.edge_h, a.mode, a.factor};
	hipEvent_t e0 = nullptr, e1 = nullptr, emid = nullptr;
	a.mid_event = nullptr;
	const bool record = h->timing && (h->timing_count++ % h->timing_stride) == 0;
	if (record) {
		if (h->events_used == h->events.size()) {
			PXZ_HIP(h, hipEventCreate(&e0));
			PXZ_HIP(h, hipEventCreate(&e1));
			PXZ_HIP(h, hipEventCreate(&emid));
			h->events.emplace_back(e0, e1);
			h->mid_events.push_back(emid);
		}
		e0 = h->events[h->events_used].first;
		e1 = h->events[h->events_used].second;
		emid = h->mid_events[h->events_used];
		++h->events_used;
		PXZ_HIP(h, hipEventRecord(e0, h->stream));
		a.mid_event = emid;  // recorded behind the first kernel of the step (pxz_last_first_kernel_ms)
	}
	// shrink_by on the headline geometry: the block-cooperative Oklab detector first, then the
	// fused kernel only stages + resamples (it still runs the generic detector on ragged-edge tiles)
	a.oklab_given = 0;
	if (a.mode == PXZ_MODE_SHRINK_BY && channels == 4 && (square_fast || general_oklab) &&
	    a.full_cols != 0 && a.full_rows != 0 && !pxz::knobs().no_oklab32) {
		// The ragged edge of the grid -- right column (edge_w x bh), bottom row (bw x edge_h), corner tile -- goes
		// through the same detector with run-time geometry, one launch per region (tile rows that are not whole
		// pixel quads are walked padded).  Without it the edge runs its chains in the generic kernel: four lanes per tile,
		// ~0.1-0.5 ms of latency per launch whatever the batch.
		const uint32_t n_frames = a.n_tiles / a.tiles_per_frame;
		struct Region { uint32_t id, w, hh, per_frame, bit; bool wanted; } regions[3] = {
		    {1u, a.edge_w, a.bh, a.full_rows, 1u, a.full_cols < a.cols},
		    {2u, a.bw, a.edge_h, a.full_cols, 2u, a.full_rows < a.rows && a.ok_rows < a.rows},
		    {3u, a.edge_w, a.edge_h, 1u, 4u, a.full_cols < a.cols && a.full_rows < a.rows}};
		size_t scratch = a.ok_bands > 4u && parks64 ? (size_t)a.n_tiles * a.ok_bands * 3328u : 0u;
		for (Region &r : regions) {
			const uint32_t wp = (r.w + 3u) & ~3u;  // rows are walked in whole quads (the padding counts as zeros)
			r.wanted = r.wanted && r.per_frame != 0u && a.bw % 4u == 0u && wp * r.hh <= 16384u && !pxz::knobs().no_oklab_edges;
			const uint32_t bands = (wp * r.hh + 255u) / 256u;
			if (r.wanted && bands > 4u) scratch = std::max(scratch, (size_t)n_frames * r.per_frame * bands * 3328u);
		}
		if (scratch != 0) {
			// a tile of more than 1024 pixels does not fit the registers between the detector's two passes: 13 dwords
			// per pixel quad in HBM
			if ((rc = ensure(h, h->okscratch, scratch)) != PXZ_OK) return rc;
			a.ok_scratch = (float *)h->okscratch.ptr;
		}
		a.oklab_given = 1;
		// (round 4) the square detectors copy every tile into its slot while they have its pixels: the shrink kernel then
		// skips the tiles that are stored at full size instead of reading them a second time
		a.clone_ahead = square_fast && a.out_px != nullptr && !pxz::knobs().oklab_v1 && !pxz::knobs().no_clone_ahead ? 1u : 0u;
		PXZ_HIP(h, pxz::launch_oklab(a, h->n_cus, h->stream));
		if (a.mid_event) {
			PXZ_HIP(h, hipEventRecord(static_cast<hipEvent_t>(a.mid_event), h->stream));
			a.mid_event = nullptr;
		}
		for (const Region &r : regions) {
			if (!r.wanted) continue;
			pxz::ShrinkArgs e = a;
			e.ok_region = r.id;
			e.ok_bands = (((r.w + 3u) & ~3u) * r.hh + 255u) / 256u;
			e.ok_count = n_frames * r.per_frame;
			PXZ_HIP(h, pxz::launch_oklab(e, h->n_cus, h->stream));
			a.ok_edges |= r.bit;
		}
	}
	if (a.mode == PXZ_MODE_SHRINK_BY && rgb_native && a.full_cols != 0 && a.full_rows != 0 && !pxz::knobs().no_oklab32) {
		// RGB, 16x16 / 32x32 / 64x64: oklab2_kernel<16 | 32, 3> / oklab_kernel<64, 0, 3> for the full tiles; a ragged edge
		// keeps its chains in the generic kernel
		if (a.ok_bands > 4u && parks64) {  // 64x64 with PXZ_OKLAB_V1: the converted tile is parked in HBM between the passes
			if ((rc = ensure(h, h->okscratch, (size_t)a.n_tiles * a.ok_bands * 3328u)) != PXZ_OK) return rc;
			a.ok_scratch = (float *)h->okscratch.ptr;
		}
		a.oklab_given = 1;
		// (as for RGBA above; an RGB lane's six bytes are two stores, which costs the 16x16 detector more than its shrink kernel
		// gains -- 8 x 8K: 1.151 against 1.141 ms; 32x32 1.052 against 1.072, 64x64 1.20 against 1.37)
		a.clone_ahead = a.out_px != nullptr && a.bw != 16u && !pxz::knobs().oklab_v1 && !pxz::knobs().no_clone_ahead ? 1u : 0u;
		PXZ_HIP(h, pxz::launch_oklab(a, h->n_cus, h->stream, 3));
		if (a.mid_event) {
			PXZ_HIP(h, hipEventRecord(static_cast<hipEvent_t>(a.mid_event), h->stream));
			a.mid_event = nullptr;
		}
	}
	if (a.mode == PXZ_MODE_SHRINK_BY && a.lab_dw == 0) {
		// the generic kernel was sized without its own detector planes: every tile must have its value by now
		const bool right = a.full_cols < a.cols, bottom = a.full_rows < a.rows && a.ok_rows < a.rows, corner = a.full_cols < a.cols && a.full_rows < a.rows;
		const bool covered = a.oklab_given && (!right || (a.ok_edges & 1u)) && (!bottom || (a.ok_edges & 2u)) && (!corner || (a.ok_edges & 4u));
		if (!covered) return fail(h, PXZ_ERR_UNSUPPORTED, "internal: Oklab values missing for a layout sized without detector planes");
	}
	// 32x32 RGBA flow: shrink32_kernel, then the worklist kernel, which also finishes every tile and
	// zeroes the worklist counter of the next launch (two counters, used alternately)
	{
		// shrink16_kernel walks 2x2 groups of tiles
		const uint32_t gcols = (a.cols + 1u) / 2u, grows = (a.rows + 1u) / 2u;
		a.div_gpf = make_fastdiv(gcols * grows);
		a.div_gcols = make_fastdiv(gcols);
		a.n_frames_x_groups = gcols * grows * (a.n_tiles / a.tiles_per_frame);
	}
	const bool fast = pxz::fast32_applicable(a, channels) || pxz::fast64_applicable(a, channels) || pxz::fast16_applicable(a, channels);
	if (fast) {
		if (!h->work_ready) {
			PXZ_HIP(h, hipMemsetAsync(h->work.ptr, 0, pxz::kWorkList * 4u, h->stream));
			h->work_slot = 0;
		}
		h->work_ready = false;  // stays false if a launch below fails: the counters are then re-zeroed next time
		a.work_slot = h->work_slot;
	}
	PXZ_HIP(h, pxz::launch_shrink(a, channels, h->n_cus, h->stream));
	if (!fast && a.mid_event) PXZ_HIP(h, hipEventRecord(static_cast<hipEvent_t>(a.mid_event), h->stream));  // the generic kernel is the step
	if (fast) {
		h->work_slot ^= 1u;
		h->work_ready = true;
	} else {
		PXZ_HIP(h, pxz::launch_finish(fin, h->stream));
	}
	if (record) PXZ_HIP(h, hipEventRecord(e1, h->stream));
	return PXZ_OK;
}

}  // namespace

namespace {
// The pixels are here on the host: a sparse look at the alpha channel (one pixel in 61 per sampled row, every 7th
// row) decides whether the kernels for transparent tiles are worth their launch (> 2 % of the samples).
// Upload of a host image for the host-buffer entry points.  The device copy gets rows of a 16-byte multiple when the
// caller's do not have one (tight rows of an RGBA image whose width is not a multiple of 4): the fast kernels and
// the block-cooperative Oklab detector want aligned rows, and the copy engine does the re-pitching for free.
int upload_image(pxz_handle *h, const uint8_t *pixels, uint32_t width, uint32_t height, uint32_t channels, uint32_t pitch_bytes,
                 uint32_t *device_pitch)
{
	const size_t row_bytes = (size_t)width * channels;
	uint32_t dp = pitch_bytes;
	if (channels == 4 && (pitch_bytes & 15u) != 0) dp = (uint32_t)((row_bytes + 15u) & ~(size_t)15u);
	const size_t bytes = (size_t)dp * (height - 1) + row_bytes;
	int rc = ensure(h, h->in, bytes);
	if (rc != PXZ_OK) return rc;
	if (dp == pitch_bytes) PXZ_HIP(h, hipMemcpyAsync(h->in.ptr, pixels, bytes, hipMemcpyHostToDevice, h->stream));
	else PXZ_HIP(h, hipMemcpy2DAsync(h->in.ptr, dp, pixels, pitch_bytes, row_bytes, height, hipMemcpyHostToDevice, h->stream));
	*device_pitch = dp;
	return PXZ_OK;
}

bool host_image_has_transparency(const uint8_t *pixels, uint32_t width, uint32_t height, uint32_t pitch_bytes)
{
	uint32_t seen = 0, looked = 0;
	for (uint32_t y = 0; y < height; y += 7)
		for (uint32_t x = (y * 13u) % 61u; x < width; x += 61, ++looked)
			seen += pixels[(size_t)y * pitch_bytes + (size_t)x * 4u + 3u] != 255u;
	return looked && seen * 50u > looked;
}
}  // namespace

extern "C" {

const char *pxz_version(void) { return "pixlzr-hip 0.1.0 (gfx950)"; }

int pxz_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

int pxz_create(int device_id, pxz_handle **out)
{
	if (!out) return PXZ_ERR_INVALID_ARG;
	*out = nullptr;
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device_id < 0 || device_id >= n) return PXZ_ERR_NO_DEVICE;
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) return PXZ_ERR_NO_DEVICE;
	if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return PXZ_ERR_NO_DEVICE;  // kernels are gfx950 only
	if (hipSetDevice(device_id) != hipSuccess) return PXZ_ERR_HIP;
	pxz_handle *h = new (std::nothrow) pxz_handle();
	if (!h) return PXZ_ERR_NOMEM;
	h->device = device_id;
	h->n_cus = prop.multiProcessorCount > 0 ? (uint32_t)prop.multiProcessorCount : 256u;
	if (!pxz::build_level_thresholds(h->thresholds, pxz::kNumThresholds)) {
		delete h;
		return PXZ_ERR_UNSUPPORTED;  // platform log2f is not a clean step around 2^(k+1/2)
	}
	void *hs = nullptr, *ds = nullptr;
	if (hipHostMalloc(&hs, 64, hipHostMallocMapped) == hipSuccess) {
		std::memset(hs, 0, 64);
		static_cast<uint32_t *>(hs)[1] = 0xffffffffu;  // listed tiles of the last launch: unknown
		if (hipHostGetDevicePointer(&ds, hs, 0) == hipSuccess) {
			h->host_stats = static_cast<uint32_t *>(hs);
			h->dev_stats = static_cast<uint32_t *>(ds);
		} else {
			(void)hipHostFree(hs);
		}
	}
	(void)hipGetLastError();  // no pinned dword: no adaptive kernel choice, nothing else changes
	*out = h;
	return PXZ_OK;
}

void pxz_destroy(pxz_handle *h)
{
	if (!h) return;
	(void)hipSetDevice(h->device);
	for (auto &kv : h->tables) {
		(void)hipFree(kv.second.d_bounds);
		(void)hipFree(kv.second.d_coeffs);
		(void)hipFree(kv.second.d_ksums);
		(void)hipFree(kv.second.d_rows);
		(void)hipFree(kv.second.d_mf64);
	}
	for (auto &kv : h->expand_tables) {
		(void)hipFree(kv.second.d_dir);
		(void)hipFree(kv.second.d_starts);
		(void)hipFree(kv.second.d_sizes);
		(void)hipFree(kv.second.d_coeffs);
		(void)hipFree(kv.second.d_xmf);
		(void)hipFree(kv.second.d_xmf16);
		(void)hipFree(kv.second.d_xmf64);
	}
	drop_tree_tables(h);
	for (DeviceBuffer *b : {&h->in, &h->val, &h->ow, &h->oh, &h->out, &h->sums, &h->chunks, &h->work, &h->qscratch, &h->qmeta, &h->status, &h->dmeta, &h->okscratch, &h->rgba, &h->slots4, &h->pk, &h->pkoff, &h->tree, &h->xlist, &h->bigscratch, &h->tree_rects[0], &h->tree_rects[1], &h->tree_count})
		if (b->ptr) (void)hipFree(b->ptr);
	for (int i = 0; i < pxz_handle::kRing; ++i)
		for (DeviceBuffer *b : {&h->ring_in[i], &h->ring_val[i], &h->ring_ow[i], &h->ring_oh[i], &h->ring_out[i], &h->ring_pk[i], &h->ring_pkoff[i]})
			if (b->ptr) (void)hipFree(b->ptr);
	for (auto &ev : h->events) {
		(void)hipEventDestroy(ev.first);
		(void)hipEventDestroy(ev.second);
	}
	for (hipEvent_t ev : h->mid_events) (void)hipEventDestroy(ev);
	if (h->host_stats) {
		(void)hipDeviceSynchronize();  // a queued launch may still write it
		(void)hipHostFree(h->host_stats);
	}
	delete h;
}

const char *pxz_last_error(const pxz_handle *h) { return h ? h->error.c_str() : "null handle"; }

int pxz_trim(pxz_handle *h)
{
	if (!h) return PXZ_ERR_INVALID_ARG;
	PXZ_HIP(h, hipSetDevice(h->device));
	PXZ_HIP(h, hipStreamSynchronize(h->stream));
	auto drop = [](DeviceBuffer &b) {
		if (b.ptr) (void)hipFree(b.ptr);
		b.ptr = nullptr;
		b.cap = 0;
	};
	for (DeviceBuffer *b : {&h->in, &h->val, &h->ow, &h->oh, &h->out, &h->sums, &h->chunks, &h->work, &h->qscratch, &h->qmeta, &h->status, &h->dmeta, &h->okscratch, &h->rgba, &h->slots4, &h->pk, &h->pkoff, &h->tree, &h->xlist, &h->bigscratch, &h->tree_rects[0], &h->tree_rects[1], &h->tree_count})
		drop(*b);
	for (int i = 0; i < pxz_handle::kRing; ++i)
		for (DeviceBuffer *b : {&h->ring_in[i], &h->ring_val[i], &h->ring_ow[i], &h->ring_oh[i], &h->ring_out[i], &h->ring_pk[i], &h->ring_pkoff[i]})
			drop(*b);
	drop_tree_tables(h);
	h->packed_len = 0;
	h->work_ready = false;  // (the worklist counters went with their buffer)
	h->qbins_clean = nullptr;
	h->dbins_clean = nullptr;
	return PXZ_OK;
}

int pxz_set_stream(pxz_handle *h, void *hip_stream)
{
	if (!h) return PXZ_ERR_INVALID_ARG;
	hipStream_t next = static_cast<hipStream_t>(hip_stream);
	if (next != h->stream) {
		// the handle's scratch buffers (and the worklist counters a launch leaves for the next one) are
		// ordered by the stream: finish what is queued on the old one before moving on
		PXZ_HIP(h, hipSetDevice(h->device));
		PXZ_HIP(h, hipStreamSynchronize(h->stream));
		h->stream = next;
	}
	return PXZ_OK;
}

int pxz_synchronize(pxz_handle *h)
{
	if (!h) return PXZ_ERR_INVALID_ARG;
	PXZ_HIP(h, hipStreamSynchronize(h->stream));
	return PXZ_OK;
}

int pxz_grid(uint32_t width, uint32_t height, uint32_t block_w, uint32_t block_h, uint32_t *cols, uint32_t *rows)
{
	if (!cols || !rows || block_w == 0 || block_h == 0) return PXZ_ERR_INVALID_ARG;
	// The reference rounds up in f64 when it splits an image (iter.rs:38-41, split.rs:45-46) and in f32 when it writes
	// and expands one (pixlzr.rs:36-46).  Up to 2^24 both are the integer ceiling; beyond, the f32 form can differ from
	// it (and the reference's own two grids from each other), so larger images are refused here and by every entry
	// point -- ONE grid, this one, is used throughout the library.
	if (width > kMaxImageSide || height > kMaxImageSide) return PXZ_ERR_UNSUPPORTED;
	*cols = (uint32_t)(((uint64_t)width + block_w - 1u) / block_w);
	*rows = (uint32_t)(((uint64_t)height + block_h - 1u) / block_h);
	return PXZ_OK;
}

// One shrink / detector pass over a batch.  identity: the closures of process() instead of shrink_by's.
// RGB batches whose tile size has an RGBA fast path are widened to RGBA (alpha 255) in scratch memory, run
// there, and their tile slots narrowed back (see rgb_to_rgba_kernel for why the results are the same).
static int run_shrink(pxz_handle *h, const pxz_frames *frames, const pxz_params *params, const uint8_t *d_pixels,
                      float *d_block_value, uint32_t *d_out_w, uint32_t *d_out_h, uint8_t *d_out_pixels, float *d_lod0,
                      float *d_lod1, bool identity)
{
	PXZ_HIP(h, hipSetDevice(h->device));
	pxz::ShrinkArgs a{};
	// RGBA, shrink_by, a tile shape the block-cooperative detector takes and at least one full tile each way: interior
	// and edge launches of oklab_kernel cover every tile (rows are re-pitched below if they are not aligned)
	const uint32_t pbw = params ? params->block_w : 0, pbh = params ? params->block_h : 0;
	const bool covers_if_rgba = frames && params && params->mode == PXZ_MODE_SHRINK_BY && pbw != 0 && pbh != 0 &&
	                            pbw % 4u == 0 && (uint64_t)pbw * pbh >= 64u && (uint64_t)pbw * pbh <= 16384u &&
	                            frames->width >= pbw && frames->height >= pbh && !pxz::knobs().no_oklab32 &&
	                            !pxz::knobs().no_oklab_general && !pxz::knobs().no_oklab_edges && !pxz::knobs().no_repitch;
	const bool oklab_covers_all = covers_if_rgba && frames->channels == 4;
	int rc = prepare(h, frames, params, d_out_pixels != nullptr, &a, oklab_covers_all);
	bool rgb_must_widen = false;
	if (rc == PXZ_ERR_UNSUPPORTED && covers_if_rgba && frames->channels == 3 && !pxz::knobs().no_widen) {
		// the RGB layout with its own detector planes does not fit LDS; the widened RGBA one without them might
		pxz_frames fw = *frames;
		fw.channels = 4;
		fw.pitch_bytes = (frames->width * 4u + 15u) & ~15u;
		fw.frame_stride_bytes = (uint64_t)fw.pitch_bytes * frames->height;
		if (prepare(h, &fw, params, d_out_pixels != nullptr, &a, true) == PXZ_OK) {
			rc = PXZ_OK;
			rgb_must_widen = true;
		} else {
			(void)prepare(h, frames, params, d_out_pixels != nullptr, &a, false);  // (restore the first error text)
		}
	}
	if (rc != PXZ_OK) return rc;
	// (also shrink_by on the tile sizes the run-time-geometry Oklab detector takes: it only exists for RGBA)
	const bool square_fast = a.bw == a.bh && (a.bw == 16 || a.bw == 32 || a.bw == 64);
	const bool general_oklab = a.mode == PXZ_MODE_SHRINK_BY && a.bw % 4u == 0 && a.bw * a.bh >= 64u && a.bw * a.bh <= 16384u;
	bool widen = frames->channels == 3 && (square_fast || general_oklab) && !pxz::knobs().no_widen;
	// (16x16, 32x32, 64x64 tiles: the fast kernels and the Oklab detector read RGB themselves, when the rows are 4-byte aligned)
	if (widen && square_fast && !rgb_must_widen && !pxz::knobs().no_native_rgb &&
	    ((reinterpret_cast<uintptr_t>(d_pixels) | frames->pitch_bytes | (frames->n_frames > 1 ? frames->frame_stride_bytes : 0)) & 3u) == 0)
		widen = false;
	if (widen && d_out_pixels && params->filter != PXZ_FILTER_NEAREST) {
		const TableSet *tsp = nullptr;
		if ((rc = get_tables(h, a.bw, a.bh, a.edge_w, a.edge_h, params->filter, &tsp)) != PXZ_OK) return rc;
		widen = tsp->opaque_stays;
	}
	if (rgb_must_widen && !widen)
		return fail(h, PXZ_ERR_UNSUPPORTED, "a %ux%u RGB tile needs more LDS than there is (and this filter cannot run it as RGBA)", a.bw, a.bh);
	pxz_frames f4 = *frames;
	const uint8_t *src = d_pixels;
	uint8_t *out_px = d_out_pixels;
	if (widen) {
		f4.channels = 4;
		f4.pitch_bytes = (frames->width * 4u + 15u) & ~15u;
		f4.frame_stride_bytes = (uint64_t)f4.pitch_bytes * frames->height;
		pxz::ShrinkArgs probe{};
		if (prepare(h, &f4, params, d_out_pixels != nullptr, &probe, covers_if_rgba) != PXZ_OK) widen = false;  // e.g. four planes of a large tile exceed LDS
	}
	if (widen) {
		if ((rc = ensure(h, h->rgba, (size_t)f4.frame_stride_bytes * frames->n_frames)) != PXZ_OK) return rc;
		const pxz::WidenArgs w{d_pixels, (uint8_t *)h->rgba.ptr,
		                       frames->n_frames > 1 ? frames->frame_stride_bytes : (uint64_t)frames->pitch_bytes * frames->height,
		                       f4.frame_stride_bytes, frames->pitch_bytes, f4.pitch_bytes, frames->width, frames->height,
		                       frames->n_frames};
		PXZ_HIP(h, pxz::launch_widen(w, h->stream));
		if ((rc = prepare(h, &f4, params, d_out_pixels != nullptr, &a, covers_if_rgba)) != PXZ_OK) return rc;
		src = (const uint8_t *)h->rgba.ptr;
		if (d_out_pixels) {
			if ((rc = ensure(h, h->slots4, (size_t)a.n_tiles * a.bw * a.bh * 4u)) != PXZ_OK) return rc;
			out_px = (uint8_t *)h->slots4.ptr;
		}
	}
	// RGBA frames whose rows (or first byte, or frame stride) are not 16-byte multiples would miss the fast kernels and
	// the block-cooperative Oklab detector: one device-to-device 2D copy into aligned scratch first (0.4 ms per GB
	// against 5-50x on the kernels)
	if (!widen && frames->channels == 4 && (square_fast || general_oklab) && !pxz::knobs().no_repitch) {
		const uint64_t fstride = frames->n_frames > 1 ? frames->frame_stride_bytes : (uint64_t)frames->pitch_bytes * frames->height;
		const bool misaligned = ((reinterpret_cast<uintptr_t>(d_pixels) | frames->pitch_bytes | (frames->n_frames > 1 ? fstride : 0)) & 15u) != 0;
		if (misaligned) {
			pxz_frames fa = *frames;
			fa.pitch_bytes = (frames->width * 4u + 15u) & ~15u;
			fa.frame_stride_bytes = (uint64_t)fa.pitch_bytes * frames->height;
			if ((rc = ensure(h, h->rgba, (size_t)fa.frame_stride_bytes * frames->n_frames)) != PXZ_OK) return rc;
			for (uint32_t n = 0; n < frames->n_frames; ++n)
				PXZ_HIP(h, hipMemcpy2DAsync((uint8_t *)h->rgba.ptr + (size_t)n * fa.frame_stride_bytes, fa.pitch_bytes,
				                            d_pixels + (size_t)n * fstride, frames->pitch_bytes, (size_t)frames->width * 4u,
				                            frames->height, hipMemcpyDeviceToDevice, h->stream));
			if ((rc = prepare(h, &fa, params, d_out_pixels != nullptr, &a, oklab_covers_all)) != PXZ_OK) return rc;
			src = (const uint8_t *)h->rgba.ptr;
		}
	}
	if (identity) {
		a.factor = 1.0f;
		a.scale2 = 1.0f;  // (x * 1) * 1 is x exactly: the identity closure
	}
	a.src = src;
	a.out_w = d_out_w;
	a.out_h = d_out_h;
	a.out_px = out_px;
	if ((rc = timed_launch(h, a, widen ? 4u : frames->channels, d_block_value, d_lod0, d_lod1)) != PXZ_OK) return rc;
	if (widen && d_out_pixels) {
		const pxz::NarrowArgs n{(const uint8_t *)h->slots4.ptr, d_out_pixels, d_out_w, d_out_h, a.n_tiles, a.bw * a.bh * 4u,
		                        a.bw * a.bh * 3u};
		PXZ_HIP(h, pxz::launch_narrow(n, h->stream));
	}
	return PXZ_OK;
}

int pxz_shrink_frames_device(pxz_handle *h, const pxz_frames *frames, const pxz_params *params,
                             const uint8_t *d_pixels, float *d_block_value, uint32_t *d_out_w, uint32_t *d_out_h,
                             uint8_t *d_out_pixels)
{
	if (!h) return PXZ_ERR_INVALID_ARG;
	if (!d_pixels || !d_block_value || !d_out_w || !d_out_h) return fail(h, PXZ_ERR_INVALID_ARG, "null device pointer");
	return run_shrink(h, frames, params, d_pixels, d_block_value, d_out_w, d_out_h, d_out_pixels, nullptr, nullptr, false);
}

int pxz_lod_frames_device(pxz_handle *h, const pxz_frames *frames, const pxz_params *params,
                          const uint8_t *d_pixels, float *d_lod0, float *d_lod1)
{
	if (!h) return PXZ_ERR_INVALID_ARG;
	if (!d_pixels || !d_lod0 || !d_lod1) return fail(h, PXZ_ERR_INVALID_ARG, "null device pointer");
	return run_shrink(h, frames, params, d_pixels, nullptr, nullptr, nullptr, nullptr, d_lod0, d_lod1, false);
}

// frames: the OUTPUT batch (its channels = bytes per output pixel); slot_channels: channels of the stored tiles
static int expand_launch(pxz_handle *h, const pxz_frames *frames, uint32_t slot_channels, const pxz_params *params,
                         const uint32_t *d_tile_w, const uint32_t *d_tile_h, const uint8_t *d_slots, uint8_t *d_out_pixels,
                         bool quiet_empty = false)
{
	if (!h) return PXZ_ERR_INVALID_ARG;
	if (!frames || !params) return fail(h, PXZ_ERR_INVALID_ARG, "null descriptor");
	pxz_params p = *params;
	p.mode = 0;
	p.factor = 0.0f;  // neither is used on the decode side
	int rc = check_frames(h, frames, &p);
	if (rc != PXZ_OK) return rc;
	if (!d_tile_w || !d_tile_h || !d_slots || !d_out_pixels) return fail(h, PXZ_ERR_INVALID_ARG, "null device pointer");
	PXZ_HIP(h, hipSetDevice(h->device));
	const uint32_t bw = p.block_w, bh = p.block_h;
	uint32_t cols, rows;
	pxz_grid(frames->width, frames->height, bw, bh, &cols, &rows);
	if ((uint64_t)cols * rows * frames->n_frames > 0xffffffffull) return fail(h, PXZ_ERR_UNSUPPORTED, "too many tiles");
	if (bw > 0xffffu || bh > 0xffffu) return fail(h, PXZ_ERR_UNSUPPORTED, "block side above 65535");
	// one wave: source pixels + horizontal-pass result + the staged windows (5 dwords per output sample of both axes)
	const uint64_t lds_bytes = (2ull * bw * bh + 5ull * (bw + bh) + 3ull) / 4ull * 16ull + 16ull;
	// (round 4) a tile image beyond LDS lives in HBM, one per wave of the grid (expand_kernel<C, false, true>)
	const bool big = lds_bytes > 160u * 1024u;
	if (big && ((uint64_t)bw * bh >= (1u << 20) || pxz::knobs().no_big_tiles))
		return fail(h, PXZ_ERR_UNSUPPORTED, "a %ux%u tile needs %llu B of LDS (limit 163840) and has too many pixels for the HBM-resident form (limit 2^20 - 1)", bw, bh, (unsigned long long)lds_bytes);
	pxz::ExpandArgs a{};
	a.tile_w = d_tile_w;
	a.tile_h = d_tile_h;
	a.slots = d_slots;
	a.dst = d_out_pixels;
	a.frame_stride = frames->n_frames > 1 ? frames->frame_stride_bytes : (uint64_t)frames->pitch_bytes * frames->height;
	a.pitch = frames->pitch_bytes;
	a.width = frames->width;
	a.height = frames->height;
	a.channels = slot_channels;
	a.bw = bw;
	a.bh = bh;
	a.cols = cols;
	a.rows = rows;
	a.tiles_per_frame = cols * rows;
	a.n_tiles = cols * rows * frames->n_frames;
	a.edge_w = frames->width - (cols - 1) * bw;
	a.edge_h = frames->height - (rows - 1) * bh;
	a.slot_bytes = bw * bh * slot_channels;
	a.filter = p.filter;
	a.out_channels = frames->channels;
	a.quiet_empty = quiet_empty ? 1u : 0u;
	const ExpandTables *et = nullptr;
	if ((rc = get_expand_tables(h, bw, bh, a.edge_w, a.edge_h, p.filter, &et)) != PXZ_OK) return rc;
	a.tabs = et->d_dir;
	a.dir_stride = et->dir_stride;
	a.starts = et->d_starts;
	a.sizes = et->d_sizes;
	a.coeffs = et->d_coeffs;
	a.fast32 = pxz::knobs().no_expand_fast32 ? 0u : 1u;
	a.xmf = a.fast32 ? et->d_xmf : nullptr;
	a.tile_dw = (2u * bw * bh + 5u * (bw + bh) + 3u) & ~3u;
	if (big) {
		uint64_t waves = (2ull << 30) / ((uint64_t)a.tile_dw * 4u);
		if (waves > 8ull * h->n_cus) waves = 8ull * h->n_cus;
		waves &= ~3ull;
		if (waves < 4) waves = 4;
		if ((rc = ensure(h, h->bigscratch, (size_t)waves * a.tile_dw * 4u)) != PXZ_OK) return rc;
		a.big_scratch = (uint32_t *)h->bigscratch.ptr;
		a.big_waves = (uint32_t)waves;
	}
	// 16x16 RGBA tiles in RGBA frames: expand16_kernel takes the 2x2 groups of full tiles (clones, powers of two), the rest -- and
	// what it leaves -- goes to expand_kernel through a list (status[1] counts it)
	const bool groups16 = a.fast32 && bw == 16 && bh == 16 && slot_channels == 4 && frames->channels == 4 && cols >= 2 && rows >= 2 &&
	                      (p.filter == 0 || et->d_xmf16 != nullptr);
	if (groups16) {
		if ((rc = ensure(h, h->xlist, (size_t)a.n_tiles * 4u)) != PXZ_OK) return rc;
		a.list = (uint32_t *)h->xlist.ptr;
		a.xmf16 = et->d_xmf16;
		a.div_gpf = make_fastdiv(((cols + 1u) / 2u) * ((rows + 1u) / 2u));
		a.div_gcols = make_fastdiv((cols + 1u) / 2u);
	}
	// 64x64 RGBA tiles in RGBA frames (the reference CLI's default block): expand64_kernel takes the full tiles stored at powers
	// of two (clones, Nearest, the two-pass convolutions), expand_kernel the rest through the list
	const bool fast64 = a.fast32 && bw == 64 && bh == 64 && ((slot_channels == 4 && frames->channels == 4) || (slot_channels == 3 && frames->channels == 3)) &&
	                    (p.filter == 0 || et->d_xmf64 != nullptr);
	if (fast64) {
		if ((rc = ensure(h, h->xlist, (size_t)a.n_tiles * 4u)) != PXZ_OK) return rc;
		a.list = (uint32_t *)h->xlist.ptr;
		a.xmf64 = et->d_xmf64;
		a.div_gpf = make_fastdiv(cols * rows);  // (tiles per frame, tile columns)
		a.div_gcols = make_fastdiv(cols);
	}
#ifdef PXZ_STAMPS
	if ((rc = ensure(h, h->status, 256)) != PXZ_OK) return rc;  // (stamps behind the flag: pxz_debug_read_status)
	a.status = (uint32_t *)h->status.ptr;
	PXZ_HIP(h, hipMemsetAsync(a.status, 0, 8, h->stream));
#else
	if ((rc = ensure(h, h->status, 8)) != PXZ_OK) return rc;
	a.status = (uint32_t *)h->status.ptr;
	PXZ_HIP(h, hipMemsetAsync(a.status, 0, 8, h->stream));
#endif
	PXZ_HIP(h, pxz::launch_expand(a, h->n_cus, h->stream));
	return PXZ_OK;
}

int pxz_expand_frames_device(pxz_handle *h, const pxz_frames *frames, const pxz_params *params, const uint32_t *d_tile_w,
                             const uint32_t *d_tile_h, const uint8_t *d_slots, uint8_t *d_out_pixels)
{
	if (!h) return PXZ_ERR_INVALID_ARG;
	if (!frames) return fail(h, PXZ_ERR_INVALID_ARG, "null descriptor");
	return expand_launch(h, frames, frames->channels, params, d_tile_w, d_tile_h, d_slots, d_out_pixels);
}

int pxz_process_frames_device(pxz_handle *h, const pxz_frames *frames, const pxz_params *params, uint32_t filter_upscale,
                              const uint8_t *d_pixels, uint8_t *d_out_rgba, uint32_t out_pitch_bytes,
                              uint64_t out_frame_stride_bytes)
{
	if (!h) return PXZ_ERR_INVALID_ARG;
	if (!frames || !params) return fail(h, PXZ_ERR_INVALID_ARG, "null descriptor");
	if (!d_pixels || !d_out_rgba) return fail(h, PXZ_ERR_INVALID_ARG, "null device pointer");
	if (filter_upscale > 4) return fail(h, PXZ_ERR_INVALID_ARG, "filter must be 0..4");
	PXZ_HIP(h, hipSetDevice(h->device));
	// 1) get_block_variance with |x - avg| and the identity (process/mod.rs:108-111) -> reduce_image_section((v, v))
	pxz_params p = *params;
	p.mode = PXZ_MODE_SHRINK_BY;
	p.factor = 1.0f;
	int rc = check_frames(h, frames, &p);
	if (rc != PXZ_OK) return rc;
	uint32_t cols, rows;
	pxz_grid(frames->width, frames->height, p.block_w, p.block_h, &cols, &rows);
	const size_t tiles = (size_t)cols * rows * frames->n_frames, slot = (size_t)p.block_w * p.block_h * frames->channels;
	if ((rc = ensure(h, h->val, tiles * 4)) != PXZ_OK) return rc;
	if ((rc = ensure(h, h->ow, tiles * 4)) != PXZ_OK) return rc;
	if ((rc = ensure(h, h->oh, tiles * 4)) != PXZ_OK) return rc;
	if ((rc = ensure(h, h->out, tiles * slot)) != PXZ_OK) return rc;
	if ((rc = run_shrink(h, frames, &p, d_pixels, (float *)h->val.ptr, (uint32_t *)h->ow.ptr, (uint32_t *)h->oh.ptr,
	                     (uint8_t *)h->out.ptr, nullptr, nullptr, true)) != PXZ_OK)
		return rc;
	// 2) .resize(w0, h0, filter_upscale) + copy_from into the RGBA output (process/mod.rs:58-63)
	pxz_frames of{frames->width, frames->height, 4, out_pitch_bytes, frames->n_frames, 0, out_frame_stride_bytes};
	pxz_params up{params->block_w, params->block_h, 0, filter_upscale, 0.0f, 0};
	return expand_launch(h, &of, frames->channels, &up, (const uint32_t *)h->ow.ptr, (const uint32_t *)h->oh.ptr,
	                     (const uint8_t *)h->out.ptr, d_out_rgba);
}

namespace {
// tree::process for any block geometry (round 3): the levels of the recursion as lists of rectangles, one launch of
// pxz::tree_rect_kernel per level (pxz_tree.hip).  The host only learns how many tiles went on to the next level -- one
// 4-byte read-back per level.
int tree_process_rects(pxz_handle *h, const pxz_frames *frames, const pxz_params &p, uint32_t filter_upscale, float threshold,
                       uint32_t mbw, uint32_t mbh, const std::vector<std::pair<uint32_t, uint32_t>> &levels, const uint8_t *d_pixels,
                       uint8_t *d_out_rgba, uint32_t out_pitch_bytes, uint64_t src_stride, uint64_t dst_stride)
{
	constexpr uint32_t kMaxSide = 128;  // pxz_tree.hip: kTreeMaxSide
	if (levels[0].first > kMaxSide || levels[0].second > kMaxSide)
		return fail(h, PXZ_ERR_UNSUPPORTED, "tree::process on the device takes blocks up to %ux%u (got %ux%u)", kMaxSide, kMaxSide,
		            levels[0].first, levels[0].second);
	// ---- every tile size of every level, per axis (split.rs:18-19 applied level after level to the sizes of the level before)
	auto split = [](uint32_t s, uint32_t b, std::set<uint32_t> &out) {
		const uint32_t c = (s + b - 1) / b;
		if (c > 1) out.insert(b);
		out.insert(s - (c - 1) * b);
	};
	std::set<uint32_t> all[2];
	{
		std::set<uint32_t> cur[2];
		split(frames->width, levels[0].first, cur[0]);
		split(frames->height, levels[0].second, cur[1]);
		for (size_t l = 0;; ++l) {
			for (int ax = 0; ax < 2; ++ax) all[ax].insert(cur[ax].begin(), cur[ax].end());
			if (l + 1 >= levels.size()) break;
			std::set<uint32_t> next[2];
			for (uint32_t v : cur[0]) split(v, levels[l + 1].first, next[0]);
			for (uint32_t v : cur[1]) split(v, levels[l + 1].second, next[1]);
			cur[0].swap(next[0]);
			cur[1].swap(next[1]);
		}
	}
	int rc;
	const auto key = std::make_tuple(frames->width, frames->height, levels[0].first, levels[0].second, mbw, mbh, p.filter, filter_upscale);
	auto it = h->tree_tables.find(key);
	if (it == h->tree_tables.end()) {
		if (h->tree_tables.size() >= 16) {  // a cache, not a log: varying geometries (tools/fuzz_tree.py) must not grow it without bound
			PXZ_HIP(h, hipStreamSynchronize(h->stream));  // (a queued launch may still read a table)
			drop_tree_tables(h);
		}
		std::set<std::tuple<uint32_t, uint32_t, uint32_t>> pairs;  // (in, out, up)
		for (int ax = 0; ax < 2; ++ax)
			for (uint32_t sz : all[ax])
				for (uint32_t m = 1; m <= 32; ++m) {
					const uint32_t o = reduced(sz, m);
					if (o == sz) continue;
					pairs.insert(std::make_tuple(sz, o, 0u));
					pairs.insert(std::make_tuple(o, sz, 1u));
				}
		std::vector<pxz::TreeAxisEntry> dir;
		std::vector<int32_t> starts, sizes;
		std::vector<int16_t> coeffs;
		for (const auto &pr : pairs) {
			const uint32_t in = std::get<0>(pr), out = std::get<1>(pr), up = std::get<2>(pr);
			pxz::AxisWindows win;
			if (!pxz::build_axis(in, out, up ? filter_upscale : p.filter, &win, up != 0)) return fail(h, PXZ_ERR_INVALID_ARG, "unknown filter");
			pxz::TreeAxisEntry e{};
			e.in = (uint16_t)in;
			e.out = (uint16_t)out;
			e.up = (uint16_t)up;
			e.window = (uint16_t)win.window;
			e.precision = (uint32_t)win.precision;
			e.starts_off = (uint32_t)starts.size();
			e.coeff_off = (uint32_t)coeffs.size();
			for (uint32_t o = 0; o < out; ++o) {
				starts.push_back(win.starts[o]);
				sizes.push_back(win.sizes.empty() ? 0 : win.sizes[o]);
			}
			coeffs.insert(coeffs.end(), win.coeffs.begin(), win.coeffs.end());
			dir.push_back(e);
		}
		if (dir.empty()) dir.push_back(pxz::TreeAxisEntry{});
		if (starts.empty()) { starts.push_back(0); sizes.push_back(0); }
		if (coeffs.empty()) coeffs.push_back(0);
		TreeTables tt;
		tt.n_dir = (uint32_t)dir.size();
		PXZ_HIP(h, hipMalloc((void **)&tt.d_dir, dir.size() * sizeof(pxz::TreeAxisEntry)));
		PXZ_HIP(h, hipMalloc((void **)&tt.d_starts, starts.size() * 4));
		PXZ_HIP(h, hipMalloc((void **)&tt.d_sizes, sizes.size() * 4));
		PXZ_HIP(h, hipMalloc((void **)&tt.d_coeffs, coeffs.size() * 2));
		PXZ_HIP(h, hipMemcpy(tt.d_dir, dir.data(), dir.size() * sizeof(pxz::TreeAxisEntry), hipMemcpyHostToDevice));
		PXZ_HIP(h, hipMemcpy(tt.d_starts, starts.data(), starts.size() * 4, hipMemcpyHostToDevice));
		PXZ_HIP(h, hipMemcpy(tt.d_sizes, sizes.data(), sizes.size() * 4, hipMemcpyHostToDevice));
		PXZ_HIP(h, hipMemcpy(tt.d_coeffs, coeffs.data(), coeffs.size() * 2, hipMemcpyHostToDevice));
		it = h->tree_tables.emplace(key, tt).first;
	}
	const TreeTables &tt = it->second;
	// ---- level 0: the frame's own grid (split.rs:37-61)
	uint32_t cols, rows;
	pxz_grid(frames->width, frames->height, levels[0].first, levels[0].second, &cols, &rows);
	if ((uint64_t)cols * rows * frames->n_frames > 0x0fffffffull) return fail(h, PXZ_ERR_UNSUPPORTED, "too many tiles");
	uint32_t n = cols * rows * frames->n_frames;
	{
		std::vector<pxz::TreeRect> r0(n);
		size_t i = 0;
		for (uint32_t f = 0; f < frames->n_frames; ++f)
			for (uint32_t ty = 0; ty < rows; ++ty)
				for (uint32_t tx = 0; tx < cols; ++tx, ++i) {
					r0[i].x = tx * levels[0].first;
					r0[i].y = ty * levels[0].second;
					r0[i].w = (uint16_t)std::min(levels[0].first, frames->width - r0[i].x);
					r0[i].h = (uint16_t)std::min(levels[0].second, frames->height - r0[i].y);
					r0[i].frame = f;
				}
		if ((rc = ensure(h, h->tree_rects[0], (size_t)n * sizeof(pxz::TreeRect))) != PXZ_OK) return rc;
		PXZ_HIP(h, hipMemcpyAsync(h->tree_rects[0].ptr, r0.data(), (size_t)n * sizeof(pxz::TreeRect), hipMemcpyHostToDevice, h->stream));
		PXZ_HIP(h, hipStreamSynchronize(h->stream));  // (r0 leaves scope)
	}
	if ((rc = ensure(h, h->tree_count, 64)) != PXZ_OK) return rc;
	pxz::TreeRectArgs a{};
	a.src = d_pixels;
	a.dst = d_out_rgba;
	a.src_frame_stride = src_stride;
	a.dst_frame_stride = dst_stride;
	a.src_pitch = frames->pitch_bytes;
	a.dst_pitch = out_pitch_bytes;
	a.channels = frames->channels;
	a.filter_down = p.filter;
	a.filter_up = filter_upscale;
	a.dir = tt.d_dir;
	a.n_dir = tt.n_dir;
	a.starts = tt.d_starts;
	a.sizes = tt.d_sizes;
	a.coeffs = tt.d_coeffs;
	std::memcpy(a.thresholds, h->thresholds, sizeof a.thresholds);
	PXZ_HIP(h, hipMemsetAsync(h->tree_count.ptr, 0, 8, h->stream));  // [0] the next level's tile count, [1] "an axis table was missing"
	for (size_t l = 0; l < levels.size() && n != 0; ++l) {
		const bool leaf_next = l + 1 == levels.size();
		const uint32_t nbw = levels[l].first >> 1, nbh = levels[l].second >> 1;
		// every tile of this level may go on: (ceil(bw / (bw >> 1)))^2 children each at most
		const uint64_t per_tile = leaf_next ? 0 : (uint64_t)((levels[l].first + nbw - 1) / nbw) * ((levels[l].second + nbh - 1) / nbh);
		const uint64_t cap = per_tile * n;
		if (cap > 0x7fffffffull) return fail(h, PXZ_ERR_UNSUPPORTED, "too many tiles");
		DeviceBuffer &cur = h->tree_rects[l & 1], &nxt = h->tree_rects[(l + 1) & 1];
		if (!leaf_next && (rc = ensure(h, nxt, (size_t)cap * sizeof(pxz::TreeRect))) != PXZ_OK) return rc;
		PXZ_HIP(h, hipMemsetAsync(h->tree_count.ptr, 0, 4, h->stream));
		a.rects = (const pxz::TreeRect *)cur.ptr;
		a.n_rects = n;
		a.threshold = std::fabs(threshold);
		a.positive = (l == 0 ? threshold >= 0.0f : true) ? 1u : 0u;  // tree.rs:37-38: the recursion passes |threshold| on
		a.next_bw = nbw;
		a.next_bh = nbh;
		a.next_is_leaf = leaf_next ? 1u : 0u;
		a.next_rects = leaf_next ? nullptr : (pxz::TreeRect *)nxt.ptr;
		a.next_count = (uint32_t *)h->tree_count.ptr;
		a.next_capacity = (uint32_t)cap;
		PXZ_HIP(h, pxz::launch_tree_rects(a, h->stream));
		uint32_t back[2] = {0, 0};
		PXZ_HIP(h, hipMemcpyAsync(back, h->tree_count.ptr, 8, hipMemcpyDeviceToHost, h->stream));
		PXZ_HIP(h, hipStreamSynchronize(h->stream));
		if (back[1] != 0) return fail(h, PXZ_ERR_INTERNAL, "tree::process: level %zu met a tile size the axis tables do not list", l);
		if (leaf_next) break;
		if (back[0] > cap) return fail(h, PXZ_ERR_HIP, "tree::process: %u tiles for a list of %llu", back[0], (unsigned long long)cap);
		n = back[0];
	}
	return PXZ_OK;
}
}  // namespace

int pxz_tree_process_frames_device(pxz_handle *h, const pxz_frames *frames, const pxz_params *params, uint32_t filter_upscale,
                                   float threshold, uint32_t min_block_w, uint32_t min_block_h, const uint8_t *d_pixels,
                                   uint8_t *d_out_rgba, uint32_t out_pitch_bytes, uint64_t out_frame_stride_bytes)
{
	if (!h) return PXZ_ERR_INVALID_ARG;
	if (!frames || !params) return fail(h, PXZ_ERR_INVALID_ARG, "null descriptor");
	if (!d_pixels || !d_out_rgba) return fail(h, PXZ_ERR_INVALID_ARG, "null device pointer");
	if (filter_upscale > 4) return fail(h, PXZ_ERR_INVALID_ARG, "filter must be 0..4");
	if (!std::isfinite(threshold)) return fail(h, PXZ_ERR_INVALID_ARG, "threshold must be finite");
	pxz_params p = *params;
	p.mode = PXZ_MODE_SHRINK_BY;
	p.factor = 1.0f;
	int rc = check_frames(h, frames, &p);
	if (rc != PXZ_OK) return rc;
	if ((uint64_t)out_pitch_bytes < (uint64_t)frames->width * 4u) return fail(h, PXZ_ERR_INVALID_ARG, "output pitch smaller than a row");
	// (the kernels store whole RGBA pixels)
	if ((reinterpret_cast<uintptr_t>(d_out_rgba) & 3u) != 0 || (out_pitch_bytes & 3u) != 0 || (frames->n_frames > 1 && (out_frame_stride_bytes & 3u) != 0))
		return fail(h, PXZ_ERR_INVALID_ARG, "the RGBA output must be 4-byte aligned (pointer, pitch and frame stride)");
	PXZ_HIP(h, hipSetDevice(h->device));
	// the levels of the recursion (tree.rs:32-36): block sizes halve while both stay above the minimum (at least 4)
	const uint32_t mbw = min_block_w > 4u ? min_block_w : 4u, mbh = min_block_h > 4u ? min_block_h : 4u;
	std::vector<std::pair<uint32_t, uint32_t>> levels;
	for (uint32_t bw = p.block_w, bh = p.block_h; bw > mbw && bh > mbh; bw >>= 1, bh >>= 1) levels.emplace_back(bw, bh);
	const uint64_t src_stride = frames->n_frames > 1 ? frames->frame_stride_bytes : (uint64_t)frames->pitch_bytes * frames->height;
	const uint64_t dst_stride = frames->n_frames > 1 ? out_frame_stride_bytes : (uint64_t)out_pitch_bytes * frames->height;
	// A level's tiles are the children of the level before: they form that level's regular grid over the frame as long as
	// a block is exactly two of the next (the recursion splits every tile from its own corner), and the per-level passes
	// below need tiles that fit the fused kernel's LDS image.  Anything else -- 50 -> 25 -> 12, the 128-px blocks of
	// src/bin/tree.rs:6 -- goes level by level over lists of rectangles (round 3).
	bool regular = !pxz::knobs().tree_rects;
	for (size_t l = 0; l + 1 < levels.size(); ++l)
		if ((levels[l].first & 1u) || (levels[l].second & 1u)) regular = false;
	if (!levels.empty() && (levels[0].first > 64u || levels[0].second > 64u)) regular = false;
	if (!regular && !levels.empty())
		return tree_process_rects(h, frames, p, filter_upscale, threshold, mbw, mbh, levels, d_pixels, d_out_rgba, out_pitch_bytes,
		                          src_stride, dst_stride);
	pxz::TreeArgs t{};
	t.src = d_pixels;
	t.dst = d_out_rgba;
	t.src_frame_stride = src_stride;
	t.dst_frame_stride = dst_stride;
	t.src_pitch = frames->pitch_bytes;
	t.dst_pitch = out_pitch_bytes;
	t.channels = frames->channels;
	if (levels.empty()) {
		// tree.rs:34-36: the image comes back as it is (as RGBA here): a plain 2-D copy / widening of the frames
		if (frames->channels == 4) {
			for (uint32_t f = 0; f < frames->n_frames; ++f)
				PXZ_HIP(h, hipMemcpy2DAsync(d_out_rgba + (size_t)f * dst_stride, out_pitch_bytes, d_pixels + (size_t)f * src_stride, frames->pitch_bytes,
				                            (size_t)frames->width * 4u, frames->height, hipMemcpyDeviceToDevice, h->stream));
		} else {
			pxz::WidenArgs wa{d_pixels, d_out_rgba, src_stride, dst_stride, frames->pitch_bytes, out_pitch_bytes, frames->width, frames->height, frames->n_frames};
			PXZ_HIP(h, pxz::launch_widen(wa, h->stream));
		}
		return PXZ_OK;
	}
	uint32_t prev_cols = 0, prev_tpf = 0;
	size_t max_tiles = 0;
	for (auto &lv : levels) {
		uint32_t c, r;
		pxz_grid(frames->width, frames->height, lv.first, lv.second, &c, &r);
		if ((uint64_t)c * r * frames->n_frames > 0xffffffffull) return fail(h, PXZ_ERR_UNSUPPORTED, "too many tiles");
		max_tiles = std::max(max_tiles, (size_t)c * r * frames->n_frames);
	}
	// per tile: the detector's own output (the stored block value is hypot(v, v), operations.rs:154) + two sets of flags
	if ((rc = ensure(h, h->tree, 4 * max_tiles + 2 * max_tiles)) != PXZ_OK) return rc;
	float *raw_value = (float *)h->tree.ptr;
	uint8_t *open_flags[2] = {(uint8_t *)h->tree.ptr + 4 * max_tiles, (uint8_t *)h->tree.ptr + 5 * max_tiles};
	const bool whole = p.block_w <= mbw || p.block_h <= mbh;
	for (size_t l = 0; l < levels.size(); ++l) {
		const uint32_t bw = levels[l].first, bh = levels[l].second;
		uint32_t cols, rows;
		pxz_grid(frames->width, frames->height, bw, bh, &cols, &rows);
		const size_t tiles = (size_t)cols * rows * frames->n_frames, slot = (size_t)bw * bh * frames->channels;
		if ((rc = ensure(h, h->val, tiles * 4)) != PXZ_OK) return rc;
		if ((rc = ensure(h, h->ow, tiles * 4)) != PXZ_OK) return rc;
		if ((rc = ensure(h, h->oh, tiles * 4)) != PXZ_OK) return rc;
		if (!whole) {
			// get_block_variance + reduce_image_section((v, v)) of every tile of this level's grid (process/mod.rs:84-95); the
			// tiles that do not take part are discarded by the decision below
			if ((rc = ensure(h, h->out, tiles * slot)) != PXZ_OK) return rc;
			pxz_params lp = p;
			lp.block_w = bw;
			lp.block_h = bh;
			if ((rc = run_shrink(h, frames, &lp, d_pixels, (float *)h->val.ptr, (uint32_t *)h->ow.ptr, (uint32_t *)h->oh.ptr,
			                     (uint8_t *)h->out.ptr, raw_value, nullptr, true)) != PXZ_OK)
				return rc;
		} else {
			PXZ_HIP(h, hipMemsetAsync(raw_value, 0, tiles * 4, h->stream));
		}
		t.value = raw_value;
		t.tile_w = (uint32_t *)h->ow.ptr;
		t.tile_h = (uint32_t *)h->oh.ptr;
		t.parent_open = l ? open_flags[(l - 1) & 1] : nullptr;
		t.open = open_flags[l & 1];
		t.bw = bw;
		t.bh = bh;
		t.cols = cols;
		t.rows = rows;
		t.tiles_per_frame = cols * rows;
		t.n_tiles = (uint32_t)tiles;
		t.edge_w = frames->width - (cols - 1) * bw;
		t.edge_h = frames->height - (rows - 1) * bh;
		t.parent_cols = prev_cols;
		t.parent_tiles_per_frame = prev_tpf;
		t.threshold = whole ? 1.0f : std::fabs(threshold);
		t.positive = whole ? 0u : ((l == 0 ? threshold >= 0.0f : true) ? 1u : 0u);  // whole: (0 >= 1) ^ false = false: nothing is pixelised
		t.last = l + 1 == levels.size() ? 1u : 0u;
		PXZ_HIP(h, pxz::launch_tree_decide(t, h->stream));
		if (!whole) {
			pxz_frames of{frames->width, frames->height, 4, out_pitch_bytes, frames->n_frames, 0, out_frame_stride_bytes};
			pxz_params up{bw, bh, 0, filter_upscale, 0.0f, 0};
			if ((rc = expand_launch(h, &of, frames->channels, &up, (const uint32_t *)h->ow.ptr, (const uint32_t *)h->oh.ptr,
			                        (const uint8_t *)h->out.ptr, d_out_rgba, true)) != PXZ_OK)
				return rc;
		}
		prev_cols = cols;
		prev_tpf = cols * rows;
	}
	return PXZ_OK;
}

int pxz_decode_frames_device(pxz_handle *h, const pxz_frames *frames, const pxz_params *params, const uint8_t *d_files,
                             const uint64_t *d_file_offsets, float *d_block_value, uint32_t *d_tile_w, uint32_t *d_tile_h,
                             uint8_t *d_slots)
{
	if (!h) return PXZ_ERR_INVALID_ARG;
	if (!frames || !params) return fail(h, PXZ_ERR_INVALID_ARG, "null descriptor");
	pxz_params p = *params;
	p.mode = 0;
	p.factor = 0.0f;
	p.filter = 0;
	pxz_frames f = *frames;
	f.pitch_bytes = f.width * f.channels;  // only the geometry of the frames matters here
	f.frame_stride_bytes = (uint64_t)f.pitch_bytes * f.height;
	int rc = check_frames(h, &f, &p);
	if (rc != PXZ_OK) return rc;
	if (!d_files || !d_file_offsets || !d_block_value || !d_tile_w || !d_tile_h || !d_slots)
		return fail(h, PXZ_ERR_INVALID_ARG, "null device pointer");
	PXZ_HIP(h, hipSetDevice(h->device));
	uint32_t cols, rows;
	pxz_grid(f.width, f.height, p.block_w, p.block_h, &cols, &rows);
	if ((uint64_t)cols * rows * f.n_frames > 0xffffffffull) return fail(h, PXZ_ERR_UNSUPPORTED, "too many tiles");
	pxz::DecodeArgs a{};
	a.files = d_files;
	a.file_offsets = reinterpret_cast<const unsigned long long *>(d_file_offsets);
	a.value = d_block_value;
	a.tile_w = d_tile_w;
	a.tile_h = d_tile_h;
	a.slots = d_slots;
	a.width = f.width;
	a.height = f.height;
	a.bw = p.block_w;
	a.bh = p.block_h;
	a.cols = cols;
	a.rows = rows;
	a.tiles_per_frame = cols * rows;
	a.n_frames = f.n_frames;
	a.n_tiles = cols * rows * f.n_frames;
	a.channels = f.channels;
	a.slot_bytes = p.block_w * p.block_h * f.channels;
	a.edge_w = f.width - (cols - 1) * p.block_w;
	a.edge_h = f.height - (rows - 1) * p.block_h;
	const size_t dmeta_cap = h->dmeta.cap;
	if ((rc = ensure(h, h->dmeta, (size_t)a.n_tiles * 16u + 4u * pxz::qoi_bins_dwords())) != PXZ_OK) return rc;  // rec_off, rec_len, perm, the bin counters
	if (h->dmeta.cap != dmeta_cap) h->dbins_clean = nullptr;  // a new allocation: nothing in it is zero
	a.rec_off = (unsigned long long *)h->dmeta.ptr;
	a.rec_len = (uint32_t *)((uint8_t *)h->dmeta.ptr + (size_t)a.n_tiles * 8u);
	a.perm = a.rec_len + a.n_tiles;
	a.bins = a.perm + a.n_tiles;
	if ((rc = ensure(h, h->status, 256)) != PXZ_OK) return rc;  // (room for the stamps of the diagnostic build: pxz_debug_read_status)
	a.status = (uint32_t *)h->status.ptr;
	PXZ_HIP(h, hipMemsetAsync(a.status, 0, 4, h->stream));
	const bool bins_clean = h->dbins_clean == a.bins;
	h->dbins_clean = nullptr;
	PXZ_HIP(h, pxz::launch_decode(a, bins_clean, h->stream));
	h->dbins_clean = a.bins;
	return PXZ_OK;
}

int pxz_decode_file(pxz_handle *h, const uint8_t *file, size_t len, uint32_t *width, uint32_t *height, uint32_t *block_w,
                    uint32_t *block_h, uint32_t *channels, uint32_t *filter_byte, float *block_value, uint32_t *tile_w,
                    uint32_t *tile_h, uint8_t *slots)
{
	if (!h) return PXZ_ERR_INVALID_ARG;
	if (!file || !width || !height || !block_w || !block_h || !channels || !filter_byte) return fail(h, PXZ_ERR_INVALID_ARG, "null pointer");
	static const uint8_t magic[9] = {'P', 'I', 'X', 'L', 'Z', 'R', 0, 0, 2};
	if (len < 26 || std::memcmp(file, magic, 9) != 0) return fail(h, PXZ_ERR_INVALID_ARG, "not a .pixlzr v0.0.2 file");
	auto be = [&](size_t o) { return ((uint32_t)file[o] << 24) | ((uint32_t)file[o + 1] << 16) | ((uint32_t)file[o + 2] << 8) | file[o + 3]; };
	*filter_byte = file[9];
	*width = be(10);
	*height = be(14);
	*block_w = be(18);
	*block_h = be(22);
	if (*width == 0 || *height == 0 || *block_w == 0 || *block_h == 0) return fail(h, PXZ_ERR_INVALID_ARG, "empty image or block in the header");
	uint32_t cols, rows;
	pxz_grid(*width, *height, *block_w, *block_h, &cols, &rows);
	const size_t first = 26 + (size_t)rows * 4;
	if (len < first + 13 + 10) return fail(h, PXZ_ERR_INVALID_ARG, "file ends inside the first record");
	*channels = file[first + 21];  // the first record's QOI header (decode_block, mod.rs:202-242)
	if (*channels != 3 && *channels != 4) return fail(h, PXZ_ERR_INVALID_ARG, "first record has %u channels", *channels);
	if (!block_value && !tile_w && !tile_h && !slots) return PXZ_OK;  // header query
	if (!block_value || !tile_w || !tile_h || !slots) return fail(h, PXZ_ERR_INVALID_ARG, "null output pointer");
	PXZ_HIP(h, hipSetDevice(h->device));
	const size_t tiles = (size_t)cols * rows, slot = (size_t)*block_w * *block_h * *channels;
	int rc;
	if ((rc = ensure(h, h->in, len + 16)) != PXZ_OK) return rc;
	if ((rc = ensure(h, h->val, tiles * 4)) != PXZ_OK) return rc;
	if ((rc = ensure(h, h->ow, tiles * 4)) != PXZ_OK) return rc;
	if ((rc = ensure(h, h->oh, tiles * 4)) != PXZ_OK) return rc;
	if ((rc = ensure(h, h->out, tiles * slot)) != PXZ_OK) return rc;
	if ((rc = ensure(h, h->chunks, 16)) != PXZ_OK) return rc;
	const uint64_t offs[2] = {0, (uint64_t)len};
	PXZ_HIP(h, hipMemcpyAsync(h->in.ptr, file, len, hipMemcpyHostToDevice, h->stream));
	PXZ_HIP(h, hipMemcpyAsync(h->chunks.ptr, offs, 16, hipMemcpyHostToDevice, h->stream));
	PXZ_HIP(h, hipMemsetAsync(h->out.ptr, 0, tiles * slot, h->stream));
	pxz_frames f{*width, *height, *channels, *width * *channels, 1, 0, 0};
	pxz_params p{*block_w, *block_h, 0, 0, 0.0f, 0};
	rc = pxz_decode_frames_device(h, &f, &p, (const uint8_t *)h->in.ptr, (const uint64_t *)h->chunks.ptr, (float *)h->val.ptr,
	                              (uint32_t *)h->ow.ptr, (uint32_t *)h->oh.ptr, (uint8_t *)h->out.ptr);
	if (rc != PXZ_OK) return rc;
	PXZ_HIP(h, hipMemcpyAsync(block_value, h->val.ptr, tiles * 4, hipMemcpyDeviceToHost, h->stream));
	PXZ_HIP(h, hipMemcpyAsync(tile_w, h->ow.ptr, tiles * 4, hipMemcpyDeviceToHost, h->stream));
	PXZ_HIP(h, hipMemcpyAsync(tile_h, h->oh.ptr, tiles * 4, hipMemcpyDeviceToHost, h->stream));
	PXZ_HIP(h, hipMemcpyAsync(slots, h->out.ptr, tiles * slot, hipMemcpyDeviceToHost, h->stream));
	uint32_t flags = 0;
	if ((rc = pxz_decode_status(h, &flags)) != PXZ_OK) return rc;
	if (flags & 2u) return fail(h, PXZ_ERR_INVALID_ARG, "malformed .pixlzr file or record");
	return PXZ_OK;
}

int pxz_decode_status(pxz_handle *h, uint32_t *flags)
{
	if (!h || !flags) return PXZ_ERR_INVALID_ARG;
	*flags = 0;
	if (!h->status.ptr) return PXZ_OK;
	PXZ_HIP(h, hipSetDevice(h->device));
	PXZ_HIP(h, hipMemcpyAsync(flags, h->status.ptr, 4, hipMemcpyDeviceToHost, h->stream));
	PXZ_HIP(h, hipStreamSynchronize(h->stream));
	return PXZ_OK;
}

int pxz_expand_image(pxz_handle *h, uint32_t width, uint32_t height, uint32_t channels, uint32_t pitch_bytes, uint32_t block_w,
                     uint32_t block_h, uint32_t filter, const uint32_t *tile_w, const uint32_t *tile_h, const uint8_t *slots,
                     uint8_t *out_pixels)
{
	if (!h) return PXZ_ERR_INVALID_ARG;
	if (!tile_w || !tile_h || !slots || !out_pixels) return fail(h, PXZ_ERR_INVALID_ARG, "null pointer");
	pxz_frames f{width, height, channels, pitch_bytes, 1, 0, 0};
	pxz_params p{block_w, block_h, 0, filter, 0.0f, 0};
	int rc = check_frames(h, &f, &p);
	if (rc != PXZ_OK) return rc;
	PXZ_HIP(h, hipSetDevice(h->device));
	uint32_t cols, rows;
	pxz_grid(width, height, block_w, block_h, &cols, &rows);
	const size_t tiles = (size_t)cols * rows, slot = (size_t)block_w * block_h * channels;
	const size_t out_bytes = (size_t)pitch_bytes * (height - 1) + (size_t)width * channels;
	if ((rc = ensure(h, h->ow, tiles * 4)) != PXZ_OK) return rc;
	if ((rc = ensure(h, h->oh, tiles * 4)) != PXZ_OK) return rc;
	if ((rc = ensure(h, h->out, tiles * slot)) != PXZ_OK) return rc;
	if ((rc = ensure(h, h->in, out_bytes)) != PXZ_OK) return rc;
	PXZ_HIP(h, hipMemcpyAsync(h->ow.ptr, tile_w, tiles * 4, hipMemcpyHostToDevice, h->stream));
	PXZ_HIP(h, hipMemcpyAsync(h->oh.ptr, tile_h, tiles * 4, hipMemcpyHostToDevice, h->stream));
	PXZ_HIP(h, hipMemcpyAsync(h->out.ptr, slots, tiles * slot, hipMemcpyHostToDevice, h->stream));
	PXZ_HIP(h, hipMemsetAsync(h->in.ptr, 0, out_bytes, h->stream));  // row padding of a pitched image stays zero
	rc = pxz_expand_frames_device(h, &f, &p, (const uint32_t *)h->ow.ptr, (const uint32_t *)h->oh.ptr, (const uint8_t *)h->out.ptr,
	                              (uint8_t *)h->in.ptr);
	if (rc != PXZ_OK) return rc;
	PXZ_HIP(h, hipMemcpyAsync(out_pixels, h->in.ptr, out_bytes, hipMemcpyDeviceToHost, h->stream));
	uint32_t bad = 0;
	if ((rc = pxz_decode_status(h, &bad)) != PXZ_OK) return rc;
	if (bad) return fail(h, PXZ_ERR_INVALID_ARG, "a tile's stored size is zero or larger than its place in the image");
	return PXZ_OK;
}

int pxz_shrink_image(pxz_handle *h, const uint8_t *pixels, uint32_t width, uint32_t height, uint32_t channels,
                     uint32_t pitch_bytes, uint32_t block_w, uint32_t block_h, uint32_t mode, uint32_t filter,
                     float factor, float *block_value, uint32_t *out_w, uint32_t *out_h, uint8_t *out_pixels)
{
	if (!h) return PXZ_ERR_INVALID_ARG;
	if (!pixels || !block_value || !out_w || !out_h) return fail(h, PXZ_ERR_INVALID_ARG, "null pointer");
	pxz_frames f{width, height, channels, pitch_bytes, 1, 0, 0};
	pxz_params p{block_w, block_h, mode, filter, factor, 0};
	int rc = check_frames(h, &f, &p);
	if (rc != PXZ_OK) return rc;
	if (channels == 4 && host_image_has_transparency(pixels, width, height, pitch_bytes)) p.reserved |= PXZ_HINT_TRANSPARENCY;
	PXZ_HIP(h, hipSetDevice(h->device));
	uint32_t cols, rows;
	pxz_grid(width, height, block_w, block_h, &cols, &rows);
	const size_t tiles = (size_t)cols * rows;
	const size_t slot = (size_t)block_w * block_h * channels;
	if ((rc = ensure(h, h->val, tiles * 4)) != PXZ_OK) return rc;
	if ((rc = ensure(h, h->ow, tiles * 4)) != PXZ_OK) return rc;
	if ((rc = ensure(h, h->oh, tiles * 4)) != PXZ_OK) return rc;
	if (out_pixels && (rc = ensure(h, h->out, tiles * slot)) != PXZ_OK) return rc;
	if ((rc = upload_image(h, pixels, width, height, channels, pitch_bytes, &f.pitch_bytes)) != PXZ_OK) return rc;
	rc = pxz_shrink_frames_device(h, &f, &p, (const uint8_t *)h->in.ptr, (float *)h->val.ptr, (uint32_t *)h->ow.ptr,
	                              (uint32_t *)h->oh.ptr, out_pixels ? (uint8_t *)h->out.ptr : nullptr);
	if (rc != PXZ_OK) return rc;
	PXZ_HIP(h, hipMemcpyAsync(block_value, h->val.ptr, tiles * 4, hipMemcpyDeviceToHost, h->stream));
	PXZ_HIP(h, hipMemcpyAsync(out_w, h->ow.ptr, tiles * 4, hipMemcpyDeviceToHost, h->stream));
	PXZ_HIP(h, hipMemcpyAsync(out_h, h->oh.ptr, tiles * 4, hipMemcpyDeviceToHost, h->stream));
	if (out_pixels) PXZ_HIP(h, hipMemcpyAsync(out_pixels, h->out.ptr, tiles * slot, hipMemcpyDeviceToHost, h->stream));
	PXZ_HIP(h, hipStreamSynchronize(h->stream));
	return PXZ_OK;
}

int pxz_shrink_image_packed(pxz_handle *h, const uint8_t *pixels, uint32_t width, uint32_t height, uint32_t channels,
                            uint32_t pitch_bytes, uint32_t block_w, uint32_t block_h, uint32_t mode, uint32_t filter,
                            float factor, float *block_value, uint32_t *out_w, uint32_t *out_h, uint64_t *packed_len)
{
	if (!h) return PXZ_ERR_INVALID_ARG;
	h->packed_len = 0;
	if (!pixels || !block_value || !out_w || !out_h || !packed_len) return fail(h, PXZ_ERR_INVALID_ARG, "null pointer");
	*packed_len = 0;
	pxz_frames f{width, height, channels, pitch_bytes, 1, 0, 0};
	pxz_params p{block_w, block_h, mode, filter, factor, 0};
	int rc = check_frames(h, &f, &p);
	if (rc != PXZ_OK) return rc;
	if (channels == 4 && host_image_has_transparency(pixels, width, height, pitch_bytes)) p.reserved |= PXZ_HINT_TRANSPARENCY;
	PXZ_HIP(h, hipSetDevice(h->device));
	uint32_t cols, rows;
	pxz_grid(width, height, block_w, block_h, &cols, &rows);
	const size_t tiles = (size_t)cols * rows;
	const size_t slot = (size_t)block_w * block_h * channels;
	const size_t most = (size_t)width * height * channels;  // no tile grows
	if ((rc = ensure(h, h->val, tiles * 4)) != PXZ_OK) return rc;
	if ((rc = ensure(h, h->ow, tiles * 4)) != PXZ_OK) return rc;
	if ((rc = ensure(h, h->oh, tiles * 4)) != PXZ_OK) return rc;
	if ((rc = ensure(h, h->out, tiles * slot)) != PXZ_OK) return rc;
	if ((rc = ensure(h, h->pk, most)) != PXZ_OK) return rc;
	if ((rc = ensure(h, h->pkoff, (tiles + 1) * 8)) != PXZ_OK) return rc;
	if ((rc = upload_image(h, pixels, width, height, channels, pitch_bytes, &f.pitch_bytes)) != PXZ_OK) return rc;
	rc = pxz_shrink_frames_device(h, &f, &p, (const uint8_t *)h->in.ptr, (float *)h->val.ptr, (uint32_t *)h->ow.ptr,
	                              (uint32_t *)h->oh.ptr, (uint8_t *)h->out.ptr);
	if (rc != PXZ_OK) return rc;
	rc = pxz_pack_tiles_device(h, (uint32_t)tiles, channels, (uint32_t)slot, (const uint32_t *)h->ow.ptr, (const uint32_t *)h->oh.ptr,
	                           (const uint8_t *)h->out.ptr, (uint64_t *)h->pkoff.ptr, (uint8_t *)h->pk.ptr, most);
	if (rc != PXZ_OK) return rc;
	uint64_t total = 0;
	PXZ_HIP(h, hipMemcpyAsync(block_value, h->val.ptr, tiles * 4, hipMemcpyDeviceToHost, h->stream));
	PXZ_HIP(h, hipMemcpyAsync(out_w, h->ow.ptr, tiles * 4, hipMemcpyDeviceToHost, h->stream));
	PXZ_HIP(h, hipMemcpyAsync(out_h, h->oh.ptr, tiles * 4, hipMemcpyDeviceToHost, h->stream));
	PXZ_HIP(h, hipMemcpyAsync(&total, (const uint64_t *)h->pkoff.ptr + tiles, 8, hipMemcpyDeviceToHost, h->stream));
	PXZ_HIP(h, hipStreamSynchronize(h->stream));
	if (total > most) return fail(h, PXZ_ERR_HIP, "packed stream longer than the image (%llu > %zu)", (unsigned long long)total, most);
	h->packed_len = total;
	*packed_len = total;
	return PXZ_OK;
}

// pxz_shrink_images / pxz_shrink_images_packed: pxz_shrink_image[_packed] over a list of equally sized host images, as a
// three-stage pipeline over three sets of device buffers -- the upload of image k + 1, the kernels of image k and the
// download of image k - 1 run at the same time (two copy streams driven by two helper threads: a copy from or to pageable
// host memory blocks its caller), so the list costs about one PCIe direction per image instead of the sum of both.
static int shrink_images_impl(pxz_handle *h, const uint8_t *const *pixels, uint32_t n_images, uint32_t width, uint32_t height,
                              uint32_t channels, uint32_t pitch_bytes, uint32_t block_w, uint32_t block_h, uint32_t mode,
                              uint32_t filter, float factor, float *const *block_value, uint32_t *const *out_w,
                              uint32_t *const *out_h, uint8_t *const *out_pixels, bool packed, uint64_t capacity,
                              uint64_t *packed_len)
{
	if (!h) return PXZ_ERR_INVALID_ARG;
	if (!pixels || !block_value || !out_w || !out_h || n_images == 0 || (packed && (!out_pixels || !packed_len)))
		return fail(h, PXZ_ERR_INVALID_ARG, "null pointer or empty list");
	for (uint32_t k = 0; k < n_images; ++k)
		if (!pixels[k] || !block_value[k] || !out_w[k] || !out_h[k] || (packed && !out_pixels[k]))
			return fail(h, PXZ_ERR_INVALID_ARG, "null pointer in the list (image %u)", k);
	pxz_frames f{width, height, channels, pitch_bytes, 1, 0, 0};
	pxz_params p{block_w, block_h, mode, filter, factor, 0};
	int rc = check_frames(h, &f, &p);
	if (rc != PXZ_OK) return rc;
	PXZ_HIP(h, hipSetDevice(h->device));
	uint32_t cols, rows;
	pxz_grid(width, height, block_w, block_h, &cols, &rows);
	const size_t tiles = (size_t)cols * rows, slot = (size_t)block_w * block_h * channels;
	const size_t most = (size_t)width * height * channels;  // no tile grows
	const size_t row_bytes = (size_t)width * channels;
	uint32_t dp = pitch_bytes;
	if (channels == 4 && (pitch_bytes & 15u) != 0) dp = (uint32_t)((row_bytes + 15u) & ~(size_t)15u);
	const size_t in_bytes = (size_t)dp * (height - 1) + row_bytes;
	const bool want_px = packed || out_pixels != nullptr;
	constexpr int R = pxz_handle::kRing;
	for (int i = 0; i < R; ++i) {
		if ((rc = ensure(h, h->ring_in[i], in_bytes)) != PXZ_OK) return rc;
		if ((rc = ensure(h, h->ring_val[i], tiles * 4)) != PXZ_OK) return rc;
		if ((rc = ensure(h, h->ring_ow[i], tiles * 4)) != PXZ_OK) return rc;
		if ((rc = ensure(h, h->ring_oh[i], tiles * 4)) != PXZ_OK) return rc;
		if (want_px && (rc = ensure(h, h->ring_out[i], tiles * slot)) != PXZ_OK) return rc;
		if (packed && (rc = ensure(h, h->ring_pk[i], most)) != PXZ_OK) return rc;
		if (packed && (rc = ensure(h, h->ring_pkoff[i], (tiles + 1) * 8)) != PXZ_OK) return rc;
	}
	hipStream_t up = nullptr, down = nullptr;
	PXZ_HIP(h, hipStreamCreateWithFlags(&up, hipStreamNonBlocking));
	if (hipStreamCreateWithFlags(&down, hipStreamNonBlocking) != hipSuccess) {
		(void)hipStreamDestroy(up);
		return fail(h, PXZ_ERR_HIP, "hipStreamCreateWithFlags failed");
	}
	std::mutex m;
	std::condition_variable cv;
	uint32_t uploaded = 0, computed = 0, downloaded = 0;
	hipError_t copy_error = hipSuccess;
	bool stop = false;
	const int device = h->device;

	auto upload_fn = [&] {
		(void)hipSetDevice(device);
		for (uint32_t k = 0; k < n_images; ++k) {
			{
				std::unique_lock<std::mutex> lk(m);
				cv.wait(lk, [&] { return stop || k < downloaded + (uint32_t)R; });  // its buffer set is free again
				if (stop) return;
			}
			void *dst = h->ring_in[k % R].ptr;
			hipError_t e = dp == pitch_bytes ? hipMemcpyAsync(dst, pixels[k], in_bytes, hipMemcpyHostToDevice, up)
			                                 : hipMemcpy2DAsync(dst, dp, pixels[k], pitch_bytes, row_bytes, height, hipMemcpyHostToDevice, up);
			if (e == hipSuccess) e = hipStreamSynchronize(up);
			std::lock_guard<std::mutex> lk(m);
			if (e != hipSuccess) {
				copy_error = e;
				stop = true;
			} else {
				uploaded = k + 1;
			}
			cv.notify_all();
			if (stop) return;
		}
	};
	auto download_fn = [&] {
		(void)hipSetDevice(device);
		for (uint32_t k = 0; k < n_images; ++k) {
			{
				std::unique_lock<std::mutex> lk(m);
				cv.wait(lk, [&] { return stop || computed > k; });
				if (stop) return;
			}
			const int i = (int)(k % R);
			hipError_t e = hipMemcpyAsync(block_value[k], h->ring_val[i].ptr, tiles * 4, hipMemcpyDeviceToHost, down);
			if (e == hipSuccess) e = hipMemcpyAsync(out_w[k], h->ring_ow[i].ptr, tiles * 4, hipMemcpyDeviceToHost, down);
			if (e == hipSuccess) e = hipMemcpyAsync(out_h[k], h->ring_oh[i].ptr, tiles * 4, hipMemcpyDeviceToHost, down);
			if (e == hipSuccess && packed) {
				uint64_t total = 0;
				e = hipMemcpyAsync(&total, (const uint64_t *)h->ring_pkoff[i].ptr + tiles, 8, hipMemcpyDeviceToHost, down);
				if (e == hipSuccess) e = hipStreamSynchronize(down);
				packed_len[k] = total;
				if (e == hipSuccess && total <= capacity && total <= most && total)
					e = hipMemcpyAsync(out_pixels[k], h->ring_pk[i].ptr, total, hipMemcpyDeviceToHost, down);
			} else if (e == hipSuccess && out_pixels && out_pixels[k]) {
				e = hipMemcpyAsync(out_pixels[k], h->ring_out[i].ptr, tiles * slot, hipMemcpyDeviceToHost, down);
			}
			if (e == hipSuccess) e = hipStreamSynchronize(down);
			std::lock_guard<std::mutex> lk(m);
			if (e != hipSuccess) {
				copy_error = e;
				stop = true;
			} else {
				downloaded = k + 1;
			}
			cv.notify_all();
			if (stop) return;
		}
	};
	// (a thread that cannot be started must not take the process down: nothing may be thrown across the C boundary)
	std::thread uploader, downloader;
	try {
		uploader = std::thread(upload_fn);
		downloader = std::thread(download_fn);
	} catch (const std::system_error &) {
		{
			std::lock_guard<std::mutex> lk(m);
			stop = true;
		}
		cv.notify_all();
		if (uploader.joinable()) uploader.join();
		(void)hipStreamDestroy(up);
		(void)hipStreamDestroy(down);
		return fail(h, PXZ_ERR_HIP, "could not start the copy threads of the pipelined boundary");
	}
	// this thread: the kernels, on the handle's stream
	rc = PXZ_OK;
	for (uint32_t k = 0; k < n_images && rc == PXZ_OK; ++k) {
		{
			std::unique_lock<std::mutex> lk(m);
			cv.wait(lk, [&] { return stop || uploaded > k; });
			if (stop) break;
		}
		const int i = (int)(k % R);
		pxz_frames fk{width, height, channels, dp, 1, 0, 0};
		pxz_params pk = p;
		if (channels == 4 && host_image_has_transparency(pixels[k], width, height, pitch_bytes)) pk.reserved |= PXZ_HINT_TRANSPARENCY;
		rc = pxz_shrink_frames_device(h, &fk, &pk, (const uint8_t *)h->ring_in[i].ptr, (float *)h->ring_val[i].ptr,
		                              (uint32_t *)h->ring_ow[i].ptr, (uint32_t *)h->ring_oh[i].ptr,
		                              want_px ? (uint8_t *)h->ring_out[i].ptr : nullptr);
		if (rc == PXZ_OK && packed)
			rc = pxz_pack_tiles_device(h, (uint32_t)tiles, channels, (uint32_t)slot, (const uint32_t *)h->ring_ow[i].ptr,
			                           (const uint32_t *)h->ring_oh[i].ptr, (const uint8_t *)h->ring_out[i].ptr,
			                           (uint64_t *)h->ring_pkoff[i].ptr, (uint8_t *)h->ring_pk[i].ptr, most);
		if (rc == PXZ_OK && hipStreamSynchronize(h->stream) != hipSuccess) rc = fail(h, PXZ_ERR_HIP, "hipStreamSynchronize failed");
		std::lock_guard<std::mutex> lk(m);
		if (rc != PXZ_OK) stop = true;
		else computed = k + 1;
		cv.notify_all();
	}
	{
		std::unique_lock<std::mutex> lk(m);
		cv.wait(lk, [&] { return stop || downloaded == n_images; });
		stop = true;  // (lets a helper that is still waiting go)
		cv.notify_all();
	}
	uploader.join();
	downloader.join();
	(void)hipStreamDestroy(up);
	(void)hipStreamDestroy(down);
	if (rc != PXZ_OK) return rc;
	if (copy_error != hipSuccess) return fail(h, PXZ_ERR_HIP, "host copy failed: %s", hipGetErrorString(copy_error));
	if (packed)
		for (uint32_t k = 0; k < n_images; ++k)
			if (packed_len[k] > capacity) return fail(h, PXZ_ERR_BUFFER_TOO_SMALL, "image %u: packed stream of %llu bytes, capacity %llu", k,
			                                          (unsigned long long)packed_len[k], (unsigned long long)capacity);
	return PXZ_OK;
}

int pxz_shrink_images(pxz_handle *h, const uint8_t *const *pixels, uint32_t n_images, uint32_t width, uint32_t height,
                      uint32_t channels, uint32_t pitch_bytes, uint32_t block_w, uint32_t block_h, uint32_t mode, uint32_t filter,
                      float factor, float *const *block_value, uint32_t *const *out_w, uint32_t *const *out_h,
                      uint8_t *const *out_pixels)
{
	return shrink_images_impl(h, pixels, n_images, width, height, channels, pitch_bytes, block_w, block_h, mode, filter, factor,
	                          block_value, out_w, out_h, out_pixels, false, 0, nullptr);
}

int pxz_shrink_images_packed(pxz_handle *h, const uint8_t *const *pixels, uint32_t n_images, uint32_t width, uint32_t height,
                             uint32_t channels, uint32_t pitch_bytes, uint32_t block_w, uint32_t block_h, uint32_t mode,
                             uint32_t filter, float factor, float *const *block_value, uint32_t *const *out_w,
                             uint32_t *const *out_h, uint8_t *const *packed, uint64_t packed_capacity, uint64_t *packed_len)
{
	return shrink_images_impl(h, pixels, n_images, width, height, channels, pitch_bytes, block_w, block_h, mode, filter, factor,
	                          block_value, out_w, out_h, packed, true, packed_capacity, packed_len);
}

int pxz_fetch_packed(pxz_handle *h, uint8_t *dst, uint64_t capacity)
{
	if (!h) return PXZ_ERR_INVALID_ARG;
	if (!dst && h->packed_len != 0) return fail(h, PXZ_ERR_INVALID_ARG, "null pointer");
	if (capacity < h->packed_len) return fail(h, PXZ_ERR_INVALID_ARG, "capacity %llu < packed length %llu",
	                                          (unsigned long long)capacity, (unsigned long long)h->packed_len);
	if (h->packed_len == 0) return PXZ_OK;
	PXZ_HIP(h, hipSetDevice(h->device));
	PXZ_HIP(h, hipMemcpyAsync(dst, h->pk.ptr, h->packed_len, hipMemcpyDeviceToHost, h->stream));
	PXZ_HIP(h, hipStreamSynchronize(h->stream));
	return PXZ_OK;
}

int pxz_oklab_pixels_device(pxz_handle *h, const uint8_t *d_rgba, uint32_t n_pixels, float *d_laba)
{
	if (!h) return PXZ_ERR_INVALID_ARG;
	if (!d_rgba || !d_laba || n_pixels == 0) return fail(h, PXZ_ERR_INVALID_ARG, "null pointer / no pixels");
	if ((reinterpret_cast<uintptr_t>(d_rgba) & 3u) || (reinterpret_cast<uintptr_t>(d_laba) & 15u))
		return fail(h, PXZ_ERR_INVALID_ARG, "pixels must be 4-byte aligned, the output 16-byte aligned");
	PXZ_HIP(h, hipSetDevice(h->device));
	PXZ_HIP(h, pxz::launch_oklab_pixels(reinterpret_cast<const uint32_t *>(d_rgba), n_pixels, d_laba, h->n_cus, h->stream));
	return PXZ_OK;
}

int pxz_pack_tiles_device(pxz_handle *h, uint32_t n_tiles, uint32_t channels, uint32_t slot_bytes,
                          const uint32_t *d_tile_w, const uint32_t *d_tile_h, const uint8_t *d_slots,
                          uint64_t *d_offsets, uint8_t *d_packed, uint64_t packed_capacity)
{
	if (!h) return PXZ_ERR_INVALID_ARG;
	if (!d_tile_w || !d_tile_h || !d_slots || !d_offsets || !d_packed || n_tiles == 0)
		return fail(h, PXZ_ERR_INVALID_ARG, "null pointer / no tiles");
	if (channels != 3 && channels != 4) return fail(h, PXZ_ERR_INVALID_ARG, "channels must be 3 or 4");
	PXZ_HIP(h, hipSetDevice(h->device));
	const uint32_t n_chunks = (n_tiles + 4095u) / 4096u;
	if ((uint64_t)slot_bytes * 4096ull > 0xffffffffull) return fail(h, PXZ_ERR_UNSUPPORTED, "slot too large for the chunked scan");
	int rc = ensure(h, h->chunks, (size_t)n_chunks * 8u);
	if (rc != PXZ_OK) return rc;
	pxz::PackArgs a{d_tile_w, d_tile_h, nullptr, d_slots, (unsigned long long *)d_offsets, (unsigned long long *)h->chunks.ptr,
	                d_packed, packed_capacity, n_tiles, n_chunks, channels, slot_bytes};
	PXZ_HIP(h, pxz::launch_pack(a, h->stream));
	return PXZ_OK;
}

int pxz_encode_frames_device(pxz_handle *h, const pxz_frames *frames, const pxz_params *params, uint32_t filter_byte,
                             const float *d_block_value, const uint32_t *d_tile_w, const uint32_t *d_tile_h,
                             const uint8_t *d_slots, uint8_t *d_out, uint64_t out_capacity, uint64_t *d_file_offsets)
{
	if (!h) return PXZ_ERR_INVALID_ARG;
	if (!frames || !params || !d_block_value || !d_tile_w || !d_tile_h || !d_slots || !d_out || !d_file_offsets)
		return fail(h, PXZ_ERR_INVALID_ARG, "null pointer");
	if (frames->channels != 3 && frames->channels != 4) return fail(h, PXZ_ERR_INVALID_ARG, "channels must be 3 or 4");
	if (params->block_w == 0 || params->block_h == 0 || frames->n_frames == 0) return fail(h, PXZ_ERR_INVALID_ARG, "bad geometry");
	PXZ_HIP(h, hipSetDevice(h->device));
	uint32_t cols, rows;
	if (pxz_grid(frames->width, frames->height, params->block_w, params->block_h, &cols, &rows) != PXZ_OK)
		return fail(h, PXZ_ERR_UNSUPPORTED, "image sides above 2^24 are not supported");
	const uint64_t tiles64 = (uint64_t)cols * rows * frames->n_frames;
	if (tiles64 > 0xffffffffull) return fail(h, PXZ_ERR_UNSUPPORTED, "too many tiles");
	const uint32_t n_tiles = (uint32_t)tiles64, c = frames->channels;
	const uint32_t slot = params->block_w * params->block_h * c;
	if ((slot & 15u) != 0 && c == 4) return fail(h, PXZ_ERR_UNSUPPORTED, "RGBA slots must be 16-byte multiples");
	// a record: 13 + 10 + at most (channels + 1) bytes per pixel + the 8-byte end marker; a chunk of the record scan holds 4096 of them
	const uint32_t px = params->block_w * params->block_h;
	const uint32_t stride = 23u + px * (c + 1u) + 8u;
	const uint32_t n_chunks = (n_tiles + 4095u) / 4096u;
	if ((uint64_t)stride * 4096ull > 0xffffffffull) return fail(h, PXZ_ERR_UNSUPPORTED, "tile too large for the chunked scan");
	int rc;
	// the encoder's units (pxz_stream.hip)
	if ((rc = ensure(h, h->qscratch, pxz::qoi_scratch_bytes(n_tiles, px, c))) != PXZ_OK) return rc;
	// perm | rec_len | bins as u32, then offsets (n+1) and chunk totals as u64
	const size_t meta_u32 = (size_t)n_tiles * 2 + pxz::qoi_bins_dwords();
	const size_t meta_bytes = ((meta_u32 * 4 + 7) & ~(size_t)7) + ((size_t)n_tiles + 1 + n_chunks) * 8;
	const size_t qmeta_cap = h->qmeta.cap;
	if ((rc = ensure(h, h->qmeta, meta_bytes)) != PXZ_OK) return rc;
	if (h->qmeta.cap != qmeta_cap) h->qbins_clean = nullptr;  // a new allocation: nothing in it is zero
	uint32_t *m32 = (uint32_t *)h->qmeta.ptr;
	unsigned long long *m64 = (unsigned long long *)((uint8_t *)h->qmeta.ptr + ((meta_u32 * 4 + 7) & ~(size_t)7));
	pxz::QoiArgs a{};
	a.slots = d_slots;
	a.w = d_tile_w;
	a.h = d_tile_h;
	a.value = d_block_value;
	a.perm = m32;
	a.rec_len = m32 + n_tiles;
	a.bins = m32 + 2 * (size_t)n_tiles;
	a.scratch = (uint8_t *)h->qscratch.ptr;
	a.offsets = m64;
	a.chunk_totals = m64 + n_tiles + 1;
	a.out = d_out;
	a.file_offsets = (unsigned long long *)d_file_offsets;
	a.capacity = out_capacity;
	a.n_tiles = n_tiles;
	a.n_chunks = n_chunks;
	a.tiles_per_frame = cols * rows;
	a.cols = cols;
	a.rows = rows;
	a.channels = c;
	a.slot_bytes = slot;
	a.hdr_bytes = 26u + rows * 4u;
	a.width = frames->width;
	a.height = frames->height;
	a.bw = params->block_w;
	a.bh = params->block_h;
	a.filter_byte = filter_byte;
	// (a launch leaves the binning counters zeroed for the next one on the same buffer)
	const bool bins_clean = h->qbins_clean == a.bins;
	h->qbins_clean = nullptr;
	PXZ_HIP(h, pxz::launch_qoi(a, bins_clean, h->n_cus, h->stream));
	h->qbins_clean = a.bins;
	return PXZ_OK;
}

int pxz_synth_frames_device(pxz_handle *h, const pxz_frames *frames, uint8_t *d_pixels, uint32_t first_frame_index,
                            uint32_t dist)
{
	if (!h) return PXZ_ERR_INVALID_ARG;
	if (!frames || !d_pixels || dist > 3) return fail(h, PXZ_ERR_INVALID_ARG, "bad synth arguments");
	if (frames->channels != 3 && frames->channels != 4) return fail(h, PXZ_ERR_INVALID_ARG, "channels must be 3 or 4");
	if (frames->height > 65535u || frames->n_frames > 65535u) return fail(h, PXZ_ERR_UNSUPPORTED, "frame too tall for the synth grid");
	PXZ_HIP(h, hipSetDevice(h->device));
	pxz::SynthArgs s{d_pixels,
	                 frames->n_frames > 1 ? frames->frame_stride_bytes : (uint64_t)frames->pitch_bytes * frames->height,
	                 frames->pitch_bytes, frames->width, frames->height, frames->channels, frames->n_frames,
	                 first_frame_index, dist};
	PXZ_HIP(h, pxz::launch_synth(s, h->stream));
	return PXZ_OK;
}

int pxz_axis_table(uint32_t in_size, uint32_t out_size, uint32_t filter, int32_t *starts, int32_t *sizes,
                   int16_t *coeffs, int32_t *window, int32_t *precision)
{
	if (in_size == 0 || out_size == 0 || filter > 4) return PXZ_ERR_INVALID_ARG;
	pxz::AxisWindows w;
	if (!pxz::build_axis(in_size, out_size, filter, &w, out_size > in_size)) return PXZ_ERR_INVALID_ARG;
	if (window) *window = w.window;
	if (precision) *precision = w.precision;
	if (starts) std::memcpy(starts, w.starts.data(), sizeof(int32_t) * out_size);
	if (sizes) std::memcpy(sizes, w.sizes.data(), sizeof(int32_t) * out_size);
	if (coeffs && !w.coeffs.empty()) std::memcpy(coeffs, w.coeffs.data(), sizeof(int16_t) * w.coeffs.size());
	return PXZ_OK;
}

// diagnostic builds only: the stamps of expand_kernel (behind its status word)
int pxz_debug_read_status(pxz_handle *h, void *dst, size_t offset, size_t bytes)
{
	if (!h || !dst || !h->status.ptr || offset + bytes > h->status.cap) return PXZ_ERR_INVALID_ARG;
	PXZ_HIP(h, hipMemcpy(dst, (const uint8_t *)h->status.ptr + offset, bytes, hipMemcpyDeviceToHost));
	return PXZ_OK;
}

// diagnostic builds only: copy bytes out of the handle's worklist/stamp buffer
int pxz_debug_read_work(pxz_handle *h, void *dst, size_t offset, size_t bytes)
{
	if (!h || !dst || !h->work.ptr || offset + bytes > h->work.cap) return PXZ_ERR_INVALID_ARG;
	PXZ_HIP(h, hipMemcpy(dst, (const uint8_t *)h->work.ptr + offset, bytes, hipMemcpyDeviceToHost));
	return PXZ_OK;
}

int pxz_enable_timing(pxz_handle *h, int on)
{
	if (!h) return PXZ_ERR_INVALID_ARG;
	h->timing = on != 0;
	h->timing_stride = on > 1 ? (uint32_t)on : 1u;  // on = n > 1: every n-th step only (an event costs ~2 us of stream time)
	h->timing_count = 0;
	h->events_used = 0;
	return PXZ_OK;
}

// average over the launches recorded since the last call (or since enable)
int pxz_last_kernel_ms(pxz_handle *h, float *ms)
{
	if (!h || !ms) return PXZ_ERR_INVALID_ARG;
	if (!h->timing || h->events_used == 0) return fail(h, PXZ_ERR_INVALID_ARG, "no timed launches recorded");
	PXZ_HIP(h, hipEventSynchronize(h->events[h->events_used - 1].second));
	double total = 0.0;
	for (size_t i = 0; i < h->events_used; ++i) {
		float t = 0.f;
		PXZ_HIP(h, hipEventElapsedTime(&t, h->events[i].first, h->events[i].second));
		total += t;
	}
	*ms = (float)(total / (double)h->events_used);
	h->events_used = 0;
	return PXZ_OK;
}

// the same for the FIRST kernel of each recorded step alone (shrink32/64/16_kernel, or oklab_kernel in shrink_by
// steps): what a per-kernel profile shows for it.  Call before pxz_last_kernel_ms (which resets the record).
int pxz_last_first_kernel_ms(pxz_handle *h, float *ms)
{
	if (!h || !ms) return PXZ_ERR_INVALID_ARG;
	if (!h->timing || h->events_used == 0) return fail(h, PXZ_ERR_INVALID_ARG, "no timed launches recorded");
	PXZ_HIP(h, hipEventSynchronize(h->events[h->events_used - 1].second));
	double total = 0.0;
	for (size_t i = 0; i < h->events_used; ++i) {
		float t = 0.f;
		PXZ_HIP(h, hipEventElapsedTime(&t, h->events[i].first, h->mid_events[i]));
		total += t;
	}
	*ms = (float)(total / (double)h->events_used);
	return PXZ_OK;
}

int pxz_handle_state(pxz_handle *h, uint32_t state[4])
{
	if (!h || !state) return PXZ_ERR_INVALID_ARG;
	state[0] = h->host_stats ? const_cast<volatile uint32_t *>(h->host_stats)[0] : 0u;
	state[1] = h->host_stats ? const_cast<volatile uint32_t *>(h->host_stats)[1] : 0xffffffffu;
	state[2] = h->last_alpha_kernel;
	state[3] = h->last_alpha_first;
	return PXZ_OK;
}

}  // extern "C"
