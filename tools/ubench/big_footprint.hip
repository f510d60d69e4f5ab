// tools/ubench/big_footprint.hip -- where does the decode rate go when ONE launch covers several GiB?
//
// Round 5 left this unexplained: the whole 32768^2 BC1 image in one launch ran at 0.75 of the HBM peak (805 us) while a 32768 x 8192
// band of the same kernel, same width and pitch, ran at 0.94 (161 us), and 16384^2 at 0.94.  This program separates the candidates
// on hipMalloc'ed buffers, through the library's device entry (no torch, no Python):
//   sweep     W = 32768, H = 8192 ... 32768: the whole image in one call / in K back-to-back calls on bands of `band` rows / a write-only
//             fill of the same shape / a band-sized call at each quarter of the big allocation
//   mock      a BC1-shaped kernel of this file (8-byte load, four 16-byte nt stores per lane) with the grid and the footprint decoupled:
//             262144 workgroups over 1 GiB (addresses wrapped) against 65536 over 1 GiB and 262144 over 4 GiB; and with the workgroup ->
//             tile map permuted (XCD-major bands) to see whether the ORDER in which a long launch walks the image matters
//   pmc FMT H [bands]   a few launches of one configuration, for `rocprofv3 --pmc` passes
// Output: one JSON object per line on stdout.  Measurement tooling: nothing here is linked into libdetexhip.so.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "../../include/detexhip.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(2); } } while (0)

typedef uint32_t v4 __attribute__((ext_vector_type(4)));
typedef uint32_t v2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
	x *= 0x9E3779B1u; x ^= x >> 15; x *= 0x85EBCA77u; x ^= x >> 13; x *= 0xC2B2AE3Du; x ^= x >> 16;
	return x;
}
__global__ void fill_random(uint32_t *dst, uint64_t n_dwords, uint32_t seed) {
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_dwords; i += (uint64_t)gridDim.x * blockDim.x)
		dst[i] = hash32((uint32_t)i * 2654435761u + (uint32_t)(i >> 32) + seed);
}

// MAP 0: workgroup b = tile b.  MAP 1: tiles wrapped into the first `param` tiles (footprint decoupled from the grid).
// MAP 2: the grid cut into chunks of `param` consecutive tiles, the chunk's tiles dealt so that the eight XCDs (workgroup b runs on
// XCD b % 8) each own a contiguous eighth of the chunk.  MAP 3: reversed order.
template <int MAP, int PX_DWORDS> __global__ __launch_bounds__(256) void mock_linear(const void *__restrict__ blocks, uint8_t *__restrict__ pixels,
		uint32_t wb_log2, uint32_t n_tiles, uint64_t pitch, uint32_t param) {
	uint32_t tile = blockIdx.x;
	if (MAP == 1) tile = tile % param;
	if (MAP == 2) { const uint32_t chunk = tile / param, r = tile - chunk * param, xcd = r & 7u, k = r >> 3; tile = chunk * param + xcd * (param >> 3) + k; }
	if (MAP == 3) tile = n_tiles - 1u - tile;
	const uint32_t i = tile * 256u + threadIdx.x;
	uint32_t a, b;
	if (PX_DWORDS == 4) { const v2 w = reinterpret_cast<const v2 *>(blocks)[i]; a = w.x; b = w.y; }
	else { const v4 w = reinterpret_cast<const v4 *>(blocks)[i]; a = w.x ^ w.z; b = w.y ^ w.w; }
	const uint32_t by = i >> wb_log2, bx = i & ((1u << wb_log2) - 1u);
	uint8_t *dst = pixels + (uint64_t)(by * 4u) * pitch + (uint64_t)bx * (4u * PX_DWORDS);
#pragma unroll
	for (int r = 0; r < 4; r++) {
#pragma unroll
		for (int k = 0; k < PX_DWORDS / 4; k++) {
			const v4 v{ a + (uint32_t)r, b, a ^ b, b + (uint32_t)k };
			__builtin_nontemporal_store(v, reinterpret_cast<v4 *>(dst + (uint64_t)r * pitch) + k);
		}
	}
}


// ---- "mix": the read stream and the write stream of the BC1-shaped mock decoupled -------------------------------------------------
// Each workgroup handles T consecutive tiles: all T block loads first (cache policy RPOL: bit 0 sc0, bit 1 sc1, bit 2 nt), then the
// 4 * T row stores (WPOL 4 = nt, 6 = sc1 nt).  The tile a load / a store goes to is wrapped into the first read_wrap / write_wrap
// tiles (powers of two), so the two footprints can be chosen independently of the grid.
template <int RPOL> __device__ __forceinline__ v2 load_policy(const v2 *p) {
	v2 r;
	if (RPOL == 0) { r = *p; }
	else if (RPOL == 1) asm volatile("global_load_dwordx2 %0, %1, off sc0" : "=v"(r) : "v"(p) : "memory");
	else if (RPOL == 2) asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(r) : "v"(p) : "memory");
	else if (RPOL == 3) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1" : "=v"(r) : "v"(p) : "memory");
	else if (RPOL == 4) asm volatile("global_load_dwordx2 %0, %1, off nt" : "=v"(r) : "v"(p) : "memory");
	else if (RPOL == 5) asm volatile("global_load_dwordx2 %0, %1, off sc0 nt" : "=v"(r) : "v"(p) : "memory");
	else if (RPOL == 6) asm volatile("global_load_dwordx2 %0, %1, off sc1 nt" : "=v"(r) : "v"(p) : "memory");
	else asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1 nt" : "=v"(r) : "v"(p) : "memory");
	return r;
}
template <int WPOL> __device__ __forceinline__ void store_policy(v4 v, v4 *p) {
	if (WPOL == 4) __builtin_nontemporal_store(v, p);
	else asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
}
template <int T, int RPOL, int WPOL, bool READ, bool WRITE> __global__ __launch_bounds__(256) void mock_mix(const v2 *__restrict__ blocks, uint8_t *__restrict__ pixels,
		uint32_t wb_log2, uint64_t pitch, uint32_t read_wrap, uint32_t write_wrap, uint32_t *__restrict__ sink) {
	const uint32_t tile0 = blockIdx.x * T;
	v2 w[T];
#pragma unroll
	for (int t = 0; t < T; t++) {
		if (READ) w[t] = load_policy<RPOL>(blocks + (uint64_t)((tile0 + t) & (read_wrap - 1u)) * 256u + threadIdx.x);
		else w[t] = v2{ tile0 + t, threadIdx.x };
	}
	if (READ && RPOL != 0) {
#pragma unroll
		for (int t = 0; t < T; t++) asm volatile("s_waitcnt vmcnt(0)" : "+v"(w[t]));
	}
	if (WRITE) {
#pragma unroll
		for (int t = 0; t < T; t++) {
			const uint32_t i = ((tile0 + t) & (write_wrap - 1u)) * 256u + threadIdx.x;
			const uint32_t by = i >> wb_log2, bx = i & ((1u << wb_log2) - 1u);
			uint8_t *dst = pixels + (uint64_t)(by * 4u) * pitch + (uint64_t)bx * 16u;
#pragma unroll
			for (int r = 0; r < 4; r++) store_policy<WPOL>(v4{ w[t].x + (uint32_t)r, w[t].y, w[t].x ^ w[t].y, w[t].y + 1u }, reinterpret_cast<v4 *>(dst + (uint64_t)r * pitch));
		}
	} else {
		uint32_t acc = 0;
#pragma unroll
		for (int t = 0; t < T; t++) acc ^= w[t].x + w[t].y;
		if (acc == 0x9E3779B9u && sink) *sink = acc;		// (practically never: keeps the loads alive)
	}
}

// ---- "readpass": shapes of the read-only pass that brings a band's blocks into the Infinity Cache (the library's read_ahead kernel is the
// 1024-lane / 4-loads-per-lane one) --------------------------------------------------------------------------------------------------------
template <int LANES, int LOADS> __global__ __launch_bounds__(LANES) void read_pass(const v4 *__restrict__ p, uint64_t n_vectors) {
	const uint64_t base = (uint64_t)blockIdx.x * (LANES * LOADS) + threadIdx.x;
	v4 sink[LOADS];
#pragma unroll
	for (int k = 0; k < LOADS; k++) {
		const uint64_t i = base + (uint64_t)LANES * (uint32_t)k;
		asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(sink[k]) : "v"(p + (i < n_vectors ? i : n_vectors - 1u)) : "memory");
	}
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
	for (int k = 0; k < LOADS; k++) asm volatile("" :: "v"(sink[k]));
}

struct Timer {
	hipStream_t stream;
	hipEvent_t e0, e1;
	Timer() { CHECK(hipStreamCreate(&stream)); CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); }
	// the first ~12 ms of work after idle run through a power-management excursion (20-40 % slower for VALU-heavy kernels: DESIGN.md section 6):
	// every experiment first keeps the device busy with its own first case for a while
	template <class F> void settle(F &&fn, double ms = 60.0) {
		CHECK(hipEventRecord(e0, stream));
		for (;;) {
			for (int k = 0; k < 4; k++) fn();
			CHECK(hipEventRecord(e1, stream)); CHECK(hipEventSynchronize(e1));
			float t = 0; CHECK(hipEventElapsedTime(&t, e0, e1));
			if (t >= ms) break;
		}
	}
	template <class F> double median_us(F &&fn, int warm = 3, int reps = 11) {
		for (int k = 0; k < warm; k++) fn();
		CHECK(hipStreamSynchronize(stream));
		std::vector<double> t;
		for (int k = 0; k < reps; k++) {
			CHECK(hipEventRecord(e0, stream));
			fn();
			CHECK(hipEventRecord(e1, stream));
			CHECK(hipEventSynchronize(e1));
			float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
			t.push_back(ms * 1000.0);
		}
		std::sort(t.begin(), t.end());
		return t[t.size() / 2];
	}
};

struct Fmt { const char *name; uint32_t texture_format, pixel_format; unsigned block_bytes, pixel_bytes; };
static const Fmt kFormats[] = {
	{ "BC1", 0x01000320u, 0x0334u, 8, 4 },
	{ "BPTC_FLOAT", 0x09802721u, 0x2721u, 16, 8 },
	{ "BPTC", 0x0B800334u, 0x0334u, 16, 4 },
	{ "BC3", 0x04800334u, 0x0334u, 16, 4 },
	{ "ETC2", 0x0D000320u, 0x0320u, 8, 4 },
	{ "ETC2_EAC", 0x0F800334u, 0x0334u, 16, 4 },
	{ "BPTC_SIGNED_FLOAT", 0x0A803721u, 0x3721u, 16, 8 },
};
static const Fmt *format_named(const char *n) { for (const Fmt &f : kFormats) if (!strcmp(f.name, n)) return &f; fprintf(stderr, "unknown format %s\n", n); exit(2); }

static void decode(const Fmt &f, const uint8_t *blocks, uint8_t *pixels, int W, int rows, hipStream_t s) {
	const int wb = W / 4, hb = rows / 4;
	if (detexhipDecompressTextureLinearDevice(f.texture_format, blocks, W, rows, wb, hb, pixels, (size_t)W * f.pixel_bytes, f.pixel_format, s, nullptr) != 0) {
		fprintf(stderr, "decode failed\n"); exit(2);
	}
}
static void decode_banded(const Fmt &f, const uint8_t *blocks, uint8_t *pixels, int W, int H, int band, hipStream_t s) {
	for (int y = 0; y < H; y += band) {
		const int rows = std::min(band, H - y);
		decode(f, blocks + (size_t)(y / 4) * (W / 4) * f.block_bytes, pixels + (size_t)y * W * f.pixel_bytes, W, rows, s);
	}
}

static double tb_per_s(const Fmt &f, int W, int H, double us) { return (double)(W / 4) * (H / 4) * (f.block_bytes + 16.0 * f.pixel_bytes) / us * 1e-6; }

static void sweep(const Fmt &f, int W, const std::vector<int> &heights, bool ext_alloc) {
	Timer t;
	const int Hmax = *std::max_element(heights.begin(), heights.end());
	const size_t nb = (size_t)(W / 4) * (Hmax / 4) * f.block_bytes, np = (size_t)W * Hmax * f.pixel_bytes;
	uint8_t *blocks, *pixels;
	CHECK(hipMalloc(&blocks, nb)); CHECK(hipMalloc(&pixels, np));
	fill_random<<<4096, 256, 0, t.stream>>>(reinterpret_cast<uint32_t *>(blocks), nb / 4, 12345u);
	CHECK(hipStreamSynchronize(t.stream));
	detexhipSetReadAhead(0);
	t.settle([&] { decode(f, blocks, pixels, W, heights[0], t.stream); });
	for (int H : heights) {
		// (one launch: the library's read-ahead banding of inputs beyond the Infinity Cache switched off; `readahead` = the entry as shipped)
		detexhipSetReadAhead(0);
		const double whole = t.median_us([&] { decode(f, blocks, pixels, W, H, t.stream); });
		detexhipSetReadAhead(1);
		const double ahead = t.median_us([&] { decode(f, blocks, pixels, W, H, t.stream); });
		detexhipSetReadAhead(0);
		printf("{\"exp\": \"sweep\", \"fmt\": \"%s\", \"W\": %d, \"H\": %d, \"GiB\": %.2f, \"whole_us\": %.1f, \"whole_TBps\": %.3f, \"readahead_us\": %.1f, \"readahead_TBps\": %.3f", f.name, W, H,
			((double)(W / 4) * (H / 4) * (f.block_bytes + 16.0 * f.pixel_bytes)) / (1ull << 30), whole, tb_per_s(f, W, H, whole), ahead, tb_per_s(f, W, H, ahead));
		for (int band : { 8192, 2048 }) {
			if (band >= H) continue;
			const double us = t.median_us([&] { decode_banded(f, blocks, pixels, W, H, band, t.stream); });
			printf(", \"bands%d_us\": %.1f, \"bands%d_TBps\": %.3f", band, us, band, tb_per_s(f, W, H, us));
		}
		// mock of the same shape, same buffers (write + read traffic identical to the decode for BC1 / BC6H)
		const uint32_t tiles = (uint32_t)((size_t)(W / 4) * (H / 4) / 256u), wb_log2 = (uint32_t)__builtin_ctz((unsigned)(W / 4));
		const uint64_t pitch = (uint64_t)W * f.pixel_bytes;
		double mock = 0;
		if (f.pixel_bytes == 4 && f.block_bytes == 8) mock = t.median_us([&] { mock_linear<0, 4><<<tiles, 256, 0, t.stream>>>(blocks, pixels, wb_log2, tiles, pitch, 0); });
		if (f.pixel_bytes == 8) mock = t.median_us([&] { mock_linear<0, 8><<<tiles, 256, 0, t.stream>>>(blocks, pixels, wb_log2, tiles, pitch, 0); });
		printf(", \"mock_us\": %.1f, \"mock_TBps\": %.3f}\n", mock, mock > 0 ? tb_per_s(f, W, H, mock) : 0.0);
		fflush(stdout);
	}
	// a band-sized call at each quarter of the big allocation: does the PLACE matter?
	const int band = 8192;
	for (int q = 0; q * band < Hmax; q++) {
		const uint8_t *b = blocks + (size_t)(q * band / 4) * (W / 4) * f.block_bytes;
		uint8_t *p = pixels + (size_t)q * band * W * f.pixel_bytes;
		const double us = t.median_us([&] { decode(f, b, p, W, band, t.stream); });
		printf("{\"exp\": \"place\", \"fmt\": \"%s\", \"W\": %d, \"band\": %d, \"quarter\": %d, \"us\": %.1f, \"TBps\": %.3f}\n", f.name, W, band, q, us, tb_per_s(f, W, band, us));
	}
	fflush(stdout);
	(void)ext_alloc;
	CHECK(hipFree(blocks)); CHECK(hipFree(pixels));
}

static void mock(int W) {
	Timer t;
	const Fmt &f = kFormats[0];
	const int H = 32768;
	const size_t nb = (size_t)(W / 4) * (H / 4) * 8, np = (size_t)W * H * 4;
	uint8_t *blocks, *pixels;
	CHECK(hipMalloc(&blocks, nb)); CHECK(hipMalloc(&pixels, np));
	fill_random<<<4096, 256, 0, t.stream>>>(reinterpret_cast<uint32_t *>(blocks), nb / 4, 777u);
	CHECK(hipStreamSynchronize(t.stream));
	const uint32_t wb_log2 = (uint32_t)__builtin_ctz((unsigned)(W / 4));
	const uint64_t pitch = (uint64_t)W * 4;
	const uint32_t full = (uint32_t)((size_t)(W / 4) * (H / 4) / 256u), quarter = full / 4;
	auto report = [&](const char *what, uint32_t tiles, double us) {
		printf("{\"exp\": \"mock\", \"what\": \"%s\", \"tiles\": %u, \"us\": %.1f, \"TBps\": %.3f}\n", what, tiles, us, (double)tiles * 256.0 * 72.0 / us * 1e-6);
		fflush(stdout);
	};
	report("identity, quarter grid over a quarter", quarter, t.median_us([&] { mock_linear<0, 4><<<quarter, 256, 0, t.stream>>>(blocks, pixels, wb_log2, quarter, pitch, 0); }));
	report("identity, full grid over the whole", full, t.median_us([&] { mock_linear<0, 4><<<full, 256, 0, t.stream>>>(blocks, pixels, wb_log2, full, pitch, 0); }));
	report("full grid wrapped into the first quarter", full, t.median_us([&] { mock_linear<1, 4><<<full, 256, 0, t.stream>>>(blocks, pixels, wb_log2, full, pitch, quarter); }));
	report("full grid wrapped into the first 1/16", full, t.median_us([&] { mock_linear<1, 4><<<full, 256, 0, t.stream>>>(blocks, pixels, wb_log2, full, pitch, quarter / 4); }));
	report("full grid reversed", full, t.median_us([&] { mock_linear<3, 4><<<full, 256, 0, t.stream>>>(blocks, pixels, wb_log2, full, pitch, 0); }));
	for (uint32_t chunk : { 64u, 512u, 4096u, 32768u })
		report(chunk == 64u ? "XCD-major chunks of 64 tiles" : chunk == 512u ? "XCD-major chunks of 512 tiles" : chunk == 4096u ? "XCD-major chunks of 4096 tiles" : "XCD-major chunks of 32768 tiles",
			full, t.median_us([&] { mock_linear<2, 4><<<full, 256, 0, t.stream>>>(blocks, pixels, wb_log2, full, pitch, chunk); }));
	// the whole in 2 / 4 / 8 / 16 back-to-back launches
	for (uint32_t parts : { 2u, 4u, 8u, 16u }) {
		const uint32_t tiles = full / parts;
		const double us = t.median_us([&] {
			for (uint32_t p = 0; p < parts; p++)
				mock_linear<0, 4><<<tiles, 256, 0, t.stream>>>(blocks + (size_t)p * tiles * 256u * 8u, pixels + (size_t)p * tiles * 256u * 64u, wb_log2, tiles, pitch, 0);
		});
		char what[64]; snprintf(what, sizeof what, "identity in %u launches", parts);
		report(what, full, us);
	}
	(void)f;
	CHECK(hipFree(blocks)); CHECK(hipFree(pixels));
}


static void mix(int W) {
	Timer t;
	const int H = 32768;
	const size_t nb = (size_t)(W / 4) * (H / 4) * 8, np = (size_t)W * H * 4;
	uint8_t *blocks, *pixels; uint32_t *sink;
	CHECK(hipMalloc(&blocks, nb)); CHECK(hipMalloc(&pixels, np)); CHECK(hipMalloc(&sink, 64));
	fill_random<<<4096, 256, 0, t.stream>>>(reinterpret_cast<uint32_t *>(blocks), nb / 4, 777u);
	CHECK(hipStreamSynchronize(t.stream));
	const uint32_t wb_log2 = (uint32_t)__builtin_ctz((unsigned)(W / 4));
	const uint64_t pitch = (uint64_t)W * 4;
	const uint32_t full = (uint32_t)((size_t)(W / 4) * (H / 4) / 256u);
	const v2 *b = reinterpret_cast<const v2 *>(blocks);
	auto report = [&](const char *what, double bytes, double us) {
		printf("{\"exp\": \"mix\", \"what\": \"%s\", \"us\": %.1f, \"TBps\": %.3f}\n", what, us, bytes / us * 1e-6);
		fflush(stdout);
	};
	const double rb = (double)full * 2048.0, wbytes = (double)full * 16384.0;
#define RUN(T, RP, WP, RD, WR, RWRAP, WWRAP) t.median_us([&] { mock_mix<T, RP, WP, RD, WR><<<full / T, 256, 0, t.stream>>>(b, pixels, wb_log2, pitch, RWRAP, WWRAP, sink); })
	report("read 512 MiB + write 4 GiB (the whole image)", rb + wbytes, RUN(1, 0, 4, true, true, full, full));
	report("read wrapped into 64 MiB + write 4 GiB", rb + wbytes, RUN(1, 0, 4, true, true, full / 8, full));
	report("read wrapped into 128 MiB + write 4 GiB", rb + wbytes, RUN(1, 0, 4, true, true, full / 4, full));
	report("read wrapped into 256 MiB + write 4 GiB", rb + wbytes, RUN(1, 0, 4, true, true, full / 2, full));
	report("read 512 MiB + write wrapped into 1 GiB", rb + wbytes, RUN(1, 0, 4, true, true, full, full / 4));
	report("read 512 MiB + write wrapped into 64 MiB", rb + wbytes, RUN(1, 0, 4, true, true, full, full / 64));
	report("no read, write 4 GiB", wbytes, RUN(1, 0, 4, false, true, full, full));
	report("no read, write wrapped into 1 GiB", wbytes, RUN(1, 0, 4, false, true, full, full / 4));
	report("read 512 MiB, no write", rb, RUN(1, 0, 4, true, false, full, full));
	report("read wrapped into 128 MiB, no write", rb, RUN(1, 0, 4, true, false, full / 4, full));
	report("read 512 MiB, no write, 4 tiles per workgroup", rb, RUN(4, 0, 4, true, false, full, full));
	report("whole image, sc1 nt stores", rb + wbytes, RUN(1, 0, 6, true, true, full, full));
	report("whole image, 2 tiles per workgroup", rb + wbytes, RUN(2, 0, 4, true, true, full, full));
	report("whole image, 4 tiles per workgroup", rb + wbytes, RUN(4, 0, 4, true, true, full, full));
	report("whole image, 8 tiles per workgroup", rb + wbytes, RUN(8, 0, 4, true, true, full, full));
	report("whole image, 16 tiles per workgroup", rb + wbytes, RUN(16, 0, 4, true, true, full, full));
	report("whole image, loads sc0", rb + wbytes, RUN(1, 1, 4, true, true, full, full));
	report("whole image, loads sc1", rb + wbytes, RUN(1, 2, 4, true, true, full, full));
	report("whole image, loads sc0 sc1", rb + wbytes, RUN(1, 3, 4, true, true, full, full));
	report("whole image, loads nt", rb + wbytes, RUN(1, 4, 4, true, true, full, full));
	report("whole image, loads sc0 nt", rb + wbytes, RUN(1, 5, 4, true, true, full, full));
	report("whole image, loads sc1 nt", rb + wbytes, RUN(1, 6, 4, true, true, full, full));
	report("whole image, loads sc0 sc1 nt", rb + wbytes, RUN(1, 7, 4, true, true, full, full));
	report("quarter-wrapped both (1.1 GiB), loads nt", rb + wbytes, RUN(1, 4, 4, true, true, full / 4, full / 4));
	report("quarter-wrapped both (1.1 GiB), loads plain", rb + wbytes, RUN(1, 0, 4, true, true, full / 4, full / 4));
	// two phases per chunk: a read-only pass over the chunk's blocks (into the memory-side cache), then the chunk's decode
	for (uint32_t chunks : { 4u, 8u, 16u, 32u }) {
		const uint32_t tiles = full / chunks;
		const double us = t.median_us([&] {
			for (uint32_t c = 0; c < chunks; c++) {
				const v2 *cb = b + (size_t)c * tiles * 256u;
				uint8_t *cp = pixels + (size_t)c * tiles * 256u * 64u;
				mock_mix<4, 0, 4, true, false><<<tiles / 4, 256, 0, t.stream>>>(cb, cp, wb_log2, pitch, tiles, tiles, sink);
				mock_mix<1, 0, 4, true, true><<<tiles, 256, 0, t.stream>>>(cb, cp, wb_log2, pitch, tiles, tiles, sink);
			}
		});
		char what[96]; snprintf(what, sizeof what, "whole image in %u chunks: read pass, then decode pass", chunks);
		report(what, rb + wbytes, us);
	}
	for (uint32_t chunks : { 4u }) {		// ... the decode pass with the product's `sc1 nt` stores; and the read pass in 16-byte loads, 1024 lanes (the library's read_ahead shape)
		const uint32_t tiles = full / chunks;
		const double us = t.median_us([&] {
			for (uint32_t c = 0; c < chunks; c++) {
				const v2 *cb = b + (size_t)c * tiles * 256u;
				uint8_t *cp = pixels + (size_t)c * tiles * 256u * 64u;
				mock_mix<4, 0, 4, true, false><<<tiles / 4, 256, 0, t.stream>>>(cb, cp, wb_log2, pitch, tiles, tiles, sink);
				mock_mix<1, 0, 6, true, true><<<tiles, 256, 0, t.stream>>>(cb, cp, wb_log2, pitch, tiles, tiles, sink);
			}
		});
		report("whole image in 4 chunks: read pass, then decode pass with sc1 nt stores", rb + wbytes, us);
	}
	// the same two passes OVERLAPPED: the read pass of chunk c + 1 on a second stream while chunk c is decoded
	{
		hipStream_t s2; CHECK(hipStreamCreate(&s2));
		for (uint32_t chunks : { 8u, 16u, 32u }) {
			const uint32_t tiles = full / chunks;
			std::vector<hipEvent_t> read_done(chunks), decode_done(chunks);
			for (auto &e : read_done) CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
			for (auto &e : decode_done) CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
			hipEvent_t start; CHECK(hipEventCreateWithFlags(&start, hipEventDisableTiming));
			const double us = t.median_us([&] {
				CHECK(hipEventRecord(start, t.stream));
				CHECK(hipStreamWaitEvent(s2, start, 0));
				for (uint32_t c = 0; c < chunks; c++) {
					const v2 *cb = b + (size_t)c * tiles * 256u;
					uint8_t *cp = pixels + (size_t)c * tiles * 256u * 64u;
					if (c >= 2) CHECK(hipStreamWaitEvent(s2, decode_done[c - 2], 0));	// at most two chunks of blocks ahead of the decode
					mock_mix<4, 0, 4, true, false><<<tiles / 4, 256, 0, s2>>>(cb, cp, wb_log2, pitch, tiles, tiles, sink);
					CHECK(hipEventRecord(read_done[c], s2));
					CHECK(hipStreamWaitEvent(t.stream, read_done[c], 0));
					mock_mix<1, 0, 4, true, true><<<tiles, 256, 0, t.stream>>>(cb, cp, wb_log2, pitch, tiles, tiles, sink);
					CHECK(hipEventRecord(decode_done[c], t.stream));
				}
			});
			char what[96]; snprintf(what, sizeof what, "whole image in %u chunks: read pass of the next chunk overlapped with the decode", chunks);
			report(what, rb + wbytes, us);
			for (auto &e : read_done) CHECK(hipEventDestroy(e));
			for (auto &e : decode_done) CHECK(hipEventDestroy(e));
			CHECK(hipEventDestroy(start));
		}
		CHECK(hipStreamDestroy(s2));
	}
#undef RUN
	CHECK(hipFree(blocks)); CHECK(hipFree(pixels)); CHECK(hipFree(sink));
}

template <int LANES, int LOADS> static void read_pass_case(Timer &t, const uint8_t *buf, size_t total, size_t band) {
	const uint64_t nv = band / 16u;
	const unsigned grid = (unsigned)((nv + (uint64_t)LANES * LOADS - 1u) / ((uint64_t)LANES * LOADS));
	size_t at = 0;
	const double us = t.median_us([&] {		// a different band every launch: the reads come out of HBM
		read_pass<LANES, LOADS><<<grid, LANES, 0, t.stream>>>(reinterpret_cast<const v4 *>(buf + at), nv);
		at = (at + band) % total;
	}, 4, 21);
	printf("{\"exp\": \"readpass\", \"lanes\": %d, \"loads_per_lane\": %d, \"band_MiB\": %zu, \"us\": %.2f, \"TBps\": %.3f}\n", LANES, LOADS, band >> 20, us, (double)band / us * 1e-6);
	fflush(stdout);
}
static void readpass() {
	Timer t;
	const size_t total = (size_t)1 << 30, band = (size_t)128 << 20;
	uint8_t *buf; CHECK(hipMalloc(&buf, total));
	fill_random<<<4096, 256, 0, t.stream>>>(reinterpret_cast<uint32_t *>(buf), total / 4, 99u);
	CHECK(hipStreamSynchronize(t.stream));
	read_pass_case<256, 2>(t, buf, total, band); read_pass_case<256, 4>(t, buf, total, band); read_pass_case<256, 8>(t, buf, total, band); read_pass_case<256, 16>(t, buf, total, band);
	read_pass_case<512, 4>(t, buf, total, band); read_pass_case<512, 8>(t, buf, total, band);
	read_pass_case<1024, 2>(t, buf, total, band); read_pass_case<1024, 4>(t, buf, total, band); read_pass_case<1024, 8>(t, buf, total, band);
	CHECK(hipFree(buf));
}

int main(int argc, char **argv) {
	const char *mode = argc > 1 ? argv[1] : "sweep";
	if (!strcmp(mode, "sweep")) {
		const Fmt *f = format_named(argc > 2 ? argv[2] : "BC1");
		std::vector<int> heights;
		for (int k = 3; k < argc; k++) heights.push_back(atoi(argv[k]));
		if (heights.empty()) heights = { 8192, 12288, 16384, 24576, 32768 };
		sweep(*f, 32768, heights, false);
	} else if (!strcmp(mode, "mock")) {
		mock(32768);
	} else if (!strcmp(mode, "readpass")) {
		readpass();
	} else if (!strcmp(mode, "mix")) {
		mix(32768);
	} else if (!strcmp(mode, "pmc")) {
		const Fmt *f = format_named(argc > 2 ? argv[2] : "BC1");
		const int H = argc > 3 ? atoi(argv[3]) : 32768, band = argc > 4 ? atoi(argv[4]) : 0, W = 32768;
		Timer t;
		const size_t nb = (size_t)(W / 4) * (H / 4) * f->block_bytes, np = (size_t)W * H * f->pixel_bytes;
		uint8_t *blocks, *pixels;
		CHECK(hipMalloc(&blocks, nb)); CHECK(hipMalloc(&pixels, np));
		fill_random<<<4096, 256, 0, t.stream>>>(reinterpret_cast<uint32_t *>(blocks), nb / 4, 12345u);
		for (int k = 0; k < 5; k++) { if (band > 0) decode_banded(*f, blocks, pixels, W, H, band, t.stream); else decode(*f, blocks, pixels, W, H, t.stream); }
		CHECK(hipStreamSynchronize(t.stream));
	} else {
		fprintf(stderr, "usage: big_footprint sweep FMT [H ...] | mock | mix | readpass | pmc FMT H [band]\n");
		return 2;
	}
	return 0;
}
