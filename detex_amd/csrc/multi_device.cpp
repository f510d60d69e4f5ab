// multi_device.cpp -- one texture over several devices (include/detexhip.h): block-row shards, no exchange for the decode,
// an optional peer-copy gather.  Host code only.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <thread>

#include "host_internal.h"

using namespace detexhip;

// ------------------------------------------------------------------------------------------------
// multi-device entry: one texture, N shards of block rows, one calling thread
// ------------------------------------------------------------------------------------------------
extern "C" int detexhipShardRows(int height_in_blocks, int n_shards, int shard, int *row0, int *row1) {
	if (n_shards <= 0 || shard < 0 || shard >= n_shards || height_in_blocks < 0 || !row0 || !row1) {
		detexSetErrorMessage("detexhipShardRows: bad arguments");
		return 1;
	}
	*row0 = (int)((int64_t)shard * height_in_blocks / n_shards);
	*row1 = (int)((int64_t)(shard + 1) * height_in_blocks / n_shards);
	return 0;
}

namespace {
// Per calling thread and shard index: stream, events, status word and grow-only staging buffers, created on first use and
// kept between calls (a shard keeps its device); detexhipReleaseThreadResources() hands them back.  Thread-local, so
// concurrent callers never share a slot and the entry points take no lock.
struct ShardSlot {
	int device = -1;
	hipStream_t stream = nullptr;
	hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
	uint32_t *d_status = nullptr;
	void *d_upload = nullptr; size_t upload_cap = 0;	// blocks uploaded from host_blocks
	void *d_band = nullptr; size_t band_cap = 0;		// decoded band of the host-output entry
	bool used = false;					// received work in the current call
	void destroy() {
		if (device < 0) return;
		int prev = -1;
		(void)hipGetDevice(&prev);
		if (hipSetDevice(device) == hipSuccess) {
			if (stream) { (void)hipStreamSynchronize(stream); (void)hipStreamDestroy(stream); }
			if (e0) (void)hipEventDestroy(e0);
			if (e1) (void)hipEventDestroy(e1);
			if (e2) (void)hipEventDestroy(e2);
			(void)hipFree(d_status); (void)hipFree(d_upload); (void)hipFree(d_band);
		}
		*this = ShardSlot{};
		if (prev >= 0) (void)hipSetDevice(prev);
	}
};
struct ShardSlots {
	ShardSlot slot[64];
	void release() { for (ShardSlot &sl : slot) sl.destroy(); }
	~ShardSlots() { release(); }
};
thread_local ShardSlots t_shards;

// makes `device` current and the slot usable on it; a slot that fails half-way is torn down completely
hipError_t prepare_slot(ShardSlot &sl, int device) {
	hipError_t e = hipSetDevice(device);
	if (e != hipSuccess) return e;
	if (sl.device == device && sl.stream) return hipSuccess;
	sl.destroy();							// the shard moved to another device (or was never set up)
	if ((e = hipSetDevice(device)) != hipSuccess) return e;
	sl.device = device;						// from here on destroy() releases whatever exists
	if ((e = hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking)) != hipSuccess || (e = hipEventCreate(&sl.e0)) != hipSuccess ||
			(e = hipEventCreate(&sl.e1)) != hipSuccess || (e = hipEventCreate(&sl.e2)) != hipSuccess ||
			(e = hipMalloc(&sl.d_status, 64)) != hipSuccess) {
		sl.destroy();
		(void)hipSetDevice(device);
		return e;
	}
	return hipSuccess;
}
hipError_t grow(void **buf, size_t *cap, size_t need) {		// on the current device
	if (need <= *cap) return hipSuccess;
	if (*buf) (void)hipFree(*buf);
	*buf = nullptr; *cap = 0;
	const size_t rounded = (need + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
	hipError_t e = hipMalloc(buf, rounded);
	if (e == hipSuccess) *cap = rounded;
	return e;
}

// Peer access device -> peer, established once per process and pair (the enable call costs milliseconds; it used to run on every
// gather).  Returns true if `device` can address `peer`'s memory directly -- the copy then travels over the link between the two
// (xGMI on an MI355X node) -- and false where the topology or the runtime does not allow it; only SUCCESS is remembered as such, a
// pair that failed is asked about again by the next call.
// DETEXHIP_PEER_ACCESS=0 in the environment: no pair is mapped -- not even a device with itself -- and every gather copy goes the way of an
// unmapped pair (hipMemcpyPeerAsync, which the runtime stages; padded bands row by row).  For nodes whose peer mappings misbehave,
// and the only way to run that branch on a box with one GPU (tests/test_gpu_host_multi.py).
std::mutex g_peer_mutex;
uint64_t g_peer_enabled[64];
bool peer_mapping_allowed() {
	static const bool allowed = [] { const char *env = getenv("DETEXHIP_PEER_ACCESS"); return !(env && env[0] == '0'); }();
	return allowed;
}
bool peer_access(int device, int peer) {			// `device` is current
	if (!peer_mapping_allowed()) return false;
	if (device == peer) return true;
	if (device < 0 || device >= 64 || peer < 0 || peer >= 64) return false;
	std::lock_guard<std::mutex> lock(g_peer_mutex);
	if (g_peer_enabled[device] >> peer & 1u) return true;
	int can = 0;
	if (hipDeviceCanAccessPeer(&can, device, peer) != hipSuccess || !can) { (void)hipGetLastError(); return false; }
	const hipError_t e = hipDeviceEnablePeerAccess(peer, 0);
	if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); return false; }
	(void)hipGetLastError();					// (hipErrorPeerAccessAlreadyEnabled is sticky otherwise)
	g_peer_enabled[device] |= (uint64_t)1 << peer;
	return true;
}

}  // namespace
namespace detexhip {
void release_shard_slots() { t_shards.release(); }
}

extern "C" int detexhipDecompressTextureLinearMultiDevice(uint32_t texture_format, const void *host_blocks, int width, int height,
		int width_in_blocks, int height_in_blocks, size_t pitch_bytes, uint32_t pixel_format, detexhipShard *shards, int n_shards,
		int gather_device, void *d_gathered, float *decode_wall_ms, float *gather_wall_ms) {
	const char *who = "detexhipDecompressTextureLinearMultiDevice";
	const FormatEntry *f = lookup_format(texture_format);
	if (!f || !pixel_format_accepted(texture_format, pixel_format)) {
		detexSetErrorMessage("%s: format 0x%08X -> pixel format 0x%08X is outside the block-decode path of libdetexhip", who, texture_format, pixel_format);
		return 1;
	}
	if (!shards || n_shards < 1 || n_shards > 64 || width < 0 || height < 0 || width_in_blocks < 0 || height_in_blocks < 0 ||
			(gather_device >= 0 && !d_gathered)) {
		detexSetErrorMessage("%s: bad arguments (1..64 shards, non-negative geometry, d_gathered with gather_device)", who);
		return 1;
	}
	const size_t px = (size_t)detexGetPixelSize(pixel_format), bs = detexGetCompressedBlockSize(texture_format);
	const size_t pitch = pitch_bytes ? pitch_bytes : (size_t)width * px;
	const size_t wb = (size_t)width_in_blocks;
	if (pitch < (size_t)width * px) { detexSetErrorMessage("%s: pitch_bytes %zu is smaller than a row (%zu bytes)", who, pitch, (size_t)width * px); return 1; }
	int prev = -1;
	(void)hipGetDevice(&prev);
	int rc = 0;
	auto fail = [&](const char *what, hipError_t e) { detexSetErrorMessage("%s: %s failed: %s", who, what, hipGetErrorString(e)); rc = 1; };
	for (int g = 0; g < n_shards; g++) { t_shards.slot[g].used = false; shards[g].peer_access = -1; }	// (out fields of shards an early failure never reaches)
	// per-shard rows, streams, status words, uploads, conversion tables (all before the timed region)
	for (int g = 0; g < n_shards && rc == 0; g++) {
		detexhipShard &sh = shards[g];
		(void)detexhipShardRows(height_in_blocks, n_shards, g, &sh.row0, &sh.row1);
		sh.decode_ms = 0.f; sh.invalid_blocks = 0;
		ShardSlot &sl = t_shards.slot[g];
		hipError_t e = prepare_slot(sl, sh.device);
		if (e != hipSuccess) { fail("device / stream setup", e); break; }
		sl.used = true;
		if (prepared_epilogue(texture_format, pixel_format, sl.stream) == -2) { rc = 1; break; }	// the half-float table of THIS device
		if ((e = hipMemsetAsync(sl.d_status, 0, 4, sl.stream)) != hipSuccess) { fail("hipMemsetAsync", e); break; }
		const size_t n = (size_t)(sh.row1 - sh.row0) * wb * bs;
		if (!sh.d_blocks) {
			if (!host_blocks) { detexSetErrorMessage("%s: shard %d has no d_blocks and host_blocks is NULL", who, g); rc = 1; break; }
			if ((e = grow(&sl.d_upload, &sl.upload_cap, n ? n : 16)) != hipSuccess) { fail("hipMalloc(blocks)", e); break; }
			if (n && (e = hipMemcpyAsync(sl.d_upload, static_cast<const uint8_t *>(host_blocks) + (size_t)sh.row0 * wb * bs, n, hipMemcpyHostToDevice,
					sl.stream)) != hipSuccess) { fail("hipMemcpyAsync(H2D)", e); break; }
		}
		sh.peer_access = gather_device < 0 ? -1 : (peer_access(sh.device, gather_device) ? 1 : 0);	// direct peer copies over xGMI where the topology allows
	}
	for (int g = 0; g < n_shards && rc == 0; g++) {		// uploads done: the timed region starts with idle devices
		hipError_t e = hipSetDevice(shards[g].device);
		if (e == hipSuccess) e = hipStreamSynchronize(t_shards.slot[g].stream);
		if (e != hipSuccess) fail("hipStreamSynchronize", e);
	}
	const auto t0 = std::chrono::steady_clock::now();
	for (int g = 0; g < n_shards && rc == 0; g++) {
		detexhipShard &sh = shards[g];
		ShardSlot &sl = t_shards.slot[g];
		hipError_t e = hipSetDevice(sh.device);
		if (e != hipSuccess) { fail("hipSetDevice", e); break; }
		const size_t y0 = (size_t)sh.row0 * 4u, y1 = ((size_t)sh.row1 * 4u < (size_t)height) ? (size_t)sh.row1 * 4u : (size_t)height;
		(void)hipEventRecord(sl.e0, sl.stream);
		if (y1 > y0 && sh.row1 > sh.row0) {
			if (detexhipDecompressTextureLinearDevice(texture_format, sh.d_blocks ? sh.d_blocks : sl.d_upload, width, (int)(y1 - y0), width_in_blocks,
					sh.row1 - sh.row0, sh.d_pixels, pitch, pixel_format, sl.stream, sl.d_status) != 0) { rc = 1; break; }
		}
		(void)hipEventRecord(sl.e1, sl.stream);
		if (gather_device >= 0 && y1 > y0) {
			// a band is (rows - 1) * pitch + width * px bytes: one flat peer copy when rows are dense, a 2-D copy otherwise (the
			// bytes between width * px and pitch belong to the caller, in the band and in the gathered image alike)
			uint8_t *dst = static_cast<uint8_t *>(d_gathered) + y0 * pitch;
			// (hipMemcpyPeerAsync names both devices, so the runtime stages the copy where there is no peer mapping; the 2-D
			// device-to-device copy needs the mapping -- without one the rows go one hipMemcpyPeerAsync each: slow, but correct)
			if (pitch == (size_t)width * px) e = hipMemcpyPeerAsync(dst, gather_device, sh.d_pixels, sh.device, (y1 - y0) * pitch, sl.stream);
			else if (sh.peer_access == 1) e = hipMemcpy2DAsync(dst, pitch, sh.d_pixels, pitch, (size_t)width * px, y1 - y0, hipMemcpyDeviceToDevice, sl.stream);
			else for (size_t y = 0; y < y1 - y0 && e == hipSuccess; y++)
				e = hipMemcpyPeerAsync(dst + y * pitch, gather_device, static_cast<const uint8_t *>(sh.d_pixels) + y * pitch, sh.device, (size_t)width * px, sl.stream);
			if (e != hipSuccess) { fail("peer copy", e); break; }
		}
		(void)hipEventRecord(sl.e2, sl.stream);
	}
	if (rc == 0) {
		for (int g = 0; g < n_shards; g++) { (void)hipSetDevice(shards[g].device); hipError_t e = hipEventSynchronize(t_shards.slot[g].e1); if (e != hipSuccess && rc == 0) fail("kernel", e); }
		const auto t1 = std::chrono::steady_clock::now();
		for (int g = 0; g < n_shards; g++) { (void)hipSetDevice(shards[g].device); hipError_t e = hipEventSynchronize(t_shards.slot[g].e2); if (e != hipSuccess && rc == 0) fail("gather", e); }
		const auto t2 = std::chrono::steady_clock::now();
		if (decode_wall_ms) *decode_wall_ms = std::chrono::duration<float, std::milli>(t1 - t0).count();
		if (gather_wall_ms) *gather_wall_ms = gather_device >= 0 ? std::chrono::duration<float, std::milli>(t2 - t0).count() : 0.f;
		for (int g = 0; g < n_shards && rc == 0; g++) {
			(void)hipSetDevice(shards[g].device);
			uint32_t st = 0;
			hipError_t e = hipMemcpy(&st, t_shards.slot[g].d_status, 4, hipMemcpyDeviceToHost);
			if (e != hipSuccess) { fail("hipMemcpy(status)", e); break; }
			shards[g].invalid_blocks = st != 0;
			(void)hipEventElapsedTime(&shards[g].decode_ms, t_shards.slot[g].e0, t_shards.slot[g].e1);
		}
	}
	// Nothing launched by this call may still be running when it returns, failed or not: the caller is free to release its
	// buffers.  (The error message of the first failure stays.)
	for (int g = 0; g < n_shards; g++) {
		ShardSlot &sl = t_shards.slot[g];
		if (sl.used && sl.stream && hipSetDevice(sl.device) == hipSuccess) (void)hipStreamSynchronize(sl.stream);
		sl.used = false;
	}
	if (prev >= 0) (void)hipSetDevice(prev);
	return rc;
}

// Host image in, host image out, over N devices: every shard uploads its band of blocks, decodes it and downloads its band
// of pixels over ITS OWN PCIe link -- the one lever left for the host-pointer tier, which a single link bounds at ~56 GB/s
// (8192^2 RGBA8: 4.8 ms of download against 0.04 ms of kernel).  One worker thread per shard (copies
// from and to pageable memory block their caller); the workers use the calling thread's slots, which it does not touch
// until they have joined.
extern "C" int detexhipDecompressTextureLinearMultiDeviceHost(uint32_t texture_format, const void *host_blocks, int width, int height,
		int width_in_blocks, int height_in_blocks, void *host_pixels, size_t pitch_bytes, uint32_t pixel_format, const int *devices, int n_shards,
		int *any_invalid, float *wall_ms) {
	const char *who = "detexhipDecompressTextureLinearMultiDeviceHost";
	const FormatEntry *f = lookup_format(texture_format);
	if (!f || !pixel_format_accepted(texture_format, pixel_format)) {
		detexSetErrorMessage("%s: format 0x%08X -> pixel format 0x%08X is outside the block-decode path of libdetexhip", who, texture_format, pixel_format);
		return 1;
	}
	if (!devices || n_shards < 1 || n_shards > 64 || width < 0 || height < 0 || width_in_blocks < 0 || height_in_blocks < 0 || !host_blocks || !host_pixels) {
		detexSetErrorMessage("%s: bad arguments (1..64 shards, non-negative geometry, host_blocks and host_pixels)", who);
		return 1;
	}
	const size_t px = (size_t)detexGetPixelSize(pixel_format), bs = detexGetCompressedBlockSize(texture_format);
	const size_t pitch = pitch_bytes ? pitch_bytes : (size_t)width * px, row_bytes = (size_t)width * px, wb = (size_t)width_in_blocks;
	if (pitch < row_bytes) { detexSetErrorMessage("%s: pitch_bytes %zu is smaller than a row (%zu bytes)", who, pitch, row_bytes); return 1; }
	// like the reference, only the pixels the block grid covers are written (texture.c:116-136)
	const size_t cov_w = (size_t)width < 4u * wb ? (size_t)width : 4u * wb;
	int prev = -1;
	(void)hipGetDevice(&prev);
	struct Work { int rc = 0; bool invalid = false; char message[256] = { 0 }; };
	Work work[64];
	ShardSlots &slots = t_shards;
	// the workers decode on behalf of the calling thread: ITS quirk mask and kernel variant apply, not the workers' own defaults
	const uint32_t decode_flags = current_spec_flags();
	const int variant = current_variant(), read_ahead = current_read_ahead();
	const auto t0 = std::chrono::steady_clock::now();
	auto run = [&](int g) {
		Work &w = work[g];
		auto fail = [&](const char *what, hipError_t e) { snprintf(w.message, sizeof w.message, "%s: shard %d: %s failed: %s", who, g, what, hipGetErrorString(e)); w.rc = 1; };
		int row0 = 0, row1 = 0;
		(void)detexhipShardRows(height_in_blocks, n_shards, g, &row0, &row1);
		const size_t y0 = (size_t)row0 * 4u, y1 = ((size_t)row1 * 4u < (size_t)height) ? (size_t)row1 * 4u : (size_t)height;
		if (row1 <= row0 || y1 <= y0 || cov_w == 0) return;
		ShardSlot &sl = slots.slot[g];
		hipError_t e = prepare_slot(sl, devices[g]);
		if (e != hipSuccess) { fail("device / stream setup", e); return; }
		if (prepared_epilogue(texture_format, pixel_format, sl.stream) == -2) { snprintf(w.message, sizeof w.message, "%s: shard %d: conversion table upload failed", who, g); w.rc = 1; return; }
		const size_t n_in = (size_t)(row1 - row0) * wb * bs, rows = y1 - y0;
		if ((e = grow(&sl.d_upload, &sl.upload_cap, n_in)) != hipSuccess || (e = grow(&sl.d_band, &sl.band_cap, rows * row_bytes)) != hipSuccess) { fail("hipMalloc", e); return; }
		if ((e = hipMemsetAsync(sl.d_status, 0, 4, sl.stream)) != hipSuccess) { fail("hipMemsetAsync", e); return; }
		if ((e = hipMemcpyAsync(sl.d_upload, static_cast<const uint8_t *>(host_blocks) + (size_t)row0 * wb * bs, n_in, hipMemcpyHostToDevice, sl.stream)) != hipSuccess) { fail("hipMemcpyAsync(H2D)", e); return; }
		if (linear_device_with(texture_format, sl.d_upload, width, (int)rows, width_in_blocks, row1 - row0, sl.d_band, row_bytes, pixel_format,
				sl.stream, sl.d_status, decode_flags, variant, read_ahead) != 0) { snprintf(w.message, sizeof w.message, "%s", detexGetErrorMessage() ? detexGetErrorMessage() : "launch failed"); w.rc = 1; }
		uint8_t *dst = static_cast<uint8_t *>(host_pixels) + y0 * pitch;
		uint32_t st = 0;
		if (w.rc == 0) {
			if (pitch == row_bytes && cov_w == (size_t)width) e = hipMemcpyAsync(dst, sl.d_band, rows * row_bytes, hipMemcpyDeviceToHost, sl.stream);
			else e = hipMemcpy2DAsync(dst, pitch, sl.d_band, row_bytes, cov_w * px, rows, hipMemcpyDeviceToHost, sl.stream);
			if (e != hipSuccess) fail("download", e);
			else if ((e = hipMemcpyAsync(&st, sl.d_status, 4, hipMemcpyDeviceToHost, sl.stream)) != hipSuccess) fail("hipMemcpyAsync(status)", e);
		}
		if ((e = hipStreamSynchronize(sl.stream)) != hipSuccess && w.rc == 0) fail("hipStreamSynchronize", e);
		w.invalid = st != 0;
	};
	if (n_shards == 1) run(0);
	else {
		// (one worker per shard, signals blocked in them; a shard whose thread the system refuses is done by the calling thread itself, after the others were started)
		std::thread workers[64];
		bool inline_shard[64] = {};
		for (int g = 0; g < n_shards; g++) inline_shard[g] = !start_helper_thread(workers[g], [&run, g]() { run(g); });
		for (int g = 0; g < n_shards; g++) if (inline_shard[g]) run(g);
		for (int g = 0; g < n_shards; g++) if (workers[g].joinable()) workers[g].join();
	}
	if (wall_ms) *wall_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
	if (prev >= 0) (void)hipSetDevice(prev);
	bool invalid = false;
	for (int g = 0; g < n_shards; g++) {
		if (work[g].rc != 0) { detexSetErrorMessage("%s", work[g].message); return 1; }
		invalid = invalid || work[g].invalid;
	}
	if (any_invalid) *any_invalid = invalid ? 1 : 0;
	if (invalid) detexSetErrorMessage("detexDecompressBlock: Decompress function for format 0x%08X returned error", texture_format);
	return 0;
}

