// host_tier.cpp -- the reference's own entry points on HOST pointers (include/detex.h): the texture drivers of texture.c:55-145, the
// 19 leaf decoders of decompress-*.c and detexDecompressBlock, over the device tier.  There is NO CPU decode in this library:
// every entry point, including the one-block leaf functions, runs the gfx950 kernels; without a usable HIP device the calls
// fail with an error message.  Host code only.
#include <cstdlib>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <atomic>
#include <mutex>
#include <thread>
#include <vector>

#include "host_internal.h"
#include "tune.h"

using namespace detexhip;

namespace {

// ---- pixel buffers owned by the library (detexhipAllocPixelBuffer) ------------------------------------------------------------------------
// Pinned, device-visible host memory handed to the caller to be used as the pixel_buffer of the texture drivers: a kernel can write straight
// into it (no staging buffer on the host side of the link, no copy-out -- what the removed registered-output path did with memory the
// library did NOT own).  Process-wide registry, a handful of entries; looked up per call under a mutex (tens of nanoseconds).
struct OwnedBuffer { uint8_t *host, *dev; size_t bytes; unsigned in_use; };
std::mutex g_owned_mutex;
std::condition_variable g_owned_idle;		// signalled when an in_use count drops to zero (detexhipFreePixelBuffer waits on it)
std::vector<OwnedBuffer> g_owned;
// A decode that reads or writes an owned buffer directly HOLDS it from the lookup until its kernel has completed (or is known never to
// complete): detexhipFreePixelBuffer from another thread waits for the count to drop instead of freeing memory a kernel is still writing.
struct OwnedHold {
	uint8_t *base = nullptr;		// host base of the held buffer (its key in the registry)
	OwnedHold() = default;
	OwnedHold(const OwnedHold &) = delete;
	OwnedHold &operator=(const OwnedHold &) = delete;
	// device view of [p, p + n) if that range lies inside one owned buffer (which is then held), else nullptr
	uint8_t *acquire(const void *p, size_t n) {
		const uint8_t *q = static_cast<const uint8_t *>(p);
		std::lock_guard<std::mutex> lock(g_owned_mutex);
		for (OwnedBuffer &b : g_owned)
			if (q >= b.host && n <= b.bytes && (size_t)(q - b.host) <= b.bytes - n) { b.in_use++; base = b.host; return b.dev + (q - b.host); }
		return nullptr;
	}
	void release() {
		if (!base) return;
		std::lock_guard<std::mutex> lock(g_owned_mutex);
		for (OwnedBuffer &b : g_owned)
			if (b.host == base) { if (--b.in_use == 0) g_owned_idle.notify_all(); break; }
		base = nullptr;
	}
	~OwnedHold() { release(); }
};

// The words of ThreadContext::d_status (device memory, zero between calls), shared with histogram.hip's 16-bin result
constexpr int kStatusWords = 32;
constexpr int kStatusHistogramBins = 16;	// [0] the status word of the staged batch / mip-chain paths; [0..15] the bins of detexhipModeHistogram
constexpr int kStatusBandCounters = 16;		// [16..19] workgroup counters of the small calls' completions (Completion::counter), one per band
constexpr int kStagedStatusWord = 24;		// [24] the status word of the staged texture path
constexpr int kDirectBands = 4;
static_assert(kStatusBandCounters >= kStatusHistogramBins && kStatusBandCounters + kDirectBands <= kStagedStatusWord && kStagedStatusWord < kStatusWords,
	"the histogram bins, the band counters and the staged status word must not overlap");

// Test hook (detexhipTestFailAfterLaunch): the next N host-tier calls of this thread return false right after their kernels were launched
thread_local int t_fail_after_launch = 0;
bool injected_failure() {
	if (t_fail_after_launch <= 0) return false;
	t_fail_after_launch--;
	detexSetErrorMessage("libdetexhip: injected failure after launch (detexhipTestFailAfterLaunch)");
	return true;
}

// ------------------------------------------------------------------------------------------------
// per-thread device context of the host-pointer tier: a stream and grow-only device staging buffers that
// detexhipReleaseThreadResources() hands back
// ------------------------------------------------------------------------------------------------
struct ThreadContext {
	bool ready = false;
	int device = -1;
	hipStream_t stream = nullptr;
	void *d_in = nullptr, *d_out = nullptr;
	size_t in_cap = 0, out_cap = 0;
	uint32_t *d_status = nullptr;	// kStatusWords words, all ZERO between calls (layout: kStatus... above)
	// A call sets `dirty` before its first launch and clears it when everything it launched is known to have finished with the device
	// words back at zero.  A call that fails in between (a launch, a copy or a wait refused by the runtime) leaves it set, and the NEXT
	// call on this thread first drains the stream and zeroes the words again (heal_if_dirty) -- instead of inheriting a raised status
	// word, a half-counted completion counter or kernels still writing into the pinned buffer it is about to fill.
	bool dirty = false;
	uint32_t ticket = 0;		// last completion ticket handed out (never 0: the completion word starts as 0)
	// small calls: a pinned host buffer the kernels read blocks from and write pixels / status into directly (see direct_exchange)
	uint8_t *h_pin = nullptr, *d_pin = nullptr;
	size_t pin_cap = 0;
	// large calls: a second stream and one event per band for the uploads that run beside the downloads (via_staging_duplex); created on first use
	hipStream_t stream_up = nullptr;
	hipEvent_t ev_up[Tune::kHostDuplexBands] = {};
	ResidentService service;	// the smallest calls, from the second in a row of one (format, target) pair on (host_resident.cpp)
	void release() {
		if (!ready) return;
		int prev = -1;
		(void)hipGetDevice(&prev);
		(void)hipSetDevice(device);
		service.release();
		(void)hipStreamSynchronize(stream);
		if (stream_up) {
			(void)hipStreamSynchronize(stream_up);
			for (hipEvent_t &e : ev_up) { if (e) (void)hipEventDestroy(e); e = nullptr; }
			(void)hipStreamDestroy(stream_up);
			stream_up = nullptr;
		}
		(void)hipFree(d_in); (void)hipFree(d_out); (void)hipFree(d_status);
		if (h_pin) (void)hipHostFree(h_pin);
		(void)hipStreamDestroy(stream);
		d_in = d_out = nullptr; d_status = nullptr; in_cap = out_cap = 0;
		h_pin = d_pin = nullptr; pin_cap = 0;
		stream = nullptr;
		ready = false; dirty = false;
		if (prev >= 0) (void)hipSetDevice(prev);
	}
	~ThreadContext() { release(); }
};
thread_local ThreadContext t_ctx;

// The host tier always runs on the context's device, whatever device the calling thread has made current since
// (e.g. torch.cuda.set_device): entry points hold one of these for their duration and the caller's device is
// restored on return.
struct DeviceScope {
	int prev = -1;
	bool ok = true;
	explicit DeviceScope(int device) {
		if (hipGetDevice(&prev) != hipSuccess) prev = -1;
		if (prev != device) {
			hipError_t e = hipSetDevice(device);
			if (e != hipSuccess) { detexSetErrorMessage("libdetexhip: hipSetDevice(%d) failed: %s", device, hipGetErrorString(e)); ok = false; }
		}
	}
	~DeviceScope() { int now = -1; if (prev >= 0 && hipGetDevice(&now) == hipSuccess && now != prev) (void)hipSetDevice(prev); }
};

bool context_ready() {
	ThreadContext &c = t_ctx;
	if (c.ready) return true;
	int count = 0;
	hipError_t e = hipGetDeviceCount(&count);
	if (e != hipSuccess || count <= 0) {
		detexSetErrorMessage("libdetexhip: no usable HIP device (%s); this library has no CPU decode path",
			e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
		return false;
	}
	if (c.device < 0) {
		ThreadSettings &s = thread_settings();		// detexhipSetDevice, else DETEXHIP_DEVICE, else 0
		if (s.device < 0) { const char *env = getenv("DETEXHIP_DEVICE"); s.device = env ? atoi(env) : 0; }
		c.device = s.device;
	}
	DeviceScope scope(c.device);
	if (!scope.ok) return false;
	auto create = [&]() -> bool {
		HIP_TRY(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking), "hipStreamCreate");
		HIP_TRY(hipMalloc(&c.d_status, kStatusWords * sizeof(uint32_t)), "hipMalloc(status)");
		// zeroed by a COPY on the context's own stream, not by hipMemset: in a fresh process the first fill makes the runtime load its own
		// fill kernels (9.7 ms in the one-shot client's API trace, profiles/r06/oneshot/) and a fill on the null stream a second hardware queue
		static const uint32_t zeros[kStatusWords] = {};
		HIP_TRY(hipMemcpyAsync(c.d_status, zeros, sizeof zeros, hipMemcpyHostToDevice, c.stream), "hipMemcpyAsync(status)");
		return true;
	};
	if (!create()) {			// nothing half-made is kept: the next call starts over
		if (c.d_status) (void)hipFree(c.d_status);
		if (c.stream) (void)hipStreamDestroy(c.stream);
		c.d_status = nullptr; c.stream = nullptr;
		return false;
	}
	c.ready = true;
	return true;
}

// The previous call on this thread failed after it had launched something: wait for whatever is still running, put the device words
// and the pinned header back to zero.  A stream that reports an error is beyond repair: the context is rebuilt from scratch.
bool heal_if_dirty() {
	ThreadContext &c = t_ctx;
	if (!c.dirty) return true;
	DeviceScope scope(c.device);
	if (!scope.ok) return false;
	if ((c.stream_up && hipStreamSynchronize(c.stream_up) != hipSuccess) || hipStreamSynchronize(c.stream) != hipSuccess ||
			hipMemset(c.d_status, 0, kStatusWords * sizeof(uint32_t)) != hipSuccess) {
		(void)hipGetLastError();
		c.release();
		return context_ready();
	}
	if (c.h_pin) memset(c.h_pin, 0, 256);
	c.dirty = false;
	return true;
}
// context_ready() + heal_if_dirty(): what every entry point of this file calls first
bool context_usable() { return context_ready() && heal_if_dirty(); }

// A call that fails after it has launched something returns with the context still `dirty` -- and, on the staged paths, possibly with copies
// out of the caller's blocks or into the caller's pixel buffer still queued (memory the CALLER pinned makes the runtime's copies truly
// asynchronous; with pageable memory they have ended when the call that asked for them returns).  Nothing of this library's may touch the
// caller's memory once the call has returned: every entry point that hands caller memory to the runtime holds one of these, declared
// after its DeviceScope.  (The device words stay as they are: the next call's heal_if_dirty puts them back to zero.)
struct DrainOnFailure {
	ThreadContext &c;
	explicit DrainOnFailure(ThreadContext &context) : c(context) {}
	DrainOnFailure(const DrainOnFailure &) = delete;
	DrainOnFailure &operator=(const DrainOnFailure &) = delete;
	~DrainOnFailure() {
		if (!c.ready || !c.dirty) return;
		if (c.stream_up) (void)hipStreamSynchronize(c.stream_up);
		(void)hipStreamSynchronize(c.stream);
		(void)hipGetLastError();
	}
};

bool reserve(void **buf, size_t *cap, size_t need) {
	if (need <= *cap) return true;
	if (*buf) HIP_TRY(hipFree(*buf), "hipFree");
	*buf = nullptr; *cap = 0;
	const size_t rounded = (need + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
	HIP_TRY(hipMalloc(buf, rounded), "hipMalloc(staging)");
	*cap = rounded;
	return true;
}

// Small calls of the host tier (the one-block leaf functions; textures up to Tune::kHostDirectBytes of blocks + pixels): the
// blocks are placed in a pinned, device-visible host buffer and the kernel reads them from there and writes pixels, ok bytes
// and the status word back into it -- ONE launch and one stream synchronisation instead of memset + upload + launch + two
// downloads (five runtime calls that cost more than the kernel's PCIe traffic for a few KiB).  Layout of the buffer:
// [status word, ok byte, completion word: 256 B][blocks, 256-byte aligned][pixels, 256-byte aligned].
// The caller does not wait for the stream either: the kernel releases a completion word in the same buffer after its last store
// and the caller polls it (wait_for_ticket; path_types.h: Completion has the measurements).
struct DirectExchange { uint8_t *h_base, *d_base; size_t in_off, out_off; };
bool direct_exchange(ThreadContext &c, size_t in_bytes, size_t out_bytes, DirectExchange *x) {
	const size_t in_off = 256, out_off = in_off + ((in_bytes + 255) & ~(size_t)255), need = out_off + ((out_bytes + 255) & ~(size_t)255);
	if (need > c.pin_cap) {
		if (c.h_pin) HIP_TRY(hipHostFree(c.h_pin), "hipHostFree");
		c.h_pin = c.d_pin = nullptr; c.pin_cap = 0;
		const size_t rounded = (need + 65535) & ~(size_t)65535;
		void *h = nullptr, *d = nullptr;
		HIP_TRY(hipHostMalloc(&h, rounded, hipHostMallocMapped), "hipHostMalloc");
		if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) { (void)hipHostFree(h); detexSetErrorMessage("libdetexhip: hipHostGetDevicePointer failed"); return false; }
		memset(h, 0, 256);			// status word, ok byte, completion word: a fresh buffer must not hold what looks like a ticket
		c.h_pin = static_cast<uint8_t *>(h); c.d_pin = static_cast<uint8_t *>(d); c.pin_cap = rounded;
	}
	*x = DirectExchange{ c.h_pin, c.d_pin, in_off, out_off };
	return true;
}
#if defined(__x86_64__) || defined(__i386__)
inline void cpu_relax() { __builtin_ia32_pause(); }
#else
inline void cpu_relax() {}
#endif
constexpr size_t kDoneOffset = 8;		// the completion word inside the exchange buffer's header
uint32_t next_ticket(ThreadContext &c) { if (++c.ticket == 0u) c.ticket = 1u; return c.ticket; }
// Spins on the completion word the kernel just launched on c.stream releases.  Every 2^14 polls (a few hundred microseconds) the
// stream is asked whether it failed or finished without the word (a kernel that faulted never publishes): no unbounded wait.
constexpr size_t kBandDoneOffset = 64;		// ... and of band k of a banded call: kBandDoneOffset + 16 * k (k < kDirectBands)
bool wait_for_ticket(ThreadContext &c, const DirectExchange &x, uint32_t ticket, size_t word_offset = kDoneOffset) {
	const uint32_t *word = reinterpret_cast<const uint32_t *>(x.h_base + word_offset);
	for (uint32_t polls = 1;; polls++) {
		if (__atomic_load_n(word, __ATOMIC_ACQUIRE) == ticket) return true;
		cpu_relax();
		if ((polls & 0x3FFFu) == 0u) {
			const hipError_t e = hipStreamQuery(c.stream);
			if (e == hipErrorNotReady) continue;
			if (e == hipSuccess && __atomic_load_n(word, __ATOMIC_ACQUIRE) == ticket) return true;
			detexSetErrorMessage("libdetexhip: the kernel did not complete: %s", e == hipSuccess ? "completion word not written" : hipGetErrorString(e));
			return false;
		}
	}
}

// shared by the 19 leaf functions and detexDecompressBlock: one block through the GPU.
// Returns 1 = decoded, 0 = the decoder returned false, -1 = HIP/runtime failure (message set).
int decode_one_block(const FormatEntry *f, const uint8_t *bitstring, uint32_t mode_mask, uint32_t flags,
		uint8_t *pixel_buffer, uint32_t pixel_format) {
	if (!context_usable()) return -1;
	ThreadContext &c = t_ctx;
	DeviceScope scope(c.device);
	if (!scope.ok) return -1;
	const size_t bs = detexGetCompressedBlockSize(f->texture_format);
	const size_t out_bytes = 16u * (size_t)detexGetPixelSize(pixel_format);
	const uint32_t decode_flags = (flags & 0x3FFFFFFFu) | current_spec_flags();
	const int epi = prepared_epilogue(f->texture_format, pixel_format, c.stream);
	if (epi == -2) return -1;
	if (c.service.wanted(f, epi)) {		// from the second call in a row on: a request to the resident kernel instead of a launch
		uint32_t payload[12] = {};
		memcpy(payload, bitstring, bs);
		payload[4] = mode_mask; payload[5] = decode_flags; payload[6] = kResidentBlock;
		bool failed = false;
		const uint32_t number = c.service.begin(f, epi);
		if (number != 0u && c.service.serve(payload, number, &failed)) {
			if (failed) return 0;
			memcpy(pixel_buffer, c.service.pixels_host(), out_bytes);
			return 1;
		}
		// (the service has switched itself off with a message; this call still gets its launch)
	}
	DirectExchange x;
	if (!direct_exchange(c, bs, out_bytes, &x)) return -1;
	x.h_base[4] = 0;								// the ok byte
	auto run = [&]() -> bool {
		const uint32_t ticket = next_ticket(c);
		SingleArgs a{ bitstring, mode_mask, decode_flags, reinterpret_cast<uint32_t *>(x.d_base + x.out_off), x.d_base + 4, c.stream, epi,
			reinterpret_cast<uint32_t *>(x.d_base + kDoneOffset), ticket };
		c.dirty = true;
		HIP_TRY(f->single(a), "kernel launch");
		if (injected_failure()) return false;
		return wait_for_ticket(c, x, ticket);
	};
	if (!run()) return -1;
	c.dirty = false;
	if (!x.h_base[4]) return 0;
	memcpy(pixel_buffer, x.h_base + x.out_off, out_bytes);
	return 1;
}

}  // namespace

namespace detexhip {
void release_thread_context() { t_ctx.release(); }
}

// Pixel buffers the kernels can write into directly (include/detexhip.h).  Portable pinned memory: visible to every device.
extern "C" void *detexhipAllocPixelBuffer(size_t bytes) {
	if (bytes == 0) bytes = 16;
	void *h = nullptr, *d = nullptr;
	hipError_t e = hipHostMalloc(&h, bytes, hipHostMallocMapped | hipHostMallocPortable);
	if (e != hipSuccess) { (void)hipGetLastError(); detexSetErrorMessage("detexhipAllocPixelBuffer: hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); return nullptr; }
	if ((e = hipHostGetDevicePointer(&d, h, 0)) != hipSuccess) {
		(void)hipGetLastError(); (void)hipHostFree(h);
		detexSetErrorMessage("detexhipAllocPixelBuffer: hipHostGetDevicePointer failed: %s", hipGetErrorString(e));
		return nullptr;
	}
	std::lock_guard<std::mutex> lock(g_owned_mutex);
	g_owned.push_back(OwnedBuffer{ static_cast<uint8_t *>(h), static_cast<uint8_t *>(d), bytes, 0u });
	return h;
}
extern "C" void detexhipFreePixelBuffer(void *p) {
	if (!p) return;
	{
		std::unique_lock<std::mutex> lock(g_owned_mutex);
		auto find = [&]() { size_t k = 0; while (k < g_owned.size() && g_owned[k].host != p) k++; return k; };
		size_t k = find();
		if (k == g_owned.size()) { detexSetErrorMessage("detexhipFreePixelBuffer: %p was not returned by detexhipAllocPixelBuffer", p); return; }
		// a decode of another thread may be reading or writing the buffer right now (OwnedHold): wait for it -- a decode always ends, its
		// waits are bounded -- rather than free memory under a running kernel; after ten seconds the buffer is LEAKED with a message
		const bool idle = g_owned_idle.wait_for(lock, std::chrono::seconds(10), [&]() { k = find(); return k == g_owned.size() || g_owned[k].in_use == 0; });
		if (k == g_owned.size()) return;			// (freed by another thread meanwhile)
		if (!idle) { detexSetErrorMessage("detexhipFreePixelBuffer: %p is still in use by a decode after 10 s; not freed", p); return; }
		g_owned.erase(g_owned.begin() + (long)k);
	}
	(void)hipHostFree(p);
}
// Test hook: the next `calls` host-pointer calls of the calling thread that reach a kernel launch return false right after it, as if the
// runtime had failed there (the kernels themselves run on); what tests/ use to check that the call after a failed one starts clean.
extern "C" void detexhipTestFailAfterLaunch(int calls) { t_fail_after_launch = calls > 0 ? calls : 0; }

extern "C" void detexhipGetResidentStats(unsigned long long *requests, unsigned long long *instances) {
	if (requests) *requests = t_ctx.service.served;
	if (instances) *instances = t_ctx.service.started;
}

extern "C" int detexhipSetDevice(int device) {
	if (t_ctx.ready && t_ctx.device != device) {	// checked BEFORE touching the current device
		detexSetErrorMessage("libdetexhip: detexhipSetDevice(%d) after this thread already used device %d "
			"(detexhipReleaseThreadResources() first)", device, t_ctx.device);
		return 1;
	}
	int count = 0;
	if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) {
		detexSetErrorMessage("libdetexhip: detexhipSetDevice(%d): no such device (%d present)", device, count);
		return 1;
	}
	thread_settings().device = device;
	t_ctx.device = device;
	return 0;
}

// ------------------------------------------------------------------------------------------------
// reference tier: host pointers (include/detex.h)
// ------------------------------------------------------------------------------------------------
#define LEAF(NAME)                                                                                          \
	extern "C" bool detexDecompressBlock##NAME(const uint8_t *bitstring, uint32_t mode_mask, uint32_t flags, \
			uint8_t *pixel_buffer) {                                                                         \
		return decode_one_block(lookup_format(DETEX_TEXTURE_FORMAT_##NAME), bitstring, mode_mask, flags,  \
			pixel_buffer, DETEX_TEXTURE_FORMAT_##NAME & 0xFFFFu) == 1;                                       \
	}
LEAF(BC1) LEAF(BC1A) LEAF(BC2) LEAF(BC3) LEAF(RGTC1) LEAF(SIGNED_RGTC1) LEAF(RGTC2) LEAF(SIGNED_RGTC2)
LEAF(BPTC_FLOAT) LEAF(BPTC_SIGNED_FLOAT) LEAF(BPTC) LEAF(ETC1) LEAF(ETC2) LEAF(ETC2_PUNCHTHROUGH) LEAF(ETC2_EAC)
LEAF(EAC_R11) LEAF(EAC_SIGNED_R11) LEAF(EAC_RG11) LEAF(EAC_SIGNED_RG11)
#undef LEAF

// texture.c:55-70
extern "C" bool detexDecompressBlock(const uint8_t *bitstring, uint32_t texture_format, uint32_t mode_mask, uint32_t flags,
		uint8_t *pixel_buffer, uint32_t pixel_format) {
	const FormatEntry *f = lookup_format(texture_format);
	if (!f) {
		detexSetErrorMessage("detexDecompressBlock: 0x%08X is not a block-compressed format of this library", texture_format);
		return false;
	}
	if (!pixel_format_accepted(texture_format, pixel_format)) {
		detexSetErrorMessage("detexDecompressBlock: conversion of format 0x%08X to pixel format 0x%08X is outside the "
			"block-decode path of libdetexhip", texture_format, pixel_format);
		return false;
	}
	const int r = decode_one_block(f, bitstring, mode_mask, flags, pixel_buffer, pixel_format);
	if (r == 0)	// same text as the reference (texture.c:63-64); HIP failures have set their own message
		detexSetErrorMessage("detexDecompressBlock: Decompress function for format 0x%08X returned error", texture_format);
	return r == 1;
}

// One call of a texture driver on its way through the host tier: what every path needs, and the paths themselves -- each either finishes
// the call (kTrue / kFalse = the reference's bool result; kFalse also for a HIP failure, with its message set) or does not apply (kNotTaken).
enum Outcome { kNotTaken, kTrue, kFalse };
struct TextureCall {
	ThreadContext &c; const FormatEntry *f; const detexTexture *texture; uint8_t *pixel_buffer; uint32_t pixel_format; bool tiled;
	size_t px, wb, hb, width, height, bs, in_bytes, out_bytes, cov_w, cov_h;

	// same text the reference leaves behind after a failed block (texture.c:63-64)
	Outcome block_failed() const { detexSetErrorMessage("detexDecompressBlock: Decompress function for format 0x%08X returned error", texture->format); return kFalse; }
	void copy_out(const uint8_t *res) const {
		if (tiled || (cov_w == width && cov_h == height)) memcpy(pixel_buffer, res, out_bytes);
		else for (size_t y = 0; y < cov_h; y++) memcpy(pixel_buffer + y * width * px, res + y * width * px, cov_w * px);
	}
	void copy_out_rows(const uint8_t *res, size_t y0, size_t y1) const {		// linear layout: image rows [y0, y1)
		if (y1 > cov_h) y1 = cov_h;
		if (y0 >= y1) return;
		if (cov_w == width) memcpy(pixel_buffer + y0 * width * px, res + y0 * width * px, (y1 - y0) * width * px);
		else for (size_t y = y0; y < y1; y++) memcpy(pixel_buffer + y * width * px, res + y * width * px, cov_w * px);
	}
	// one launch of the one-level form of the mip-chain kernel (any geometry, publishes `completion`): block rows [r0, r1) of the texture,
	// blocks at `blocks_base`, pixels into an image that starts at `pixels_base`
	bool launch_rows(int epi, const uint8_t *blocks_base, uint8_t *pixels_base, uint32_t *status, size_t r0, size_t r1, const Completion &completion, bool *empty) const {
		const size_t y0 = r0 * 4u < height ? r0 * 4u : height, y1 = r1 * 4u < height ? r1 * 4u : height;
		LevelsArgs a{};
		a.status = status; a.stream = c.stream; a.epi = epi; a.decode_flags = current_spec_flags(); a.completion = completion;
		a.table.n_levels = 1;
		LevelDesc &lv = a.table.level[0];
		lv.blocks = blocks_base + r0 * wb * bs; lv.pixels = pixels_base + y0 * width * px; lv.pitch = width * px;
		lv.width_in_blocks = (uint32_t)wb; lv.n_blocks = (uint32_t)(wb * (r1 - r0)); lv.width = (uint32_t)width; lv.height = (uint32_t)(y1 - y0);
		a.table.wg_start[0] = 0; a.table.wg_start[1] = (lv.n_blocks + 255u) / 256u;
		// (rows of blocks that lie below the image are still DECODED -- nothing of them is stored, but an invalid block among them makes the
		// reference's result false, texture.c:122-128, and so it does here)
		*empty = lv.n_blocks == 0;
		if (*empty) return true;
		HIP_TRY(f->levels(a), "kernel launch");
		return true;
	}

	// The smallest textures (either layout), from the second call in a row of one (format, target) pair on: a request to the resident
	// kernel instead of a launch (host_resident.cpp).  Any other call ends the row -- and the instance that served it.
	Outcome via_resident_service() const {
		if (!(in_bytes + out_bytes <= Tune::kHostDirectBytes && wb * hb <= kResidentMaxBlocks && in_bytes <= kResidentBlockBytes && out_bytes <= kResidentPixelBytes)) {
			(void)c.service.wanted(nullptr, -1);
			return kNotTaken;
		}
		const int epi = prepared_epilogue(texture->format, pixel_format, c.stream);
		if (epi == -2) return kFalse;
		if (!c.service.wanted(f, epi)) return kNotTaken;
		// (up to one tile: the blocks travel as tagged chunks the kernel reads along with its polls -- path_types.h: kResidentTagged)
		const bool tagged = wb * hb <= 256u;
		const uint32_t payload[12] = { (uint32_t)width, (uint32_t)height, (uint32_t)wb, (uint32_t)hb, 0xFFFFFFFFu, current_spec_flags(),
			tagged ? kResidentTagged : kResidentTexture, tiled ? 1u : 0u };
		bool failed = false;
		const uint32_t number = c.service.begin(f, epi);
		if (number == 0u) return kNotTaken;
		if (tagged) c.service.pack_tagged(texture->data, in_bytes, number);
		else memcpy(c.service.blocks_host(), texture->data, in_bytes);
		if (!c.service.serve(payload, number, &failed)) return kNotTaken;	// (the service has switched itself off with a message; this call still gets its launch)
		copy_out(c.service.pixels_host());
		return failed ? block_failed() : kTrue;
	}

	// The caller's pixel buffer is one the library handed out (detexhipAllocPixelBuffer: pinned, device-visible): linear textures with up to
	// Tune::kOwnedDirectBytes of pixels are written by the kernel straight into it -- blocks through the pinned exchange buffer, completion
	// polled, nothing copied out.  Shader stores in the kernels' 1 KiB runs cross the link at the DMA engines' rate at these sizes (1 MiB
	// 19 us of data, 4 MiB 87 vs 84; tools/ubench/host_midsize.hip), so the serial kernel -> download step of the staged path and the
	// copy-out of the pinned exchange both disappear.  Larger textures take the staged path, whose download into pinned memory is the
	// fastest copy there is.
	Outcome via_owned_pixel_buffer() const {
		if (tiled || out_bytes > Tune::kOwnedDirectBytes) return kNotTaken;
		// (the kernels' row stores need the alignment the device tier asks for -- device_tier.cpp: palign; an odd sub-range of an owned buffer
		// takes the copying paths like any other pointer)
		const size_t palign = px == 3 ? 1 : (px < 4 ? px : 4);
		if (reinterpret_cast<uintptr_t>(pixel_buffer) % palign != 0) return kNotTaken;
		OwnedHold pixels_hold, blocks_hold;
		uint8_t *dev = pixels_hold.acquire(pixel_buffer, out_bytes);
		if (!dev) return kNotTaken;
		// (the blocks may live in such a buffer too -- it is ordinary pinned memory: then the kernel reads them where they are, whatever
		// their size; blocks that have to be copied into the exchange buffer first are limited like the staged path's pinned input)
		const uint8_t *dev_blocks = reinterpret_cast<uintptr_t>(texture->data) % bs == 0 ? blocks_hold.acquire(texture->data, in_bytes) : nullptr;	// (a block is one 8 / 16-byte load)
		if (!dev_blocks && in_bytes > Tune::kHostPinnedInputBytes) return kNotTaken;
		const int epi = prepared_epilogue(texture->format, pixel_format, c.stream);
		if (epi == -2) return kFalse;
		DirectExchange x;
		if (!direct_exchange(c, dev_blocks ? 0 : in_bytes, 0, &x)) return kFalse;
		if (!dev_blocks) { memcpy(x.h_base + x.in_off, texture->data, in_bytes); dev_blocks = x.d_base + x.in_off; }
		*reinterpret_cast<volatile uint32_t *>(x.h_base) = 0;
		const uint32_t ticket = next_ticket(c);
		bool empty = false;
		c.dirty = true;
		if (!launch_rows(epi, dev_blocks, dev, reinterpret_cast<uint32_t *>(x.d_base), 0, hb,
				Completion{ reinterpret_cast<uint32_t *>(x.d_base + kDoneOffset), c.d_status + kStatusBandCounters, ticket }, &empty)) return drained(kFalse);
		if (injected_failure()) return drained(kFalse);
		if (!empty && !wait_for_ticket(c, x, ticket)) return drained(kFalse);
		c.dirty = false;
		return *reinterpret_cast<volatile uint32_t *>(x.h_base) != 0 ? block_failed() : kTrue;
	}
	// A failure after a launch on a path whose kernel writes into memory that is NOT the thread context's (a held owned buffer): the holds
	// are about to be dropped, so whatever is still running is waited for here (the context stays `dirty` for the next call to clean up).
	Outcome drained(Outcome r) const { (void)hipStreamSynchronize(c.stream); (void)hipGetLastError(); return r; }

	// Blocks + pixels up to Tune::kHostDirectBytes: the kernel reads the blocks from, and writes pixels and status into, pinned host memory
	// (direct_exchange); the caller polls a completion word and copies the pixels out.  Above a quarter MiB of pixels the linear layout
	// goes in kDirectBands bands of block rows -- a band is one contiguous range of blocks and
	// of image rows, texture.c:115-141 -- one launch and one completion word each, all launched at once: band k is copied out while the
	// kernels of the later bands are still writing theirs across the link (the copy-out of freshly written pinned memory runs at ~20 GB/s
	// on one host thread, half the link's rate).
	Outcome via_pinned_exchange() const {
		if (in_bytes + out_bytes > Tune::kHostDirectBytes) return kNotTaken;
		DirectExchange x;
		if (!direct_exchange(c, in_bytes, out_bytes, &x)) return kFalse;
		memcpy(x.h_base + x.in_off, texture->data, in_bytes);
		volatile uint32_t *h_status = reinterpret_cast<volatile uint32_t *>(x.h_base);
		*h_status = 0;
		uint32_t *d_st = reinterpret_cast<uint32_t *>(x.d_base);
		if (tiled) {
			c.dirty = true;
			if (detexhipDecompressTextureTiledDevice(texture->format, x.d_base + x.in_off, (int)wb, (int)hb, x.d_base + x.out_off, pixel_format, c.stream, d_st) != 0) return kFalse;
			if (injected_failure()) return kFalse;
			if (hipStreamSynchronize(c.stream) != hipSuccess) { detexSetErrorMessage("libdetexhip: hipStreamSynchronize failed"); return kFalse; }
			c.dirty = false;
			copy_out(x.h_base + x.out_off);
			return *h_status != 0 ? block_failed() : kTrue;
		}
		const int epi = prepared_epilogue(texture->format, pixel_format, c.stream);
		if (epi == -2) return kFalse;
		const int bands = (out_bytes > ((size_t)256 << 10) && hb >= (size_t)(2 * kDirectBands)) ? kDirectBands : 1;
		uint32_t tickets[kDirectBands];
		size_t band_y1[kDirectBands];
		c.dirty = true;
		for (int b = 0; b < bands; b++) {
			const size_t r0 = (size_t)b * hb / (size_t)bands, r1 = (size_t)(b + 1) * hb / (size_t)bands;
			band_y1[b] = b + 1 == bands ? height : (r1 * 4u < height ? r1 * 4u : height);
			tickets[b] = next_ticket(c);
			bool empty = false;
			if (!launch_rows(epi, x.d_base + x.in_off, x.d_base + x.out_off, d_st, r0, r1,
					Completion{ reinterpret_cast<uint32_t *>(x.d_base + (bands == 1 ? kDoneOffset : kBandDoneOffset + 16u * (size_t)b)), c.d_status + kStatusBandCounters + b, tickets[b] }, &empty)) return kFalse;
			if (empty) tickets[b] = 0;
		}
		if (injected_failure()) return kFalse;
		size_t y_done = 0;
		for (int b = 0; b < bands; b++) {
			if (tickets[b] != 0 && !wait_for_ticket(c, x, tickets[b], bands == 1 ? kDoneOffset : kBandDoneOffset + 16u * (size_t)b)) return kFalse;
			copy_out_rows(x.h_base + x.out_off, y_done, band_y1[b]);
			y_done = band_y1[b];
		}
		c.dirty = false;
		return *h_status != 0 ? block_failed() : kTrue;
	}

	// Larger textures: staged through device memory.  Per call (measured piece by piece, tools/ubench/host_midsize.hip): the status word
	// is NOT zeroed by a memset (the device word is zero between calls: a call that raised it zeroes it again afterwards), it is read from
	// PINNED memory (a 4-byte copy into pageable memory costs 25 us, into pinned memory 13, none at all for up to 2^20 blocks: below), and
	// blocks of up to Tune::kHostPinnedInputBytes reach the kernel through the pinned buffer, read across the link as it decodes (a memcpy
	// of 512 KiB: 4 us; the runtime's copy out of pageable memory: 27).  The pixels travel by the runtime's device-to-host copy into the
	// caller's pageable buffer: at these sizes it pins the pages and runs at the link's rate, which no copy loop of one host thread reaches.
	// (upload -> one launch -> download on the thread's stream; the PCIe download of the pixels bounds it: 8192^2 BC1 256 MiB at 55 GB/s =
	// 4.8 ms, upload 0.6 ms, kernel 0.04 ms.  A band pipeline -- upload k+1 | kernel k | download k-1 on three streams -- was built and
	// measured in round 2: 5.40 vs 5.45 ms, a download that shares the link with an upload runs at 41-50 GB/s; two bands on two streams
	// at 1024^2 in round 5: 135 vs 121 us.  Neither was kept.)
	// Large textures: upload and download at the same time.  The link is full duplex, but a copy between PAGEABLE memory and the device blocks the
	// thread that asks for it (the "async" call returns when the copy is done: profiles/r05/host_paths.txt) -- so a helper thread uploads the
	// blocks band by band (block rows) on a second stream, recording an event per band, while this thread, band by band, makes its stream wait
	// for the band's event, launches the band's decode and downloads the band's pixels.  Whole-block geometries only (every band is then a
	// texture of its own to the device entry); the status word is shared by the bands' launches and read once at the end.
	Outcome via_staging_duplex() const {
		constexpr int B = Tune::kHostDuplexBands;
		// (worth it from 32 MiB of blocks on -- 0.6 ms of upload to hide, of which the helper thread and the eight bands cost ~0.3: BC1 4096^2,
		// 8 MiB of blocks, loses 0.2 ms, BC7 4096^2, 16 MiB, 0.08; BC1 8192^2, 32 MiB, gains 0.18, 64 MiB 0.5-0.55)
		if (Tune::kHostDuplexBytes == 0 || out_bytes < Tune::kHostDuplexBytes || in_bytes < Tune::kHostDuplexBytes || hb < (size_t)(2 * B)) return kNotTaken;
		// (DETEXHIP_HOST_DUPLEX=0 in the environment: never -- for hosts that do not want a library to start threads; read once per process)
		static const bool allowed = []() { const char *e = getenv("DETEXHIP_HOST_DUPLEX"); return !(e && *e == '0'); }();
		if (!allowed) return kNotTaken;
		if (!tiled && !(width == 4u * wb && height == 4u * hb)) return kNotTaken;
		auto try_hip = [](hipError_t e, const char *what) { if (e != hipSuccess) detexSetErrorMessage("libdetexhip: %s failed: %s", what, hipGetErrorString(e)); return e == hipSuccess; };
		if (!c.stream_up) {
			if (!try_hip(hipStreamCreateWithFlags(&c.stream_up, hipStreamNonBlocking), "hipStreamCreate")) { c.stream_up = nullptr; return kFalse; }
			bool all = true;
			for (hipEvent_t &e : c.ev_up)
				if (all && !try_hip(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate")) { e = nullptr; all = false; }
			if (!all) {		// (all or nothing: the next call starts over)
				for (hipEvent_t &e : c.ev_up) { if (e) (void)hipEventDestroy(e); e = nullptr; }
				(void)hipStreamDestroy(c.stream_up);
				c.stream_up = nullptr;
				return kFalse;
			}
		}
		DirectExchange x;
		if (!direct_exchange(c, 0, 0, &x)) return kFalse;
		if (!reserve(&c.d_out, &c.out_cap, out_bytes) || !reserve(&c.d_in, &c.in_cap, in_bytes)) return kFalse;
		// (the status word stays in device memory here and is fetched into the pinned header at the end: a call of this size does not notice the 13 us,
		// and a texture full of invalid blocks would pay a trip across the link per wave for a word in pinned memory -- see via_staging)
		volatile uint32_t *h_status = reinterpret_cast<volatile uint32_t *>(x.h_base + 16);
		uint32_t *d_status = c.d_status + kStagedStatusWord;
		*h_status = 0xFFFFFFFFu;
		c.dirty = true;
		const uint8_t *src = static_cast<const uint8_t *>(texture->data);
		uint8_t *d_in = static_cast<uint8_t *>(c.d_in), *d_out = static_cast<uint8_t *>(c.d_out);
		const size_t in_row = wb * bs, out_row = tiled ? wb * 16u * px : 4u * width * px;		// bytes per block row
		auto row_of = [&](int b) { return hb * (size_t)b / (size_t)B; };
		std::atomic<int> uploaded{ 1 };			// bands whose upload has been issued and whose event is recorded (-1: the uploader failed)
		hipError_t up_error = hipSuccess;
		const int device = c.device;
		// (band 0 goes up from THIS thread, on the main stream, while the helper -- whose first HIP call costs a few hundred microseconds -- starts)
		std::thread uploader;
		const bool started = start_helper_thread(uploader, [&, device]() {
			hipError_t e = hipSetDevice(device);
			for (int b = 1; b < B && e == hipSuccess; b++) {
				const size_t r0 = row_of(b), r1 = row_of(b + 1);
				e = hipMemcpyAsync(d_in + r0 * in_row, src + r0 * in_row, (r1 - r0) * in_row, hipMemcpyHostToDevice, c.stream_up);
				if (e == hipSuccess) e = hipEventRecord(c.ev_up[b], c.stream_up);
				if (e == hipSuccess) uploaded.store(b + 1, std::memory_order_release);
			}
			if (e != hipSuccess) { up_error = e; uploaded.store(-1, std::memory_order_release); }
		});
		if (!started) {		// no thread to be had: nothing has been launched yet, the serial path takes the call
			c.dirty = false;
			return kNotTaken;
		}
		// band b's decode: behind its upload (band 0's went up on this stream; the others' events), one call of the device entry
		auto launch_band = [&](int b) -> bool {
			int seen;
			while ((seen = uploaded.load(std::memory_order_acquire)) >= 0 && seen <= b) std::this_thread::yield();
			if (seen < 0) return false;
			const size_t r0 = row_of(b), r1 = row_of(b + 1);
			if (b > 0 && !try_hip(hipStreamWaitEvent(c.stream, c.ev_up[b], 0), "hipStreamWaitEvent")) return false;
			const int rc = tiled ? detexhipDecompressTextureTiledDevice(texture->format, d_in + r0 * in_row, (int)wb, (int)(r1 - r0), d_out + r0 * out_row, pixel_format, c.stream, d_status)
				: detexhipDecompressTextureLinearDevice(texture->format, d_in + r0 * in_row, (int)width, (int)(4u * (r1 - r0)), (int)wb, (int)(r1 - r0), d_out + r0 * out_row,
					width * px, pixel_format, c.stream, d_status);
			return rc == 0;
		};
		// Band b + 1 is launched BEFORE band b's download is asked for: the download call blocks this thread until the copy is done, and a launch
		// issued only then leaves the link idle for a launch latency per band (8 x ~30 us of a 5 ms call).  Same stream, so the download of band b
		// runs behind the decode of band b + 1 -- which it does not need, and which takes microseconds.
		bool fine = try_hip(hipMemcpyAsync(d_in, src, row_of(1) * in_row, hipMemcpyHostToDevice, c.stream), "hipMemcpyAsync(H2D)") && launch_band(0);
		for (int b = 0; b < B && fine; b++) {
			if (b + 1 < B) fine = launch_band(b + 1);
			const size_t r0 = row_of(b), r1 = row_of(b + 1);
			fine = fine && try_hip(hipMemcpyAsync(pixel_buffer + r0 * out_row, d_out + r0 * out_row, (r1 - r0) * out_row, hipMemcpyDeviceToHost, c.stream), "hipMemcpyAsync(D2H)");
		}
		uploader.join();
		if (uploaded.load(std::memory_order_acquire) < 0) { detexSetErrorMessage("libdetexhip: hipMemcpyAsync(H2D) failed: %s", hipGetErrorString(up_error)); fine = false; }
		if (!fine || injected_failure()) return kFalse;
		if (!try_hip(hipMemcpyAsync(const_cast<uint32_t *>(h_status), d_status, 4, hipMemcpyDeviceToHost, c.stream), "hipMemcpyAsync(D2H)")) return kFalse;
		if (!try_hip(hipStreamSynchronize(c.stream), "hipStreamSynchronize")) return kFalse;
		if (*h_status == 0) { c.dirty = false; return kTrue; }
		// (the device word is zero between calls: restore that before reporting)
		if (!try_hip(hipMemsetAsync(d_status, 0, 4, c.stream), "hipMemsetAsync") || !try_hip(hipStreamSynchronize(c.stream), "hipStreamSynchronize")) return kFalse;
		c.dirty = false;
		return block_failed();
	}

	Outcome via_staging() const {
		const bool pinned_in = in_bytes <= Tune::kHostPinnedInputBytes;
		DirectExchange x;
		if (!direct_exchange(c, pinned_in ? in_bytes : 0, 0, &x)) return kFalse;
		if (!reserve(&c.d_out, &c.out_cap, out_bytes) || (!pinned_in && !reserve(&c.d_in, &c.in_cap, in_bytes))) return kFalse;
		uint8_t *d_out = static_cast<uint8_t *>(c.d_out);
		const uint8_t *d_in;
		auto try_hip = [](hipError_t e, const char *what) { if (e != hipSuccess) detexSetErrorMessage("libdetexhip: %s failed: %s", what, hipGetErrorString(e)); return e == hipSuccess; };
		c.dirty = true;		// (from the first thing handed to the runtime on)
		if (pinned_in) {
			memcpy(x.h_base + x.in_off, texture->data, in_bytes);
			d_in = x.d_base + x.in_off;
		} else {
			if (!try_hip(hipMemcpyAsync(c.d_in, texture->data, in_bytes, hipMemcpyHostToDevice, c.stream), "hipMemcpyAsync(H2D)")) return kFalse;
			d_in = static_cast<const uint8_t *>(c.d_in);
		}
		// The status word: up to 2^18 blocks (2048^2) it lives in the pinned header itself -- only a wave that holds a failed block touches it (one
		// load, at most one store across the link), and the call saves the copy that would fetch it (13 us of a call of at most 0.4 ms); beyond
		// that it stays in device memory and is fetched into the pinned word: every wave of a texture FULL of invalid blocks makes that trip across
		// the link (random BC6H, an eighth of whose blocks are reserved modes: 4096^2, 16384 waves, 5.6 ms instead of 2.6 with the word in pinned
		// memory -- round 6, when the limit was still 2^20 blocks).
		const bool pinned_status = wb * hb <= ((size_t)1 << 18);
		volatile uint32_t *h_status = reinterpret_cast<volatile uint32_t *>(x.h_base + 16);		// (header of the exchange buffer: [0] status of the direct path, [8] its completion word)
		uint32_t *d_status = pinned_status ? reinterpret_cast<uint32_t *>(x.d_base + 16) : c.d_status + kStagedStatusWord;
		*h_status = pinned_status ? 0u : 0xFFFFFFFFu;
		const int rc = tiled ? detexhipDecompressTextureTiledDevice(texture->format, d_in, (int)wb, (int)hb, d_out, pixel_format, c.stream, d_status)
			: detexhipDecompressTextureLinearDevice(texture->format, d_in, (int)width, (int)height, (int)wb, (int)hb, d_out, width * px, pixel_format, c.stream, d_status);
		if (rc != 0) return kFalse;
		if (injected_failure()) return kFalse;
		if (tiled || (cov_w == width && cov_h == height)) {
			if (!try_hip(hipMemcpyAsync(pixel_buffer, d_out, out_bytes, hipMemcpyDeviceToHost, c.stream), "hipMemcpyAsync(D2H)")) return kFalse;
		} else if (cov_w > 0 && cov_h > 0) {
			if (!try_hip(hipMemcpy2DAsync(pixel_buffer, width * px, d_out, width * px, cov_w * px, cov_h, hipMemcpyDeviceToHost, c.stream), "hipMemcpy2DAsync(D2H)")) return kFalse;
		}
		if (!pinned_status && !try_hip(hipMemcpyAsync(const_cast<uint32_t *>(h_status), d_status, 4, hipMemcpyDeviceToHost, c.stream), "hipMemcpyAsync(D2H)")) return kFalse;
		if (!try_hip(hipStreamSynchronize(c.stream), "hipStreamSynchronize")) return kFalse;
		if (*h_status == 0) { c.dirty = false; return kTrue; }
		if (!pinned_status) {		// (the device word is zero between calls: restore that before reporting)
			if (!try_hip(hipMemsetAsync(d_status, 0, 4, c.stream), "hipMemsetAsync") || !try_hip(hipStreamSynchronize(c.stream), "hipStreamSynchronize")) return kFalse;
		}
		c.dirty = false;
		return block_failed();
	}
};

// shared body of the two texture drivers (texture.c:77-98, 105-145): argument checks and the reference's non-decode edges here, the
// decode in TextureCall above
static bool decompress_texture(const detexTexture *texture, uint8_t *pixel_buffer, uint32_t pixel_format, bool tiled) {
	const char *who = tiled ? "detexDecompressTextureTiled" : "detexDecompressTextureLinear";
	const size_t px = (size_t)detexGetPixelSize(pixel_format);
	if (texture->width < 0 || texture->height < 0 || texture->width_in_blocks < 0 || texture->height_in_blocks < 0) {
		detexSetErrorMessage("%s: negative texture dimensions", who);
		return false;
	}
	const size_t wb = (size_t)texture->width_in_blocks, hb = (size_t)texture->height_in_blocks;
	const size_t width = (size_t)texture->width, height = (size_t)texture->height;
	const size_t out_bytes = tiled ? wb * hb * 16u * px : width * height * px;
	if (!detexFormatIsCompressed(texture->format)) {
		if (tiled) { detexSetErrorMessage("detexDecompressTextureTiled: Cannot handle uncompressed texture format"); return false; }
		// texture.c:108-111 hands uncompressed textures to detexConvertPixels; only its identity
		// edge (convert.c:1087-1092) belongs to this path.
		const uint32_t src = detexGetPixelFormat(texture->format);
		const bool same8 = (src == DETEX_PIXEL_FORMAT_RGBA8 || src == DETEX_PIXEL_FORMAT_RGBX8) &&
			(pixel_format == DETEX_PIXEL_FORMAT_RGBA8 || pixel_format == DETEX_PIXEL_FORMAT_RGBX8);
		if (src == pixel_format || same8) { memcpy(pixel_buffer, texture->data, out_bytes); return true; }
		detexSetErrorMessage("%s: pixel conversion 0x%08X -> 0x%08X is outside the block-decode path of libdetexhip", who, src, pixel_format);
		return false;
	}
	const FormatEntry *f = lookup_format(texture->format);
	if (!f || !pixel_format_accepted(texture->format, pixel_format)) {
		// the reference fails every block here: all-zero image and false
		memset(pixel_buffer, 0, out_bytes);
		detexSetErrorMessage("%s: format 0x%08X -> pixel format 0x%08X is outside the block-decode path of libdetexhip", who,
			texture->format, pixel_format);
		return false;
	}
	if (out_bytes == 0 || wb * hb == 0) return true;
	if (!context_usable()) return false;
	ThreadContext &c = t_ctx;
	DeviceScope scope(c.device);
	if (!scope.ok) return false;
	DrainOnFailure drain(c);
	const size_t bs = detexGetCompressedBlockSize(texture->format);
	TextureCall call{ c, f, texture, pixel_buffer, pixel_format, tiled, px, wb, hb, width, height, bs, wb * hb * bs, out_bytes,
		// The reference writes only the pixels its block grid covers and the image contains (texture.c:116-136): when the
		// grid is smaller than the image, the rest of the caller's buffer is left untouched, not overwritten with staging bytes.
		tiled ? 0 : (width < 4u * wb ? width : 4u * wb), tiled ? 0 : (height < 4u * hb ? height : 4u * hb) };
	// by size: the resident service (up to 1024 blocks, from the second call in a row on), a pixel buffer the library handed out (written
	// directly, up to 8 MiB), the pinned exchange (up to 1.25 MiB in all), staging through device memory.  (The caller's OWN buffer is never
	// registered with the runtime: its lifetime is not the library's to know -- DESIGN.md section 5.)
	Outcome r = call.via_resident_service();
	if (r == kNotTaken) r = call.via_owned_pixel_buffer();
	if (r == kNotTaken) r = call.via_pinned_exchange();
	if (r == kNotTaken) r = call.via_staging_duplex();
	if (r == kNotTaken) r = call.via_staging();
	return r == kTrue;
}

// 8f-3 host tier: what a caller of detexLoadKTXFileWithMipmaps does level by level (one
// detexDecompressTextureLinear per level), as one staging copy in, ONE launch, one copy out.
extern "C" bool detexhipDecompressTexturesLinear(const detexTexture *const *textures, int n_textures,
		uint8_t *const *pixel_buffers, uint32_t pixel_format) {
	const char *who = "detexhipDecompressTexturesLinear";
	if (n_textures <= 0) return true;
	if (n_textures > kMaxLevels) { detexSetErrorMessage("%s: at most %d textures per call", who, kMaxLevels); return false; }
	const uint32_t format = textures[0]->format;
	const FormatEntry *f = lookup_format(format);
	const size_t px = (size_t)detexGetPixelSize(pixel_format);
	if (!f || !pixel_format_accepted(format, pixel_format)) {
		for (int l = 0; l < n_textures; l++) memset(pixel_buffers[l], 0, (size_t)textures[l]->width * (size_t)textures[l]->height * px);
		detexSetErrorMessage("%s: format 0x%08X -> pixel format 0x%08X is outside the block-decode path of libdetexhip", who, format, pixel_format);
		return false;
	}
	if (!context_usable()) return false;
	ThreadContext &c = t_ctx;
	DeviceScope scope(c.device);
	if (!scope.ok) return false;
	DrainOnFailure drain(c);
	const size_t bs = detexGetCompressedBlockSize(format);
	size_t in_off[kMaxLevels], out_off[kMaxLevels], in_total = 0, out_total = 0;
	for (int l = 0; l < n_textures; l++) {
		if (textures[l]->format != format) { detexSetErrorMessage("%s: all textures must share one format", who); return false; }
		in_off[l] = in_total; out_off[l] = out_total;
		in_total += ((size_t)textures[l]->width_in_blocks * (size_t)textures[l]->height_in_blocks * bs + 255) & ~(size_t)255;
		out_total += ((size_t)textures[l]->width * (size_t)textures[l]->height * px + 255) & ~(size_t)255;
	}
	if (!reserve(&c.d_in, &c.in_cap, in_total ? in_total : 256) || !reserve(&c.d_out, &c.out_cap, out_total ? out_total : 256)) return false;
	detexhipLevel lv[kMaxLevels];
	c.dirty = true;
	for (int l = 0; l < n_textures; l++) {
		const detexTexture *t = textures[l];
		const size_t nbytes = (size_t)t->width_in_blocks * (size_t)t->height_in_blocks * bs;
		if (nbytes) HIP_TRY(hipMemcpyAsync(static_cast<uint8_t *>(c.d_in) + in_off[l], t->data, nbytes, hipMemcpyHostToDevice, c.stream), "hipMemcpyAsync(H2D)");
		lv[l] = detexhipLevel{ static_cast<uint8_t *>(c.d_in) + in_off[l], static_cast<uint8_t *>(c.d_out) + out_off[l],
			(size_t)t->width * px, t->width, t->height, t->width_in_blocks, t->height_in_blocks };
	}
	if (detexhipDecompressLevelsLinearDevice(format, lv, n_textures, pixel_format, c.stream, c.d_status) != 0) return false;
	if (injected_failure()) return false;
	uint32_t status = 0;
	for (int l = 0; l < n_textures; l++) {
		const size_t nbytes = (size_t)textures[l]->width * (size_t)textures[l]->height * px;
		if (nbytes) HIP_TRY(hipMemcpyAsync(pixel_buffers[l], static_cast<uint8_t *>(c.d_out) + out_off[l], nbytes, hipMemcpyDeviceToHost, c.stream), "hipMemcpyAsync(D2H)");
	}
	HIP_TRY(hipMemcpyAsync(&status, c.d_status, 4, hipMemcpyDeviceToHost, c.stream), "hipMemcpyAsync(D2H)");
	HIP_TRY(hipMemsetAsync(c.d_status, 0, 4, c.stream), "hipMemsetAsync");		// (zero between calls)
	HIP_TRY(hipStreamSynchronize(c.stream), "hipStreamSynchronize");
	c.dirty = false;
	if (status != 0) {
		detexSetErrorMessage("detexDecompressBlock: Decompress function for format 0x%08X returned error", format);
		return false;
	}
	return true;
}

// The batched form of the per-block API on HOST pointers: what a client that loops over detexDecompressBlock<FMT> (detex.h:435-531;
// 0.03 us per call in the reference, 5-8 us per call here: every call is a trip to the GPU) calls ONCE instead.  n_blocks
// independent blocks, mode_mask and flags honoured per block exactly as the leaf functions do; pixels = 16 pixels per block in the
// format's native pixel format, block after block; ok[i] (optional) = the leaf function's bool for block i; a failed block is
// zero-filled.  Returns true if every block decoded (the texture drivers' convention, texture.c:125-128,144), else false with the
// reference's error text.  One block goes the leaf functions' way (resident kernel from the second call in a row on); batches whose
// blocks + pixels + ok bytes fit Tune::kHostDirectBytes are exchanged through pinned host memory (one launch, completion polled);
// larger ones are staged through the thread's device buffers (upload, one launch, download).
extern "C" bool detexhipDecompressBlocks(uint32_t texture_format, const uint8_t *blocks, size_t n_blocks, uint32_t mode_mask, uint32_t flags,
		uint8_t *pixels, uint8_t *ok) {
	const char *who = "detexhipDecompressBlocks";
	const FormatEntry *f = lookup_format(texture_format);
	if (!f) { detexSetErrorMessage("%s: 0x%08X is not a block-compressed format of this library", who, texture_format); return false; }
	if (n_blocks == 0) return true;
	if (!blocks || !pixels) { detexSetErrorMessage("%s: NULL blocks / pixels", who); return false; }
	if (n_blocks > 0xFFFFFF00ull) { detexSetErrorMessage("%s: too many blocks", who); return false; }
	const uint32_t pixel_format = texture_format & 0xFFFFu;
	const size_t bs = detexGetCompressedBlockSize(texture_format), out_per_block = 16u * (size_t)detexGetPixelSize(pixel_format);
	auto failed_text = [&]() { detexSetErrorMessage("detexDecompressBlock: Decompress function for format 0x%08X returned error", texture_format); };
	if (n_blocks == 1) {
		const int r = decode_one_block(f, blocks, mode_mask, flags, pixels, pixel_format);
		if (ok && r >= 0) ok[0] = r == 1 ? 1 : 0;
		if (r == 0) { memset(pixels, 0, out_per_block); failed_text(); }
		return r == 1;
	}
	if (!context_usable()) return false;
	ThreadContext &c = t_ctx;
	DeviceScope scope(c.device);
	if (!scope.ok) return false;
	DrainOnFailure drain(c);
	(void)c.service.wanted(nullptr, -1);		// not a call for the resident kernel: ends a row of small calls
	const size_t in_bytes = n_blocks * bs, out_bytes = n_blocks * out_per_block;
	const uint32_t decode_flags = (flags & 0x3FFFFFFFu) | current_spec_flags();
	BatchArgs a{};
	a.n = n_blocks; a.mode_mask = mode_mask; a.flags = decode_flags; a.stream = c.stream; a.checked = true; a.epi = kEpiNone; a.resident = 0;
	if (in_bytes + out_bytes + n_blocks <= Tune::kHostDirectBytes) {
		// [header][blocks][pixels + ok bytes]: the ok bytes follow the pixels (256-byte aligned) inside the exchange buffer's output part
		const size_t ok_off = (out_bytes + 255) & ~(size_t)255;
		DirectExchange x;
		if (!direct_exchange(c, in_bytes, ok_off + n_blocks, &x)) return false;
		memcpy(x.h_base + x.in_off, blocks, in_bytes);
		*reinterpret_cast<volatile uint32_t *>(x.h_base) = 0;
		const uint32_t ticket = next_ticket(c);
		a.blocks = x.d_base + x.in_off; a.pixels = x.d_base + x.out_off; a.ok = x.d_base + x.out_off + ok_off;
		a.status = reinterpret_cast<uint32_t *>(x.d_base);
		a.completion = Completion{ reinterpret_cast<uint32_t *>(x.d_base + kDoneOffset), c.d_status + kStatusBandCounters, ticket };
		c.dirty = true;
		HIP_TRY(f->blocks(a), "kernel launch");
		if (injected_failure()) return false;
		if (!wait_for_ticket(c, x, ticket)) return false;
		c.dirty = false;
		memcpy(pixels, x.h_base + x.out_off, out_bytes);
		if (ok) memcpy(ok, x.h_base + x.out_off + ok_off, n_blocks);
		if (*reinterpret_cast<volatile uint32_t *>(x.h_base) != 0) { failed_text(); return false; }
		return true;
	}
	// staged: the ok bytes are decoded into the tail of the output staging buffer and come back with their own copy
	const size_t ok_off = (out_bytes + 255) & ~(size_t)255;
	if (!reserve(&c.d_in, &c.in_cap, in_bytes) || !reserve(&c.d_out, &c.out_cap, ok_off + n_blocks)) return false;
	uint8_t *d_out = static_cast<uint8_t *>(c.d_out);
	c.dirty = true;
	HIP_TRY(hipMemcpyAsync(c.d_in, blocks, in_bytes, hipMemcpyHostToDevice, c.stream), "hipMemcpyAsync(H2D)");
	a.blocks = c.d_in; a.pixels = d_out; a.ok = d_out + ok_off; a.status = c.d_status;
	HIP_TRY(f->blocks(a), "kernel launch");
	if (injected_failure()) return false;
	uint32_t status = 0;
	HIP_TRY(hipMemcpyAsync(pixels, d_out, out_bytes, hipMemcpyDeviceToHost, c.stream), "hipMemcpyAsync(D2H)");
	if (ok) HIP_TRY(hipMemcpyAsync(ok, d_out + ok_off, n_blocks, hipMemcpyDeviceToHost, c.stream), "hipMemcpyAsync(D2H)");
	HIP_TRY(hipMemcpyAsync(&status, c.d_status, 4, hipMemcpyDeviceToHost, c.stream), "hipMemcpyAsync(D2H)");
	HIP_TRY(hipMemsetAsync(c.d_status, 0, 4, c.stream), "hipMemsetAsync");		// (zero between calls)
	HIP_TRY(hipStreamSynchronize(c.stream), "hipStreamSynchronize");
	c.dirty = false;
	if (status != 0) { failed_text(); return false; }
	return true;
}

// 8f-4 host tier: histogram[m] = number of blocks whose detexGetMode<FMT> is m (bin 15: reserved codes)
extern "C" bool detexhipModeHistogram(uint32_t texture_format, const uint8_t *blocks, size_t n_blocks, uint32_t histogram[16]) {
	const FormatEntry *f = lookup_format(texture_format);
	if (!f) { detexSetErrorMessage("detexhipModeHistogram: 0x%08X is not a block-compressed format of this library", texture_format); return false; }
	if (!context_usable()) return false;
	ThreadContext &c = t_ctx;
	DeviceScope scope(c.device);
	if (!scope.ok) return false;
	DrainOnFailure drain(c);
	const size_t nbytes = n_blocks * detexGetCompressedBlockSize(texture_format);
	if (!reserve(&c.d_in, &c.in_cap, nbytes ? nbytes : 256)) return false;
	c.dirty = true;
	if (nbytes) HIP_TRY(hipMemcpyAsync(c.d_in, blocks, nbytes, hipMemcpyHostToDevice, c.stream), "hipMemcpyAsync(H2D)");
	uint32_t *d_hist = c.d_status;		// words 0 .. kStatusHistogramBins - 1 of the status allocation, zeroed again below (the words are zero between calls)
	if (detexhipModeHistogramDevice(texture_format, c.d_in, n_blocks, d_hist, c.stream) != 0) return false;
	HIP_TRY(hipMemcpyAsync(histogram, d_hist, kStatusHistogramBins * sizeof(uint32_t), hipMemcpyDeviceToHost, c.stream), "hipMemcpyAsync(D2H)");
	HIP_TRY(hipMemsetAsync(d_hist, 0, kStatusHistogramBins * sizeof(uint32_t), c.stream), "hipMemsetAsync");
	HIP_TRY(hipStreamSynchronize(c.stream), "hipStreamSynchronize");
	c.dirty = false;
	return true;
}

extern "C" bool detexDecompressTextureTiled(const detexTexture *texture, uint8_t *pixel_buffer, uint32_t pixel_format) {
	return decompress_texture(texture, pixel_buffer, pixel_format, true);
}

extern "C" bool detexDecompressTextureLinear(const detexTexture *texture, uint8_t *pixel_buffer, uint32_t pixel_format) {
	return decompress_texture(texture, pixel_buffer, pixel_format, false);
}

// The same two drivers under names that do not collide with the reference's: for a libdetex that forwards its own
// detexDecompressTextureLinear / Tiled here while both libraries are linked.
extern "C" bool detexhipHostDecompressTextureLinear(const detexTexture *texture, uint8_t *pixel_buffer, uint32_t pixel_format) {
	return decompress_texture(texture, pixel_buffer, pixel_format, false);
}
extern "C" bool detexhipHostDecompressTextureTiled(const detexTexture *texture, uint8_t *pixel_buffer, uint32_t pixel_format) {
	return decompress_texture(texture, pixel_buffer, pixel_format, true);
}

