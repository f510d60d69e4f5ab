// host_resident.cpp -- host side of the resident service kernel (kernels_resident.h; protocol and measurements: path_types.h
// ResidentMail).  Host code only.
//
// Nothing here can wait forever: the kernel leaves by itself (idle time, total time, poll count) and always says so in
// ResidentMail::state; the host polls `done` AND `state`, starts a new instance when the old one has left with the request
// unanswered, and every 2^14 polls asks the stream whether the kernel faulted.
#include <atomic>
#include <cstdlib>
#include <cstring>
#if defined(__x86_64__)
#include <emmintrin.h>
#endif

#include "host_internal.h"

namespace detexhip {

namespace {
std::atomic<int> g_idle_us{ -1 };		// -1 = not read yet
constexpr uint32_t kMailBytes = 256;
static_assert(sizeof(ResidentMail) == kMailBytes, "ResidentMail is four 64-byte lines");
constexpr size_t kBufferBytes = kMailBytes + (size_t)kResidentBlockBytes + (size_t)kResidentPixelBytes;
#if defined(__x86_64__) || defined(__i386__)
inline void cpu_relax() { __builtin_ia32_pause(); }
#else
inline void cpu_relax() {}
#endif
}  // namespace

int resident_idle_microseconds() {
	int v = g_idle_us.load(std::memory_order_relaxed);
	if (v < 0) {
		const char *env = getenv("DETEXHIP_RESIDENT_US");
		v = env ? atoi(env) : 100;		// (also what an application's hipDeviceSynchronize() can wait longer right after a small call)
		if (v < 0) v = 0;
		if (v > 1000000) v = 1000000;
		g_idle_us.store(v, std::memory_order_relaxed);
	}
	return v;
}

bool ResidentService::wanted(const FormatEntry *fmt, int epilogue) {
	const bool repeat = fmt != nullptr && prev_f == fmt && prev_epi == epilogue;
	prev_f = fmt; prev_epi = epilogue;
	// A call that does not continue the row ends the running instance NOW (a stop request and its answer: one round trip across the
	// link) instead of leaving it to spin out its idle time: what such a call does next -- hipMalloc / hipFree of staging buffers, a
	// table upload -- synchronises with the device and would wait behind that kernel, as would other streams sharing its hardware queue.
	// (quiet: a failure here marks the service broken, but THIS call goes on to its own launch and may well succeed -- it must not leave
	// a 'resident kernel failed' text behind a call that returns true)
	if (!repeat && launched) (void)stop(true);
	if (!repeat || broken || resident_idle_microseconds() <= 0) return false;
	int device = 0;
	if (hipGetDevice(&device) != hipSuccess) return false;
	return prepare(device);			// (the caller fills blocks_host() before serve())
}

uint8_t *ResidentService::blocks_host() { return h_buf + kMailBytes; }
const uint8_t *ResidentService::pixels_host() const { return h_buf + kMailBytes + kResidentBlockBytes; }

bool ResidentService::prepare(int device) {
	if (ready) return true;
	auto make = [&]() -> bool {
		HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking), "hipStreamCreate");
		void *h = nullptr, *d = nullptr;
		// (coherent pinned memory or no service: with non-coherent memory the kernel's polls and the host's view of `done` / `state` are
		// not guaranteed to meet, and the calls would only be rescued by their timeouts)
		HIP_TRY(hipHostMalloc(&h, kBufferBytes, hipHostMallocMapped | hipHostMallocCoherent), "hipHostMalloc(coherent)");
		h_buf = static_cast<uint8_t *>(h);
		HIP_TRY(hipHostGetDevicePointer(&d, h, 0), "hipHostGetDevicePointer");
		d_buf = static_cast<uint8_t *>(d);
		memset(h_buf, 0, kMailBytes);
		HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d_words), 64), "hipMalloc");
		HIP_TRY(hipMemset(d_words, 0, 64), "hipMemset");
		int khz = 0;
		if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device) == hipSuccess && khz >= 1000) ticks_per_us = (uint64_t)khz / 1000u;
		else (void)hipGetLastError();
		return true;
	};
	if (!make()) {			// nothing half-made is kept
		if (d_words) (void)hipFree(d_words);
		if (h_buf) (void)hipHostFree(h_buf);
		if (stream) (void)hipStreamDestroy(stream);
		d_words = nullptr; h_buf = d_buf = nullptr; stream = nullptr;
		broken = true;
		return false;
	}
	ready = true;
	return true;
}

bool ResidentService::launch(uint32_t start_seq) {
	ResidentMail *mail = reinterpret_cast<ResidentMail *>(h_buf);
	ResidentLaunch l{};
	instance = (instance + 1u) & 0x3FFFFFFFu;
	l.args.mail = reinterpret_cast<ResidentMail *>(d_buf);
	l.args.blocks = d_buf + kMailBytes;
	l.args.pixels = d_buf + kMailBytes + kResidentBlockBytes;
	l.args.words = d_words;
	l.args.start_seq = start_seq;
	l.args.instance = instance;
	l.args.idle_ticks = (uint64_t)resident_idle_microseconds() * ticks_per_us;
	l.args.max_ticks = 10000000ull * ticks_per_us;		// ten seconds: a new instance takes over with the next request
	{ static const char *env = getenv("DETEXHIP_RESIDENT_SPECULATIVE_POLLS"); l.args.speculative_polls = env ? (uint32_t)atoi(env) : 16u; }
	l.stream = stream; l.epi = epi;
	__atomic_store_n(&mail->state, instance << 2 | kResidentRunning, __ATOMIC_RELEASE);	// (no instance is running: the line is the host's for now)
	HIP_TRY(f->service(l), "kernel launch (resident)");
	launched = true;
	started++;
	return true;
}

void ResidentService::post(const uint32_t payload[12], uint32_t number) {
	ResidentMail *mail = reinterpret_cast<ResidentMail *>(h_buf);
	for (int k = 0; k < 4; k++) {
#if defined(__x86_64__)
		_mm_store_si128(reinterpret_cast<__m128i *>(mail->chunk[k]), _mm_set_epi32((int)payload[3 * k + 2], (int)payload[3 * k + 1], (int)payload[3 * k], (int)number));
#else
		typedef uint32_t v4u __attribute__((vector_size(16)));
		*reinterpret_cast<volatile v4u *>(mail->chunk[k]) = v4u{ number, payload[3 * k], payload[3 * k + 1], payload[3 * k + 2] };
#endif
	}
	__atomic_thread_fence(__ATOMIC_SEQ_CST);	// the request is out before `state` is looked at (the kernel announces its leaving, then looks again)
}

// ends the running instance (a request it must see before the next one may be posted: the next one may be for another format's kernel)
bool ResidentService::stop(bool quiet) {
	if (!launched) return true;
	ResidentMail *mail = reinterpret_cast<ResidentMail *>(h_buf);
	uint32_t payload[12] = {};
	payload[6] = kResidentStop;
	if (++seq == 0u) seq = 1u;
	post(payload, seq);
	const uint32_t gone = instance << 2 | kResidentExited;
	for (uint32_t polls = 1; __atomic_load_n(&mail->state, __ATOMIC_ACQUIRE) != gone; polls++) {
		cpu_relax();
		if ((polls & 0x3FFFu) == 0u) {
			const hipError_t e = hipStreamQuery(stream);
			if (e == hipErrorNotReady) continue;
			if (e != hipSuccess) {
				if (!quiet) detexSetErrorMessage("libdetexhip: the resident kernel failed: %s", hipGetErrorString(e));
				(void)hipGetLastError();
				launched = false; broken = true;
				return false;
			}
			break;		// the stream is idle: no instance left, whatever the word says
		}
	}
	launched = false;
	return true;
}

uint32_t ResidentService::begin(const FormatEntry *fmt, int epilogue) {
	if (!ready) { detexSetErrorMessage("libdetexhip: resident service used before it was prepared"); return 0u; }
	if (launched && (f != fmt || epi != epilogue) && !stop()) return 0u;
	f = fmt; epi = epilogue;
	before = seq;
	if (++seq == 0u) seq = 1u;
	return seq;
}

// 16-byte chunks {eight block bytes, request number, 0}: whatever moment the kernel reads one at, its tag says whose data it is
void ResidentService::pack_tagged(const void *blocks, size_t bytes, uint32_t number) {
	const uint8_t *src = static_cast<const uint8_t *>(blocks);
	uint8_t *dst = blocks_host();
	for (size_t k = 0; k < bytes / 8u; k++) {
		uint32_t w[2];
		memcpy(w, src + 8u * k, 8);
#if defined(__x86_64__)
		_mm_store_si128(reinterpret_cast<__m128i *>(dst + 16u * k), _mm_set_epi32(0, (int)number, (int)w[1], (int)w[0]));
#else
		typedef uint32_t v4u __attribute__((vector_size(16)));
		*reinterpret_cast<volatile v4u *>(dst + 16u * k) = v4u{ w[0], w[1], number, 0u };
#endif
	}
}

bool ResidentService::serve(const uint32_t payload[12], uint32_t number, bool *failed) {
	if (!ready || number != seq) { detexSetErrorMessage("libdetexhip: resident service: serve() without begin()"); return false; }
	ResidentMail *mail = reinterpret_cast<ResidentMail *>(h_buf);
	__atomic_store_n(&mail->status, 0u, __ATOMIC_RELAXED);
	post(payload, seq);
	if (!launched && !launch(before)) { broken = true; return false; }
	int relaunches = 0;
	for (uint32_t polls = 1;; polls++) {
		if (__atomic_load_n(&mail->done, __ATOMIC_ACQUIRE) == seq) break;
		bool gone = __atomic_load_n(&mail->state, __ATOMIC_ACQUIRE) == (instance << 2 | kResidentExited);
		if (!gone && (polls & 0x3FFFu) == 0u) {
			const hipError_t e = hipStreamQuery(stream);
			if (e != hipErrorNotReady && e != hipSuccess) {
				detexSetErrorMessage("libdetexhip: the resident kernel failed: %s", hipGetErrorString(e));
				launched = false; broken = true;
				return false;
			}
			gone = e == hipSuccess;		// the stream is idle: no instance is running
		}
		if (gone) {
			// the instance left (idle time, total time) -- after this request, or without having seen it
			if (__atomic_load_n(&mail->done, __ATOMIC_ACQUIRE) == seq) { launched = false; break; }
			if (++relaunches > 3) { detexSetErrorMessage("libdetexhip: the resident kernel leaves without answering"); launched = false; broken = true; return false; }
			if (!launch(before)) { broken = true; return false; }
			continue;
		}
		cpu_relax();
	}
	*failed = __atomic_load_n(&mail->status, __ATOMIC_ACQUIRE) != 0u;
	served++;
	return true;
}

void ResidentService::release() {
	if (!ready) return;
	(void)stop();
	(void)hipStreamSynchronize(stream);
	(void)hipFree(d_words);
	(void)hipHostFree(h_buf);
	(void)hipStreamDestroy(stream);
	d_words = nullptr; h_buf = d_buf = nullptr; stream = nullptr;
	ready = launched = false;
	prev_f = nullptr; prev_epi = -1;
}

}  // namespace detexhip

// 0 = the small calls of the host tier never leave a kernel behind (a launch per call); otherwise the time a resident kernel waits for
// the next request before it leaves.  Process-wide; takes effect with the next instance started.  Returns the previous value.
extern "C" int detexhipSetResidentIdleMicroseconds(int microseconds) {
	const int before = detexhip::resident_idle_microseconds();
	if (microseconds < 0) microseconds = 0;
	if (microseconds > 1000000) microseconds = 1000000;
	detexhip::g_idle_us.store(microseconds, std::memory_order_relaxed);
	return before;
}
