// tools/ubench/host_latency.hip -- what one small launch costs a host-pointer caller on this box, and which way of waiting for it is
// the cheapest (measurement tool, not product; decides how host_tier.cpp completes its small calls).
//   usage: host_latency [spin]        `spin`: hipSetDeviceFlags(hipDeviceScheduleSpin) before the context is created
// The kernel stands in for the small-texture decode: it reads `in_bytes` from a pinned host buffer (or takes 16 bytes as an argument),
// writes `out_bytes` into pinned host memory and then releases a completion word there.  Per variant: us per call (median and mean over
// 2000 calls after 200 warm-ups), and for the plain variant the split launch-call / wait.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <csetjmp>
#include <csignal>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// 256 threads; each copies its share of `in` to `out` scaled up 8x (a BC1 decode writes 8 bytes per byte read), then thread 0 of the
// LAST workgroup to finish publishes `ticket` in *done (system scope), after all stores of the grid
__global__ void work(const uint32_t *in, uint32_t in_dwords, uint4 arg, uint32_t *out, uint32_t out_dwords, uint32_t *done, uint32_t ticket, uint32_t *counter) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	const uint32_t seed = in_dwords ? in[i % in_dwords] : arg.x;
	for (uint32_t k = i; k < out_dwords; k += gridDim.x * 256u) out[k] = seed + k;
	if (done) {
		__threadfence_system();
		__syncthreads();
		if (threadIdx.x == 0) {
			const uint32_t finished = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1u;
			if (finished == gridDim.x) {
				__hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				__hip_atomic_store(done, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
			}
		}
	}
}

// A RESIDENT workgroup instead of a launch per call: thread 0 polls a request word in pinned host memory (system-scope loads across the
// link), the workgroup does the same work as `work`, releases the request's number in *done and goes back to polling.  It leaves on
// request 0xFFFFFFFF, after `idle_ticks` of the 100 MHz clock without a request, after `max_ticks` in total, or after `max_polls`
// polls (three independent bounds: the measurement cannot hang the device).
__global__ __launch_bounds__(256) void resident(uint32_t *request, uint32_t *done, const uint32_t *in, uint32_t in_dwords, uint32_t *out, uint32_t out_dwords,
		uint64_t idle_ticks, uint64_t max_ticks, uint64_t max_polls) {
	__shared__ uint32_t s_seq;
	uint32_t last = 0;
	const uint64_t t_start = wall_clock64();
	uint64_t t_last = t_start, polls = 0;
	for (;;) {
		if (threadIdx.x == 0) {
			uint32_t r;
			for (;;) {
				r = __hip_atomic_load(request, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
				if (r != last) break;
				const uint64_t now = wall_clock64();
				if (now - t_last > idle_ticks || now - t_start > max_ticks || ++polls > max_polls) { r = 0xFFFFFFFFu; break; }
			}
			s_seq = r;
		}
		__syncthreads();
		const uint32_t r = s_seq;
		__syncthreads();
		if (r == 0xFFFFFFFFu) break;
		last = r;
		const uint32_t seed = in_dwords ? __hip_atomic_load(in + threadIdx.x % in_dwords, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : r;
		for (uint32_t k = threadIdx.x; k < out_dwords; k += 256u) out[k] = seed + k;
		__threadfence_system();
		__syncthreads();
		if (threadIdx.x == 0) {
			__hip_atomic_store(done, r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
			t_last = wall_clock64();
		}
	}
	if (threadIdx.x == 0) __hip_atomic_store(done + 16, 0xE0E0E0E0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);	// "gone"
}

// can the CPU write fine-grained DEVICE memory directly (large BAR)?  probed under a SIGSEGV handler
static sigjmp_buf g_probe_jmp;
static void probe_segv(int) { siglongjmp(g_probe_jmp, 1); }
static bool cpu_can_write(volatile uint32_t *p) {
	struct sigaction sa, old_segv, old_bus;
	memset(&sa, 0, sizeof sa); sa.sa_handler = probe_segv; sigemptyset(&sa.sa_mask);
	sigaction(SIGSEGV, &sa, &old_segv); sigaction(SIGBUS, &sa, &old_bus);
	bool ok = false;
	if (sigsetjmp(g_probe_jmp, 1) == 0) { p[0] = 0x12345678u; ok = p[0] == 0x12345678u; }
	sigaction(SIGSEGV, &old_segv, nullptr); sigaction(SIGBUS, &old_bus, nullptr);
	return ok;
}

struct Stat { double median, mean, p90; };
template <class F> Stat measure(F &&call, int n = 2000, int warm = 200) {
	for (int i = 0; i < warm; i++) call();
	std::vector<double> t(n);
	for (int i = 0; i < n; i++) { const double t0 = now_us(); call(); t[i] = now_us() - t0; }
	double sum = 0; for (double v : t) sum += v;
	std::sort(t.begin(), t.end());
	return { t[n / 2], sum / n, t[n * 9 / 10] };
}

int main(int argc, char **argv) {
	setvbuf(stdout, nullptr, _IOLBF, 0);
	const bool spin = argc > 1 && !strcmp(argv[1], "spin");
	const bool only_bar = argc > 1 && !strcmp(argv[1], "bar");
	if (spin) CK(hipSetDeviceFlags(hipDeviceScheduleSpin));
	CK(hipSetDevice(0));
	hipStream_t s;
	CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
	uint8_t *h = nullptr, *d = nullptr;
	CK(hipHostMalloc(&h, 4u << 20, hipHostMallocMapped));
	CK(hipHostGetDevicePointer((void **)&d, h, 0));
	uint32_t *counter; CK(hipMalloc(&counter, 4)); CK(hipMemset(counter, 0, 4));
	volatile uint32_t *h_done = reinterpret_cast<volatile uint32_t *>(h);
	uint32_t *d_done = reinterpret_cast<uint32_t *>(d);
	const uint32_t *d_in = reinterpret_cast<const uint32_t *>(d + 256);
	uint32_t *d_out = reinterpret_cast<uint32_t *>(d + (1u << 20));
	hipEvent_t ev, ev_blocking;
	CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
	CK(hipEventCreateWithFlags(&ev_blocking, hipEventDisableTiming | hipEventBlockingSync));
	printf("host_latency: scheduling %s\n", spin ? "hipDeviceScheduleSpin" : "default (auto)");
	struct Case { const char *name; uint32_t in_dwords, out_dwords, grid; } cases[] = {
		{ "one block (16 B argument -> 64 B)", 0, 16, 1 }, { "64x64 BC1 (2 KiB -> 16 KiB)", 512, 4096, 1 }, { "256x256 BC1 (32 KiB -> 256 KiB)", 8192, 65536, 16 } };
	uint32_t ticket = 0;
	for (const Case &c : cases) {
		if (only_bar) break;
		printf("-- %s\n", c.name);
		const uint4 arg = { 1, 2, 3, 4 };
		auto launch = [&](uint32_t *done, uint32_t t) { hipLaunchKernelGGL(work, dim3(c.grid), dim3(256), 0, s, d_in, c.in_dwords, arg, d_out, c.out_dwords, done, t, counter); };
		Stat st = measure([&] { launch(nullptr, 0); CK(hipStreamSynchronize(s)); });
		printf("   launch + hipStreamSynchronize                 median %6.2f  mean %6.2f  p90 %6.2f us\n", st.median, st.mean, st.p90);
		{	// split of the same
			double tl = 0, tw = 0; const int n = 1000;
			for (int i = 0; i < n; i++) { const double t0 = now_us(); launch(nullptr, 0); const double t1 = now_us(); CK(hipStreamSynchronize(s)); tl += t1 - t0; tw += now_us() - t1; }
			printf("      of which: launch call %.2f us, wait %.2f us (means)\n", tl / n, tw / n);
		}
		st = measure([&] { launch(nullptr, 0); while (hipStreamQuery(s) == hipErrorNotReady) {} });
		printf("   launch + hipStreamQuery spin                  median %6.2f  mean %6.2f  p90 %6.2f us\n", st.median, st.mean, st.p90);
		st = measure([&] { launch(nullptr, 0); CK(hipEventRecord(ev, s)); CK(hipEventSynchronize(ev)); });
		printf("   launch + event record + hipEventSynchronize   median %6.2f  mean %6.2f  p90 %6.2f us\n", st.median, st.mean, st.p90);
		st = measure([&] { launch(nullptr, 0); CK(hipEventRecord(ev, s)); while (hipEventQuery(ev) == hipErrorNotReady) {} });
		printf("   launch + event record + hipEventQuery spin    median %6.2f  mean %6.2f  p90 %6.2f us\n", st.median, st.mean, st.p90);
		st = measure([&] { ++ticket; launch(d_done, ticket); while (*h_done != ticket) { __builtin_ia32_pause(); } });
		printf("   launch + poll a word the kernel releases      median %6.2f  mean %6.2f  p90 %6.2f us\n", st.median, st.mean, st.p90);
		{	// split
			double tl = 0, tw = 0; const int n = 1000;
			for (int i = 0; i < n; i++) { ++ticket; const double t0 = now_us(); launch(d_done, ticket); const double t1 = now_us(); while (*h_done != ticket) { __builtin_ia32_pause(); } tl += t1 - t0; tw += now_us() - t1; }
			printf("      of which: launch call %.2f us, poll %.2f us (means)\n", tl / n, tw / n);
		}
		st = measure([&] { ++ticket; hipLaunchKernelGGL(work, dim3(c.grid), dim3(256), 0, 0, d_in, c.in_dwords, arg, d_out, c.out_dwords, d_done, ticket, counter); while (*h_done != ticket) { __builtin_ia32_pause(); } });
		printf("   the same on the NULL stream                   median %6.2f  mean %6.2f  p90 %6.2f us\n", st.median, st.mean, st.p90);
		CK(hipDeviceSynchronize());
		{	// a pre-instantiated graph of the one kernel; the ticket travels in pinned memory (no per-call parameter update)
			// kernel variant for the graph: ticket read from h[64]
			hipGraph_t graph; hipGraphExec_t exec;
			CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
			launch(nullptr, 0);
			CK(hipStreamEndCapture(s, &graph));
			CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
			st = measure([&] { CK(hipGraphLaunch(exec, s)); CK(hipStreamSynchronize(s)); });
			printf("   hipGraphLaunch + hipStreamSynchronize          median %6.2f  mean %6.2f  p90 %6.2f us\n", st.median, st.mean, st.p90);
			CK(hipGraphExecDestroy(exec)); CK(hipGraphDestroy(graph));
		}
		st = measure([&] { memcpy(h + 256, h + (2u << 20), c.in_dwords * 4u); ++ticket; launch(d_done, ticket); while (*h_done != ticket) { __builtin_ia32_pause(); } memcpy(h + (3u << 20), h + (1u << 20), c.out_dwords * 4u); });
		printf("   poll variant + the two host memcpys           median %6.2f  mean %6.2f  p90 %6.2f us\n", st.median, st.mean, st.p90);
	}
	{	// resident workgroup: request word at h + 128, done word at h + 0, "gone" marker at h + 64
		volatile uint32_t *h_req = reinterpret_cast<volatile uint32_t *>(h + 128);
		volatile uint32_t *h_gone = reinterpret_cast<volatile uint32_t *>(h + 64);
		volatile uint32_t *h_in = reinterpret_cast<volatile uint32_t *>(h + 256);
		volatile uint32_t *h_out = reinterpret_cast<volatile uint32_t *>(h + (1u << 20));
		for (const Case &c : cases) {
			if (c.grid != 1) continue;
			*h_req = 0; *h_done = 0; *h_gone = 0;
			__sync_synchronize();
			// bounds: 20 ms idle, 1 s in total, 2^28 polls
			hipLaunchKernelGGL(resident, dim3(1), dim3(256), 0, s, reinterpret_cast<uint32_t *>(d + 128), d_done, d_in, c.in_dwords, d_out, c.out_dwords, 2000000ull, 100000000ull, 1ull << 28);
			uint32_t seq = 0, wrong = 0;
			Stat st = measure([&] {
				++seq;
				if (c.in_dwords) h_in[0] = seq * 7u;
				__atomic_store_n(const_cast<uint32_t *>(h_req), seq, __ATOMIC_RELEASE);
				while (*h_done != seq) { __builtin_ia32_pause(); }
				const uint32_t want = c.in_dwords ? seq * 7u : seq;
				if (h_out[0] != want || h_out[c.out_dwords - 1] != (c.in_dwords ? h_in[255 % c.in_dwords] : seq) + c.out_dwords - 1) wrong++;
			});
			Stat st2 = measure([&] {
				++seq;
				memcpy(h + 256, h + (2u << 20), c.in_dwords * 4u);
				__atomic_store_n(const_cast<uint32_t *>(h_req), seq, __ATOMIC_RELEASE);
				while (*h_done != seq) { __builtin_ia32_pause(); }
				memcpy(h + (3u << 20), h + (1u << 20), c.out_dwords * 4u);
			}, 2000, 0);
			__atomic_store_n(const_cast<uint32_t *>(h_req), 0xFFFFFFFFu, __ATOMIC_RELEASE);
			const double t0 = now_us();
			CK(hipStreamSynchronize(s));
			printf("-- %s: RESIDENT workgroup, request + completion words in pinned memory\n", c.name);
			printf("   request -> completion seen by the host          median %6.2f  mean %6.2f  p90 %6.2f us   (wrong results: %u)\n", st.median, st.mean, st.p90, wrong);
			printf("   ... + the two host memcpys                      median %6.2f  mean %6.2f  p90 %6.2f us\n", st2.median, st2.mean, st2.p90);
			printf("   stop request -> stream idle %.1f us; gone marker %08x\n", now_us() - t0, *h_gone);
		}
	}
	{	// the same resident workgroup with the request word AND the input in fine-grained device memory that the CPU writes through the BAR
		int large_bar = 0;
		(void)hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, 0);
		uint8_t *v = nullptr;
		const hipError_t e = hipExtMallocWithFlags((void **)&v, 1u << 20, hipDeviceMallocFinegrained);
		printf("-- large BAR attribute %d, hipExtMallocWithFlags(fine-grained) %s\n", large_bar, hipGetErrorString(e));
		if (e == hipSuccess && cpu_can_write(reinterpret_cast<volatile uint32_t *>(v))) {
			volatile uint32_t *v_req = reinterpret_cast<volatile uint32_t *>(v + 128);
			volatile uint32_t *v_in = reinterpret_cast<volatile uint32_t *>(v + 256);
			volatile uint32_t *h_out = reinterpret_cast<volatile uint32_t *>(h + (1u << 20));
			for (const Case &c : cases) {
				if (c.grid != 1) continue;
				*v_req = 0; *h_done = 0;
				__sync_synchronize();
				hipLaunchKernelGGL(resident, dim3(1), dim3(256), 0, s, reinterpret_cast<uint32_t *>(v + 128), d_done, reinterpret_cast<const uint32_t *>(v + 256), c.in_dwords, d_out, c.out_dwords, 2000000ull, 100000000ull, 1ull << 28);
				uint32_t seq = 0, wrong = 0;
				Stat st = measure([&] {
					++seq;
					for (uint32_t k = 0; k < c.in_dwords; k++) v_in[k] = seq * 7u + k;		// the whole input through the BAR
					__builtin_ia32_sfence();			// (the BAR mapping is write-combining: order the input before the request word ...)
					__atomic_store_n(const_cast<uint32_t *>(v_req), seq, __ATOMIC_RELEASE);
					__builtin_ia32_sfence();			// (... and push the request word out of the write-combining buffer now)
					const double t_post = now_us();
					while (*h_done != seq) { __builtin_ia32_pause(); if (now_us() - t_post > 200000.0) { printf("   no answer to request %u within 0.2 s (done word %u): giving up\n", seq, *h_done); exit(2); } }
					const uint32_t want = c.in_dwords ? seq * 7u : seq;
					if (h_out[0] != want) wrong++;
				});
				__atomic_store_n(const_cast<uint32_t *>(v_req), 0xFFFFFFFFu, __ATOMIC_RELEASE);
				CK(hipStreamSynchronize(s));
				printf("-- %s: RESIDENT workgroup, request + input written by the CPU into device memory (BAR), output + completion in pinned memory\n", c.name);
				printf("   request -> completion seen by the host          median %6.2f  mean %6.2f  p90 %6.2f us   (wrong results: %u)\n", st.median, st.mean, st.p90, wrong);
			}
		} else {
			printf("   the CPU cannot write that memory: not measured\n");
		}
		if (v) (void)hipFree(v);
	}
	return 0;
}
