// tools/ubench/host_midsize.hip -- measurement tool, not product: where the time of a mid-size host-pointer decode goes (VERDICT r04 item 3:
// 256^2 .. 4096^2 textures run at ~2x their PCIe floor).  A stand-in kernel with BC1's traffic (8 bytes read, 64 written per lane) replaces the
// decoder (its arithmetic is noise at these sizes); per size the runtime primitives are timed one by one and then whole call sequences:
//   A  the library's staged path as of round 4: memsetAsync(status) + H2D(pageable) + kernel + D2H(pageable) + D2H(status -> stack) + sync
//   B  lean staged: H2D + kernel + D2H + D2H(status -> pinned word) + sync                (no memset: the status word is kept zero between calls)
//   C  blocks through pinned memory, read by the kernel across the link: memcpy(in -> pinned) + kernel + D2H(pageable) + sync, status word
//      written by the kernel straight into pinned memory
//   D  the library's direct path: memcpy in, kernel reads and writes pinned host memory, completion word polled, memcpy out (one thread)
//   E  D in K bands (one launch and one completion word each), the caller and ONE helper thread copying finished bands out alternately
//   F  the caller's pixel buffer is itself pinned (hipHostRegister'ed once, outside the timing): memcpy in, kernel writes into it, poll -- no copy out
// Output: one JSON line per size (microseconds, median of `reps`).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
typedef uint32_t v4 __attribute__((ext_vector_type(4)));

// lane i: reads 8 bytes, writes 64 (four 16-byte stores); band of `n` lanes starting at `first`; the last workgroup publishes `ticket`
__global__ __launch_bounds__(256) void standin(const uint2 *in, v4 *out, uint32_t first, uint32_t n, uint32_t *status, uint32_t *done, uint32_t *counter, uint32_t ticket) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i < n) {
		const uint2 b = in[first + i];
		for (int k = 0; k < 4; k++) out[(size_t)(first + i) * 4 + k] = v4{ b.x + k, b.y, b.x ^ b.y, 0xFF000000u | k };
		if (b.x == 0xDEADBEEFu && status) *status = 1;
	}
	if (done) {
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
		__syncthreads();
		if (threadIdx.x == 0 && __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1u == gridDim.x) {
			__hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			__hip_atomic_store(done, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
		}
	}
}
__global__ void empty_kernel() {}
// the decode kernels' store shape: every store instruction of a wave covers one contiguous 1 KiB run (lane l writes 16 bytes at run + 16 l)
__global__ __launch_bounds__(256) void fill_runs(v4 *out, uint32_t n_vec, uint32_t *done, uint32_t *counter, uint32_t ticket) {
	const uint32_t wave = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
	for (int k = 0; k < 4; k++) {
		const uint32_t e = (wave * 4u + (uint32_t)k) * 64u + lane;
		if (e < n_vec) __builtin_nontemporal_store(v4{ e, 1u, 2u, 3u }, out + e);
	}
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
	__syncthreads();
	if (threadIdx.x == 0 && __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1u == gridDim.x) {
		__hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		__hip_atomic_store(done, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
	}
}

template <class F> static double median_us(int reps, F &&fn) {
	std::vector<double> t(reps);
	fn(); fn();
	for (int r = 0; r < reps; r++) { const double t0 = now_us(); fn(); t[r] = now_us() - t0; }
	std::sort(t.begin(), t.end());
	return t[reps / 2];
}
static void poll(volatile uint32_t *w, uint32_t ticket) { while (__atomic_load_n(w, __ATOMIC_ACQUIRE) != ticket) __builtin_ia32_pause(); }

int main(int argc, char **argv) {
	const int max_side = argc > 1 ? atoi(argv[1]) : 4096;
	hipStream_t s;
	CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
	const size_t OUT_MAX = (size_t)max_side * max_side * 4, IN_MAX = OUT_MAX / 8;
	uint8_t *d_in, *d_out; uint32_t *d_words;
	CK(hipMalloc(&d_in, IN_MAX)); CK(hipMalloc(&d_out, OUT_MAX)); CK(hipMalloc(&d_words, 256)); CK(hipMemset(d_words, 0, 256));
	uint8_t *h_in = (uint8_t *)aligned_alloc(4096, IN_MAX), *h_out = (uint8_t *)aligned_alloc(4096, OUT_MAX), *h_reg = (uint8_t *)aligned_alloc(4096, OUT_MAX);
	for (size_t k = 0; k < IN_MAX; k++) h_in[k] = (uint8_t)(k * 2654435761u >> 11);
	memset(h_out, 1, OUT_MAX); memset(h_reg, 2, OUT_MAX);
	uint8_t *p_in, *p_out; uint32_t *p_words;
	CK(hipHostMalloc((void **)&p_in, IN_MAX, hipHostMallocMapped)); CK(hipHostMalloc((void **)&p_out, OUT_MAX, hipHostMallocMapped));
	CK(hipHostMalloc((void **)&p_words, 4096, hipHostMallocMapped | hipHostMallocCoherent)); memset(p_words, 0, 4096);
	CK(hipHostRegister(h_reg, OUT_MAX, hipHostRegisterDefault));
	void *dv; uint8_t *pd_in, *pd_out, *rd_out; uint32_t *pd_words;
	CK(hipHostGetDevicePointer(&dv, p_in, 0)); pd_in = (uint8_t *)dv; CK(hipHostGetDevicePointer(&dv, p_out, 0)); pd_out = (uint8_t *)dv;
	CK(hipHostGetDevicePointer(&dv, p_words, 0)); pd_words = (uint32_t *)dv; CK(hipHostGetDevicePointer(&dv, h_reg, 0)); rd_out = (uint8_t *)dv;
	uint32_t ticket = 0;
	// helper thread of sequence E: spins on a job word while a call is in flight, parked on nothing fancier than a spin (measurement only)
	struct Job { std::atomic<uint32_t> go{ 0 }, done{ 0 }; const uint8_t *src; uint8_t *dst; size_t band_bytes; int bands; volatile uint32_t *words; uint32_t base_ticket; std::atomic<bool> quit{ false }; } job;
	std::thread helper([&] {
		uint32_t seen = 0;
		for (;;) {
			while (job.go.load(std::memory_order_acquire) == seen) { if (job.quit.load()) return; __builtin_ia32_pause(); }
			seen = job.go.load(std::memory_order_acquire);
			for (int b = 1; b < job.bands; b += 2) { poll(job.words + 16 * b, job.base_ticket + (uint32_t)b); memcpy(job.dst + (size_t)b * job.band_bytes, job.src + (size_t)b * job.band_bytes, job.band_bytes); }
			job.done.store(seen, std::memory_order_release);
		}
	});
	const int reps = 60;
	{
		const double launch_sync = median_us(200, [&] { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s); CK(hipStreamSynchronize(s)); });
		const double memset_sync = median_us(200, [&] { CK(hipMemsetAsync(d_words, 0, 4, s)); CK(hipStreamSynchronize(s)); });
		uint32_t st = 0;
		const double d2h4_stack = median_us(200, [&] { CK(hipMemcpyAsync(&st, d_words, 4, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); });
		const double d2h4_pinned = median_us(200, [&] { CK(hipMemcpyAsync(p_words + 64, d_words, 4, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); });
		printf("{\"primitives_us\": {\"empty_launch_sync\": %.2f, \"memsetAsync4_sync\": %.2f, \"d2h_4B_to_stack_sync\": %.2f, \"d2h_4B_to_pinned_sync\": %.2f}}\n", launch_sync, memset_sync, d2h4_stack, d2h4_pinned);
	}
	{	// the runtime's copies out of / into PAGEABLE memory against size: where does it switch from staging to pinning the caller's pages?
		printf("{\"pageable_copy_curve_us\": {");
		bool first = true;
		for (size_t kib : { 64, 128, 256, 384, 512, 768, 1024, 1536, 2048, 3072, 4096, 8192 }) {
			const size_t n = kib << 10;
			if (n > OUT_MAX || n > IN_MAX * 8) break;
			const double d = median_us(30, [&] { CK(hipMemcpyAsync(h_out, d_out, n, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); });
			const double u = n <= IN_MAX ? median_us(30, [&] { CK(hipMemcpyAsync(d_in, h_in, n, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); }) : -1.0;
			const double dp = median_us(30, [&] { CK(hipMemcpyAsync(p_out, d_out, n, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); });
			printf("%s\"%zuKiB\": {\"d2h\": %.1f, \"h2d\": %.1f, \"d2h_pinned\": %.1f}", first ? "" : ", ", kib, d, u, dp);
			first = false;
		}
		printf("}}\n");
	}
	for (int side = 256; side <= max_side; side *= 2) {
		const size_t out_bytes = (size_t)side * side * 4, in_bytes = out_bytes / 8;
		const uint32_t n = (uint32_t)(in_bytes / 8), grid = (n + 255u) / 256u;
		uint32_t st = 0;
		auto kern = [&](const uint8_t *in, uint8_t *out, uint32_t first, uint32_t count, uint32_t *status, uint32_t *done, uint32_t tk) {
			hipLaunchKernelGGL(standin, dim3((count + 255u) / 256u), dim3(256), 0, s, (const uint2 *)in, (v4 *)out, first, count, status, done, d_words + 8, tk);
		};
		const double h2d = median_us(reps, [&] { CK(hipMemcpyAsync(d_in, h_in, in_bytes, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); });
		const double d2h = median_us(reps, [&] { CK(hipMemcpyAsync(h_out, d_out, out_bytes, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); });
		const double d2h_pin = median_us(reps, [&] { CK(hipMemcpyAsync(p_out, d_out, out_bytes, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); });
		const double cp_in = median_us(reps, [&] { memcpy(p_in, h_in, in_bytes); });
		const double cp_out = median_us(reps, [&] { memcpy(h_out, p_out, out_bytes); });
		const double k_dev = median_us(reps, [&] { kern(d_in, d_out, 0, n, d_words, nullptr, 0); CK(hipStreamSynchronize(s)); });
		const double k_host = median_us(reps, [&] { const uint32_t tk = ++ticket; kern(pd_in, pd_out, 0, n, pd_words, pd_words + 16, tk); poll(p_words + 16, tk); });
		const double A = median_us(reps, [&] {
			CK(hipMemsetAsync(d_words, 0, 4, s)); CK(hipMemcpyAsync(d_in, h_in, in_bytes, hipMemcpyHostToDevice, s)); kern(d_in, d_out, 0, n, d_words, nullptr, 0);
			CK(hipMemcpyAsync(h_out, d_out, out_bytes, hipMemcpyDeviceToHost, s)); CK(hipMemcpyAsync(&st, d_words, 4, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); });
		const double B = median_us(reps, [&] {
			CK(hipMemcpyAsync(d_in, h_in, in_bytes, hipMemcpyHostToDevice, s)); kern(d_in, d_out, 0, n, d_words, nullptr, 0);
			CK(hipMemcpyAsync(h_out, d_out, out_bytes, hipMemcpyDeviceToHost, s)); CK(hipMemcpyAsync(p_words + 64, d_words, 4, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); });
		const double C = median_us(reps, [&] {
			memcpy(p_in, h_in, in_bytes); p_words[0] = 0; kern(pd_in, d_out, 0, n, pd_words, nullptr, 0);
			CK(hipMemcpyAsync(h_out, d_out, out_bytes, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); });
		const double D = median_us(reps, [&] {
			memcpy(p_in, h_in, in_bytes); p_words[0] = 0; const uint32_t tk = ++ticket; kern(pd_in, pd_out, 0, n, pd_words, pd_words + 16, tk); poll(p_words + 16, tk);
			memcpy(h_out, p_out, out_bytes); });
		double E[3] = { 0, 0, 0 };
		int e_idx = 0;
		for (int bands : { 2, 4, 8 }) {
			if ((n / bands) % 256u) { e_idx++; continue; }
			E[e_idx++] = median_us(reps, [&] {
				memcpy(p_in, h_in, in_bytes); p_words[0] = 0;
				const uint32_t base = ticket + 1; ticket += (uint32_t)bands;
				job.src = p_out; job.dst = h_out; job.band_bytes = out_bytes / bands; job.bands = bands; job.words = p_words + 16; job.base_ticket = base;
				const uint32_t g = job.go.load() + 1; job.go.store(g, std::memory_order_release);
				for (int b = 0; b < bands; b++)
					hipLaunchKernelGGL(standin, dim3(grid / bands), dim3(256), 0, s, (const uint2 *)pd_in, (v4 *)pd_out, (uint32_t)b * (n / bands), n / bands, pd_words, pd_words + 16 + 16 * b, d_words + 8 + b, base + (uint32_t)b);
				for (int b = 0; b < bands; b += 2) { poll(p_words + 16 + 16 * b, base + (uint32_t)b); memcpy(h_out + (size_t)b * job.band_bytes, p_out + (size_t)b * job.band_bytes, job.band_bytes); }
				while (job.done.load(std::memory_order_acquire) != g) __builtin_ia32_pause(); });
		}
		const double F = median_us(reps, [&] {
			memcpy(p_in, h_in, in_bytes); p_words[0] = 0; const uint32_t tk = ++ticket; kern(pd_in, rd_out, 0, n, pd_words, pd_words + 16, tk); poll(p_words + 16, tk); });
		// shader stores into host memory in the decode kernels' store shape (1 KiB runs), and what registering the caller's buffer would cost per call
		const uint32_t n_vec = (uint32_t)(out_bytes / 16);
		const double runs_pinned = median_us(reps, [&] { const uint32_t tk = ++ticket; hipLaunchKernelGGL(fill_runs, dim3((n_vec + 1023u) / 1024u), dim3(256), 0, s, (v4 *)pd_out, n_vec, pd_words + 16, d_words + 8, tk); poll(p_words + 16, tk); });
		const double runs_device = median_us(reps, [&] { hipLaunchKernelGGL(fill_runs, dim3((n_vec + 1023u) / 1024u), dim3(256), 0, s, (v4 *)d_out, n_vec, d_words + 12, d_words + 8, 1u); CK(hipStreamSynchronize(s)); });
		// G: the caller's pageable buffer registered FOR THE CALL (four buffers in rotation, so that no registration is reused), run-shaped shader
		// stores straight into it, completion polled, unregistered; G2: the same with the registration kept (a caller that reuses its buffer)
		uint8_t *user[4];
		for (int k = 0; k < 4; k++) { user[k] = (uint8_t *)malloc(out_bytes + 4096) + 64 * (k + 1); memset(user[k], 7, out_bytes); }	// (deliberately not page-aligned)
		int turn = 0;
		bool g_ok = true;
		const double G = median_us(reps, [&] {
			uint8_t *u = user[turn++ & 3];
			memcpy(p_in, h_in, in_bytes);
			CK(hipHostRegister(u, out_bytes, hipHostRegisterDefault));
			void *du = nullptr; CK(hipHostGetDevicePointer(&du, u, 0));
			const uint32_t tk = ++ticket;
			hipLaunchKernelGGL(fill_runs, dim3((n_vec + 1023u) / 1024u), dim3(256), 0, s, (v4 *)du, n_vec, pd_words + 16, d_words + 8, tk); poll(p_words + 16, tk);
			CK(hipHostUnregister(u));
			g_ok = g_ok && ((uint32_t *)u)[4] == 1u && ((uint32_t *)(u + out_bytes - 16))[0] == n_vec - 1u; });
		CK(hipHostRegister(user[0], out_bytes, hipHostRegisterDefault));
		void *du0 = nullptr; CK(hipHostGetDevicePointer(&du0, user[0], 0));
		const double G2 = median_us(reps, [&] {
			memcpy(p_in, h_in, in_bytes); const uint32_t tk = ++ticket;
			hipLaunchKernelGGL(fill_runs, dim3((n_vec + 1023u) / 1024u), dim3(256), 0, s, (v4 *)du0, n_vec, pd_words + 16, d_words + 8, tk); poll(p_words + 16, tk); });
		CK(hipHostUnregister(user[0]));
		printf("{\"side\": %d, \"G_register_per_call_run_stores_into_caller_buffer\": %.1f, \"G2_registration_kept\": %.1f, \"G_results_ok\": %s}\n", side, G, G2, g_ok ? "true" : "false");
		uint8_t *h_tmp = (uint8_t *)aligned_alloc(4096, out_bytes); memset(h_tmp, 3, out_bytes);
		const double reg_us = median_us(15, [&] { CK(hipHostRegister(h_tmp, out_bytes, hipHostRegisterDefault)); CK(hipHostUnregister(h_tmp)); });
		free(h_tmp);
		printf("{\"side\": %d, \"shader_stores_1KiB_runs_into_pinned_poll\": %.1f, \"same_into_device_sync\": %.1f, \"hipHostRegister_plus_Unregister\": %.1f}\n", side, runs_pinned, runs_device, reg_us);
		const bool same = memcmp(h_out, h_reg, out_bytes) == 0;
		printf("{\"side\": %d, \"in_KiB\": %zu, \"out_KiB\": %zu, \"h2d_pageable\": %.1f, \"d2h_pageable\": %.1f, \"d2h_pinned\": %.1f, \"memcpy_in_to_pinned\": %.1f, \"memcpy_out_from_pinned\": %.1f, "
			"\"kernel_device_sync\": %.1f, \"kernel_host_both_ways_poll\": %.1f, \"A_staged_r04\": %.1f, \"B_staged_lean\": %.1f, \"C_pinned_in_d2h_out\": %.1f, \"D_direct\": %.1f, "
			"\"E_direct_bands2_helper\": %.1f, \"E_bands4\": %.1f, \"E_bands8\": %.1f, \"F_pinned_user_buffer\": %.1f, \"results_equal\": %s}\n",
			side, in_bytes >> 10, out_bytes >> 10, h2d, d2h, d2h_pin, cp_in, cp_out, k_dev, k_host, A, B, C, D, E[0], E[1], E[2], F, same ? "true" : "false");
		fflush(stdout);
	}
	job.quit.store(true); helper.join();
	return 0;
}
