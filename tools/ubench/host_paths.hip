// tools/ubench/host_paths.hip -- what the host-pointer tier can reach on this box (measurement tool, not product):
// D2H / H2D rates for pageable, registered (hipHostRegister) and pinned memory, the cost of registering, the rate of
// a kernel storing straight into mapped host memory, and chunked overlap of H2D with D2H on two streams.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void fill_nt(uint4 *dst, size_t n) {
	typedef uint32_t v4 __attribute__((ext_vector_type(4)));
	for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
		v4 v = { (uint32_t)i, 1u, 2u, 3u };
		__builtin_nontemporal_store(v, reinterpret_cast<v4 *>(dst) + i);
	}
}

int main() {
	const size_t OUT = 256u << 20, IN = 32u << 20;
	void *d_out, *d_in;
	CK(hipMalloc(&d_out, OUT)); CK(hipMalloc(&d_in, IN));
	CK(hipMemset(d_out, 1, OUT));
	char *pageable_out = (char *)aligned_alloc(4096, OUT), *pageable_in = (char *)aligned_alloc(4096, IN);
	memset(pageable_out, 0, OUT); memset(pageable_in, 3, IN);
	hipStream_t s0, s1;
	CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
	for (int rep = 0; rep < 3; rep++) {
		double t = now(); CK(hipMemcpy(pageable_out, d_out, OUT, hipMemcpyDeviceToHost)); double a = now() - t;
		t = now(); CK(hipMemcpy(d_in, pageable_in, IN, hipMemcpyHostToDevice)); double b = now() - t;
		printf("pageable: D2H 256 MiB %.3f ms (%.1f GB/s)   H2D 32 MiB %.3f ms (%.1f GB/s)\n", a * 1e3, OUT / a / 1e9, b * 1e3, IN / b / 1e9);
	}
	{	// is hipMemcpyAsync from/to pageable memory asynchronous to the host?
		double t = now(); CK(hipMemcpyAsync(pageable_out, d_out, OUT, hipMemcpyDeviceToHost, s0)); double issued = now() - t;
		CK(hipStreamSynchronize(s0)); double done = now() - t;
		printf("pageable async D2H: call returns after %.3f ms, complete after %.3f ms\n", issued * 1e3, done * 1e3);
	}
	for (int rep = 0; rep < 2; rep++) {	// pageable, 8 chunks of 32 MiB back to back on one stream
		double t = now();
		for (int k = 0; k < 8; k++) CK(hipMemcpyAsync(pageable_out + (size_t)k * (OUT / 8), (char *)d_out + (size_t)k * (OUT / 8), OUT / 8, hipMemcpyDeviceToHost, s0));
		CK(hipStreamSynchronize(s0));
		double a = now() - t;
		printf("pageable D2H in 8 chunks of 32 MiB: %.3f ms (%.1f GB/s)\n", a * 1e3, OUT / a / 1e9);
	}
	for (int rep = 0; rep < 2; rep++) {	// two host threads: pageable H2D (32 MiB) while pageable D2H (256 MiB)
		double t = now();
		std::thread up([&] { CK(hipSetDevice(0)); CK(hipMemcpyAsync(d_in, pageable_in, IN, hipMemcpyHostToDevice, s1)); CK(hipStreamSynchronize(s1)); });
		CK(hipMemcpyAsync(pageable_out, d_out, OUT, hipMemcpyDeviceToHost, s0)); CK(hipStreamSynchronize(s0));
		double a = now() - t;
		up.join();
		double b = now() - t;
		printf("pageable D2H 256 MiB with a second thread uploading 32 MiB: D2H done %.3f ms, both done %.3f ms\n", a * 1e3, b * 1e3);
	}
	for (int rep = 0; rep < 3; rep++) {
		double t = now(); CK(hipHostRegister(pageable_out, OUT, hipHostRegisterDefault)); double reg = now() - t;
		t = now(); CK(hipMemcpyAsync(pageable_out, d_out, OUT, hipMemcpyDeviceToHost, s0)); CK(hipStreamSynchronize(s0)); double cp = now() - t;
		void *dev_view = nullptr; CK(hipHostGetDevicePointer(&dev_view, pageable_out, 0));
		t = now(); hipLaunchKernelGGL(fill_nt, dim3(2048), dim3(256), 0, s0, (uint4 *)dev_view, OUT / 16); CK(hipStreamSynchronize(s0)); double zc = now() - t;
		t = now(); CK(hipHostUnregister(pageable_out)); double unreg = now() - t;
		printf("registered: register %.3f ms, D2H %.3f ms (%.1f GB/s), kernel stores to host %.3f ms (%.1f GB/s), unregister %.3f ms\n", reg * 1e3, cp * 1e3,
			OUT / cp / 1e9, zc * 1e3, OUT / zc / 1e9, unreg * 1e3);
	}
	{
		char *pin_out, *pin_in;
		CK(hipHostMalloc((void **)&pin_out, OUT, hipHostMallocDefault)); CK(hipHostMalloc((void **)&pin_in, IN, hipHostMallocDefault));
		memset(pin_in, 5, IN);
		for (int rep = 0; rep < 3; rep++) {
			double t = now(); CK(hipMemcpyAsync(pin_out, d_out, OUT, hipMemcpyDeviceToHost, s0)); CK(hipStreamSynchronize(s0)); double a = now() - t;
			t = now(); CK(hipMemcpyAsync(d_in, pin_in, IN, hipMemcpyHostToDevice, s0)); CK(hipStreamSynchronize(s0)); double b = now() - t;
			t = now();
			CK(hipMemcpyAsync(pin_out, d_out, OUT, hipMemcpyDeviceToHost, s0)); CK(hipMemcpyAsync(d_in, pin_in, IN, hipMemcpyHostToDevice, s1));
			CK(hipStreamSynchronize(s0)); CK(hipStreamSynchronize(s1)); double both = now() - t;
			printf("pinned: D2H %.3f ms (%.1f GB/s)  H2D %.3f ms (%.1f GB/s)  both directions at once %.3f ms\n", a * 1e3, OUT / a / 1e9, b * 1e3, IN / b / 1e9, both * 1e3);
		}
		// pinned bounce -> pageable with N host threads copying
		for (int threads : { 1, 4, 8, 16 }) {
			double t = now();
			std::vector<std::thread> th;
			for (int k = 0; k < threads; k++) th.emplace_back([&, k] { const size_t c = OUT / threads; memcpy(pageable_out + k * c, pin_out + k * c, c); });
			for (auto &x : th) x.join();
			double a = now() - t;
			printf("host memcpy pinned -> pageable 256 MiB, %d threads: %.3f ms (%.1f GB/s)\n", threads, a * 1e3, OUT / a / 1e9);
		}
		// chunked: D2H into a pinned ring on a stream, host threads drain finished chunks into the pageable buffer
		for (int threads : { 4, 8 }) {
			const size_t CH = 16u << 20; const int n = (int)(OUT / CH);
			std::vector<hipEvent_t> ev(n);
			for (auto &evk : ev) CK(hipEventCreateWithFlags(&evk, hipEventDisableTiming));
			double t = now();
			for (int k = 0; k < n; k++) { CK(hipMemcpyAsync(pin_out + k * CH, (char *)d_out + k * CH, CH, hipMemcpyDeviceToHost, s0)); CK(hipEventRecord(ev[k], s0)); }
			for (int k = 0; k < n; k++) {
				CK(hipEventSynchronize(ev[k]));
				std::vector<std::thread> th;
				for (int j = 0; j < threads; j++) th.emplace_back([&, j, k] { const size_t c = CH / threads; memcpy(pageable_out + k * CH + j * c, pin_out + k * CH + j * c, c); });
				for (auto &x : th) x.join();
			}
			double a = now() - t;
			printf("chunked D2H (16 MiB) into pinned + %d-thread drain to pageable: %.3f ms (%.1f GB/s)\n", threads, a * 1e3, OUT / a / 1e9);
		}
	}
	return 0;
}
