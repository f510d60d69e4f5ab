// histogram.hip -- SURVEY.md 8f-4: block-mode histograms (the reference's detexGetMode<FMT> helpers, decompress-bc.c:63-69,
// decompress-etc.c:183-190,370-395,721-742, decompress-bptc.c:603-610, decompress-bptc-float.c:647-658) computed on the GPU
// over a whole block stream: the kernels, their launcher and the two device-tier entry points.
#include <hip/hip_runtime.h>

#include "host_internal.h"
#include "kernels.h"

namespace detexhip {

// ---- 8f-4: mode classification ------------------------------------------------------------------
// Bin numbers are the reference's detexGetMode<FMT> return values; bin 15 collects the reserved
// BPTC / BPTC_FLOAT codes (where the reference returns -1).  Formats without modes use bin 0.
// (classes: path_types.h kClass...)

DH uint32_t etc2_mode_of(uint32_t w0, bool has_individual) {	// decompress-etc.c:370-395
	const uint32_t b0 = w0 & 0xFFu, b1 = (w0 >> 8) & 0xFFu, b2 = (w0 >> 16) & 0xFFu, b3 = w0 >> 24;
	if (has_individual && !(b3 & 2u)) return 0u;
	const bool ovr = (uint32_t)((int32_t)(b0 >> 3) + sbfe(b0, 0, 3)) > 31u;
	const bool ovg = (uint32_t)((int32_t)(b1 >> 3) + sbfe(b1, 0, 3)) > 31u;
	const bool ovb = (uint32_t)((int32_t)(b2 >> 3) + sbfe(b2, 0, 3)) > 31u;
	return ovr ? 2u : (ovg ? 3u : (ovb ? 4u : 1u));
}

template <int CLASS> DH uint32_t block_mode(const uint32_t *w) {	// w = the block's 2 or 4 dwords
	if constexpr (CLASS == kClassS3TC) return (w[0] & 0xFFFFu) > (w[0] >> 16) ? 0u : 1u;
	else if constexpr (CLASS == kClassS3TCat8) return (w[2] & 0xFFFFu) > (w[2] >> 16) ? 0u : 1u;
	else if constexpr (CLASS == kClassETC1) return (w[0] >> 25) & 1u;
	else if constexpr (CLASS == kClassETC2) return etc2_mode_of(w[0], true);
	else if constexpr (CLASS == kClassETC2PT) return etc2_mode_of(w[0], false);
	else if constexpr (CLASS == kClassETC2at8) return etc2_mode_of(w[2], true);
	else if constexpr (CLASS == kClassBPTC) return (w[0] & 0xFFu) ? (uint32_t)__builtin_ctz(w[0] & 0xFFu) : 15u;
	else if constexpr (CLASS == kClassBPTCFloat) {
		const uint32_t low2 = w[0] & 3u, low5 = w[0] & 0x1Fu;
		const uint32_t m = low2 < 2u ? low2 : (low2 == 2u ? 2u + (low5 >> 2) : 10u + (low5 >> 2));
		return m > 13u ? 15u : m;
	} else return 0u;
}

// Persistent grid-stride kernel.  Every lane counts into its OWN column of a [16 modes][1024 lanes] LDS table with
// one ds_add_u32 per block (address = column + mode * 4 KiB: conflict-free, no return value, one VALU op) -- the
// round-1 kernel issued 16 ballots + popcounts per block and was SALU-bound.  Eight blocks per lane per trip, all
// loads issued before the first is classified: the kernel only reads, so its speed is the bytes it keeps in flight.
// Workgroups of 1024 lanes (four waves per SIMD): with 256-lane workgroups, one per CU, the kernel took 19.9 us for 4 Mi
// BC7 blocks where 64 MiB at the HBM read rate need 10.6 -- one wave per SIMD does not keep enough loads in flight.  The
// combine is device-scope atomics on the one 64-byte line of the 16 result words (~8.6 ns each, serialised), two adjacent
// bins per 64-bit atomic, so the grid stays at a few hundred workgroups (launch_mode_histogram below has the sweep: 192-256).
// (A ticketed "last workgroup sums per-workgroup slots" combine was measured too: 30+ us with agent-scope fences -- every
// workgroup writes back / invalidates its XCD's L2 -- and 15-21 us with completion-ordered relaxed atomics, not provably
// ordered.)
constexpr int kHistogramLanes = 1024;
// The combine adds two adjacent 32-bit bins with ONE 64-bit atomic (half as many serialised device-scope atomics: the
// accumulating entry 12.4 us per call against 14.4 with 32-bit adds, 4 Mi BC7 blocks).  No carry can cross while every bin
// stays below 2^32: always true for the zeroing entry (a call counts fewer than 2^32 blocks), and the documented limit of
// the accumulating one (include/detexhip.h) -- past it an even bin would carry into its odd neighbour instead of wrapping.
template <int CLASS, int BLOCK_DWORDS>
__global__ __launch_bounds__(kHistogramLanes) void mode_histogram(const uint32_t *__restrict__ blocks, uint32_t n_blocks,
		uint32_t *__restrict__ hist) {
	typedef typename BlockWord<4 * BLOCK_DWORDS>::type Word;
	constexpr int UNROLL = 8;
	constexpr uint32_t LANES = kHistogramLanes;
	__shared__ uint32_t bins[16][LANES];
	__shared__ uint32_t totals[16];
#pragma unroll
	for (int m = 0; m < 16; m++) bins[m][threadIdx.x] = 0u;		// own column: no barrier needed before the counting
	uint32_t *column = &bins[0][threadIdx.x];
	const uint32_t stride = gridDim.x * LANES;
	for (uint64_t i = blockIdx.x * LANES + threadIdx.x; i < n_blocks; i += (uint64_t)stride * UNROLL) {	// 64-bit: n_blocks may be close to 2^32
		Word v[UNROLL];
		bool live[UNROLL];
#pragma unroll
		for (int k = 0; k < UNROLL; k++) {
			const uint64_t j = i + (uint64_t)k * stride;
			live[k] = j < n_blocks;
			v[k] = live[k] ? reinterpret_cast<const Word *>(blocks)[j] : Word{};
		}
#pragma unroll
		for (int k = 0; k < UNROLL; k++) {
			uint32_t w[BLOCK_DWORDS];
			__builtin_memcpy(w, &v[k], sizeof w);
			if (live[k]) __hip_atomic_fetch_add(column + block_mode<CLASS>(w) * LANES, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		}
	}
	__syncthreads();
	// wave m (of 16) sums row m: 16 LDS reads per lane (consecutive lanes, consecutive words), then a wave reduction
	const uint32_t m = threadIdx.x >> 6, lane = threadIdx.x & 63u;
	uint32_t sum = 0;
#pragma unroll
	for (int l = 0; l < (int)(LANES / 64u); l++) sum += bins[m][64u * (uint32_t)l + lane];
#pragma unroll
	for (int step = 32; step >= 1; step >>= 1) sum += (uint32_t)__shfl_xor((int)sum, step, 64);
	if (lane == 0) totals[m] = sum;
	__syncthreads();
	if (threadIdx.x < 8u) {
		const uint32_t lo = totals[2u * threadIdx.x], hi = totals[2u * threadIdx.x + 1u];
		if ((reinterpret_cast<uintptr_t>(hist) & 7u) == 0) {
			if (lo | hi) atomicAdd(reinterpret_cast<unsigned long long *>(hist + 2u * threadIdx.x), (unsigned long long)lo | ((unsigned long long)hi << 32));
		} else {
			if (lo) atomicAdd(&hist[2u * threadIdx.x], lo);
			if (hi) atomicAdd(&hist[2u * threadIdx.x + 1u], hi);
		}
	}
}

template <int CLASS, int DWORDS> static hipError_t launch_histogram(const void *blocks, size_t n, uint32_t *hist, hipStream_t stream, bool zero_first) {
	hipError_t e = zero_first ? hipMemsetAsync(hist, 0, 16 * sizeof(uint32_t), stream) : hipSuccess;
	if (e != hipSuccess || n == 0) return e;
	// 1024-lane workgroups, eight loads in flight per lane; every further workgroup adds serialised global atomics at the end.
	// Measured, 4 Mi / 16 Mi blocks, us per call incl. the memset: BC7 grid 96: 15.3, 128: 14.0 / 40.5,
	// 192: 14.0, 256: 14.3 / 42.2, 384: 14.7; ETC2 128: 11.8 / 32.3, 192: 10.6, 256: 10.4 / 24.4, 384: 11.4
	// (round 2 began at 22.0 and 18.0 with 256 workgroups of 256 lanes).
	const unsigned max_grid = DWORDS == 2 ? 256u : 192u;
	const size_t tiles = (n + kHistogramLanes - 1) / kHistogramLanes;
	const unsigned grid = (unsigned)(tiles < max_grid ? tiles : max_grid);
	hipLaunchKernelGGL((mode_histogram<CLASS, DWORDS>), dim3(grid), dim3(kHistogramLanes), 0, stream, static_cast<const uint32_t *>(blocks), (uint32_t)n, hist);
	return hipGetLastError();
}

hipError_t launch_mode_histogram(int histogram_class, int block_dwords, const void *blocks, size_t n, uint32_t *hist, hipStream_t stream, bool zero_first) {
	switch (histogram_class) {		// (a class fixes the block size, except "no modes")
	case kClassS3TC: return launch_histogram<kClassS3TC, 2>(blocks, n, hist, stream, zero_first);
	case kClassS3TCat8: return launch_histogram<kClassS3TCat8, 4>(blocks, n, hist, stream, zero_first);
	case kClassETC1: return launch_histogram<kClassETC1, 2>(blocks, n, hist, stream, zero_first);
	case kClassETC2: return launch_histogram<kClassETC2, 2>(blocks, n, hist, stream, zero_first);
	case kClassETC2PT: return launch_histogram<kClassETC2PT, 2>(blocks, n, hist, stream, zero_first);
	case kClassETC2at8: return launch_histogram<kClassETC2at8, 4>(blocks, n, hist, stream, zero_first);
	case kClassBPTC: return launch_histogram<kClassBPTC, 4>(blocks, n, hist, stream, zero_first);
	case kClassBPTCFloat: return launch_histogram<kClassBPTCFloat, 4>(blocks, n, hist, stream, zero_first);
	default: return block_dwords == 2 ? launch_histogram<kClassNone, 2>(blocks, n, hist, stream, zero_first) : launch_histogram<kClassNone, 4>(blocks, n, hist, stream, zero_first);
	}
}

}  // namespace detexhip

using namespace detexhip;

static int mode_histogram_device(uint32_t texture_format, const void *d_blocks, size_t n_blocks, uint32_t *d_hist, void *stream, bool zero_first) {
	const FormatEntry *f = lookup_format(texture_format);
	if (!f) { detexSetErrorMessage("detexhipModeHistogramDevice: 0x%08X is not a block-compressed format of this library", texture_format); return 1; }
	if (n_blocks > 0xFFFFFF00ull || !d_hist || reinterpret_cast<uintptr_t>(d_blocks) % detexGetCompressedBlockSize(texture_format) != 0) {
		detexSetErrorMessage("detexhipModeHistogramDevice: bad arguments (d_hist NULL, d_blocks not block-aligned, or too many blocks)");
		return 1;
	}
	hipError_t e = launch_mode_histogram(f->histogram_class, (int)detexGetCompressedBlockSize(texture_format) / 4, d_blocks, n_blocks, d_hist, static_cast<hipStream_t>(stream), zero_first);
	if (e != hipSuccess) { detexSetErrorMessage("libdetexhip: kernel launch failed: %s", hipGetErrorString(e)); return 1; }
	return 0;
}
// ---- read-ahead of a block range into the memory-side cache (device_tier.cpp: linear_device_with) -----------------------------------------------
// A read-only pass over [p, p + bytes): every 16-byte vector loaded once with the default cache policy -- which allocates in the 256 MiB
// Infinity Cache -- and dropped.  Workgroup = 1024 lanes x 4 vectors = 64 KiB, every wave instruction one contiguous 1 KiB run, all four
// loads of a lane in flight together (tools/ubench/big_footprint.hip: 512 MiB in 85 us = 6.3 TB/s with four loads per lane, 99 us with one).
namespace detexhip {
__global__ __launch_bounds__(1024) void read_ahead(const u32x4 *__restrict__ p, uint64_t n_vectors) {
	const uint64_t base = (uint64_t)blockIdx.x * 4096u + threadIdx.x;
	const u32x4 *q[4];
#pragma unroll
	for (int k = 0; k < 4; k++) { const uint64_t i = base + 1024u * (uint32_t)k; q[k] = p + (i < n_vectors ? i : n_vectors - 1u); }
	// (the values are not wanted, only the lines' arrival in the cache: the loads are `asm volatile` so that the compiler keeps them --
	// written as plain loads into an unused sink the whole kernel was compiled to s_endpgm -- and waited for before the wave ends)
	u32x4 a, b, c, d;
	asm volatile("global_load_dwordx4 %0, %4, off\n\t"
		"global_load_dwordx4 %1, %5, off\n\t"
		"global_load_dwordx4 %2, %6, off\n\t"
		"global_load_dwordx4 %3, %7, off\n\t"
		"s_waitcnt vmcnt(0)"
		: "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(q[0]), "v"(q[1]), "v"(q[2]), "v"(q[3]) : "memory");
}
hipError_t launch_read_ahead(const void *p, size_t bytes, hipStream_t stream) {
	const uint64_t n_vectors = bytes / 16u;
	if (n_vectors == 0) return hipSuccess;
	hipLaunchKernelGGL(read_ahead, dim3((unsigned)((n_vectors + 4095u) / 4096u)), dim3(1024), 0, stream, static_cast<const u32x4 *>(p), n_vectors);
	return hipGetLastError();
}
}  // namespace detexhip

extern "C" int detexhipModeHistogramDevice(uint32_t texture_format, const void *d_blocks, size_t n_blocks, uint32_t *d_hist, void *stream) {
	return mode_histogram_device(texture_format, d_blocks, n_blocks, d_hist, stream, true);
}
extern "C" int detexhipModeHistogramAccumulateDevice(uint32_t texture_format, const void *d_blocks, size_t n_blocks, uint32_t *d_hist, void *stream) {
	return mode_histogram_device(texture_format, d_blocks, n_blocks, d_hist, stream, false);
}
