// tools/ubench/hbm_ref.hip -- reference points for the HBM roofline, measured on the box a bench run uses:
// a write-only fill with the store shape of the decode kernels (four non-temporal 16-byte stores per lane, each wave
// instruction one contiguous 1 KiB run) and a 16-byte-vector copy.  Built as tools/ubench/libhbmref.so; bench.py loads it
// (ctypes) to report roofline.ref_fill_GBps / ref_copy_GBps beside the decode kernel, tools/gpu_hbm_ref.py sweeps sizes and
// data patterns.  Measurement tooling: nothing here is linked into libdetexhip.so.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint32_t v4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
	x *= 0x9E3779B1u; x ^= x >> 15; x *= 0x85EBCA77u; x ^= x >> 13; x *= 0xC2B2AE3Du; x ^= x >> 16;
	return x;
}

// PATTERN 0 zeros . 1 one constant dword . 2 uniform random dwords . 3 random, but the upper half of every second dword 0 (the
// X component of FLOAT_RGBX16 pixels) . 4 random, constant within each 16-byte vector . 5 random bytes < 64 (few set bits)
// . 6 random with every second BYTE zero
template <int PATTERN> __device__ __forceinline__ v4 make_vector(uint32_t vec, uint32_t seed) {
	if (PATTERN == 0) return v4{ 0u, 0u, 0u, 0u };
	if (PATTERN == 1) return v4{ 0x12345678u, 0x12345678u, 0x12345678u, 0x12345678u };
	const uint32_t a = hash32(4u * vec + seed), b = hash32(4u * vec + 1u + seed), c = hash32(4u * vec + 2u + seed), d = hash32(4u * vec + 3u + seed);
	if (PATTERN == 2) return v4{ a, b, c, d };
	if (PATTERN == 3) return v4{ a, b & 0xFFFFu, c, d & 0xFFFFu };
	if (PATTERN == 4) return v4{ a, a, a, a };
	if (PATTERN == 5) return v4{ a & 0x3F3F3F3Fu, b & 0x3F3F3F3Fu, c & 0x3F3F3F3Fu, d & 0x3F3F3F3Fu };
	return v4{ a & 0x00FF00FFu, b & 0x00FF00FFu, c & 0x00FF00FFu, d & 0x00FF00FFu };
}

// one workgroup = 256 lanes x 4 vectors = 16 KiB, written as four wave-wide 1 KiB runs per wave (rows 4 KiB apart, like the
// texel rows of a 1024-pixel-wide tile)
template <int PATTERN, bool NT> __global__ __launch_bounds__(256) void fill_kernel(v4 *__restrict__ dst, uint64_t n_vectors, uint32_t seed) {
	const uint64_t base = (uint64_t)blockIdx.x * 1024u;
#pragma unroll
	for (int r = 0; r < 4; r++) {
		const uint64_t i = base + (uint64_t)r * 256u + threadIdx.x;
		if (i < n_vectors) {
			const v4 v = make_vector<PATTERN>((uint32_t)i, seed);
			if (NT) __builtin_nontemporal_store(v, dst + i);
			else dst[i] = v;
		}
	}
}

// the same fill through stores that are only dword-aligned (what an odd texture width does to every other pixel row)
typedef v4 v4_dword_aligned __attribute__((aligned(4)));
__global__ __launch_bounds__(256) void fill_unaligned_kernel(uint8_t *__restrict__ dst, uint64_t n_vectors, uint32_t seed) {
	const uint64_t base = (uint64_t)blockIdx.x * 1024u;
#pragma unroll
	for (int r = 0; r < 4; r++) {
		const uint64_t i = base + (uint64_t)r * 256u + threadIdx.x;
		if (i < n_vectors) __builtin_nontemporal_store(make_vector<2>((uint32_t)i, seed), reinterpret_cast<v4_dword_aligned *>(dst + 16u * i));
	}
}

// the fill laid out like a decoded image: workgroup b writes four rows of `pitch_vectors`-wide image rows, 256 vectors (4 KiB)
// of each -- the texel rows of one 256-block tile -- instead of 16 consecutive KiB
template <int PATTERN> __global__ __launch_bounds__(256) void fill_image_kernel(v4 *__restrict__ dst, uint32_t pitch_vectors, uint32_t tiles_per_row, uint32_t seed) {
	const uint32_t ty = blockIdx.x / tiles_per_row, tx = blockIdx.x - ty * tiles_per_row;
	v4 *p = dst + (uint64_t)(4u * ty) * pitch_vectors + (uint64_t)tx * 256u + threadIdx.x;
#pragma unroll
	for (int r = 0; r < 4; r++)
		__builtin_nontemporal_store(make_vector<PATTERN>(blockIdx.x * 1024u + r * 256u + threadIdx.x, seed), p + (uint64_t)r * pitch_vectors);
}

// image layout with ONE store per lane.  SHAPE 0: a workgroup covers 64 vectors (1 KiB) of four image rows, wave w writing row w (the
// decode kernels' tile with the texel rows dealt to the waves instead of to the lanes' registers); SHAPE 1: a workgroup writes 256
// vectors (4 KiB) of ONE image row, workgroups in row-major order; SHAPE 2: as 0 but 128 lanes x 2 tiles (two stores per lane,
// the wave's two runs 1 KiB apart in its row)
template <int SHAPE, bool NT> __global__ __launch_bounds__(256) void fill_image_lane_kernel(v4 *__restrict__ dst, uint32_t pitch_vectors, uint32_t tiles_per_row, uint32_t seed) {
	const uint32_t ty = blockIdx.x / tiles_per_row, tx = blockIdx.x - ty * tiles_per_row;
	if (SHAPE == 0) {
		v4 *p = dst + (uint64_t)(4u * ty + (threadIdx.x >> 6)) * pitch_vectors + (uint64_t)tx * 64u + (threadIdx.x & 63u);
		const v4 v = make_vector<2>(blockIdx.x * 256u + threadIdx.x, seed);
		if (NT) __builtin_nontemporal_store(v, p); else *p = v;
	} else if (SHAPE == 1) {
		v4 *p = dst + (uint64_t)ty * pitch_vectors + (uint64_t)tx * 256u + threadIdx.x;
		const v4 v = make_vector<2>(blockIdx.x * 256u + threadIdx.x, seed);
		if (NT) __builtin_nontemporal_store(v, p); else *p = v;
	} else {
		v4 *p = dst + (uint64_t)(4u * ty + (threadIdx.x >> 6)) * pitch_vectors + (uint64_t)tx * 128u + (threadIdx.x & 63u);
#pragma unroll
		for (int k = 0; k < 2; k++) {
			const v4 v = make_vector<2>(blockIdx.x * 512u + k * 256u + threadIdx.x, seed);
			if (NT) __builtin_nontemporal_store(v, p + 64 * k); else p[64 * k] = v;
		}
	}
}
extern "C" __attribute__((visibility("default"))) int hbmref_fill_image_lane(void *dst, size_t width_bytes, size_t height, size_t pitch_bytes, int shape, int nontemporal, uint32_t seed, void *stream) {
	if (pitch_bytes == 0) pitch_bytes = width_bytes;
	if (width_bytes == 0 || height == 0 || width_bytes % 4096u || height % 4u || pitch_bytes % 16u || pitch_bytes < width_bytes || (reinterpret_cast<uintptr_t>(dst) & 15u)) return 1;
	const uint32_t pitch_vectors = (uint32_t)(pitch_bytes / 16u);
	hipStream_t s = static_cast<hipStream_t>(stream);
	v4 *d = static_cast<v4 *>(dst);
	const uint32_t per_row = (uint32_t)(width_bytes / (shape == 0 ? 1024u : shape == 1 ? 4096u : 2048u));
	const dim3 grid((unsigned)(per_row * (shape == 1 ? height : height / 4u))), block(256);
#define LANE_CASE(S) case S: if (nontemporal) hipLaunchKernelGGL((fill_image_lane_kernel<S, true>), grid, block, 0, s, d, pitch_vectors, per_row, seed); \
	else hipLaunchKernelGGL((fill_image_lane_kernel<S, false>), grid, block, 0, s, d, pitch_vectors, per_row, seed); break;
	switch (shape) { LANE_CASE(0) LANE_CASE(1) LANE_CASE(2) default: return 1; }
#undef LANE_CASE
	return hipGetLastError() == hipSuccess ? 0 : 1;
}

// the decode kernels' own shape (every lane four stores, one per texel row) with LANES-wide workgroups; ROTATE: wave w starts with row w
// (the four waves of a workgroup write four different rows at any moment)
template <int LANES, bool ROTATE> __global__ __launch_bounds__(LANES) void fill_image_group_kernel(v4 *__restrict__ dst, uint32_t pitch_vectors, uint32_t tiles_per_row, uint32_t seed) {
	const uint32_t ty = blockIdx.x / tiles_per_row, tx = blockIdx.x - ty * tiles_per_row;
	v4 *p = dst + (uint64_t)(4u * ty) * pitch_vectors + (uint64_t)tx * LANES + threadIdx.x;
	const uint32_t w = ROTATE ? (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0u;
#pragma unroll
	for (int r = 0; r < 4; r++) {
		const uint32_t row = (r + w) & 3u;
		__builtin_nontemporal_store(make_vector<2>(blockIdx.x * (4u * LANES) + r * LANES + threadIdx.x, seed), p + (uint64_t)row * pitch_vectors);
	}
}
extern "C" __attribute__((visibility("default"))) int hbmref_fill_image_group(void *dst, size_t width_bytes, size_t height, int lanes, int rotate, uint32_t seed, void *stream) {
	if (width_bytes == 0 || height == 0 || width_bytes % 4096u || height % 4u || (reinterpret_cast<uintptr_t>(dst) & 15u)) return 1;
	const uint32_t pitch_vectors = (uint32_t)(width_bytes / 16u), per_row = (uint32_t)(width_bytes / (16u * lanes));
	hipStream_t s = static_cast<hipStream_t>(stream);
	v4 *d = static_cast<v4 *>(dst);
	const dim3 grid((unsigned)(per_row * (height / 4u)));
	if (lanes == 64) hipLaunchKernelGGL((fill_image_group_kernel<64, false>), grid, dim3(64), 0, s, d, pitch_vectors, per_row, seed);
	else if (lanes == 128) hipLaunchKernelGGL((fill_image_group_kernel<128, false>), grid, dim3(128), 0, s, d, pitch_vectors, per_row, seed);
	else if (lanes == 512) hipLaunchKernelGGL((fill_image_group_kernel<512, false>), grid, dim3(512), 0, s, d, pitch_vectors, per_row, seed);
	else if (lanes == 1024) hipLaunchKernelGGL((fill_image_group_kernel<1024, false>), grid, dim3(1024), 0, s, d, pitch_vectors, per_row, seed);
	else if (lanes == 256 && rotate) hipLaunchKernelGGL((fill_image_group_kernel<256, true>), grid, dim3(256), 0, s, d, pitch_vectors, per_row, seed);
	else if (lanes == 256) hipLaunchKernelGGL((fill_image_group_kernel<256, false>), grid, dim3(256), 0, s, d, pitch_vectors, per_row, seed);
	else return 1;
	return hipGetLastError() == hipSuccess ? 0 : 1;
}

// shape 0 of fill_image_lane_kernel (wave w = texel row w of a 64-vector tile, one store per lane per tile) on a PERSISTENT grid: workgroups loop
// over tiles b, b + gridDim.x, ...  SYNC 0: nothing between iterations; 1: a workgroup barrier; 2: the wave waits for its store's acknowledgement
// (s_waitcnt vmcnt(0)) before the next one.  And FOUR: the decode kernels' four-stores-per-lane tile with that wait between the stores (grid = tiles).
template <int SYNC> __global__ __launch_bounds__(256) void fill_image_lane_persistent_kernel(v4 *__restrict__ dst, uint32_t pitch_vectors, uint32_t tiles_per_row, uint32_t n_tiles, uint32_t seed) {
	for (uint32_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
		const uint32_t ty = t / tiles_per_row, tx = t - ty * tiles_per_row;
		v4 *p = dst + (uint64_t)(4u * ty + (threadIdx.x >> 6)) * pitch_vectors + (uint64_t)tx * 64u + (threadIdx.x & 63u);
		__builtin_nontemporal_store(make_vector<2>(t * 256u + threadIdx.x, seed), p);
		if (SYNC == 1) __syncthreads();
		if (SYNC == 2) __builtin_amdgcn_s_waitcnt(0x0F70);	// vmcnt(0) on gfx9 encodings: lgkmcnt / expcnt left at their maxima
	}
}
__global__ __launch_bounds__(256) void fill_image_four_waited_kernel(v4 *__restrict__ dst, uint32_t pitch_vectors, uint32_t tiles_per_row, uint32_t seed) {
	const uint32_t ty = blockIdx.x / tiles_per_row, tx = blockIdx.x - ty * tiles_per_row;
	v4 *p = dst + (uint64_t)(4u * ty) * pitch_vectors + (uint64_t)tx * 256u + threadIdx.x;
#pragma unroll
	for (int r = 0; r < 4; r++) {
		__builtin_nontemporal_store(make_vector<2>(blockIdx.x * 1024u + r * 256u + threadIdx.x, seed), p + (uint64_t)r * pitch_vectors);
		__builtin_amdgcn_s_waitcnt(0x0F70);
	}
}
extern "C" __attribute__((visibility("default"))) int hbmref_fill_image_persistent(void *dst, size_t width_bytes, size_t height, int sync, int groups, uint32_t seed, void *stream) {
	if (width_bytes == 0 || height == 0 || width_bytes % 4096u || height % 4u || (reinterpret_cast<uintptr_t>(dst) & 15u)) return 1;
	const uint32_t pitch_vectors = (uint32_t)(width_bytes / 16u);
	hipStream_t s = static_cast<hipStream_t>(stream);
	v4 *d = static_cast<v4 *>(dst);
	if (sync == 4) {
		const uint32_t per_row = (uint32_t)(width_bytes / 4096u);
		hipLaunchKernelGGL(fill_image_four_waited_kernel, dim3((unsigned)(per_row * (height / 4u))), dim3(256), 0, s, d, pitch_vectors, per_row, seed);
		return hipGetLastError() == hipSuccess ? 0 : 1;
	}
	const uint32_t per_row = (uint32_t)(width_bytes / 1024u), n_tiles = per_row * (uint32_t)(height / 4u);
	const dim3 grid((unsigned)groups), block(256);
	if (sync == 0) hipLaunchKernelGGL((fill_image_lane_persistent_kernel<0>), grid, block, 0, s, d, pitch_vectors, per_row, n_tiles, seed);
	else if (sync == 1) hipLaunchKernelGGL((fill_image_lane_persistent_kernel<1>), grid, block, 0, s, d, pitch_vectors, per_row, n_tiles, seed);
	else if (sync == 2) hipLaunchKernelGGL((fill_image_lane_persistent_kernel<2>), grid, block, 0, s, d, pitch_vectors, per_row, n_tiles, seed);
	else return 1;
	return hipGetLastError() == hipSuccess ? 0 : 1;
}

// the linear fill with ROWS vectors per lane (ROWS = 1: a workgroup writes 4 KiB and every wave issues ONE store, the shape of
// torch's fill kernel)
// WAVE_CONTIGUOUS: a wave's ROWS stores cover ROWS consecutive KiB (instead of one KiB in each of ROWS 4 KiB pieces)
template <int ROWS, bool NT, bool WAVE_CONTIGUOUS = false> __global__ __launch_bounds__(256) void fill_rows_kernel(v4 *__restrict__ dst, uint64_t n_vectors, uint32_t seed) {
	const uint64_t base = (uint64_t)blockIdx.x * (256u * ROWS);
#pragma unroll
	for (int r = 0; r < ROWS; r++) {
		const uint64_t i = WAVE_CONTIGUOUS ? base + ((threadIdx.x >> 6) * ROWS + r) * 64u + (threadIdx.x & 63u) : base + (uint64_t)r * 256u + threadIdx.x;
		if (i < n_vectors) {
			const v4 v = make_vector<2>((uint32_t)i, seed);
			if (NT) __builtin_nontemporal_store(v, dst + i);
			else dst[i] = v;
		}
	}
}

// one-wave workgroups: 64 lanes x ROWS stores (is it the stores per wave or the bytes per workgroup that decide the rate?)
template <int ROWS> __global__ __launch_bounds__(64) void fill_wave_kernel(v4 *__restrict__ dst, uint64_t n_vectors, uint32_t seed) {
	const uint64_t base = (uint64_t)blockIdx.x * (64u * ROWS);
#pragma unroll
	for (int r = 0; r < ROWS; r++) {
		const uint64_t i = base + (uint64_t)r * 64u + threadIdx.x;
		if (i < n_vectors) __builtin_nontemporal_store(make_vector<2>((uint32_t)i, seed), dst + i);
	}
}

template <bool NT> __global__ __launch_bounds__(256) void copy_kernel(v4 *__restrict__ dst, const v4 *__restrict__ src, uint64_t n_vectors) {
	const uint64_t base = (uint64_t)blockIdx.x * 1024u;
	v4 v[4];
#pragma unroll
	for (int r = 0; r < 4; r++) {
		const uint64_t i = base + (uint64_t)r * 256u + threadIdx.x;
		v[r] = NT ? __builtin_nontemporal_load(src + (i < n_vectors ? i : n_vectors - 1u)) : src[i < n_vectors ? i : n_vectors - 1u];
	}
#pragma unroll
	for (int r = 0; r < 4; r++) {
		const uint64_t i = base + (uint64_t)r * 256u + threadIdx.x;
		if (i < n_vectors) { if (NT) __builtin_nontemporal_store(v[r], dst + i); else dst[i] = v[r]; }
	}
}

extern "C" __attribute__((visibility("default"))) int hbmref_fill(void *dst, size_t bytes, int pattern, int nontemporal, uint32_t seed, void *stream) {
	const uint64_t n = bytes / 16u;
	if (n == 0 || (reinterpret_cast<uintptr_t>(dst) & 15u)) return 1;
	const dim3 grid((unsigned)((n + 1023u) / 1024u)), block(256);
	hipStream_t s = static_cast<hipStream_t>(stream);
	v4 *d = static_cast<v4 *>(dst);
#define FILL(P) case P: if (nontemporal) hipLaunchKernelGGL((fill_kernel<P, true>), grid, block, 0, s, d, n, seed); \
	else hipLaunchKernelGGL((fill_kernel<P, false>), grid, block, 0, s, d, n, seed); break;
	switch (pattern) { FILL(0) FILL(1) FILL(2) FILL(3) FILL(4) FILL(5) FILL(6) default: return 1; }
#undef FILL
	return hipGetLastError() == hipSuccess ? 0 : 1;
}

// the fill of fill_kernel (pattern 2) with an explicit cache policy on the store: POLICY bit 0 = sc0, bit 1 = sc1, bit 2 = nt
// (gfx940+: sc0 / sc1 give the coherence scope, nt the non-temporal hint; __builtin_nontemporal_store emits `nt` alone = 4)
template <int POLICY> __global__ __launch_bounds__(256) void fill_policy_kernel(v4 *__restrict__ dst, uint64_t n_vectors, uint32_t seed) {
	const uint64_t base = (uint64_t)blockIdx.x * 1024u;
#pragma unroll
	for (int r = 0; r < 4; r++) {
		const uint64_t i = base + (uint64_t)r * 256u + threadIdx.x;
		if (i < n_vectors) {
			const v4 v = make_vector<2>((uint32_t)i, seed);
			v4 *p = dst + i;
			if constexpr (POLICY == 0) asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(p), "v"(v) : "memory");
			else if constexpr (POLICY == 1) asm volatile("global_store_dwordx4 %0, %1, off sc0" :: "v"(p), "v"(v) : "memory");
			else if constexpr (POLICY == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
			else if constexpr (POLICY == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
			else if constexpr (POLICY == 4) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(p), "v"(v) : "memory");
			else if constexpr (POLICY == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0 nt" :: "v"(p), "v"(v) : "memory");
			else if constexpr (POLICY == 6) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" :: "v"(p), "v"(v) : "memory");
			else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" :: "v"(p), "v"(v) : "memory");
		}
	}
}
extern "C" __attribute__((visibility("default"))) int hbmref_fill_policy(void *dst, size_t bytes, int policy, uint32_t seed, void *stream) {
	const uint64_t n = bytes / 16u;
	if (n == 0 || (reinterpret_cast<uintptr_t>(dst) & 15u)) return 1;
	const dim3 grid((unsigned)((n + 1023u) / 1024u)), block(256);
	hipStream_t s = static_cast<hipStream_t>(stream);
	v4 *d = static_cast<v4 *>(dst);
#define POL(P) case P: hipLaunchKernelGGL((fill_policy_kernel<P>), grid, block, 0, s, d, n, seed); break;
	switch (policy) { POL(0) POL(1) POL(2) POL(3) POL(4) POL(5) POL(6) POL(7) default: return 1; }
#undef POL
	return hipGetLastError() == hipSuccess ? 0 : 1;
}

// the same 4 KiB per wave written in different SHAPES per store instruction (sc1 nt): 0 = 1 KiB runs (lane j, store k -> vector 64 k + j),
// 1 = 64-byte pieces 256 B apart (16 (j / 4) + 4 k + j % 4: what a transposition inside quads of lanes gives a block-major kernel),
// 2 = 16-byte pieces 64 B apart (4 j + k: a lane storing its own block), 3 = 128-byte lines 512 B apart (32 (j / 8) + 8 k + j % 8)
template <int SHAPE> __global__ __launch_bounds__(256) void fill_shape_kernel(v4 *__restrict__ dst, uint64_t n_vectors, uint32_t seed) {
	const uint32_t lane = threadIdx.x & 63u;
	const uint64_t base = ((uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6)) * 256u;
#pragma unroll
	for (uint32_t k = 0; k < 4; k++) {
		const uint32_t o = SHAPE == 0 ? 64u * k + lane : SHAPE == 1 ? 16u * (lane >> 2) + 4u * k + (lane & 3u) : SHAPE == 2 ? 4u * lane + k : 32u * (lane >> 3) + 8u * k + (lane & 7u);
		const uint64_t i = base + o;
		if (i < n_vectors) {
			const v4 v = make_vector<2>((uint32_t)i, seed);
			asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" :: "v"(dst + i), "v"(v) : "memory");
		}
	}
}
extern "C" __attribute__((visibility("default"))) int hbmref_fill_shape(void *dst, size_t bytes, int shape, uint32_t seed, void *stream) {
	const uint64_t n = bytes / 16u;
	if (n == 0 || (reinterpret_cast<uintptr_t>(dst) & 15u)) return 1;
	const dim3 grid((unsigned)((n + 1023u) / 1024u)), block(256);
	hipStream_t s = static_cast<hipStream_t>(stream);
	v4 *d = static_cast<v4 *>(dst);
	switch (shape) {
	case 0: hipLaunchKernelGGL((fill_shape_kernel<0>), grid, block, 0, s, d, n, seed); break;
	case 1: hipLaunchKernelGGL((fill_shape_kernel<1>), grid, block, 0, s, d, n, seed); break;
	case 2: hipLaunchKernelGGL((fill_shape_kernel<2>), grid, block, 0, s, d, n, seed); break;
	case 3: hipLaunchKernelGGL((fill_shape_kernel<3>), grid, block, 0, s, d, n, seed); break;
	default: return 1;
	}
	return hipGetLastError() == hipSuccess ? 0 : 1;
}

extern "C" __attribute__((visibility("default"))) int hbmref_copy(void *dst, const void *src, size_t bytes, int nontemporal, void *stream) {
	const uint64_t n = bytes / 16u;
	if (n == 0 || ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15u)) return 1;
	const dim3 grid((unsigned)((n + 1023u) / 1024u)), block(256);
	hipStream_t s = static_cast<hipStream_t>(stream);
	if (nontemporal) hipLaunchKernelGGL((copy_kernel<true>), grid, block, 0, s, static_cast<v4 *>(dst), static_cast<const v4 *>(src), n);
	else hipLaunchKernelGGL((copy_kernel<false>), grid, block, 0, s, static_cast<v4 *>(dst), static_cast<const v4 *>(src), n);
	return hipGetLastError() == hipSuccess ? 0 : 1;
}

// dst may be any dword-aligned address
extern "C" __attribute__((visibility("default"))) int hbmref_fill_unaligned(void *dst, size_t bytes, uint32_t seed, void *stream) {
	const uint64_t n = bytes / 16u;
	if (n == 0 || (reinterpret_cast<uintptr_t>(dst) & 3u)) return 1;
	hipLaunchKernelGGL(fill_unaligned_kernel, dim3((unsigned)((n + 1023u) / 1024u)), dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<uint8_t *>(dst), n, seed);
	return hipGetLastError() == hipSuccess ? 0 : 1;
}

// width_bytes x height image (width a multiple of 4096 bytes, height of 4 rows), rows pitch_bytes apart (0 = dense; a multiple
// of 16), pattern 0 or 2
// lds_bytes: unused dynamic LDS per workgroup -- caps the resident workgroups per CU (160 KiB per CU), the launch-time handle the
// decode kernels use as well
static unsigned g_fill_image_lds = 0;
extern "C" __attribute__((visibility("default"))) void hbmref_set_fill_image_lds(unsigned lds_bytes) { g_fill_image_lds = lds_bytes; }
extern "C" __attribute__((visibility("default"))) int hbmref_fill_image(void *dst, size_t width_bytes, size_t height, size_t pitch_bytes, int pattern, uint32_t seed, void *stream) {
	if (pitch_bytes == 0) pitch_bytes = width_bytes;
	if (width_bytes == 0 || height == 0 || width_bytes % 4096u || height % 4u || pitch_bytes % 16u || pitch_bytes < width_bytes || (reinterpret_cast<uintptr_t>(dst) & 15u)) return 1;
	const uint32_t tiles_per_row = (uint32_t)(width_bytes / 4096u), pitch_vectors = (uint32_t)(pitch_bytes / 16u);
	const dim3 grid((unsigned)(tiles_per_row * (height / 4u))), block(256);
	hipStream_t s = static_cast<hipStream_t>(stream);
	if (pattern == 0) hipLaunchKernelGGL((fill_image_kernel<0>), grid, block, g_fill_image_lds, s, static_cast<v4 *>(dst), pitch_vectors, tiles_per_row, seed);
	else hipLaunchKernelGGL((fill_image_kernel<2>), grid, block, g_fill_image_lds, s, static_cast<v4 *>(dst), pitch_vectors, tiles_per_row, seed);
	return hipGetLastError() == hipSuccess ? 0 : 1;
}

extern "C" __attribute__((visibility("default"))) int hbmref_fill_rows(void *dst, size_t bytes, int rows, int nontemporal, uint32_t seed, void *stream) {
	const uint64_t n = bytes / 16u;
	if (n == 0 || (reinterpret_cast<uintptr_t>(dst) & 15u)) return 1;
	hipStream_t s = static_cast<hipStream_t>(stream);
	v4 *d = static_cast<v4 *>(dst);
	const dim3 block(256);
#define ROWS_CASE(R) case R: { const dim3 grid((unsigned)((n + 256u * R - 1u) / (256u * R))); \
	if (nontemporal == 2) hipLaunchKernelGGL((fill_rows_kernel<R, true, true>), grid, block, 0, s, d, n, seed); \
	else if (nontemporal) hipLaunchKernelGGL((fill_rows_kernel<R, true>), grid, block, 0, s, d, n, seed); \
	else hipLaunchKernelGGL((fill_rows_kernel<R, false>), grid, block, 0, s, d, n, seed); break; }
	if (nontemporal == 3) {		// one-wave workgroups, non-temporal
		const dim3 grid((unsigned)((n + 64u * rows - 1u) / (64u * rows)));
		if (rows == 4) hipLaunchKernelGGL((fill_wave_kernel<4>), grid, dim3(64), 0, s, d, n, seed);
		else if (rows == 1) hipLaunchKernelGGL((fill_wave_kernel<1>), grid, dim3(64), 0, s, d, n, seed);
		else return 1;
		return hipGetLastError() == hipSuccess ? 0 : 1;
	}
	switch (rows) { ROWS_CASE(1) ROWS_CASE(2) ROWS_CASE(4) ROWS_CASE(8) default: return 1; }	// nontemporal = 2: non-temporal, wave-contiguous
#undef ROWS_CASE
	return hipGetLastError() == hipSuccess ? 0 : 1;
}
