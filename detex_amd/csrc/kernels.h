// kernels.h -- the gfx950 launch skeletons shared by every block format.
//
// Mapping: ONE LANE = ONE 4x4 BLOCK, a wavefront covers 64 consecutive
// blocks of the row-major block stream, a 256-thread workgroup 256 of them.
//   load   lane i reads block i: 64 lanes x 8/16 B = one 512 B / 1 KiB fully coalesced
//          global_load_dwordx2/x4 per wave, no LDS needed because nothing is shared
//   decode 16 texels in registers (4*P dwords, P = bytes per pixel)
//   store  linear layout: texel row r of the 64 blocks is 64 x 4P contiguous bytes of image
//          row 4*by+r  ->  four (P=8: eight) wave-wide global_store_dwordx4, each a single
//          1 KiB contiguous run (full 128 B lines, no partial-line writes, no read-for-ownership)
//          tiled layout: each lane owns 16*P contiguous bytes
// Invalid blocks are zero-filled and raise *status (texture.c:125-128 semantics): one relaxed
// agent-scope load + (only while it still reads 0) one store per wave, never an atomic RMW.
#pragma once
#include "dev_common.h"

namespace detexhip {

template <int BYTES> struct BlockWord;
template <> struct BlockWord<8> { using type = uint2; };
template <> struct BlockWord<16> { using type = uint4; };

template <int P> struct RowWord;			// 4 pixels of P bytes
template <> struct RowWord<1> { using type = uint32_t; };
template <> struct RowWord<2> { using type = uint2; };
template <> struct RowWord<4> { using type = uint4; };

template <int P, bool NT> DH void store_row(uint8_t *dst, const uint32_t *d) {
	if constexpr (P == 1) {
		if (NT) __builtin_nontemporal_store(d[0], reinterpret_cast<uint32_t *>(dst));
		else *reinterpret_cast<uint32_t *>(dst) = d[0];
	} else if constexpr (P == 2) {
		typedef uint32_t v2 __attribute__((ext_vector_type(2)));
		v2 v = { d[0], d[1] };
		if (NT) __builtin_nontemporal_store(v, reinterpret_cast<v2 *>(dst));
		else *reinterpret_cast<v2 *>(dst) = v;
	} else {
		typedef uint32_t v4 __attribute__((ext_vector_type(4)));
#pragma unroll
		for (int k = 0; k < P / 4; k++) {
			v4 v = { d[4 * k], d[4 * k + 1], d[4 * k + 2], d[4 * k + 3] };
			if (NT) __builtin_nontemporal_store(v, reinterpret_cast<v4 *>(dst) + k);
			else reinterpret_cast<v4 *>(dst)[k] = v;
		}
	}
}

// raise the "some block was invalid" word without an atomic RMW storm (see header comment)
DH void raise_status(bool bad, uint32_t *status) {
	if (status == nullptr) return;
	if (__builtin_amdgcn_ballot_w64(bad) == 0) return;		// wave-uniform
	if (bad && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u)
		__hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- linear layout, fast path: width % 4 == 0, 16-byte aligned rows ---------------------------
template <class Dec, bool NT>
__global__ __launch_bounds__(256) void decode_linear(const void *__restrict__ blocks,
		uint8_t *__restrict__ pixels, uint32_t width_in_blocks, uint32_t n_blocks, uint64_t pitch,
		uint32_t *__restrict__ status) {
	constexpr int P = Dec::kPixelBytes;
	using Word = typename BlockWord<Dec::kBlockBytes>::type;
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= n_blocks) return;
	const Word blk = reinterpret_cast<const Word *>(blocks)[i];
	const uint32_t by = i / width_in_blocks, bx = i - by * width_in_blocks;
	uint32_t d[4 * P];
	const bool ok = Dec::template decode<false>(blk, 0xFFFFFFFFu, 0u, d);
	if (!ok) {
#pragma unroll
		for (int k = 0; k < 4 * P; k++) d[k] = 0u;
	}
	uint8_t *dst = pixels + (uint64_t)(by * 4u) * pitch + (uint64_t)bx * (4u * P);
#pragma unroll
	for (int r = 0; r < 4; r++) store_row<P, NT>(dst + (uint64_t)r * pitch, d + r * P);
	raise_status(!ok, status);
}

// ---- linear layout, clipped / unaligned path (texture.c:116-120,132-136) ----------------------
// Any width/height/pitch/pointer alignment down to the pixel size; texels outside the image
// are dropped.  Per-pixel stores: this path is for edge geometry, not for throughput.
template <int P> DH void store_pixel(uint8_t *dst, const uint32_t *row, int x) {
	if constexpr (P == 1) *dst = (uint8_t)(row[0] >> (8 * x));
	else if constexpr (P == 2) *reinterpret_cast<uint16_t *>(dst) = (uint16_t)(row[x >> 1] >> (16 * (x & 1)));
	else if constexpr (P == 4) *reinterpret_cast<uint32_t *>(dst) = row[x];
	else { reinterpret_cast<uint32_t *>(dst)[0] = row[2 * x]; reinterpret_cast<uint32_t *>(dst)[1] = row[2 * x + 1]; }
}

template <class Dec>
__global__ __launch_bounds__(256) void decode_linear_clipped(const void *__restrict__ blocks,
		uint8_t *__restrict__ pixels, uint32_t width_in_blocks, uint32_t n_blocks, uint32_t width,
		uint32_t height, uint64_t pitch, uint32_t *__restrict__ status) {
	constexpr int P = Dec::kPixelBytes;
	using Word = typename BlockWord<Dec::kBlockBytes>::type;
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= n_blocks) return;
	const Word blk = reinterpret_cast<const Word *>(blocks)[i];
	const uint32_t by = i / width_in_blocks, bx = i - by * width_in_blocks;
	uint32_t d[4 * P];
	const bool ok = Dec::template decode<false>(blk, 0xFFFFFFFFu, 0u, d);
	if (!ok) {
#pragma unroll
		for (int k = 0; k < 4 * P; k++) d[k] = 0u;
	}
#pragma unroll
	for (int r = 0; r < 4; r++) {
		const uint32_t y = by * 4u + r;
		if (y >= height) continue;
		uint8_t *dst = pixels + (uint64_t)y * pitch + (uint64_t)bx * (4u * P);
#pragma unroll
		for (int x = 0; x < 4; x++)
			if (bx * 4u + x < width) store_pixel<P>(dst + x * P, d + r * P, x);
	}
	raise_status(!ok, status);
}

// ---- block-major output (detexDecompressTextureTiled, texture.c:77-98) and the batched form of
// the per-block API (mode_mask / flags honoured, per-block ok byte) ----------------------------
template <class Dec, bool CHECKED>
__global__ __launch_bounds__(256) void decode_blocks(const void *__restrict__ blocks,
		uint8_t *__restrict__ pixels, uint32_t n_blocks, uint32_t mode_mask, uint32_t flags,
		uint8_t *__restrict__ ok_out, uint32_t *__restrict__ status) {
	constexpr int P = Dec::kPixelBytes;
	using Word = typename BlockWord<Dec::kBlockBytes>::type;
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= n_blocks) return;
	const Word blk = reinterpret_cast<const Word *>(blocks)[i];
	uint32_t d[4 * P];
	const bool ok = Dec::template decode<CHECKED>(blk, mode_mask, flags, d);
	if (!ok) {
#pragma unroll
		for (int k = 0; k < 4 * P; k++) d[k] = 0u;
	}
	uint32_t *dst = reinterpret_cast<uint32_t *>(pixels + (uint64_t)i * (16u * P));
	if constexpr (P >= 4) {
		typedef uint32_t v4 __attribute__((ext_vector_type(4)));
#pragma unroll
		for (int k = 0; k < P; k++) reinterpret_cast<v4 *>(dst)[k] = v4{ d[4 * k], d[4 * k + 1], d[4 * k + 2], d[4 * k + 3] };
	} else {
#pragma unroll
		for (int k = 0; k < 4 * P; k++) dst[k] = d[k];
	}
	if (ok_out) ok_out[i] = ok ? 1 : 0;
	raise_status(!ok, status);
}

}  // namespace detexhip
