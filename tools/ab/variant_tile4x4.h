// variant_tile4x4.h -- the north_star's literal tile shape, kept as a measured A/B variant.
//
// BASELINE.json's north_star sketches "one wavefront owns a tile of 4x4 blocks, the compressed
// blocks are staged into LDS with coalesced HBM loads, the 16 output texels are scattered with
// wave-wide coalesced stores".  This file implements exactly that for BC1 so it can be timed
// against the lane-per-block mapping of kernels.h (bench.py --variant 1; results in profiles/AB_RECORD.md).  Layout: workgroup = 4 waves = 4 horizontally adjacent tiles = 16x4 blocks;
// 64 threads stage the 64 blocks (4 rows x 128 contiguous bytes) into 512 B of LDS; then lane l
// of wave w owns texel row (l >> 2) & 3 of block (4w + (l & 3), l >> 4): it re-reads its 8-byte
// block from LDS (4 lanes broadcast-read the same address), rebuilds the palette and writes ONE
// global_store_dwordx4.  Per store instruction the wave writes 16 image rows x 64 contiguous
// bytes (half a 128 B line each), versus 1 KiB contiguous per store in the default mapping.
#pragma once
#include "dev_common.h"
#include "decode_s3tc_rgtc.h"
#include "ab_traits.h"

namespace detexhip {

__global__ __launch_bounds__(256) void decode_linear_tile4x4_bc1(const uint2 *__restrict__ blocks,
		uint8_t *__restrict__ pixels, uint32_t width_in_blocks, uint64_t pitch) {
	__shared__ uint2 tile[64];				// [block row 0..3][block col 0..15]
	const uint32_t t = threadIdx.x;
	const uint32_t bx0 = blockIdx.x * 16u, by0 = blockIdx.y * 4u;
	if (t < 64u) tile[t] = blocks[(uint64_t)(by0 + (t >> 4)) * width_in_blocks + bx0 + (t & 15u)];
	__syncthreads();
	const uint32_t wave = t >> 6, lane = t & 63u;
	const uint32_t bcol = wave * 4u + (lane & 3u), brow = lane >> 4, trow = (lane >> 2) & 3u;
	const uint2 blk = tile[brow * 16u + bcol];
	uint32_t p[4];
	s3tc_palette(blk.x, (blk.x & 0xFFFFu) > (blk.x >> 16), 0xFF000000u, p);
	const uint32_t idx = (blk.y >> (8u * trow)) & 0xFFu;
	typedef uint32_t v4 __attribute__((ext_vector_type(4)));
	v4 out;
#pragma unroll
	for (int x = 0; x < 4; x++)
		out[x] = select4(bit_to_mask(idx, 2 * x), bit_to_mask(idx, 2 * x + 1), p[0], p[1], p[2], p[3]);
	uint8_t *dst = pixels + (uint64_t)((by0 + brow) * 4u + trow) * pitch + (uint64_t)(bx0 + bcol) * 16u;
	*reinterpret_cast<v4 *>(dst) = out;
}

template <> struct Tile4x4<DecBC1> {
	static constexpr bool kAvailable = true;
	// BC1 has no invalid blocks, so the status word is never raised
	static hipError_t launch(const void *blocks, uint8_t *pixels, uint32_t wb, uint32_t hb, uint64_t pitch, uint32_t *,
			hipStream_t stream) {
		hipLaunchKernelGGL(decode_linear_tile4x4_bc1, dim3(wb / 16u, hb / 4u), dim3(256), 0, stream,
			static_cast<const uint2 *>(blocks), pixels, wb, pitch);
		return hipGetLastError();
	}
};

}  // namespace detexhip
