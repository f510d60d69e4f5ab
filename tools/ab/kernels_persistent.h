// ab/kernels_persistent.h -- the linear kernel on a PERSISTENT grid (profiles/AB_RECORD.md): workgroups decode tiles
// blockIdx.x, blockIdx.x + gridDim.x, ... on a grid that just fills the chip, the format tables are copied once per
// resident workgroup and the next tile's block is requested before the current one is decoded.  Measured 1-20 % slower than
// one workgroup per tile for every format (BC7: 58.3 vs 54.3 us once its tables had shrunk to 3.6 KiB), so the product
// library does not contain it; only compiled into the measurement build, make lib-ab (variants 6 / 7 of ab_dispatch.h).
#pragma once
#include "kernels.h"

namespace detexhip {

template <class Dec, int EPI>
__global__ __launch_bounds__(256, WavesPerSimd<Dec>::value) void decode_linear_persistent(const void *__restrict__ blocks,
		uint8_t *__restrict__ pixels, uint32_t width_in_blocks, uint32_t n_blocks, uint64_t pitch, uint32_t *__restrict__ status) {
	constexpr int ROW = EpilogueOf<Dec, EPI>::kRowDwords;
	static_assert(ROW != 8, "64-bit pixels leave through the per-wave transpose of decode_linear; not needed for this A/B");
	using Word = typename BlockWord<Dec::kBlockBytes>::type;
	// the first block is requested BEFORE the table copy and waited for after its barrier
	const uint32_t first = blockIdx.x * 256u + threadIdx.x;
	Word blk = reinterpret_cast<const Word *>(blocks)[first < n_blocks ? first : n_blocks - 1u];
	prepare_tables<Dec>();
	prepare_epilogue<Dec, EPI>();
	pin_block(blk);
	const uint32_t n_tiles = (n_blocks + 255u) >> 8;
	// per-lane LDS rows need no barrier between tiles: a lane only reads what it wrote
	for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
		const Word cur = blk;
		const uint32_t i_next = (tile + gridDim.x) * 256u + threadIdx.x;
		blk = reinterpret_cast<const Word *>(blocks)[i_next < n_blocks ? i_next : n_blocks - 1u];	// requested now ...
		const uint32_t i = tile * 256u + threadIdx.x;
		if (i < n_blocks) {
			uint32_t o[4 * ROW];
			const bool ok = decode_word<Dec, EPI, false>(cur, 0xFFFFFFFFu, 0u, o);
			uint32_t by, bx;
			split_index(i, width_in_blocks, by, bx);
			uint8_t *dst = pixels + (uint64_t)(by * 4u) * pitch + (uint64_t)bx * (4u * ROW);
			if (stores_enabled(o)) {
#pragma unroll
				for (int r = 0; r < 4; r++) store_row<ROW, 4>(dst + (uint64_t)r * pitch, o + r * ROW);
			}
			raise_status(!ok, status);
		}
		if constexpr (Tune::kBc7Prio != 0) __builtin_amdgcn_s_setprio(0);	// the next tile starts at the bottom again
		pin_block(blk);											// ... waited for after this tile
	}
}

}  // namespace detexhip
