// ab/kernels_store_shape.h -- decode kernels in the store shapes of profiles/AB_RECORD.md round 6 ("is the write rate a matter of how a tile's
// texel rows are dealt out?"): an image-layout fill in which wave w of a 64-block workgroup writes texel row w -- ONE store per lane -- reaches
// 6.85-6.98 TB/s where the decode kernels' four stores per lane reach 6.2-6.4 (tools/gpu_store_lanes.py).  Three decoders in such shapes, and
// BC6H with several tiles per workgroup; all bit-exact, all slower than the product where it matters (blocks out of HBM: a lane that stores one
// texel row holds a quarter of a block, so a CU has 4 KiB of blocks in flight instead of 16).  Only compiled into the measurement build, make
// lib-ab (variants 8-11 of ab_dispatch.h); the product library contains none of this.
#pragma once
#include "kernels.h"

namespace detexhip {

template <class D, class = void> struct HasTables { static constexpr bool value = false; };
template <class D> struct HasTables<D, decltype(D::prepare(), void())> { static constexpr bool value = true; };

// variant 8: the texel rows of a tile dealt to the WAVES instead of to the lanes' registers.  A workgroup covers 64 consecutive blocks; lane l of
// wave w decodes block l and stores its texel row w: one store instruction per lane, each a 1 KiB run.  The price is every block loaded and
// decoded by four lanes (each copy of the decoder keeps only what its row needs: a scalar switch over four inlined copies).
template <class Dec, int EPI>
__global__ __launch_bounds__(256) void decode_linear_rowwave(const void *__restrict__ blocks, uint8_t *__restrict__ pixels, uint32_t width_in_blocks,
		uint32_t n_blocks, uint64_t pitch, uint32_t *__restrict__ status, uint32_t decode_flags) {
	constexpr int ROW = EpilogueOf<Dec, EPI>::kRowDwords;
	using Word = typename BlockWord<Dec::kBlockBytes>::type;
	prepare_tables<Dec>();
	prepare_epilogue<Dec, EPI>();
	const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	const uint32_t i = blockIdx.x * 64u + (threadIdx.x & 63u);
	Word blk = reinterpret_cast<const Word *>(blocks)[i < n_blocks ? i : n_blocks - 1u];
	pin_block(blk);
	if (i >= n_blocks) return;
	uint32_t by, bx;
	split_index(i, width_in_blocks, by, bx);
	uint8_t *dst = pixels + (uint64_t)(by * 4u + w) * pitch + (uint64_t)bx * (4u * ROW);
	bool ok = true;
	auto one = [&](auto r) {
		uint32_t o[4 * ROW];
		ok = decode_word<Dec, EPI, false>(blk, 0xFFFFFFFFu, decode_flags, o);
		store_row<ROW, StorePolicy<Dec>::value>(dst, o + decltype(r)::value * ROW);
	};
	switch (w) {
	case 0: one(std::integral_constant<int, 0>{}); break;
	case 1: one(std::integral_constant<int, 1>{}); break;
	case 2: one(std::integral_constant<int, 2>{}); break;
	default: one(std::integral_constant<int, 3>{}); break;
	}
	if (w == 0) raise_status(!ok, status);
}

// variant 9: the cooperative form of the same shape for decoders with a row-split form (RowSplitOf<Dec>: BC1 / BC1A, native target): wave 0
// loads the 64 blocks and decodes what a block's sixteen texels share (palette + selector word) into LDS; behind the barrier wave w picks and
// stores texel row w.  One global load and one palette per block, one store per lane.
template <class Dec> struct RowSplitOf { static constexpr bool kAvailable = false; };
template <class Dec>
__global__ __launch_bounds__(256) void decode_linear_rowsplit(const void *__restrict__ blocks, uint8_t *__restrict__ pixels, uint32_t width_in_blocks,
		uint32_t n_blocks, uint64_t pitch) {
	using RS = RowSplitOf<Dec>;
	using Word = typename BlockWord<Dec::kBlockBytes>::type;
	constexpr int N = RS::kSharedDwords;
	__shared__ uint32_t shared[N][64];
	const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63u;
	const uint32_t i = blockIdx.x * 64u + lane;
	if (w == 0) {
		Word blk = reinterpret_cast<const Word *>(blocks)[i < n_blocks ? i : n_blocks - 1u];
		pin_block(blk);
		uint32_t s[N];
		RS::row_setup(blk, s);
#pragma unroll
		for (int k = 0; k < N; k++) shared[k][lane] = s[k];
	}
	// (the destination is computed while wave 0 works)
	uint32_t by, bx;
	split_index(i, width_in_blocks, by, bx);
	uint8_t *dst = pixels + (uint64_t)(by * 4u + w) * pitch + (uint64_t)bx * 16u;
	__syncthreads();
	if (i >= n_blocks) return;
	uint32_t s[N], o[4];
#pragma unroll
	for (int k = 0; k < N; k++) s[k] = shared[k][lane];
	RS::row_texels(s, w, o);
	store_row<4, StorePolicy<Dec>::value>(dst, o);
}

// variant 10: decode_linear's 32-bit-pixel path with one-wave workgroups (64 blocks, 4 KiB of pixels per workgroup, four stores per lane) -- is
// it the bytes a workgroup has in flight that the write path dislikes?  (No.)  Decoders without format tables only.
template <class Dec, int EPI>
__global__ __launch_bounds__(64) void decode_linear_onewave(const void *__restrict__ blocks, uint8_t *__restrict__ pixels, uint32_t width_in_blocks,
		uint32_t n_blocks, uint64_t pitch, uint32_t *__restrict__ status, uint32_t decode_flags) {
	constexpr int ROW = EpilogueOf<Dec, EPI>::kRowDwords;
	using Word = typename BlockWord<Dec::kBlockBytes>::type;
	const uint32_t i = blockIdx.x * 64u + threadIdx.x;
	Word blk = reinterpret_cast<const Word *>(blocks)[i < n_blocks ? i : n_blocks - 1u];
	pin_block(blk);
	if (i >= n_blocks) return;
	uint32_t o[4 * ROW];
	const bool ok = decode_word<Dec, EPI, false>(blk, 0xFFFFFFFFu, decode_flags, o);
	uint32_t by, bx;
	split_index(i, width_in_blocks, by, bx);
	uint8_t *dst = pixels + (uint64_t)(by * 4u) * pitch + (uint64_t)bx * (4u * ROW);
#pragma unroll
	for (int r = 0; r < 4; r++) store_row<ROW, StorePolicy<Dec>::value>(dst + (uint64_t)r * pitch, o + r * ROW);
	raise_status(!ok, status);
}

// variant 11: the 64-bit-pixel path of decode_linear with TILES consecutive tiles per workgroup.  All TILES blocks of a lane are requested
// before the table copy; the first tile waits for its block behind that copy's barrier as in the product, the later ones found theirs long ago
// -- with the blocks out of HBM only one tile in TILES pays the round trip, and the tables are copied once for TILES tiles.  (BC6H 102 ->
// 105-115 us: a workgroup that lives twice as long spreads its store bursts instead of hiding its reads.)
template <class Dec, int EPI, int TILES>
__global__ __launch_bounds__(256, WavesPerSimd<Dec>::value) void decode_linear_wide_tiles(const void *__restrict__ blocks,
		uint8_t *__restrict__ pixels, uint32_t width_in_blocks, uint32_t n_blocks, uint64_t pitch,
		uint32_t *__restrict__ status, uint32_t decode_flags) {
	constexpr int ROW = EpilogueOf<Dec, EPI>::kRowDwords;
	static_assert(ROW == 8, "64-bit pixels only");
	using Word = typename BlockWord<Dec::kBlockBytes>::type;
	const uint32_t i0 = blockIdx.x * (256u * TILES) + threadIdx.x;
	Word blk[TILES];
#pragma unroll
	for (int t = 0; t < TILES; t++) {
		const uint32_t i = i0 + 256u * t;
		blk[t] = reinterpret_cast<const Word *>(blocks)[i < n_blocks ? i : n_blocks - 1u];
	}
	prepare_tables<Dec>();
	prepare_epilogue<Dec, EPI>();
#pragma unroll
	for (int t = 0; t < TILES; t++) {
		if (blockIdx.x * (256u * TILES) + 256u * t >= n_blocks) break;	// (uniform: the stream's last workgroup)
		pin_block(blk[t]);
		const uint32_t i = i0 + 256u * t;
		const bool live = i < n_blocks;
		uint32_t o[4 * ROW];
		bool ok = true;
		if (live) ok = decode_word<Dec, EPI, false>(blk[t], 0xFFFFFFFFu, decode_flags, o);
		store_rows_wide_pixels(pixels, pitch, width_in_blocks, i - (threadIdx.x & 63u), n_blocks, live, o);
		if (live) raise_status(!ok, status);
	}
}

}  // namespace detexhip
