// kernels_extra.h -- the "next" rows of SURVEY.md section 8(f) that sit directly on the hot path:
//   8f-3  mip-chain batching: every level of a chain (ktx.c:108-171 yields them as separate small
//         textures) decoded by ONE launch over a descriptor table, so the small levels do not each
//         pay a launch + a mostly empty grid;
//   and the one-block kernel behind the reference's per-block entry points.  (8f-4, the mode histograms, is histogram.hip.)
#pragma once
#include "dev_common.h"
#include "kernels.h"

namespace detexhip {

// (LevelDesc / LevelTable / kMaxLevels: path_types.h)

// ---- one block, for the reference's per-block entry points (detexDecompressBlock<FMT>, detex.h:435-531) -----------------------
// The block travels as a KERNEL ARGUMENT and the sixteen pixels + the decoder's bool go straight into pinned host memory: a
// call is one launch and one synchronisation, with no read across PCIe at all (the batched kernel fetching its single block
// from host memory measured 19.8 us per call, upload + launch + two downloads 16.7).  Every lane decodes the same block (the
// table copy and the per-lane LDS rows of the BPTC decoders want the whole workgroup); lane 0 stores.
// Completion: the host does not wait for the stream but polls *done, a word in the same pinned buffer, which lane 0 releases at
// system scope after its stores (path_types.h: Completion).
template <class Dec, int EPI>
__global__ __launch_bounds__(256) void decode_single(const typename BlockWord<Dec::kBlockBytes>::type blk, uint32_t mode_mask, uint32_t flags,
		uint32_t *__restrict__ pixels, uint8_t *__restrict__ ok_out, uint32_t *__restrict__ done, uint32_t ticket) {
	constexpr int ROW = EpilogueOf<Dec, EPI>::kRowDwords;
	prepare_tables<Dec>();
	prepare_epilogue<Dec, EPI>();
	uint32_t o[4 * ROW];
	const bool ok = decode_word<Dec, EPI, true>(blk, mode_mask, flags, o);
	if (threadIdx.x == 0) {
		u32x4 *out = reinterpret_cast<u32x4 *>(pixels);		// (16-byte stores: the buffer is 256-byte aligned, host_tier.cpp: direct_exchange)
#pragma unroll
		for (int k = 0; k < ROW; k++) out[k] = u32x4{ o[4 * k], o[4 * k + 1], o[4 * k + 2], o[4 * k + 3] };
		*ok_out = ok ? 1 : 0;
		if (done) __hip_atomic_store(done, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);	// (release: the stores above are visible to the host first)
	}
}

// the workgroup's part of a grid-wide completion (every thread of the workgroup calls this after its last store; path_types.h)
DH void publish_completion(const Completion &c) {
	if (c.done == nullptr) return;				// kernel argument: uniform
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "");		// system scope: this thread's stores (pixels, status word) are on their way to host memory
	__syncthreads();
	if (threadIdx.x == 0) {
		const uint32_t finished = __hip_atomic_fetch_add(c.counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1u;
		if (finished == gridDim.x) {
			__hip_atomic_store(c.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			__hip_atomic_store(c.done, c.ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
		}
	}
}

// ---- a small batch of independent blocks for the host tier's batched per-block call (include/detexhip.h: detexhipDecompressBlocks; the
// loop a client of the leaf functions detex.h:435-531 would otherwise write): blocks, pixels, ok bytes and the status word all live in
// pinned host memory, mode_mask and flags are honoured per block exactly as the leaf functions do, output is block-major (a lane's
// 16-byte stores fill its block's 64 / 128 contiguous bytes), and the last workgroup releases the completion word the caller polls.
// Batches too large for the pinned exchange go through decode_blocks<.., CHECKED = true> on device buffers instead.
template <class Dec, int EPI>
__global__ __launch_bounds__(256) void decode_blocks_direct(const void *__restrict__ blocks, uint8_t *__restrict__ pixels, uint32_t n_blocks, uint32_t mode_mask,
		uint32_t flags, uint8_t *__restrict__ ok_out, uint32_t *__restrict__ status, const Completion completion) {
	constexpr int ROW = EpilogueOf<Dec, EPI>::kRowDwords;
	prepare_tables<Dec>();
	prepare_epilogue<Dec, EPI>();
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i < n_blocks) {
		uint32_t o[4 * ROW];
		const bool ok = decode_block<Dec, EPI, true>(blocks, i, mode_mask, flags, o);
		u32x4 *out = reinterpret_cast<u32x4 *>(pixels) + (uint64_t)i * ROW;
#pragma unroll
		for (int k = 0; k < ROW; k++) out[k] = u32x4{ o[4 * k], o[4 * k + 1], o[4 * k + 2], o[4 * k + 3] };
		ok_out[i] = ok ? 1 : 0;
		raise_status(!ok, status);
	}
	publish_completion(completion);
}

// one workgroup's 256 blocks of one level; `fetch(i)` delivers block i (from the level's stream, or a word the caller already holds)
template <class Dec, int EPI, class Fetch> DH void decode_level_tile_from(const LevelDesc &lv, uint32_t i, uint32_t *__restrict__ status, uint32_t decode_flags, Fetch &&fetch) {
	constexpr int ROW = EpilogueOf<Dec, EPI>::kRowDwords;
	if constexpr (ROW == 8) {
		if (lv.fast) {		// workgroup-uniform (a workgroup never spans two levels): 64-bit pixels leave through the LDS transpose
			const bool live = i < lv.n_blocks;
			uint32_t o[4 * ROW];
			bool ok = true;
			if (live) ok = decode_word<Dec, EPI, false>(fetch(i), 0xFFFFFFFFu, decode_flags, o);
			store_rows_wide_pixels(lv.pixels, lv.pitch, lv.width_in_blocks, i - (threadIdx.x & 63u), lv.n_blocks, live, o);
			if (live) raise_status(!ok, status);
			return;
		}
	}
	if (i >= lv.n_blocks) return;
	uint32_t o[4 * ROW];
	const bool ok = decode_word<Dec, EPI, false>(fetch(i), 0xFFFFFFFFu, decode_flags, o);
	uint32_t by, bx;
	split_index(i, lv.width_in_blocks, by, bx);
	uint8_t *dst = lv.pixels + (uint64_t)(by * 4u) * lv.pitch + (uint64_t)bx * (4u * ROW);
	if (lv.fast) {
#pragma unroll
		for (int r = 0; r < 4; r++) store_row<ROW, 4>(dst + (uint64_t)r * lv.pitch, o + r * ROW);
	} else {
#pragma unroll
		for (int r = 0; r < 4; r++) {
			if (by * 4u + r >= lv.height) continue;
#pragma unroll
			for (int x = 0; x < 4; x++)
				if (bx * 4u + x < lv.width) store_pixel<ROW>(dst + (uint64_t)r * lv.pitch + x * ROW, o + r * ROW, x);
		}
	}
	raise_status(!ok, status);
}
template <class Dec, int EPI> DH void decode_level_tile(const LevelDesc &lv, uint32_t i, uint32_t *__restrict__ status, uint32_t decode_flags) {
	decode_level_tile_from<Dec, EPI>(lv, i, status, decode_flags, [&](uint32_t k) { return load_block<Dec>(lv.blocks, k); });
}

// Also the kernel of the host tier's small textures (one level, blocks and pixels in pinned host memory): `completion` then names the
// word the last workgroup releases for the polling caller (path_types.h: Completion; {nullptr, ...} otherwise).
template <class Dec, int EPI>
__global__ __launch_bounds__(256) void decode_levels(const LevelTable table, uint32_t *__restrict__ status, uint32_t decode_flags, const Completion completion) {
	prepare_tables<Dec>();
	prepare_epilogue<Dec, EPI>();
	// workgroup -> level: wave-uniform scalar search over <= 16 entries
	uint32_t l = 0;
	for (uint32_t k = 1; k < table.n_levels; k++) l = blockIdx.x >= table.wg_start[k] ? k : l;
	decode_level_tile<Dec, EPI>(table.level[l], (blockIdx.x - table.wg_start[l]) * 256u + threadIdx.x, status, decode_flags);
	publish_completion(completion);
}

}  // namespace detexhip
