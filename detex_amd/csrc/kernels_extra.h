// kernels_extra.h -- the "next" rows of SURVEY.md section 8(f) that sit directly on the hot path:
//   8f-3  mip-chain batching: every level of a chain (ktx.c:108-171 yields them as separate small
//         textures) decoded by ONE launch over a descriptor table, so the small levels do not each
//         pay a launch + a mostly empty grid;
//   8f-4  block-mode histograms (the reference's detexGetMode<FMT> helpers, decompress-bc.c:63-69,
//         decompress-etc.c:183-190,370-395,721-742, decompress-bptc.c:603-610,
//         decompress-bptc-float.c:647-658) computed on the GPU over a whole block stream.
#pragma once
#include "dev_common.h"
#include "kernels.h"

namespace detexhip {

// ---- 8f-3: one launch over up to 16 levels ------------------------------------------------------
constexpr int kMaxLevels = 16;
struct LevelDesc {
	const void *blocks; uint8_t *pixels; uint64_t pitch;
	uint32_t width_in_blocks, n_blocks, width, height;
	uint32_t fast;			// 4-aligned geometry + vector-aligned rows: wave-wide row stores
	uint32_t pad;
};
struct LevelTable {
	uint32_t n_levels;
	uint32_t wg_start[kMaxLevels + 1];	// first workgroup of each level; [n_levels] = grid size
	LevelDesc level[kMaxLevels];
};

// ---- one block, for the reference's per-block entry points (detexDecompressBlock<FMT>, detex.h:435-531) -----------------------
// The block travels as a KERNEL ARGUMENT and the sixteen pixels + the decoder's bool go straight into pinned host memory: a
// call is one launch and one synchronisation, with no read across PCIe at all (the batched kernel fetching its single block
// from host memory measured 19.8 us per call, upload + launch + two downloads 16.7).  Every lane decodes the same block (the
// table copy and the per-lane LDS rows of the BPTC decoders want the whole workgroup); lane 0 stores.
template <class Dec, int EPI>
__global__ __launch_bounds__(256) void decode_single(const typename BlockWord<Dec::kBlockBytes>::type blk, uint32_t mode_mask, uint32_t flags,
		uint32_t *__restrict__ pixels, uint8_t *__restrict__ ok_out) {
	constexpr int ROW = EpilogueOf<Dec, EPI>::kRowDwords;
	prepare_tables<Dec>();
	prepare_epilogue<Dec, EPI>();
	uint32_t o[4 * ROW];
	const bool ok = decode_word<Dec, EPI, true>(blk, mode_mask, flags, o);
	if (threadIdx.x == 0) {
#pragma unroll
		for (int k = 0; k < 4 * ROW; k++) pixels[k] = o[k];
		*ok_out = ok ? 1 : 0;
	}
}

template <class Dec, int EPI>
__global__ __launch_bounds__(256) void decode_levels(const LevelTable table, uint32_t *__restrict__ status, uint32_t decode_flags) {
	constexpr int ROW = EpilogueOf<Dec, EPI>::kRowDwords;
	prepare_tables<Dec>();
	prepare_epilogue<Dec, EPI>();
	// workgroup -> level: wave-uniform scalar search over <= 16 entries
	uint32_t l = 0;
	for (uint32_t k = 1; k < table.n_levels; k++) l = blockIdx.x >= table.wg_start[k] ? k : l;
	const LevelDesc &lv = table.level[l];
	const uint32_t i = (blockIdx.x - table.wg_start[l]) * 256u + threadIdx.x;
	if constexpr (ROW == 8) {
		if (lv.fast) {		// workgroup-uniform (a workgroup never spans two levels): 64-bit pixels leave through the LDS transpose
			const bool live = i < lv.n_blocks;
			uint32_t o[4 * ROW];
			bool ok = true;
			if (live) ok = decode_block<Dec, EPI, false>(lv.blocks, i, 0xFFFFFFFFu, decode_flags, o);
			store_rows_wide_pixels(lv.pixels, lv.pitch, lv.width_in_blocks, i - (threadIdx.x & 63u), lv.n_blocks, live, o);
			if (live) raise_status(!ok, status);
			return;
		}
	}
	if (i >= lv.n_blocks) return;
	uint32_t o[4 * ROW];
	const bool ok = decode_block<Dec, EPI, false>(lv.blocks, i, 0xFFFFFFFFu, decode_flags, o);
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

// ---- 8f-4: mode classification ------------------------------------------------------------------
// Bin numbers are the reference's detexGetMode<FMT> return values; bin 15 collects the reserved
// BPTC / BPTC_FLOAT codes (where the reference returns -1).  Formats without modes use bin 0.
enum : int { kClassS3TC = 0, kClassS3TCat8, kClassETC1, kClassETC2, kClassETC2PT, kClassETC2at8, kClassBPTC, kClassBPTCFloat, kClassNone };

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
// bins per 64-bit atomic, so the grid stays at a few hundred workgroups (detexhip.hip has the sweep: 192-256).
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

}  // namespace detexhip
