// kernels_sorted.h -- mode-sorted linear decode for the multi-mode formats (BC7 first).
//
// A wavefront of 64 independent BC7 blocks holds a mix of the eight modes, so the ordinary kernel runs
// the data-driven all-modes decoder (decode_bptc.h) on every lane: ~2700 issue cycles per wave, against
// 1100-1650 for a decoder compiled for ONE mode (field positions, widths and subset count constant).
// This kernel makes waves mode-uniform instead of making the decoder mode-agnostic:
//   1. lane = block as usual; every lane classifies its block (mode 0-7, reserved, or "past the end")
//   2. counting sort of the workgroup's 256 blocks by class through LDS (two ds_add per lane + an
//      exclusive prefix over <= 16 counters); lane j then owns the j-th block in class order
//   3. a wave whose 64 blocks share one class (wave-uniform test, scalar jump) runs that class's
//      specialised decoder; a wave straddling class boundaries runs the all-modes decoder
//   4. decoded texel rows go back through LDS to the lane that owns the block's position in the image,
//      and leave the chip exactly as in decode_linear: four wave-wide 1 KiB-contiguous streaming stores
// MEASURED (BC7 8192^2, stream U, profiles/r01): 124 us against 76 us for the ordinary kernel -- kept only as
// kernel variant 5 for the A/B record.  Why it loses: class boundaries almost never fall on wave boundaries
// (mode 0 is 128 +- 8 of a workgroup's 256 blocks), so typically one wave of four comes out uniform and the
// other three still run the all-modes decoder (SQ_INSTS_VALU 657 per wave vs ~600 unsorted), while four
// extra workgroup barriers and 32 KiB of LDS per workgroup (5 workgroups per CU) leave the SIMDs with too
// few issuing waves.  A larger sort domain (1024 blocks) was costed at < 10 % fewer instructions.
#pragma once
#include "kernels.h"
#include "decode_bptc.h"
#include "decode_bptc_r01.h"
#include "ab_traits.h"

namespace detexhip {

template <class Dec, int EPI, bool NT>
__global__ void decode_linear_sorted(const void *__restrict__ blocks, uint8_t *__restrict__ pixels, uint32_t width_in_blocks, uint32_t n_blocks,
		uint64_t pitch, uint32_t *__restrict__ status);

template <> struct ClassSorted<DecBPTC> {
	static constexpr bool kAvailable = true;
	static constexpr int kClasses = 9;			// modes 0-7, 8 = reserved (decompress-bptc.c:361)
	typedef uint4 Word;
	static DH uint32_t classify(const uint4 &blk) {
		const uint32_t low = blk.x & 0xFFu;
		return low ? (uint32_t)__builtin_ctz(low) : 8u;
	}
	// all lanes of the calling wave hold class c (wave-uniform)
	static DH bool decode_uniform(uint32_t c, const uint4 &blk, uint32_t (&d)[16]) {
		switch (c) {
		case 0: return r01::DecBPTCMode<0, 1>::template decode<false>(blk, 0xFFFFFFFFu, 0u, d);
		case 1: return r01::DecBPTCMode<1, 1>::template decode<false>(blk, 0xFFFFFFFFu, 0u, d);
		case 2: return r01::DecBPTCMode<2, 1>::template decode<false>(blk, 0xFFFFFFFFu, 0u, d);
		case 3: return r01::DecBPTCMode<3, 1>::template decode<false>(blk, 0xFFFFFFFFu, 0u, d);
		case 4: return r01::DecBPTCMode<4, 1>::template decode<false>(blk, 0xFFFFFFFFu, 0u, d);
		case 5: return r01::DecBPTCMode<5, 1>::template decode<false>(blk, 0xFFFFFFFFu, 0u, d);
		case 6: return r01::DecBPTCMode<6, 1>::template decode<false>(blk, 0xFFFFFFFFu, 0u, d);
		case 7: return r01::DecBPTCMode<7, 1>::template decode<false>(blk, 0xFFFFFFFFu, 0u, d);
		default: return false;				// reserved: zero-filled, raises the status word
		}
	}
	typedef r01::DecBPTCRegisterFields Generic;			// straddling waves: all-modes decoder
	template <int EPI> static hipError_t launch(const void *blocks, uint8_t *pixels, uint32_t wb, uint32_t n, uint64_t pitch, uint32_t *status, hipStream_t stream) {
		hipLaunchKernelGGL((decode_linear_sorted<DecBPTC, EPI, true>), dim3((n + 255u) / 256u), dim3(256), 0, stream, blocks, pixels, wb, n, pitch, status);
		return hipGetLastError();
	}
};

template <class Dec, int EPI, bool NT>
__global__ __launch_bounds__(256) void decode_linear_sorted(const void *__restrict__ blocks,
		uint8_t *__restrict__ pixels, uint32_t width_in_blocks, uint32_t n_blocks, uint64_t pitch,
		uint32_t *__restrict__ status) {
	typedef ClassSorted<Dec> S;
	typedef typename S::Word Word;
	constexpr int P = Dec::kPixelBytes;
	constexpr int ROW = EpilogueOf<Dec, EPI>::kRowDwords;
	static_assert(P == 4 && ROW == 4, "32-bit pixels, 16-byte rows");
	constexpr uint32_t kDead = S::kClasses;			// lanes past the end of the stream
	static_assert(S::kClasses + 1 <= 16, "counter table");
	typedef uint32_t v4 __attribute__((ext_vector_type(4)));
	__shared__ uint32_t count[16], cursor[16];
	// the sorted blocks are dead once every lane has fetched its own, the staged rows are born after the
	// decode: both live in one buffer (a workgroup barrier separates the two uses)
	__shared__ v4 exchange[4 * 256];
	Word *sorted_block = reinterpret_cast<Word *>(exchange);			// [256]
	uint16_t *sorted_meta = reinterpret_cast<uint16_t *>(exchange + 256);		// [256] owning lane | class << 8
	v4 (*stage)[256] = reinterpret_cast<v4 (*)[256]>(exchange);			// [texel row][owning lane]

	const uint32_t tid = threadIdx.x;
	if (tid < 16u) count[tid] = 0u;
	prepare_tables<typename S::Generic>();
	__syncthreads();

	// 1. classify
	const uint32_t i = blockIdx.x * 256u + tid;
	const bool live = i < n_blocks;
	Word blk = {};
	if (live) blk = reinterpret_cast<const Word *>(blocks)[i];
	const uint32_t cls = live ? S::classify(blk) : kDead;
	// 2. counting sort by class.  Wave-aggregated: one scalar loop over the classes present in the wave; each
	//    pass ballots the class's lanes, ranks them with mbcnt and lets their first lane add the whole count
	//    (256 lanes hammering <= 10 LDS counters with individual atomics serialise badly)
	uint32_t rank = 0, peers = 0;
	{
		uint64_t todo = __builtin_amdgcn_ballot_w64(true);
		while (todo) {
			const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)cls, (int)__builtin_ctzll(todo));
			const uint64_t members = __builtin_amdgcn_ballot_w64(cls == c);
			if (cls == c) {
				rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(members >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)members, 0u));
				peers = (uint32_t)__builtin_popcountll(members);
				if (rank == 0u) atomicAdd(&count[c], peers);
			}
			todo &= ~members;
		}
	}
	__syncthreads();
	if (tid < 16u) {
		uint32_t below = 0;
#pragma unroll
		for (uint32_t c = 0; c < 16u; c++) below += c < tid ? count[c] : 0u;
		cursor[tid] = below;
	}
	__syncthreads();
	uint32_t slot = 0;
	if (rank == 0u) slot = atomicAdd(&cursor[cls], peers);			// the wave's run inside its class
	{
		// hand the run start from each class's first lane to its peers
		uint64_t todo = __builtin_amdgcn_ballot_w64(true);
		while (todo) {
			const int leader = (int)__builtin_ctzll(todo);
			const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)cls, leader);
			const uint32_t start = (uint32_t)__builtin_amdgcn_readlane((int)slot, leader);
			const uint64_t members = __builtin_amdgcn_ballot_w64(cls == c);
			if (cls == c) slot = start + rank;
			todo &= ~members;
		}
	}
	sorted_block[slot] = blk;
	sorted_meta[slot] = (uint16_t)(tid | (cls << 8));
	__syncthreads();
	// Which wave takes which quarter of the class order rotates with the workgroup: the quarter that straddles
	// class boundaries (all-modes decoder, about twice the work) would otherwise always be wave 3, and the
	// waves of a workgroup are spread one per SIMD in a fixed order -- one SIMD would get all the slow waves.
	const uint32_t take = (tid + ((blockIdx.x & 3u) << 6)) & 255u;
	const Word mine = sorted_block[take];
	const uint32_t meta = sorted_meta[take], owner = meta & 0xFFu, c = meta >> 8;
	__syncthreads();			// every lane holds its block: the buffer may be reused for the rows

	// 3. decode: specialised when the whole wave holds one class
	uint32_t d[4 * P];
	bool ok = true;
	const uint32_t c0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
	if (__builtin_amdgcn_ballot_w64(c != c0) == 0) {
		if (c0 < kDead) ok = S::decode_uniform(c0, mine, d);
	} else if (c < kDead) {
		ok = S::Generic::template decode<false>(mine, 0xFFFFFFFFu, 0u, d);
	}
	if (!ok || c >= kDead) {
#pragma unroll
		for (int k = 0; k < 4 * P; k++) d[k] = 0u;
	}
	uint32_t o[4 * ROW];
	EpilogueOf<Dec, EPI>::apply(d, o);
	// 4. rows back to the lane that owns the block's place in the image
#pragma unroll
	for (int r = 0; r < 4; r++) stage[r][owner] = v4{ o[4 * r], o[4 * r + 1], o[4 * r + 2], o[4 * r + 3] };
	raise_status(!ok, status);
	__syncthreads();
	if (!live) return;
	uint32_t by, bx;
	split_index(i, width_in_blocks, by, bx);
	uint8_t *dst = pixels + (uint64_t)(by * 4u) * pitch + (uint64_t)bx * (4u * ROW);
#pragma unroll
	for (int r = 0; r < 4; r++) {
		const v4 row = stage[r][tid];
		if (NT) __builtin_nontemporal_store(row, reinterpret_cast<v4 *>(dst + (uint64_t)r * pitch));
		else *reinterpret_cast<v4 *>(dst + (uint64_t)r * pitch) = row;
	}
}

}  // namespace detexhip
