// kernels_resident.h -- the resident service kernel of the host tier's smallest calls (path_types.h: ResidentMail has the protocol and
// the measurements; host side: host_resident.cpp).  Four workgroups; the first polls the request line in pinned host memory, decodes
// one-block requests and one-tile textures alone and wakes the others through device memory for larger ones.  The kernel cannot
// outlive its use: it leaves after `idle_ticks` without a request, after `max_ticks` in total or after 2^22 polls in one wait, whichever comes
// first, and says so in ResidentMail::state before it goes (the host then starts a new one with its next request).
#pragma once
#include "dev_common.h"
#include "kernels.h"
#include "kernels_extra.h"

namespace detexhip {

// s_req layout: [0..11] payload words, [12] request number, [13] 1 = leave
// leading workgroup, thread 0: wait for the next request of the host
DH void resident_wait_host(const ResidentArgs &a, uint32_t last, uint64_t t_last, uint64_t t_start, uint32_t *s_req) {
	bool leaving = false;
	for (uint32_t polls = 0;; polls++) {
		u32x4 c0, c1, c2, c3;
		load_system_4x16(a.mail->chunk, c0, c1, c2, c3);
		const uint32_t seq = c0.x;
		if (seq != last && c1.x == seq && c2.x == seq && c3.x == seq) {
			if (leaving) __hip_atomic_store(&a.mail->state, a.instance << 2 | kResidentRunning, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			const uint32_t payload[12] = { c0.y, c0.z, c0.w, c1.y, c1.z, c1.w, c2.y, c2.z, c2.w, c3.y, c3.z, c3.w };
#pragma unroll
			for (int k = 0; k < 12; k++) s_req[k] = payload[k];
			s_req[12] = seq;
			s_req[13] = payload[6] == kResidentStop ? 1u : 0u;
			if (payload[6] == kResidentStop) break;
			if (payload[6] == kResidentTexture && payload[2] * payload[3] > 256u) {		// more than one tile: the other workgroups take theirs
#pragma unroll
				for (int k = 0; k < 12; k++) a.words[4 + k] = payload[k];
				__hip_atomic_store(a.words, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
			}
			return;
		}
		if (leaving) { s_req[13] = 1u; break; }		// announced, looked once more, nothing there
		const uint64_t now = wall_clock64();
		if (now - t_last > a.idle_ticks || now - t_start > a.max_ticks || polls > (1u << 22)) {
			// announce first, then look again: a host that posts while this workgroup is on its way out either sees the announcement
			// (and waits for Running or Exited) or has its request seen by the next poll
			__hip_atomic_store(&a.mail->state, a.instance << 2 | kResidentExiting, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "");
			leaving = true;
		}
	}
	__hip_atomic_store(a.words + 1, a.instance, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);		// the other workgroups leave too
	__hip_atomic_store(&a.mail->state, a.instance << 2 | kResidentExited, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// the other workgroups, thread 0: wait for the leading workgroup
DH void resident_wait_leader(const ResidentArgs &a, uint32_t last, uint64_t t_start, uint32_t *s_req) {
	for (;;) {
		const uint32_t seq = __hip_atomic_load(a.words, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
		if ((int32_t)(seq - last) > 0) {
#pragma unroll
			for (int k = 0; k < 12; k++) s_req[k] = a.words[4 + k];
			s_req[12] = seq; s_req[13] = 0u;
			return;
		}
		if (__hip_atomic_load(a.words + 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == a.instance || wall_clock64() - t_start > 2u * a.max_ticks) {
			s_req[13] = 1u;
			return;
		}
		__builtin_amdgcn_s_sleep(4);
	}
}

template <class Dec, int EPI>
__global__ __launch_bounds__(256) void decode_resident(const ResidentArgs a) {
	constexpr int ROW = EpilogueOf<Dec, EPI>::kRowDwords;
	using Word = typename BlockWord<Dec::kBlockBytes>::type;
	prepare_tables<Dec>();
	prepare_epilogue<Dec, EPI>();
	__shared__ uint32_t s_req[16];
	uint32_t last = a.start_seq;
	const uint64_t t_start = wall_clock64();
	uint64_t t_last = t_start;			// (thread 0's copy is the one that counts)
	for (;;) {
		if (threadIdx.x == 0) {
			if (blockIdx.x == 0) resident_wait_host(a, last, t_last, t_start, s_req);
			else resident_wait_leader(a, last, t_start, s_req);
		}
		__syncthreads();
		uint32_t q[7];
#pragma unroll
		for (int k = 0; k < 7; k++) q[k] = s_req[k];
		const uint32_t seq = s_req[12], leave = s_req[13];
		__syncthreads();			// thread 0 writes s_req again in the next round
		if (leave) break;
		last = seq;
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");	// system scope: nothing an earlier request left in the caches is read for this one
		uint32_t n_tiles = 1;
		if (q[6] == kResidentBlock) {
			// every lane decodes the same block, lane 0 stores (kernels_extra.h: decode_single)
			Word blk;
			if constexpr (sizeof(Word) == 16) blk = Word{ q[0], q[1], q[2], q[3] };
			else blk = Word{ q[0], q[1] };
			uint32_t o[4 * ROW];
			const bool ok = decode_word<Dec, EPI, true>(blk, q[4], q[5], o);
			if (threadIdx.x == 0) {
				u32x4 *out = reinterpret_cast<u32x4 *>(a.pixels);		// (16-byte stores: a quarter of the writes across the link)
#pragma unroll
				for (int k = 0; k < ROW; k++) out[k] = u32x4{ o[4 * k], o[4 * k + 1], o[4 * k + 2], o[4 * k + 3] };
				if (!ok) __hip_atomic_store(&a.mail->status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
		} else {
			LevelDesc lv;
			lv.blocks = a.blocks; lv.pixels = a.pixels; lv.pitch = (uint64_t)q[0] * ROW;	// ROW dwords per four pixels = bytes per pixel
			lv.width = q[0]; lv.height = q[1]; lv.width_in_blocks = q[2]; lv.n_blocks = q[2] * q[3];
			lv.fast = (q[0] & 3u) == 0u && (q[1] & 3u) == 0u && q[2] * 4u == q[0] && q[3] * 4u == q[1];
			n_tiles = lv.n_blocks > 256u ? (lv.n_blocks + 255u) / 256u : 1u;
			if (blockIdx.x >= n_tiles) continue;		// (workgroup-uniform) woken for a texture with fewer tiles than workgroups
			decode_level_tile<Dec, EPI>(lv, blockIdx.x * 256u + threadIdx.x, &a.mail->status, q[5]);
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "");	// system scope: this thread's stores are on their way to host memory
		__syncthreads();
		if (threadIdx.x == 0) {
			bool publish = n_tiles <= 1u;
			if (!publish && __hip_atomic_fetch_add(a.words + 2, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1u == n_tiles) {
				__hip_atomic_store(a.words + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				publish = true;
			}
			if (publish) __hip_atomic_store(&a.mail->done, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
			t_last = wall_clock64();
		}
	}
}

}  // namespace detexhip
