// kernels_resident.h -- the resident service kernel of the host tier's smallest calls (path_types.h: ResidentMail has the protocol and
// the measurements; host side: host_resident.cpp).  Four workgroups; the first (the leader) polls the request line in pinned host
// memory, decodes one-block requests and one-tile textures alone and wakes the others through device memory for larger ones.  The
// kernel cannot outlive its use: it leaves after `idle_ticks` without a request, after `max_ticks` in total or after 2^22 polls in one
// wait, whichever comes first, and says so in ResidentMail::state before it goes (the host then starts a new one with its next request).
#pragma once
#include "dev_common.h"
#include "kernels.h"
#include "kernels_extra.h"

namespace detexhip {


// s_req layout: [0..11] payload words, [12] request number, [13] 0 = nothing yet, 1 = a request, 2 = leave
struct ResidentLeader {		// thread 0's state between polls
	uint32_t polls = 0;
	bool leaving = false;
	uint64_t t_start, t_last;
};

// leader, thread 0: what the request line (four chunks c[]) read by this poll means
DH uint32_t resident_leader_step(const ResidentArgs &a, const u32x4 (&c)[4], uint32_t last, ResidentLeader &me, uint32_t *s_req) {
	const uint32_t seq = c[0].x;
	bool leave = false;
	if (seq != last && c[1].x == seq && c[2].x == seq && c[3].x == seq) {
		if (me.leaving) __hip_atomic_store(&a.mail->state, a.instance << 2 | kResidentRunning, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		me.leaving = false;
		me.polls = 0;
		const uint32_t payload[12] = { c[0].y, c[0].z, c[0].w, c[1].y, c[1].z, c[1].w, c[2].y, c[2].z, c[2].w, c[3].y, c[3].z, c[3].w };
#pragma unroll
		for (int k = 0; k < 12; k++) s_req[k] = payload[k];
		s_req[12] = seq;
		if (payload[6] != kResidentStop) {
			if (payload[6] == kResidentTexture && payload[2] * payload[3] > 256u) {		// more than one tile: the other workgroups take theirs
#pragma unroll
				for (int k = 0; k < 12; k++) a.words[4 + k] = payload[k];
				__hip_atomic_store(a.words, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
			}
			return 1u;
		}
		leave = true;
	}
	if (!leave && !me.leaving) {
		const uint64_t now = wall_clock64();
		if (now - me.t_last > a.idle_ticks || now - me.t_start > a.max_ticks || ++me.polls > (1u << 22)) {
			// announce first, then look again (the next poll): a host that posts while this workgroup is on its way out either sees the
			// announcement (and waits for Running or Exited) or has its request seen by that poll
			__hip_atomic_store(&a.mail->state, a.instance << 2 | kResidentExiting, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "");
			me.leaving = true;
		}
		return 0u;
	}
	// a stop request, or announced + looked once more + nothing there
	__hip_atomic_store(a.words + 1, a.instance, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);		// the other workgroups leave too
	__hip_atomic_store(&a.mail->state, a.instance << 2 | kResidentExited, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
	return 2u;
}

// A request for more than one tile is acknowledged by EVERY workgroup of the instance -- those without a tile of it as well -- and the
// last of the kResidentWorkgroups acknowledgements publishes `done`: the host posts its next request only after `done`, so the
// leader cannot overwrite the payload in words[4..15] while a slow follower is still copying it, and a follower that wakes up late
// cannot run with a stale request number and a newer payload.  The counter (words[2..3] as ONE 64-bit word) carries its request:
// {instance, request number} << 2 | acknowledgements so far -- what a previous request or a previous instance left there (an
// instance that went away before all of its workgroups had answered) has another key and is simply replaced.  (thread 0 only)
DH bool resident_acknowledge(const ResidentArgs &a, uint32_t seq) {
	unsigned long long *word = reinterpret_cast<unsigned long long *>(a.words + 2);
	const unsigned long long key = ((unsigned long long)(a.instance & 0x3FFFFFFFu) << 32 | seq) << 2;
	unsigned long long seen = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	for (;;) {
		const bool mine = (seen & ~3ull) == key;
		const bool last = mine && (seen & 3ull) == kResidentWorkgroups - 1u;
		const unsigned long long next = last ? 0ull : (mine ? seen + 1ull : key | 1ull);
		if (__hip_atomic_compare_exchange_strong(word, &seen, next, __ATOMIC_ACQ_REL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return last;
	}
}
static_assert(kResidentWorkgroups == 4, "resident_acknowledge counts to four in two bits");

// the other workgroups, thread 0: wait for the leader
DH void resident_wait_leader(const ResidentArgs &a, uint32_t last, uint64_t t_start, uint32_t *s_req) {
	for (;;) {
		const uint32_t seq = __hip_atomic_load(a.words, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
		if ((int32_t)(seq - last) > 0) {
#pragma unroll
			for (int k = 0; k < 12; k++) s_req[k] = a.words[4 + k];
			s_req[12] = seq; s_req[13] = 1u;
			return;
		}
		if (__hip_atomic_load(a.words + 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == a.instance || wall_clock64() - t_start > 2u * a.max_ticks) {
			s_req[13] = 2u;
			return;
		}
		__builtin_amdgcn_s_sleep(4);
	}
}

template <class Dec, int EPI>
__global__ __launch_bounds__(256) void decode_resident(const ResidentArgs a) {
	constexpr int ROW = EpilogueOf<Dec, EPI>::kRowDwords;
	using Word = typename BlockWord<Dec::kBlockBytes>::type;
	constexpr uint32_t kChunkBytes = Dec::kBlockBytes * 2u;		// a lane's tagged chunks: 16 bytes per eight block bytes
	prepare_tables<Dec>();
	prepare_epilogue<Dec, EPI>();
	__shared__ uint32_t s_req[16];
	const bool leader = blockIdx.x == 0;
	uint32_t last = a.start_seq;
	ResidentLeader me;
	me.t_start = me.t_last = wall_clock64();
	uint32_t speculative = 0;		// (workgroup-uniform) polls left in which the leader's lanes read their chunks along with the request line
	const uint8_t *my_chunks = static_cast<const uint8_t *>(a.blocks) + threadIdx.x * kChunkBytes;
	for (;;) {
		u32x4 t0 = {}, t1 = {};
		const bool with_chunks = leader && speculative != 0u;
		if (leader) {
			u32x4 c[4] = {};
			// (one instruction stream for the wave: the lanes other than 0 aim their four line reads at device memory and ignore them)
			if (with_chunks) load_system_4x16_and_2x16(threadIdx.x == 0 ? static_cast<const void *>(a.mail->chunk) : static_cast<const void *>(a.words), my_chunks, c[0], c[1], c[2], c[3], t0, t1);
			else if (threadIdx.x == 0) load_system_4x16(a.mail->chunk, c[0], c[1], c[2], c[3]);
			if (threadIdx.x == 0) s_req[13] = resident_leader_step(a, c, last, me, s_req);
		} else if (threadIdx.x == 0) {
			resident_wait_leader(a, last, me.t_start, s_req);
		}
		__syncthreads();
		uint32_t q[7];
#pragma unroll
		for (int k = 0; k < 7; k++) q[k] = s_req[k];
		const uint32_t seq = s_req[12], what = s_req[13], s_tiled = s_req[7];
		__syncthreads();			// thread 0 writes s_req again in the next round
		if (what == 0u) { speculative -= speculative != 0u ? 1u : 0u; continue; }
		if (what == 2u) break;
		last = seq;
		speculative = q[6] == kResidentTagged ? a.speculative_polls : 0u;	// (a row of tagged requests is likely to go on)
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
			const bool block_major = s_tiled != 0u;		// detexDecompressTextureTiled: block i's sixteen pixels at pixels + i * 16 * pixel size
			const uint32_t i = blockIdx.x * 256u + threadIdx.x;
			Word blk;
			if (q[6] == kResidentTagged) {
				// (only the leader gets here.)  The chunks this poll read are this request's if they carry its number; otherwise they are
				// read now -- after the request was seen, so they are
				const bool mine = threadIdx.x >= lv.n_blocks || (t0.z == seq && (Dec::kBlockBytes == 8 || t1.z == seq));
				if (!__syncthreads_and(with_chunks && mine)) load_system_2x16(my_chunks, t0, t1);
				if constexpr (sizeof(Word) == 16) blk = Word{ t0.x, t0.y, t1.x, t1.y };
				else blk = Word{ t0.x, t0.y };
			} else {
				n_tiles = lv.n_blocks > 256u ? (lv.n_blocks + 255u) / 256u : 1u;
				blk = load_block<Dec>(lv.blocks, i < lv.n_blocks ? i : 0u);
			}
			if (blockIdx.x >= n_tiles) {
				// (workgroup-uniform) woken for a texture with fewer tiles than workgroups: nothing to decode, but acknowledged below
			} else if (block_major) {
				if (i < lv.n_blocks) {
					uint32_t o[4 * ROW];
					const bool ok = decode_word<Dec, EPI, false>(blk, 0xFFFFFFFFu, q[5], o);
					u32x4 *out = reinterpret_cast<u32x4 *>(a.pixels + (uint64_t)i * (16u * ROW));
#pragma unroll
					for (int k = 0; k < ROW; k++) out[k] = u32x4{ o[4 * k], o[4 * k + 1], o[4 * k + 2], o[4 * k + 3] };
					raise_status(!ok, &a.mail->status);
				}
			} else {
				decode_level_tile_from<Dec, EPI>(lv, i, &a.mail->status, q[5], [&](uint32_t) { return blk; });
			}
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "");	// system scope: this thread's stores are on their way to host memory
		__syncthreads();
		if (threadIdx.x == 0) {
			const bool publish = n_tiles <= 1u || resident_acknowledge(a, seq);
			if (publish) __hip_atomic_store(&a.mail->done, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
			me.t_last = wall_clock64();
		}
	}
}

}  // namespace detexhip
