// tests/host_emul/hip_host_shim.h -- TEST-ONLY emulation of the handful of gfx950 builtins the
// device decoders use, so that tests/test_host_emulation.py can compile detex_amd/csrc/decode_*.h
// with g++ and run the *device decode logic* on the CPU against the oracle in containers without
// a GPU.  Never compiled into libdetexhip.so; the -m gpu tests remain the parity tests proper.
#pragma once
#include <stdint.h>
#include <algorithm>
#define DH inline
#define __constant__ static const
struct uint2 { uint32_t x, y; };
struct uint4 { uint32_t x, y, z, w; };
using std::min;
using std::max;
static inline uint32_t __builtin_amdgcn_ubfe(uint32_t v, uint32_t off, uint32_t w) {
	off &= 31; w &= 31;			// v_bfe_u32 uses the low 5 bits of offset and width
	return w == 0 ? 0u : (v >> off) & ((1u << w) - 1u);
}
static inline int32_t __builtin_amdgcn_sbfe(int32_t v, uint32_t off, uint32_t w) {
	off &= 31; w &= 31;
	if (w == 0) return 0;
	const uint32_t f = ((uint32_t)v >> off) & ((1u << w) - 1u);
	return (f & (1u << (w - 1))) ? (int32_t)(f | ~((1u << w) - 1u)) : (int32_t)f;
}
static inline uint32_t __builtin_amdgcn_perm(uint32_t s0, uint32_t s1, uint32_t sel) {
	const uint64_t pool = ((uint64_t)s0 << 32) | s1;
	uint32_t r = 0;
	for (int i = 0; i < 4; i++) {
		const uint32_t c = (sel >> (8 * i)) & 0xFF;
		uint32_t b;
		if (c <= 7) b = (uint32_t)(pool >> (8 * c)) & 0xFF;
		else if (c == 0x0C) b = 0x00;
		else if (c >= 0x0D) b = 0xFF;
		else b = ((pool >> (16 * (c - 8) + 15)) & 1) ? 0xFF : 0x00;	// 8..11: sign of a 16-bit half
		r |= b << (8 * i);
	}
	return r;
}
static inline uint32_t __builtin_amdgcn_alignbit(uint32_t hi, uint32_t lo, uint32_t s) {
	return (uint32_t)(((((uint64_t)hi) << 32) | lo) >> (s & 31));
}
static inline uint32_t __brev(uint32_t v) { uint32_t r = 0; for (int i = 0; i < 32; i++) r |= ((v >> i) & 1u) << (31 - i); return r; }
static inline int32_t __mul24(int32_t a, int32_t b) {
	const int32_t sa = (int32_t)((uint32_t)a << 8) >> 8, sb = (int32_t)((uint32_t)b << 8) >> 8;
	return (int32_t)((int64_t)sa * sb);
}
// wave-uniform votes: the emulation runs one "lane" at a time, so a ballot is just the predicate -- OR the vote of the
// imaginary other lanes of the wave (emul_other_lanes_vote, set by the test): with it set every "does any lane of the wave ...?"
// answers yes, i.e. each block is decoded the way it is inside a maximally mixed wave (BC7: three subsets, alpha, two index
// streams expanded for every block whatever its own mode)
static unsigned long long emul_other_lanes_vote = 0ull;
static inline unsigned long long __builtin_amdgcn_ballot_w64(bool p) { return (p ? 1ull : 0ull) | emul_other_lanes_vote; }
