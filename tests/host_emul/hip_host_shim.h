// tests/host_emul/hip_host_shim.h -- TEST-ONLY emulation of what the device decoders take from the compiler and from
// detex_amd/csrc/gfx950_prims.h, so that tests/test_host_logic.py can compile detex_amd/csrc/decode_*.h with g++ and run the
// *device decode logic* on the CPU against the oracle in containers without a GPU: the handful of amdgcn builtins by their own
// names, the launch environment (one emulated lane: threadIdx, __shared__, barriers), and -- under gfx950_prims.h's include guard,
// so that the product header is skipped -- plain C++ statements of that header's inline-assembly / vector-extension primitives.
// Included BEFORE any product header.  Never compiled into libdetexhip.so; the -m gpu tests remain the parity tests proper.
#pragma once
#define DETEXHIP_GFX950_PRIMS_H 1	// detex_amd/csrc/gfx950_prims.h is replaced by this file
#include <stdint.h>
#include <algorithm>
#define DETEX_HD static inline
#define DH inline
#define DETEX_UMUL24(a, b) (((a) & 0xFFFFFFu) * ((b) & 0xFFFFFFu))
#define __constant__ static const
// the launch environment: ONE lane at a time.  `__shared__` objects are plain statics; the harness walks threadIdx.x over 0..255 around
// prepare_tables<Dec>() (so the workgroup's table copies are made by the device code itself) and decodes with threadIdx.x = 0
#define __shared__ static
struct EmulIdx { unsigned x, y, z; };
static EmulIdx threadIdx = { 0, 0, 0 };
static inline void __syncthreads() {}
#define __builtin_amdgcn_fence(order, scope) ((void)0)
static inline void __builtin_amdgcn_wave_barrier() {}
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

static inline int __builtin_amdgcn_readfirstlane(int v) { return v; }
// lanes of `mask` below the calling lane, which is lane 0
static inline uint32_t __builtin_amdgcn_mbcnt_lo(uint32_t, uint32_t base) { return base; }
static inline uint32_t __builtin_amdgcn_mbcnt_hi(uint32_t, uint32_t base) { return base; }

// ---- gfx950_prims.h in plain C++ --------------------------------------------------------------------------------------------------
namespace detexhip {
struct u32x2 { uint32_t x, y; };
struct u32x4 { uint32_t x, y, z, w; };
DH uint32_t opaque(uint32_t m) { return m; }
template <class... T> DH void pin_vgpr(T &...) {}
DH uint32_t bfi(uint32_t m, uint32_t a, uint32_t b) { return (a & m) | (b & ~m); }
DH uint32_t and_or(uint32_t a, uint32_t b, uint32_t c) { return (a & b) | c; }
DH uint32_t or3(uint32_t a, uint32_t b, uint32_t c) { return a | b | c; }
DH uint32_t and3(uint32_t a, uint32_t b, uint32_t c) { return a & b & c; }
template <int S> DH uint32_t high_half_shl(uint32_t v) { return (v >> 16) << S; }
DH uint32_t nonzero_as_one(uint32_t v) { return v ? 1u : 0u; }
DH int32_t med3_i32(int32_t x, int32_t lo, int32_t hi) { return x < lo ? lo : (x > hi ? hi : x); }
DH uint32_t pk_add16(uint32_t a, uint32_t b) { return ((a + b) & 0xFFFFu) | (((a >> 16) + (b >> 16)) << 16); }
DH uint32_t pk_sub16(uint32_t a, uint32_t b) { return ((a - b) & 0xFFFFu) | (((a >> 16) - (b >> 16)) << 16); }
DH uint32_t pk_ashr16(uint32_t a, int s) {
	return ((uint32_t)((int32_t)(int16_t)(a & 0xFFFFu) >> s) & 0xFFFFu) | ((uint32_t)((int32_t)(int16_t)(a >> 16) >> s) << 16);
}
DH uint32_t pk_mul16(uint32_t a, uint32_t b) { return ((a * b) & 0xFFFFu) | (((a >> 16) * (b >> 16)) << 16); }
DH uint32_t pk_max16(uint32_t a, uint32_t b) {
	const int32_t al = (int16_t)(a & 0xFFFFu), ah = (int16_t)(a >> 16), bl = (int16_t)(b & 0xFFFFu), bh = (int16_t)(b >> 16);
	return ((uint32_t)std::max(al, bl) & 0xFFFFu) | ((uint32_t)std::max(ah, bh) << 16);
}
DH uint32_t pk_min16(uint32_t a, uint32_t b) {
	const int32_t al = (int16_t)(a & 0xFFFFu), ah = (int16_t)(a >> 16), bl = (int16_t)(b & 0xFFFFu), bh = (int16_t)(b >> 16);
	return ((uint32_t)std::min(al, bl) & 0xFFFFu) | ((uint32_t)std::min(ah, bh) << 16);
}
DH uint32_t pk_lshl16(uint32_t a, int s) { return ((a << s) & 0xFFFFu) | ((((a >> 16) << s) & 0xFFFFu) << 16); }
DH uint32_t pk_lshr16(uint32_t a, int s) { return ((a & 0xFFFFu) >> s) | (((a >> 16) >> s) << 16); }
DH uint32_t sat_u8_pk16(uint32_t a) {
	const int32_t lo = (int16_t)(a & 0xFFFFu), hi = (int16_t)(a >> 16);
	return (uint32_t)std::min(std::max(lo, 0), 255) | ((uint32_t)std::min(std::max(hi, 0), 255) << 8) | 0xDEAD0000u;	// poison the unspecified half
}
DH uint32_t pk_mad_u16(uint32_t a, uint32_t b, uint32_t c) {
	return (((a & 0xFFFFu) * (b & 0xFFFFu) + (c & 0xFFFFu)) & 0xFFFFu) | ((((a >> 16) * (b >> 16) + (c >> 16)) & 0xFFFFu) << 16);
}
DH uint32_t pk_sub_u16(uint32_t a, uint32_t b) { return pk_sub16(a, b); }
DH uint32_t pk_mad_u16_bhi(uint32_t a, uint32_t b, uint32_t c) {
	const uint32_t w = b >> 16;
	return (((a & 0xFFFFu) * w + (c & 0xFFFFu)) & 0xFFFFu) | ((((a >> 16) * w + (c >> 16)) & 0xFFFFu) << 16);
}
DH uint32_t pk_mad_u16_blo(uint32_t a, uint32_t b, uint32_t c) {
	const uint32_t w = b & 0xFFFFu;
	return (((a & 0xFFFFu) * w + (c & 0xFFFFu)) & 0xFFFFu) | ((((a >> 16) * w + (c >> 16)) & 0xFFFFu) << 16);
}
DH uint32_t pk_lshl_v(uint32_t s, uint32_t a) {
	return (((a & 0xFFFFu) << (s & 15u)) & 0xFFFFu) | ((((a >> 16) << ((s >> 16) & 15u)) & 0xFFFFu) << 16);
}
DH uint32_t pk_lshr_v(uint32_t s, uint32_t a) { return ((a & 0xFFFFu) >> (s & 15u)) | (((a >> 16) >> ((s >> 16) & 15u)) << 16); }
template <int POLICY, int STAGE> DH void stage_priority() {}
template <int N> DH void sleep_cycles64() {}
}  // namespace detexhip
