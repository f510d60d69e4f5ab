// bptc_common.h -- what the BPTC (BC7) and BPTC_FLOAT (BC6H) decoders share: the bit-packed partition /
// anchor constants of bptc_tables.inc in __constant__ memory, the closed-form index weights, and packed
// 2 x u16 arithmetic helpers.  (Reference: bptc-tables.c:23-201, decompress-bptc.c:182-193.)
#pragma once
#include "dev_common.h"
#include "path_types.h"
#include "bptc_tables.inc"

namespace detexhip {

// (the spec-conformance switches kFlagSpec... carried in the decoders' `flags` are in path_types.h)

// [0..63] two-subset partitions, [64..127] three-subset partitions; 2-bit subset field per texel
__constant__ uint32_t kPartition2Bit[128] = { DETEXHIP_P2X_WORDS, DETEXHIP_P3_WORDS };
// anchor2 | anchor3_second << 4 | anchor3_third << 8
__constant__ uint16_t kAnchorWords[64] = { DETEXHIP_ANCHOR_WORDS };
// one-bit-per-texel form of the two-subset partitions (BC6H)
__constant__ uint16_t kPartition1Bit[64] = { DETEXHIP_P2_WORDS };

// the same constants for compile-time table derivation (decode_bptc.h builds its LDS tables from them)
constexpr uint32_t kPartition2BitCx[128] = { DETEXHIP_P2X_WORDS, DETEXHIP_P3_WORDS };
constexpr uint16_t kAnchorWordsCx[64] = { DETEXHIP_ANCHOR_WORDS };
constexpr uint16_t kPartition1BitCx[64] = { DETEXHIP_P2_WORDS };

// packed 2 x u16 arithmetic in one VGPR (v_pk_mad_u16 / v_pk_sub_u16): lanes wrap mod 2^16
typedef uint16_t pk16 __attribute__((vector_size(4)));
DH pk16 as_pk16(uint32_t v) { pk16 r; __builtin_memcpy(&r, &v, 4); return r; }
DH uint32_t from_pk16(pk16 v) { uint32_t r; __builtin_memcpy(&r, &v, 4); return r; }
DH uint32_t pk_mad_u16(uint32_t a, uint32_t b, uint32_t c) { return from_pk16(as_pk16(a) * as_pk16(b) + as_pk16(c)); }
DH uint32_t pk_sub_u16(uint32_t a, uint32_t b) { return from_pk16(as_pk16(a) - as_pk16(b)); }
// both 16-bit lanes of a multiplied by the HIGH half of b (VOP3P op_sel broadcast: no v_perm_b32 needed to
// duplicate a weight into both lanes), plus c
#if defined(__HIPCC__)
DH uint32_t pk_mad_u16_bhi(uint32_t a, uint32_t b, uint32_t c) {
	uint32_t r;
	asm("v_pk_mad_u16 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(r) : "v"(a), "v"(b), "v"(c));
	return r;
}
// ... by the LOW half of b
DH uint32_t pk_mad_u16_blo(uint32_t a, uint32_t b, uint32_t c) {
	uint32_t r;
	asm("v_pk_mad_u16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(b), "v"(c));
	return r;
}
// per-lane shift amounts (< 16) in the 16-bit lanes of s: v_pk_lshlrev_b16 / v_pk_lshrrev_b16 (vector shifts rather
// than inline asm, so that compile-time-constant amounts become inline operands instead of VGPRs)
typedef uint16_t pk_u16x2 __attribute__((ext_vector_type(2)));
DH pk_u16x2 to_u16x2(uint32_t v) { pk_u16x2 r; __builtin_memcpy(&r, &v, 4); return r; }
DH uint32_t of_u16x2(pk_u16x2 v) { uint32_t r; __builtin_memcpy(&r, &v, 4); return r; }
DH uint32_t pk_lshl_v(uint32_t s, uint32_t a) { return of_u16x2(to_u16x2(a) << to_u16x2(s)); }
DH uint32_t pk_lshr_v(uint32_t s, uint32_t a) { return of_u16x2(to_u16x2(a) >> to_u16x2(s)); }
#else
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
#endif

// Weight of an n-bit index as one multiply-add: t = (64*i + d/2) * ceil(65536/d) < 2^24 and the
// weight is byte 2 of t (bptc-tables.c aWeight2/3/4 in closed form, proven in tests/test_host_logic.py).
struct WeightMad { uint32_t mul, add; };
constexpr uint32_t bptc_weight_mul(uint32_t bits) { return bits == 2 ? 1398144u : (bits == 3 ? 599232u : 279680u); }
constexpr uint32_t bptc_weight_add(uint32_t bits) { return bits == 2 ? 21846u : (bits == 3 ? 28089u : 30590u); }
DH WeightMad weight_mad(uint32_t bits) {
	WeightMad w;
	w.mul = bits == 2 ? 1398144u : (bits == 3 ? 599232u : 279680u);
	w.add = bits == 2 ? 21846u : (bits == 3 ? 28089u : 30590u);
	return w;
}

// workgroup LDS copy used by BC6H (dev_common.h: prepare_tables): anchor_p1[i] = kAnchorWords[i] | kPartition1Bit[i] << 16
#if defined(__HIPCC__)
DH uint32_t *bptc_anchor_p1_lds() { __shared__ uint32_t t[64]; return t; }
DH void bptc_anchor_p1_prepare() {		// no barrier: the caller ends its own prepare() with one
	const uint32_t k = threadIdx.x;
	if (k < 64u) bptc_anchor_p1_lds()[k] = (uint32_t)kAnchorWords[k] | ((uint32_t)kPartition1Bit[k] << 16);
}
DH uint32_t bptc_anchor_p1(uint32_t i) { return bptc_anchor_p1_lds()[i]; }
#else
DH void bptc_anchor_p1_prepare() {}
DH uint32_t bptc_anchor_p1(uint32_t i) { return (uint32_t)kAnchorWords[i] | ((uint32_t)kPartition1Bit[i] << 16); }
#endif

}  // namespace detexhip
