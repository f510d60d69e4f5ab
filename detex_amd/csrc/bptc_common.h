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

// (packed 2 x u16 arithmetic -- pk_mad_u16, pk_mad_u16_bhi / _blo, pk_sub_u16, pk_lshl_v / pk_lshr_v: gfx950_prims.h)

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
DH uint32_t *bptc_anchor_p1_lds() { __shared__ uint32_t t[64]; return t; }
DH void bptc_anchor_p1_prepare() {		// no barrier: the caller ends its own prepare() with one
	const uint32_t k = threadIdx.x;
	if (k < 64u) bptc_anchor_p1_lds()[k] = (uint32_t)kAnchorWords[k] | ((uint32_t)kPartition1Bit[k] << 16);
}
DH uint32_t bptc_anchor_p1(uint32_t i) { return bptc_anchor_p1_lds()[i]; }

}  // namespace detexhip
