// ab/decode_bptc_r01.h -- the ROUND-1 BC7 decoders, kept only for A/B measurements against decode_bptc.h
// (measurement build only, make lib-ab; never part of the product library).  Original header follows.
//
// BPTC (BC7), all eight modes, one lane per block, gfx950.
//
// The reference decodes with an 8-way mode switch, a bit-at-a-time 128-bit reader and per-texel
// table lookups (decompress-bptc.c:354-512).  A wavefront of 64 independent blocks would execute
// every taken mode path serially, so this decoder is a single DIVERGENCE-FREE data-driven path:
//   * everything about a mode that does not depend on the block (field positions and widths, expansion
//     shifts, index widths, weight constants) is a Bc7Layout record derived at compile time from the
//     eight mode descriptors and fetched per lane from a workgroup LDS copy;
//   * the block's bits and the (up to three) subsets' blend operands live in per-lane LDS rows
//     (dev_common.h: LaneRows), so a field is two dwords + v_alignbit_b32 and a texel's subset is one
//     ds_read_b128 instead of register-select chains;
//   * endpoints are expanded to 8 bits with SWAR byte math, the two index streams are read through two
//     32-bit windows each (one shift per texel), a weight is one v_mad_u32_u24, and a texel is blended two
//     channels per v_pk_mad_u16.
// Partition / anchor tables are the bit-packed words of bptc_tables.inc (__constant__, LDS copy per workgroup).
//
// Reference quirk reproduced: in mode 6 the second P-bit (block bit 64) reads 0.
#pragma once
#include "dev_common.h"
#include "decode_s3tc_rgtc.h"
#include "bptc_tables.inc"

namespace detexhip {
namespace r01 {

// [0..63] two-subset partitions, [64..127] three-subset partitions; 2-bit subset field per texel
__constant__ uint32_t kPartition2Bit[128] = { DETEXHIP_P2X_WORDS, DETEXHIP_P3_WORDS };
// anchor2 | anchor3_second << 4 | anchor3_third << 8
__constant__ uint16_t kAnchorWords[64] = { DETEXHIP_ANCHOR_WORDS };
// one-bit-per-texel form of the two-subset partitions (BC6H)
__constant__ uint16_t kPartition1Bit[64] = { DETEXHIP_P2_WORDS };

// per-mode layout (decompress-bptc.c:24-43 comment table, :45-71, :134, :195-225, :265-267)
constexpr uint32_t bc7_desc(uint32_t ns, uint32_t pb, uint32_t rb, uint32_t isb, uint32_t cb, uint32_t ab, uint32_t epb,
		uint32_t spb, uint32_t ib, uint32_t ib2) {
	return ns | (pb << 2) | (rb << 5) | (isb << 7) | (cb << 8) | (ab << 11) | (epb << 15) | (spb << 16) | (ib << 17) | (ib2 << 20);
}

// Everything about a mode that does not depend on the block's contents, precomputed (decompress-bptc.c:24-43,
// :45-71, :134-180): bit positions of the fields, their widths, the shift amounts of the 8-bit expansion, the
// index widths and their weight constants.  For a per-lane mode the decoder fetches this record from LDS with a
// few ds_read_b128 instead of deriving it from the descriptor word with ~50 VALU instructions per block.
struct alignas(16) Bc7Layout {
	uint32_t pos_part, pb, pos_rot, rb, pos_isel, isb;		// header fields
	uint32_t pos_r, pos_g, pos_b, pos_a, pos_p, pos_idx, pos_idx2;	// channel words, P-bits, index streams
	uint32_t cb, ab, off[6], offa[4];				// field widths, e*cb, e*ab
	uint32_t has_p, epb, p_word_mask, p_double;			// p_word_mask: quirk A-2 (mode 6 keeps one P-bit); p_double: shared P-bit
	uint32_t c_up, c_down, c_keep, a_up, a_down;
	uint32_t alpha_keep, alpha_set;					// modes 0-3: alpha = 255
	uint32_t ns, part_base, ib, ib2;				// part_base: +64 selects the three-subset table
	uint32_t w_mul, w_add, w2_mul, w2_add;				// weight = byte 2 of index * mul + add, for ib and ib2
};
constexpr uint32_t bc7_weight_mul(uint32_t bits) { return bits == 2 ? 1398144u : (bits == 3 ? 599232u : 279680u); }
constexpr uint32_t bc7_weight_add(uint32_t bits) { return bits == 2 ? 21846u : (bits == 3 ? 28089u : 30590u); }
constexpr Bc7Layout bc7_layout(uint32_t mode, uint32_t desc) {
	const uint32_t ns = desc & 3u, pb = (desc >> 2) & 7u, rb = (desc >> 5) & 3u, isb = (desc >> 7) & 1u;
	const uint32_t cb = (desc >> 8) & 7u, ab = (desc >> 11) & 15u, epb = (desc >> 15) & 1u, spb = (desc >> 16) & 1u;
	const uint32_t ib = (desc >> 17) & 7u, ib2 = (desc >> 20) & 3u;
	Bc7Layout L = {};
	L.pos_part = mode + 1u; L.pb = pb;
	L.pos_rot = L.pos_part + pb; L.rb = rb;
	L.pos_isel = L.pos_rot + rb; L.isb = isb;
	const uint32_t chan = 2u * ns * cb;
	L.pos_r = L.pos_isel + isb; L.pos_g = L.pos_r + chan; L.pos_b = L.pos_g + chan; L.pos_a = L.pos_b + chan;
	L.pos_p = L.pos_a + 2u * ns * ab;
	L.pos_idx = L.pos_p + epb * 2u * ns + spb * ns;
	L.pos_idx2 = L.pos_idx + 16u * ib - ns;
	L.cb = cb; L.ab = ab;
	for (uint32_t e = 0; e < 6u; e++) L.off[e] = e * cb;
	for (uint32_t e = 0; e < 4u; e++) L.offa[e] = e * ab;
	L.has_p = epb | spb; L.epb = epb;
	L.p_word_mask = mode == 6u ? 1u : 0xFFFFFFFFu;
	L.p_double = (spb && !epb) ? 0xFFFFFFFFu : 0u;
	const uint32_t cprec = cb + L.has_p, aprec = ab + epb;
	L.c_up = 8u - cprec; L.c_down = (2u * cprec - 8u) & 31u;
	L.c_keep = 0x010101u * ((1u << L.c_up) - 1u);
	L.a_up = (8u - aprec) & 31u; L.a_down = (2u * aprec - 8u) & 31u;
	L.alpha_keep = mode < 4u ? 0u : 0xFFFFFFFFu; L.alpha_set = mode < 4u ? 0xFF000000u : 0u;
	L.ns = ns; L.part_base = ns == 3u ? 64u : 0u; L.ib = ib; L.ib2 = ib2;
	L.w_mul = bc7_weight_mul(ib); L.w_add = bc7_weight_add(ib);
	L.w2_mul = bc7_weight_mul(ib2); L.w2_add = bc7_weight_add(ib2);
	return L;
}
constexpr uint32_t kBc7Desc[8] = {
	//       subsets part rot isel colour alpha endpoint-P shared-P index index2
	bc7_desc(3, 4, 0, 0, 4, 0, 1, 0, 3, 0), bc7_desc(2, 6, 0, 0, 6, 0, 0, 1, 3, 0), bc7_desc(3, 6, 0, 0, 5, 0, 0, 0, 2, 0),
	bc7_desc(2, 6, 0, 0, 7, 0, 1, 0, 2, 0), bc7_desc(1, 0, 2, 1, 5, 6, 0, 0, 2, 3), bc7_desc(1, 0, 2, 0, 7, 8, 0, 0, 2, 2),
	bc7_desc(1, 0, 0, 0, 7, 7, 1, 0, 4, 0), bc7_desc(2, 6, 0, 0, 5, 5, 1, 0, 2, 0) };
__constant__ Bc7Layout kBc7Layouts[8] = {
	bc7_layout(0, kBc7Desc[0]), bc7_layout(1, kBc7Desc[1]), bc7_layout(2, kBc7Desc[2]), bc7_layout(3, kBc7Desc[3]),
	bc7_layout(4, kBc7Desc[4]), bc7_layout(5, kBc7Desc[5]), bc7_layout(6, kBc7Desc[6]), bc7_layout(7, kBc7Desc[7]),
};
static_assert(sizeof(Bc7Layout) % 16 == 0, "record is fetched with 16-byte LDS reads");

// workgroup copies in LDS (dev_common.h: prepare_tables).  anchor_p1[i] = kAnchorWords[i] | kPartition1Bit[i] << 16
struct BptcTables { uint32_t part2[128]; uint32_t anchor_p1[64]; Bc7Layout layout[8]; };
#if defined(__HIPCC__)
DH BptcTables &bptc_tables() { __shared__ BptcTables t; return t; }
DH void bptc_prepare(bool with_layouts) {
	BptcTables &t = bptc_tables();
	const uint32_t k = threadIdx.x;
	if (k < 128u) t.part2[k] = kPartition2Bit[k];
	else if (k < 192u) t.anchor_p1[k - 128u] = (uint32_t)kAnchorWords[k - 128u] | ((uint32_t)kPartition1Bit[k - 128u] << 16);
	if (with_layouts) {
		constexpr uint32_t kWords = 8u * sizeof(Bc7Layout) / 4u;
		const uint32_t *src = reinterpret_cast<const uint32_t *>(kBc7Layouts);
		uint32_t *dst = reinterpret_cast<uint32_t *>(t.layout);
		for (uint32_t w = k; w < kWords; w += 256u) dst[w] = src[w];
	}
	__syncthreads();
}
DH uint32_t bptc_part2(uint32_t i) { return bptc_tables().part2[i]; }
DH uint32_t bptc_anchor_p1(uint32_t i) { return bptc_tables().anchor_p1[i]; }
DH const Bc7Layout &bptc_layout(uint32_t m) {
	// one opaque byte offset per lane: the record's fields then come as immediate offsets of a few wide LDS
	// reads (left to itself the compiler rebuilds mode * sizeof + field offset with a v_mad per field)
	uint32_t off = m * (uint32_t)sizeof(Bc7Layout);
	asm("" : "+v"(off));
	return *reinterpret_cast<const Bc7Layout *>(reinterpret_cast<const char *>(bptc_tables().layout) + off);
}
#else
DH void bptc_prepare(bool) {}
DH uint32_t bptc_part2(uint32_t i) { return kPartition2Bit[i]; }
DH uint32_t bptc_anchor_p1(uint32_t i) { return (uint32_t)kAnchorWords[i] | ((uint32_t)kPartition1Bit[i] << 16); }
DH const Bc7Layout &bptc_layout(uint32_t m) { return kBc7Layouts[m]; }
#endif

// weight(index) for a per-lane index width: (64*i + d/2) / d as multiply-shift (dev_common.h)
struct WeightParams { uint32_t half, magic; };
DH WeightParams weight_params(uint32_t bits) {
	WeightParams w;
	w.half = ((1u << bits) - 1u) >> 1;
	w.magic = bits == 2 ? 21846u : (bits == 3 ? 9363u : 4370u);
	return w;
}
DH uint32_t weight_of(uint32_t index, const WeightParams &w) { return DETEX_UMUL24((index << 6) + w.half, w.magic) >> 16; }

// packed 2 x u16 arithmetic in one VGPR (v_pk_mad_u16 / v_pk_sub_u16): lanes wrap mod 2^16
typedef uint16_t pk16 __attribute__((vector_size(4)));
DH pk16 as_pk16(uint32_t v) { pk16 r; __builtin_memcpy(&r, &v, 4); return r; }
DH uint32_t from_pk16(pk16 v) { uint32_t r; __builtin_memcpy(&r, &v, 4); return r; }
DH uint32_t pk_mad_u16(uint32_t a, uint32_t b, uint32_t c) { return from_pk16(as_pk16(a) * as_pk16(b) + as_pk16(c)); }
DH uint32_t pk_sub_u16(uint32_t a, uint32_t b) { return from_pk16(as_pk16(a) - as_pk16(b)); }

// One subset's endpoint pair prepared for blending.  With e0, e1 in 0..255 and w in 0..64 the
// reference's ((64-w)*e0 + w*e1 + 32) >> 6 (decompress-bptc.c:182-193) equals the high byte of
//     256*e0 + 128 + 4*w*(e1 - e0)          (range 128 .. 65408: fits a 16-bit lane, exact mod 2^16)
// so a texel is two v_pk_mad_u16 (R,G and B,A lanes) and one v_perm_b32 that gathers the four
// high bytes (and applies the mode 4/5 channel rotation for free).
struct BlendPair { uint32_t base_rg, base_ba, diff_rg, diff_ba; };
DH BlendPair blend_pair(uint32_t e0, uint32_t e1) {
	const uint32_t rg0 = perm(0u, e0, 0x0C010C00u), ba0 = perm(0u, e0, 0x0C030C02u);	// zero-extended channel pairs
	const uint32_t rg1 = perm(0u, e1, 0x0C010C00u), ba1 = perm(0u, e1, 0x0C030C02u);
	BlendPair p;
	p.base_rg = (rg0 << 8) | 0x00800080u;
	p.base_ba = (ba0 << 8) | 0x00800080u;
	p.diff_rg = pk_sub_u16(rg1, rg0);
	p.diff_ba = pk_sub_u16(ba1, ba0);
	return p;
}

// Weight of an n-bit index as one multiply-add: t = (64*i + d/2) * ceil(65536/d) < 2^24 and the
// weight is byte 2 of t (bptc-tables.c aWeight2/3/4 in closed form, proven in tests/test_host_logic.py).
struct WeightMad { uint32_t mul, add; };
DH WeightMad weight_mad(uint32_t bits) {
	WeightMad w;
	w.mul = bits == 2 ? 1398144u : (bits == 3 ? 599232u : 279680u);
	w.add = bits == 2 ? 21846u : (bits == 3 ? 28089u : 30590u);
	return w;
}

// FIXED_MODE >= 0 instantiates the decoder for one mode with every layout parameter a compile-time
// constant (used by wave-uniform fast paths); FIXED_MODE = -1 is the per-lane data-driven form.
// IMPL selects the texel stage (A/B, profiles/AB_RECORD.md):
//   0  subset endpoints picked per texel with v_bfi_b32 chains (registers only)
//   1  subset endpoints kept in per-lane LDS rows, one ds_read_b128 per texel; colour / alpha index
//      streams (instead of primary / secondary + per-texel swaps); weights by one v_mad_u32_u24
//   2  as 1, and block fields are fetched from an LDS copy of the block (two dwords + v_alignbit)
template <int FIXED_MODE, int IMPL = 2> struct DecBPTCMode {
	static constexpr int kBlockBytes = 16, kPixelBytes = 4;
	static DH void prepare() { bptc_prepare(true); }

	template <bool CHECKED> static DH bool decode(uint4 blk, uint32_t mode_mask, uint32_t flags, uint32_t (&d)[16]) {
		const uint32_t low = blk.x & 0xFFu;
		if (low == 0) return false;				// reserved (decompress-bptc.c:229-237, 361)
		const uint32_t mode = FIXED_MODE >= 0 ? (uint32_t)FIXED_MODE : (uint32_t)__builtin_ctz(low);
		if (CHECKED) {						// :363-369
			if (!(mode_mask & (1u << mode))) return false;
			if (mode >= 4 && (flags & kFlagOpaqueOnly)) return false;
			if (mode < 4 && (flags & kFlagNonOpaqueOnly)) return false;
		}
		// the mode's layout record: compile-time for a fixed mode, one LDS record for a per-lane mode
		constexpr Bc7Layout kFixed = bc7_layout(FIXED_MODE >= 0 ? FIXED_MODE : 0, kBc7Desc[FIXED_MODE >= 0 ? FIXED_MODE : 0]);
		const Bc7Layout &L = FIXED_MODE >= 0 ? kFixed : bptc_layout(mode);
		const uint32_t ns = L.ns, cb = L.cb, ab = L.ab, epb = L.epb, has_p = L.has_p, ib = L.ib, ib2 = L.ib2;
		const Bits128 b = { { blk.x, blk.y, blk.z, blk.w } };
		// IMPL 2: the block's dwords as per-lane LDS rows (rows 4, 5 read as zero: bits beyond 127)
		LaneRows<uint32_t, IMPL == 2 ? 6 : 1, 71> rows;
		if (IMPL == 2) {
			rows.put(0, blk.x); rows.put(1, blk.y); rows.put(2, blk.z); rows.put(3, blk.w); rows.put(4, 0u); rows.put(5, 0u);
		}
		auto field32 = [&](uint32_t at) -> uint32_t {
			if (IMPL != 2) return extract32(b, at);
			const uint32_t k = at >> 5;
			return __builtin_amdgcn_alignbit(rows.get(k + 1u), rows.get(k), at);
		};

		// header fields all lie in the first 14 bits
		const uint32_t part = ubfe(blk.x, L.pos_part, L.pb);
		const uint32_t rot = ubfe(blk.x, L.pos_rot, L.rb);
		const uint32_t isel = ubfe(blk.x, L.pos_isel, L.isb);

		// endpoint fields: all R, then all G, then all B, then all A (each 2*ns values) -- :74-132
		const uint32_t wr = field32(L.pos_r), wg = field32(L.pos_g), wb = field32(L.pos_b), wa = field32(L.pos_a);
		uint32_t pw = field32(L.pos_p) & L.p_word_mask;	// P-bits; QUIRK A-2: mode 6 keeps only the first (:142-146)
		const uint32_t pos = L.pos_idx;

		// expand to 8 bits: append the P-bit, shift the MSB to bit 7, replicate the top bits (:136-180)
		const uint32_t c_up = L.c_up, c_down = L.c_down, c_keep = L.c_keep, a_up = L.a_up, a_down = L.a_down;
		// P-bit of endpoint e at bit e: per-endpoint P-bits as stored, a shared P-bit (mode 1) doubled
		const uint32_t pw_e = bfi(L.p_double, ((pw & 1u) * 3u) | ((pw & 2u) * 6u), pw) & (0u - has_p);
		// subsets actually present in this wave (wave-uniform): endpoints of absent subsets are not expanded.  The
		// uniform-random stream always has three-subset blocks (modes 0, 2) in every wave; encoder output mostly does not.
		const uint32_t wave_subsets = FIXED_MODE >= 0 ? L.ns : (IMPL == 0 ? 3u
			: (__builtin_amdgcn_ballot_w64(ns == 3u) ? 3u : (__builtin_amdgcn_ballot_w64(ns == 2u) ? 2u : 1u)));
		// likewise the alpha fields: modes 0-3 are opaque
		const bool wave_alpha = FIXED_MODE >= 0 ? FIXED_MODE >= 4 : (IMPL == 0 || __builtin_amdgcn_ballot_w64(mode >= 4u) != 0);
		uint32_t ep[6] = {};
#pragma unroll
		for (int e = 0; e < 6; e++) {
			if ((uint32_t)(e >> 1) >= wave_subsets) break;
			const uint32_t off = L.off[e];
			// (three-input logic goes through v_bitop3_b32 -- dev_common.h: and_or / or3 -- at 2.5 cycles instead of 4.4)
			uint32_t x = or3(ubfe(wr, off, cb), ubfe(wg, off, cb) << 8, ubfe(wb, off, cb) << 16);
			const uint32_t p = ubfe(pw_e, e, 1);
			x = and_or(0u - p, 0x010101u, x << has_p);
			x = and_or(x >> c_down, c_keep, x << c_up);		// each byte holds cprec bits: nothing crosses a byte
			uint32_t a = 0xFF000000u;				// :176-179; modes with alpha have at most two subsets
			if (e < 4 && wave_alpha) {
				a = ubfe(wa, L.offa[e], ab);
				a = and_or(p, epb, a << epb);
				a = ((a << a_up) | (a >> a_down)) << 24;	// bits above the byte fall off the top
				a = and_or(a, L.alpha_keep, L.alpha_set);	// modes 0-3 are opaque
			}
			ep[e] = x | a;
		}

		// partition + anchors (:391-400)
		const uint32_t pword = ns == 1u ? 0u : bptc_part2(part + L.part_base);
		const uint32_t an = bptc_anchor_p1(part);
		const uint32_t a1 = ns == 2u ? (an & 0xFu) : ubfe(an, 4, 4), a2 = ubfe(an, 8, 4);
		const uint32_t amask = 1u | (ns >= 2u ? (1u << a1) : 0u) | (ns == 3u ? (1u << a2) : 0u);
		// gather the high bytes of the four 16-bit sums; rotation swaps A with R/G/B (:497-508)
		const uint32_t gather = rot == 0u ? 0x07050301u : (rot == 1u ? 0x01050307u : (rot == 2u ? 0x03050701u : 0x05070301u));
		// index streams, LSB-first: primary (16*ib - ns bits), then, for modes 4/5, the secondary one
		// (16*ib2 - 1 bits) -- :401-480
		const uint32_t pos2 = L.pos_idx2;
		const bool two = ib2 != 0u, swap = two && isel != 0u;
		const bool any_two = FIXED_MODE >= 0 ? (FIXED_MODE == 4 || FIXED_MODE == 5) : (__builtin_amdgcn_ballot_w64(two) != 0);

		if (IMPL == 0) {
			const BlendPair s0 = blend_pair(ep[0], ep[1]), s1 = blend_pair(ep[2], ep[3]), s2 = blend_pair(ep[4], ep[5]);
			uint32_t plo = extract32(b, pos), phi = extract32(b, pos + 32u);
			uint32_t slo = 0, shi = 0;
			if (any_two) { slo = extract32(b, pos2); shi = extract32(b, pos2 + 32u); }
			// colour uses the secondary indices when the index-selection bit is set (:374-375, 452-480)
			const WeightParams wp_a = weight_params(ib), wp_b = weight_params(two ? ib2 : ib);
#pragma unroll
			for (int i = 0; i < 16; i++) {
				const uint32_t width = ib - ((amask >> i) & 1u);	// anchor texels store one bit less
				const uint32_t w_a = weight_of(ubfe(plo, 0, width), wp_a);
				plo = __builtin_amdgcn_alignbit(phi, plo, width);
				phi >>= width;
				uint32_t w_rg = DETEX_UMUL24(w_a, 0x00040004u), w_ba = w_rg;	// 4*w in both 16-bit lanes
				if (any_two) {
					const uint32_t width2 = (ib2 - (i == 0 ? 1u : 0u)) & 31u;
					const uint32_t w_b = two ? weight_of(ubfe(slo, 0, width2), wp_b) : w_a;
					slo = __builtin_amdgcn_alignbit(shi, slo, width2);
					shi >>= width2;
					const uint32_t wc = swap ? w_b : w_a, wal = swap ? w_a : w_b;
					w_rg = DETEX_UMUL24(wc, 0x00040004u);
					w_ba = (wc | (wal << 16)) << 2;
				}
				const uint32_t m1 = bit_to_mask(pword, 2 * i), m2 = bit_to_mask(pword, 2 * i + 1);
				const uint32_t base_rg = bfi(m2, s2.base_rg, bfi(m1, s1.base_rg, s0.base_rg));
				const uint32_t base_ba = bfi(m2, s2.base_ba, bfi(m1, s1.base_ba, s0.base_ba));
				const uint32_t diff_rg = bfi(m2, s2.diff_rg, bfi(m1, s1.diff_rg, s0.diff_rg));
				const uint32_t diff_ba = bfi(m2, s2.diff_ba, bfi(m1, s1.diff_ba, s0.diff_ba));
				d[i] = perm(pk_mad_u16(diff_ba, w_ba, base_ba), pk_mad_u16(diff_rg, w_rg, base_rg), gather);
			}
			return true;
		}

		// ---- IMPL 1 / 2 ----
		// per-subset blend operands {base_rg, base_ba, 4*diff_rg, 4*diff_ba} as LDS rows of this lane
		LaneRows<uint4, 3, 72> subsets;
#pragma unroll
		for (int s = 0; s < 3; s++) {
			if ((uint32_t)s >= wave_subsets) break;
			const uint32_t e0 = ep[2 * s], e1 = ep[2 * s + 1];
			const uint32_t rg0 = perm(0u, e0, 0x0C010C00u), ba0 = perm(0u, e0, 0x0C030C02u);
			const uint32_t rg1 = perm(0u, e1, 0x0C010C00u), ba1 = perm(0u, e1, 0x0C030C02u);
			uint4 row;
			row.x = (rg0 << 8) | 0x00800080u;
			row.y = (ba0 << 8) | 0x00800080u;
			row.z = pk_sub_u16(rg1 << 2, rg0 << 2);		// 4*(e1-e0) per 16-bit lane (mod 2^16)
			row.w = pk_sub_u16(ba1 << 2, ba0 << 2);
			subsets.put(s, row);
		}
		// colour stream C and alpha stream A: C is the primary stream unless the index-selection bit
		// swaps them (:374-375, 452-480); only modes 4/5 have an A stream (one subset, anchor = texel 0)
		const uint32_t sm = cond_to_mask(swap);
		const uint32_t ibc = bfi(sm, ib2, ib), iba = bfi(sm, ib, ib2);
		const uint32_t pos_c = bfi(sm, pos2, pos), pos_a = bfi(sm, pos, pos2);
		// Each stream is read through two 32-bit windows: texels 0-7 consume at most 31 bits (texel 0 is
		// always an anchor), texels 8-15 start where they ended -- so advancing a stream is one plain shift.
		const uint32_t half_c = 8u * ibc - (uint32_t)__builtin_popcount(amask & 0xFFu);
		const uint32_t c_lo = field32(pos_c), c_hi = field32(pos_c + half_c);
		const WeightMad wm_c = { bfi(sm, L.w2_mul, L.w_mul), bfi(sm, L.w2_add, L.w_add) };
		// one wave-uniform branch around two straight-line loops (a per-texel branch costs more than it skips)
		if (any_two) {
			const uint32_t half_a = (8u * iba - 1u) & 31u;
			const uint32_t a_lo = field32(pos_a), a_hi = field32(pos_a + half_a);
			const WeightMad wm_a = { bfi(sm, L.w_mul, L.w2_mul), bfi(sm, L.w_add, L.w2_add) };
			const uint32_t sel_ba = two ? 0x0C060C02u : 0x0C020C02u;	// alpha weight from the A stream, or the colour weight again
			const uint32_t width_a = iba & 31u;
			uint32_t cw = c_lo, aw = a_lo;
#pragma unroll
			for (int i = 0; i < 16; i++) {
				if (i == 8) { cw = c_hi; aw = a_hi; }
				const uint32_t width = ibc - ((amask >> i) & 1u);	// anchor texels store one bit less
				const uint32_t tc = DETEX_UMUL24(ubfe(cw, 0, width), wm_c.mul) + wm_c.add;	// weight = byte 2
				cw >>= width;
				const uint32_t wa = i == 0 ? ((iba - 1u) & 31u) : width_a;
				const uint32_t ta = DETEX_UMUL24(ubfe(aw, 0, wa), wm_a.mul) + wm_a.add;
				aw >>= wa;
				const uint4 s = subsets.get(ubfe(pword, 2 * i, 2));
				d[i] = perm(pk_mad_u16(s.w, perm(ta, tc, sel_ba), s.y), pk_mad_u16(s.z, perm(tc, tc, 0x0C020C02u), s.x), gather);
			}
		} else {
			uint32_t cw = c_lo;
#pragma unroll
			for (int i = 0; i < 16; i++) {
				if (i == 8) cw = c_hi;
				const uint32_t width = ibc - ((amask >> i) & 1u);
				const uint32_t tc = DETEX_UMUL24(ubfe(cw, 0, width), wm_c.mul) + wm_c.add;
				cw >>= width;
				const uint32_t w = perm(tc, tc, 0x0C020C02u);		// weight in both 16-bit lanes
				const uint4 s = subsets.get(ubfe(pword, 2 * i, 2));
				d[i] = perm(pk_mad_u16(s.w, w, s.y), pk_mad_u16(s.z, w, s.x), gather);
			}
		}
		return true;
	}
};
using DecBPTCLdsFields = DecBPTCMode<-1>;
using DecBPTCRegisterSelect = DecBPTCMode<-1, 0>;
using DecBPTCRegisterFields = DecBPTCMode<-1, 1>;

}  // namespace r01
}  // namespace detexhip
