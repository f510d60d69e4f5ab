// decode_etc_eac.h -- ETC1 / ETC2 / ETC2 punchthrough / ETC2+EAC and EAC R11/RG11 (+-signed), gfx950.
//
// One lane = one block.  The colour codec is restructured around per-sub-block 4-entry
// palettes of packed RGBA dwords: individual/differential blocks, T blocks and H blocks all
// reduce to "texel = palette[sub-block][2-bit selector]", so they share one branch-free texel
// loop (v_bfe_i32 lane masks + v_bitop3_b32 selects); T and H are first reduced to two 12-bit colours
// and a distance and share their arithmetic; only planar blocks take a separate path (row / column
// increments).  Colour arithmetic runs two signed 16-bit lanes per VGPR with the saturating pack
// v_sat_pk_u8_i16 as the 0..255 clamp.  The selector planes are big-endian and texels are numbered
// column-major.  ETC tables are literal operands (v_perm_b32 pools); the EAC modifier
// table lives in __constant__ memory with a workgroup LDS copy (north_star).
#pragma once
#include "dev_common.h"
#include "decode_s3tc_rgtc.h"
#include "bptc_common.h"

namespace detexhip {

enum : uint32_t {	// detex.h:383-395
	kMaskEtcIndividual = 0x1, kMaskEtcDifferential = 0x2, kMaskEtcT = 0x4, kMaskEtcH = 0x8, kMaskEtcPlanar = 0x10,
};

// ETC1 intensity modifiers {small, large} per codeword, one byte each (decompress-etc.c:25-34
// holds them as signed rows {a, b, -a, -b}); looked up with v_perm_b32.
#define ETC_SMALL_LO 0x0D090502u	/* 2, 5, 9, 13 */
#define ETC_SMALL_HI 0x2F211812u	/* 18, 24, 33, 47 */
#define ETC_LARGE_LO 0x2A1D1108u	/* 8, 17, 29, 42 */
#define ETC_LARGE_HI 0xB76A503Cu	/* 60, 80, 106, 183 */
// ETC2 T/H distances (decompress-etc.c:200): 3, 6, 11, 16, 23, 32, 41, 64
#define ETC_DIST_LO 0x100B0603u
#define ETC_DIST_HI 0x40292017u

// EAC modifier tables (decompress-eac.c:21-38).  Row t is {-m0,-m1,-m2,-m3, m0-1,m1-1,m2-1,m3-1};
// only the four magnitudes are stored, one nibble each, m0 in the low nibble.
__constant__ uint16_t kEacMagnitudes[16] = {
	0xF963, 0xDA73, 0xD852, 0xD642, 0xC863, 0xB973, 0xB874, 0xB853,
	0xA862, 0xA852, 0xA842, 0xA752, 0xA743, 0xA321, 0x9864, 0x9753,
};

// workgroup copy of the table in LDS (dev_common.h: prepare_tables)
DH uint16_t *eac_rows_lds() { __shared__ uint16_t rows[16]; return rows; }
DH void eac_prepare() {
	if (threadIdx.x < 16u) eac_rows_lds()[threadIdx.x] = kEacMagnitudes[threadIdx.x];
	__syncthreads();
}
DH uint32_t eac_row(uint32_t t) { return eac_rows_lds()[t]; }


// Arithmetic of this file runs two signed 16-bit lanes per VGPR (dev_common.h: pk_add16 / pk_sub16 /
// pk_ashr16 / sat_u8_pk16): a colour is held as RB = (R, B) lanes plus G, where the G's of two
// colours (or of two neighbouring texels) share one register.  sat_u8_pk16 is the 0..255 clamp of both
// lanes and leaves them in bytes 0, 1, from where one v_perm_b32 assembles an RGBA texel (selector
// byte 0x0D yields the opaque alpha 0xFF, 0x0C yields 0).
template <uint32_t ALPHA> struct EtcGather {
	static constexpr uint32_t kA = ALPHA ? 0x0D000000u : 0x0C000000u;
	// rb: R in byte 0, B in byte 1 (saturated form); g: G in byte GB of g
	template <int GB> static DH uint32_t sat(uint32_t g, uint32_t rb) { return perm(g, rb, kA | 0x00010000u | ((4u + GB) << 8)); }
	// rb: unsaturated lanes (R in byte 0, B in byte 2); g lanes likewise (G in byte 0 or 2)
	template <int GB> static DH uint32_t raw(uint32_t g, uint32_t rb) { return perm(g, rb, kA | 0x00020000u | ((4u + GB) << 8)); }
};

// texel loop shared by every non-planar mode.  pal0 = sub-block 0 palette, pal1 = sub-block 1;
// flip selects the 2x4 / 4x2 split (decompress-etc.c:143-178).  Texel p is column-major
// (x = p>>2, y = p&3) and lands at row-major index y*4+x (decompress-etc.c:83).
DH void etc_texels(uint32_t word, bool flip, const uint32_t (&pal0)[4], const uint32_t (&pal1)[4], uint32_t (&d)[16]) {
	uint32_t palb[4], palc[4];	// quadrant (x<2,y>=2) and (x>=2,y<2)
	const uint32_t fm = cond_to_mask(flip);
#pragma unroll
	for (int k = 0; k < 4; k++) {
		palb[k] = bfi(fm, pal1[k], pal0[k]);
		palc[k] = bfi(fm, pal0[k], pal1[k]);
	}
#pragma unroll
	for (int p = 0; p < 16; p++) {
		const int x = p >> 2, y = p & 3;
		const uint32_t m_lo = bit_to_mask(word, p), m_hi = bit_to_mask(word, 16 + p);
		const uint32_t(&q)[4] = x < 2 ? (y < 2 ? pal0 : palb) : (y < 2 ? palc : pal1);
		d[y * 4 + x] = select4(m_lo, m_hi, q[0], q[1], q[2], q[3]);
	}
}

// ---- ETC2 planar mode (decompress-etc.c:287-317) ----------------------------------------------------------
// O, H, V in 6-7-6 bits, MSB-replicated to 8; texel(x, y) = clamp255((x(H-O) + y(V-O) + 4O + 2) >> 2).  Every term fits a
// signed 16-bit lane (|sum| <= 2550): (R, B) share a register, G sits in the low lane of a second one.
struct EtcPlanar { uint32_t o_rb, dh_rb, dv_rb, o_g, dh_g, dv_g; };	// o_* = 4*O + 2
DH EtcPlanar etc_planar_setup(uint32_t W, uint32_t word) {
	const uint32_t ro = (W >> 25) & 0x3Fu, go = ((W >> 18) & 0x40u) | ((W >> 17) & 0x3Fu);
	const uint32_t bo = ((W >> 11) & 0x20u) | ((W >> 8) & 0x18u) | ((W >> 7) & 0x7u);
	const uint32_t rh = ((W >> 1) & 0x3Eu) | (W & 1u), gh = word >> 25, bh = (word >> 19) & 0x3Fu;
	const uint32_t rv = (word >> 13) & 0x3Fu, gv = (word >> 6) & 0x7Fu, bv = word & 0x3Fu;
	uint32_t o_rb = ro | (bo << 16), h_rb = rh | (bh << 16), v_rb = rv | (bv << 16);
	o_rb = (o_rb << 2) | ((o_rb >> 4) & 0x00030003u);
	h_rb = (h_rb << 2) | ((h_rb >> 4) & 0x00030003u);
	v_rb = (v_rb << 2) | ((v_rb >> 4) & 0x00030003u);
	const uint32_t o_g = (go << 1) | (go >> 6), h_g = (gh << 1) | (gh >> 6), v_g = (gv << 1) | (gv >> 6);
	EtcPlanar c;
	c.dh_rb = pk_sub16(h_rb, o_rb); c.dv_rb = pk_sub16(v_rb, o_rb);
	c.dh_g = (h_g - o_g) & 0xFFFFu; c.dv_g = (v_g - o_g) & 0xFFFFu;
	c.o_rb = (o_rb << 2) + 0x00020002u; c.o_g = 4u * o_g + 2u;
	return c;
}
// all sixteen texels in the block's own lane: three packed adds shared between channels / neighbours, an arithmetic
// shift and the saturating pack per texel
template <uint32_t ALPHA> DH void etc_planar_block(const EtcPlanar &c, uint32_t (&d)[16]) {
	typedef EtcGather<ALPHA> G;
	uint32_t rb_row = c.o_rb;
	uint32_t gg_row = pack16(c.o_g, c.o_g + c.dh_g);		// G of texels x = 0, 1
	const uint32_t gg_dv = pack16(c.dv_g, c.dv_g), gg_2dh = pack16(2u * c.dh_g, 2u * c.dh_g);
#pragma unroll
	for (int y = 0; y < 4; y++) {
		if (y) { rb_row = pk_add16(rb_row, c.dv_rb); gg_row = pk_add16(gg_row, gg_dv); }
		const uint32_t g01 = sat_u8_pk16(pk_ashr16(gg_row, 2)), g23 = sat_u8_pk16(pk_ashr16(pk_add16(gg_row, gg_2dh), 2));
		uint32_t rb = rb_row;
		d[y * 4 + 0] = G::template sat<0>(g01, sat_u8_pk16(pk_ashr16(rb, 2)));
		rb = pk_add16(rb, c.dh_rb);
		d[y * 4 + 1] = G::template sat<1>(g01, sat_u8_pk16(pk_ashr16(rb, 2)));
		rb = pk_add16(rb, c.dh_rb);
		d[y * 4 + 2] = G::template sat<0>(g23, sat_u8_pk16(pk_ashr16(rb, 2)));
		rb = pk_add16(rb, c.dh_rb);
		d[y * 4 + 3] = G::template sat<1>(g23, sat_u8_pk16(pk_ashr16(rb, 2)));
	}
}
// one texel (x, y) of a planar block
template <uint32_t ALPHA> DH uint32_t etc_planar_texel(const EtcPlanar &c, uint32_t x, uint32_t y) {
	const uint32_t xx = DETEX_UMUL24(x, 0x10001u), yy = DETEX_UMUL24(y, 0x10001u);
	const uint32_t rb = pk_mad_u16(c.dh_rb, xx, pk_mad_u16(c.dv_rb, yy, c.o_rb));
	const uint32_t g = pk_mad_u16(c.dh_g, xx, pk_mad_u16(c.dv_g, yy, c.o_g));		// low lane
	return EtcGather<ALPHA>::template sat<0>(sat_u8_pk16(pk_ashr16(g, 2)), sat_u8_pk16(pk_ashr16(rb, 2)));
}

// Planar blocks are rare among other blocks (1 in 36 of a random stream, the smooth patches of real textures), yet a wave
// with a single one executes the whole 16-texel planar path: ~125 VALU instructions on top of the ~300 of the palette modes.
// When a wave holds at most eight of them, the owners only derive the six coefficients and park them in LDS; the wave's
// lanes then compute ONE texel each (lane t: block t / 16, texel t % 16) and the owners read their sixteen texels back.
// Waves with more planar blocks (43 % of the blocks of the reference's ETC2 fixtures are planar) decode them in their own
// lanes, on the path they always took.  Same run, 8192^2, stream U: ETC2_EAC 52.9 -> 50.4 us, ETC2 44.7 -> 43.1.  Purely wave-local: LDS operations of one wave
// complete in order, no workgroup barrier; works for any set of active lanes (tasks are dealt to the active ones).
// most planar blocks per wave that are decoded cooperatively (8 = two passes of 64 texels; thresholds 4 / 8 / 16 measured equal
// on the random stream; 0 = always in their own lanes: the measurement builds' control)
constexpr int kEtcPlanarShared = Tune::kEtcPlanarShared;
constexpr int kEtcPlanarSlots = kEtcPlanarShared > 0 ? kEtcPlanarShared : 1;
struct alignas(16) EtcPlanarSlab { uint32_t coef[kEtcPlanarSlots][8]; uint32_t texel[kEtcPlanarSlots * 16]; };	// texel[16 * b + 4 * y + x]
DH EtcPlanarSlab &etc_planar_slab() { __shared__ EtcPlanarSlab slabs[4]; return slabs[threadIdx.x >> 6]; }
DH void wave_lds_sync() {
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
DH uint32_t lanes_below(uint64_t mask) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u)); }
// true (wave-uniform) when this wave's planar blocks are decoded cooperatively
DH bool etc_planar_shared(bool mode_planar, uint64_t &owners) {
	owners = __builtin_amdgcn_ballot_w64(mode_planar);
	return owners != 0 && __builtin_popcountll(owners) <= kEtcPlanarShared;
}
template <uint32_t ALPHA> DH void etc_planar_wave(bool mode_planar, uint64_t owners, uint32_t W, uint32_t word, uint32_t (&d)[16]) {
	const uint32_t count = (uint32_t)__builtin_popcountll(owners);
	EtcPlanarSlab &slab = etc_planar_slab();
	const uint32_t rank = lanes_below(owners);
	if (mode_planar) {
		const EtcPlanar c = etc_planar_setup(W, word);
		*reinterpret_cast<u32x4 *>(&slab.coef[rank][0]) = u32x4{ c.o_rb, c.dh_rb, c.dv_rb, c.o_g };
		*reinterpret_cast<u32x2 *>(&slab.coef[rank][4]) = u32x2{ c.dh_g, c.dv_g };
	}
	wave_lds_sync();
	const uint64_t active = __builtin_amdgcn_ballot_w64(true);
	const uint32_t n_active = (uint32_t)__builtin_popcountll(active);
	for (uint32_t t = lanes_below(active); t < 16u * count; t += n_active) {
		const u32x4 ka = *reinterpret_cast<const u32x4 *>(&slab.coef[t >> 4][0]);
		const u32x2 kb = *reinterpret_cast<const u32x2 *>(&slab.coef[t >> 4][4]);
		const EtcPlanar c = { ka.x, ka.y, ka.z, ka.w, kb.x, kb.y };
		slab.texel[t] = etc_planar_texel<ALPHA>(c, t & 3u, (t >> 2) & 3u);
	}
	wave_lds_sync();
	if (mode_planar) {
#pragma unroll
		for (int k = 0; k < 4; k++) {
			const u32x4 v = *reinterpret_cast<const u32x4 *>(&slab.texel[16u * rank + 4u * (uint32_t)k]);
			d[4 * k] = v.x; d[4 * k + 1] = v.y; d[4 * k + 2] = v.z; d[4 * k + 3] = v.w;
		}
	}
	wave_lds_sync();		// the slab is free again before this wave's next block
}

// KIND: 0 = ETC1, 1 = ETC2, 2 = ETC2 punchthrough.  ALPHA = alpha bits of opaque texels
// (0xFF000000, or 0 when an EAC alpha plane is merged afterwards).
// decompress-etc.c:89-180 (ETC1), :202-367 (ETC2), :472-717 (punchthrough).
template <int KIND, uint32_t ALPHA, bool CHECKED>
DH bool etc_colour(uint32_t w0, uint32_t w1, uint32_t mode_mask, uint32_t flags, uint32_t (&d)[16]) {
	static_assert(ALPHA == 0u || ALPHA == 0xFF000000u, "alpha is all-or-nothing");
	typedef EtcGather<ALPHA> G;
	const uint32_t b0 = w0 & 0xFFu, b1 = (w0 >> 8) & 0xFFu, b2 = (w0 >> 16) & 0xFFu, b3 = w0 >> 24;
	const uint32_t W = bswap32(w0);			// b0 in bits 31:24 ... b3 in bits 7:0
	const uint32_t word = bswap32(w1);
	const bool diffbit = (b3 & 2u) != 0;		// ETC1/ETC2: differential; punchthrough: opaque
	// 5-bit base + 3-bit two's-complement delta; out of 0..31 selects T / H / planar (:324-366)
	const int32_t sr = (int32_t)(b0 >> 3) + sbfe(b0, 0, 3);
	const int32_t sg = (int32_t)(b1 >> 3) + sbfe(b1, 0, 3);
	const int32_t sb = (int32_t)(b2 >> 3) + sbfe(b2, 0, 3);
	const bool ovr = (uint32_t)sr > 31u, ovg = (uint32_t)sg > 31u, ovb = (uint32_t)sb > 31u;
	const bool individual = (KIND != 2) && !diffbit;
	const bool opaque = (KIND != 2) || diffbit;
	bool mode_t = false, mode_h = false, mode_planar = false;
	if (KIND == 0) {
		if (CHECKED && !(mode_mask & (diffbit ? kMaskEtcDifferential : kMaskEtcIndividual))) return false;
		if (diffbit && (ovr || ovg || ovb)) return false;		// :111-122 invalid in ETC1
	} else {
		mode_t = !individual && ovr;
		mode_h = !individual && !ovr && ovg;
		mode_planar = !individual && !ovr && !ovg && ovb;
		if (CHECKED) {
			if (KIND == 2) {
				if (opaque && (flags & kFlagNonOpaqueOnly)) return false;
				if (!opaque && (flags & kFlagOpaqueOnly)) return false;
				if (mode_planar && (flags & kFlagNonOpaqueOnly)) return false;
			} else if (!individual && (mode_mask & ~kMaskEtcIndividual) == 0) {
				return false;
			}
			const uint32_t need = individual ? kMaskEtcIndividual : (mode_t ? kMaskEtcT : (mode_h ? kMaskEtcH
				: (mode_planar ? kMaskEtcPlanar : kMaskEtcDifferential)));
			if (!(mode_mask & need)) return false;
		}
	}
	uint64_t planar_owners = 0;
	const bool planar_shared = KIND != 0 && kEtcPlanarShared > 0 && etc_planar_shared(mode_planar, planar_owners);	// wave-uniform
	if (mode_planar && !planar_shared) {
		etc_planar_block<ALPHA>(etc_planar_setup(W, word), d);
		return true;
	}
	uint32_t pal0[4], pal1[4];
	bool flip = (b3 & 1u) != 0;
	if (mode_t || mode_h) {
		// :202-285 / :565-649: two RGB444 base colours and a distance -> four paint colours.
		// Both layouts are reduced to 12-bit colours c1, c2 (r<<8 | g<<4 | b) and a distance index, then share
		// the arithmetic:  T = {c1, c2+d, c2, c2-d},  H = {c1+d, c1-d, c2+d, c2-d}.
		const uint32_t c1_t = ((W >> 17) & 0xC00u) | ubfe(W, 16, 10);
		const uint32_t c1_h = ((W >> 19) & 0xFE0u) | ((W >> 16) & 0x18u) | ((W >> 15) & 0x7u);
		const uint32_t c1 = mode_t ? c1_t : c1_h;
		const uint32_t c2 = ubfe(W, mode_t ? 4u : 3u, 12);
		// H orders by the 24-bit expanded colours (:236); nibble replication is monotonic, so c1 >= c2 decides the same
		const uint32_t di_t = ((b3 & 0xCu) >> 1) | (b3 & 1u), di_h = (b3 & 4u) | ((b3 & 1u) << 1) | (c1 >= c2 ? 1u : 0u);
		const uint32_t di = mode_t ? di_t : di_h;
		const uint32_t dd = perm(ETC_DIST_HI, ETC_DIST_LO, DETEX_UMUL24(di, 0x10001u) | 0x0C000C00u);	// (dist, dist)
		const uint32_t rb1 = DETEX_UMUL24((c1 >> 8) | ((c1 & 0xFu) << 16), 17u), rb2 = DETEX_UMUL24((c2 >> 8) | ((c2 & 0xFu) << 16), 17u);
		const uint32_t g12 = DETEX_UMUL24(ubfe(c1, 4, 4) | (ubfe(c2, 4, 4) << 16), 17u);	// (G1, G2)
		const uint32_t gp = sat_u8_pk16(pk_add16(g12, dd)), gm = sat_u8_pk16(pk_sub16(g12, dd));
		const uint32_t c1p = G::template sat<0>(gp, sat_u8_pk16(pk_add16(rb1, dd)));
		const uint32_t c1m = G::template sat<0>(gm, sat_u8_pk16(pk_sub16(rb1, dd)));
		const uint32_t c2p = G::template sat<1>(gp, sat_u8_pk16(pk_add16(rb2, dd)));
		const uint32_t c2m = G::template sat<1>(gm, sat_u8_pk16(pk_sub16(rb2, dd)));
		const uint32_t c1z = G::template raw<0>(g12, rb1), c2z = G::template raw<2>(g12, rb2);
		const uint32_t tm = cond_to_mask(mode_t);
		pal0[0] = bfi(tm, c1z, c1p);
		pal0[1] = bfi(tm, c2p, c1m);
		pal0[2] = bfi(tm, c2z, c2p);
		pal0[3] = c2m;
		if (!opaque) pal0[2] = 0u;		// punchthrough: selector 2 is fully transparent black (:483-499)
#pragma unroll
		for (int k = 0; k < 4; k++) pal1[k] = pal0[k];
		flip = false;
	} else {
		// individual (4+4 bits, replicated) or differential (5 bits + signed 3-bit delta) bases, all three
		// channels at once in the bytes of x = R | G << 8 | B << 16 (:143-160)
		const uint32_t x = w0 & 0xFFFFFFu;
		const uint32_t i0 = (x & 0xF0F0F0u) | ((x >> 4) & 0x0F0F0Fu), i1 = (x & 0x0F0F0Fu) | ((x << 4) & 0xF0F0F0u);
		const uint32_t d0 = (x & 0xF8F8F8u) | ((x >> 5) & 0x070707u);
		// per byte: (b >> 3) + (b & 3) - (b & 4) = base + delta, in 0..31 for every block that reaches this path
		// (an out-of-range sum selected T / H / planar, or failed ETC1, above), so no borrow crosses bytes
		const uint32_t t = ((x >> 3) & 0x1F1F1Fu) + (x & 0x030303u) - (x & 0x040404u);
		const uint32_t d1 = ((t << 3) & 0xF8F8F8u) | ((t >> 2) & 0x070707u);
		const uint32_t im = cond_to_mask(individual);
		const uint32_t base0 = bfi(im, i0, d0), base1 = bfi(im, i1, d1);
		const uint32_t rb0 = perm(0u, base0, 0x0C020C00u), rb1 = perm(0u, base1, 0x0C020C00u);	// (R, B) lanes
		const uint32_t g01 = perm(base1, base0, 0x0C050C01u);						// (G0, G1)
		// intensity modifiers of both sub-blocks in one lookup each: (small0, small1), (large0, large1)
		const uint32_t tsel = ubfe(b3, 5, 3) | (ubfe(b3, 2, 3) << 16) | 0x0C000C00u;
		uint32_t s01 = perm(ETC_SMALL_HI, ETC_SMALL_LO, tsel);
		const uint32_t l01 = perm(ETC_LARGE_HI, ETC_LARGE_LO, tsel);
		if (!opaque) s01 = 0u;			// punchthrough non-opaque: modifiers {0, large, (transparent), -large}
		const uint32_t s00 = perm(0u, s01, 0x01000100u), s11 = perm(0u, s01, 0x03020302u);
		const uint32_t l00 = perm(0u, l01, 0x01000100u), l11 = perm(0u, l01, 0x03020302u);
		const uint32_t gps = sat_u8_pk16(pk_add16(g01, s01)), gms = sat_u8_pk16(pk_sub16(g01, s01));
		const uint32_t gpl = sat_u8_pk16(pk_add16(g01, l01)), gml = sat_u8_pk16(pk_sub16(g01, l01));
		pal0[0] = G::template sat<0>(gps, sat_u8_pk16(pk_add16(rb0, s00)));
		pal0[1] = G::template sat<0>(gpl, sat_u8_pk16(pk_add16(rb0, l00)));
		pal0[2] = G::template sat<0>(gms, sat_u8_pk16(pk_sub16(rb0, s00)));
		pal0[3] = G::template sat<0>(gml, sat_u8_pk16(pk_sub16(rb0, l00)));
		pal1[0] = G::template sat<1>(gps, sat_u8_pk16(pk_add16(rb1, s11)));
		pal1[1] = G::template sat<1>(gpl, sat_u8_pk16(pk_add16(rb1, l11)));
		pal1[2] = G::template sat<1>(gms, sat_u8_pk16(pk_sub16(rb1, s11)));
		pal1[3] = G::template sat<1>(gml, sat_u8_pk16(pk_sub16(rb1, l11)));
		if (!opaque) { pal0[2] = 0u; pal1[2] = 0u; }
	}
	etc_texels(word, flip, pal0, pal1, d);
	// the few planar blocks of this wave went through the palette path with meaningless palettes; their texels come now
	if (KIND != 0 && planar_shared) etc_planar_wave<ALPHA>(mode_planar, planar_owners, W, word, d);
	return true;
}

// ---- EAC ------------------------------------------------------------------------------------------
// 64-bit big-endian EAC word: base(8) multiplier(4) table(4) then sixteen 3-bit selectors MSB-first,
// column-major (decompress-eac.c:44-50, 111-128).
struct EacWord { uint32_t base, mult, row, sel_a, sel_b; };	// row = nibble-packed magnitudes
DH EacWord eac_word(uint32_t w0, uint32_t w1) {
	const uint32_t hi = bswap32(w0), lo = bswap32(w1);
	EacWord e;
	e.base = hi >> 24;
	e.mult = (hi >> 20) & 0xFu;
	e.row = eac_row((hi >> 16) & 0xFu);
	e.sel_a = ((hi & 0xFFFFu) << 8) | (lo >> 24);	// texels 0-7, texel 0 in bits 23:21
	e.sel_b = lo & 0xFFFFFFu;			// texels 8-15
	return e;
}
DH uint32_t eac_selector(const EacWord &e, int p) { return ubfe(p < 8 ? e.sel_a : e.sel_b, 21 - 3 * (p & 7), 3); }
// ETC2_EAC alpha plane (decompress-eac.c:54-86): alpha = clamp255(base + modifier * multiplier),
// multiplier 0 allowed (A-8).  Builds the 8-entry alpha table once, then one v_perm per texel.
DH void eac_alpha_overlay(uint32_t w0, uint32_t w1, uint32_t (&d)[16]) {
	const EacWord e = eac_word(w0, w1);
	// the eight table values base - m*mult (selectors 0-3) and base + (m-1)*mult (4-7), two per VGPR in signed
	// 16-bit lanes (|m*mult| <= 225), clamped by the saturating pack
	const uint32_t m01 = (e.row & 0xFu) | ((e.row << 12) & 0xF0000u), m23 = ((e.row >> 8) & 0xFu) | ((e.row << 4) & 0xF0000u);
	const uint32_t mult2 = DETEX_UMUL24(e.mult, 0x10001u), base2 = DETEX_UMUL24(e.base, 0x10001u);
	const uint32_t mm01 = pk_mul16(m01, mult2), mm23 = pk_mul16(m23, mult2), bm = pk_sub16(base2, mult2);
	const uint32_t lo = perm(sat_u8_pk16(pk_sub16(base2, mm23)), sat_u8_pk16(pk_sub16(base2, mm01)), 0x05040100u);
	const uint32_t hi = perm(sat_u8_pk16(pk_add16(bm, mm23)), sat_u8_pk16(pk_add16(bm, mm01)), 0x05040100u);
#pragma unroll
	for (int p = 0; p < 16; p++) {
		const int idx = (p & 3) * 4 + (p >> 2);
		d[idx] |= perm(hi, lo, eac_selector(e, p)) << 24;
	}
}

// one 11-bit channel -> eight dwords of two 16-bit texels (row-major pairs).
// unsigned: decompress-eac.c:111-128; signed: :159-201 (base -128 is invalid, A-5).
template <bool SIGNED> DH bool eac11_channel(uint32_t w0, uint32_t w1, uint32_t (&pairs)[8]) {
	const EacWord e = eac_word(w0, w1);
	// the eight table values, two per VGPR in signed 16-bit lanes: raw = base - m*mult8 (selectors 0-3) and
	// base + (m-1)*mult8 (4-7); |raw| <= 2044 + 15*120 fits, and so does every later step
	const uint32_t mult8 = e.mult ? e.mult * 8u : 1u;
	const uint32_t base = SIGNED ? (uint32_t)(sbfe(e.base, 0, 8) * 8) & 0xFFFFu : e.base * 8u + 4u;
	const uint32_t mult2 = pack16(mult8, mult8), base2 = pack16(base, base), bm = pk_sub16(base2, mult2);
	const uint32_t m01 = (e.row & 0xFu) | ((e.row << 12) & 0xF0000u), m23 = ((e.row >> 8) & 0xFu) | ((e.row << 4) & 0xF0000u);
	const uint32_t mm01 = pk_mul16(m01, mult2), mm23 = pk_mul16(m23, mult2);
	uint32_t t[4] = { pk_sub16(base2, mm01), pk_sub16(base2, mm23), pk_add16(bm, mm01), pk_add16(bm, mm23) };
#pragma unroll
	for (int k = 0; k < 4; k++) {
		if (SIGNED) {		// :159-201: clamp to +-1023, 11-bit magnitude widened to 16 bits by bit replication, sign restored
			const uint32_t v = pk_min16(pk_max16(t[k], 0xFC01FC01u), 0x03FF03FFu);
			const uint32_t m = pk_max16(v, pk_sub16(0u, v));
			const uint32_t wide = pk_lshl16(m, 5) | pk_lshr16(m, 5);
			const uint32_t sg = pk_ashr16(v, 15);
			t[k] = pk_sub16(wide ^ sg, sg);
		} else {		// :111-128: clamp to 0..2047, widen by bit replication
			const uint32_t v = pk_min16(pk_max16(t[k], 0u), 0x07FF07FFu);
			t[k] = pk_lshl16(v, 5) | pk_lshr16(v, 6);
		}
	}
	// low-byte / high-byte tables of the 8 values for v_perm lookups
	const uint32_t lo_l = perm(t[1], t[0], 0x06040200u), hi_l = perm(t[1], t[0], 0x07050301u);
	const uint32_t lo_h = perm(t[3], t[2], 0x06040200u), hi_h = perm(t[3], t[2], 0x07050301u);
#pragma unroll
	for (int y = 0; y < 4; y++) {
		// row y holds column-major texels p = 4x + y
		const uint32_t sel = eac_selector(e, y) | (eac_selector(e, 4 + y) << 8) | (eac_selector(e, 8 + y) << 16) |
			(eac_selector(e, 12 + y) << 24);
		const uint32_t l4 = perm(lo_h, lo_l, sel), h4 = perm(hi_h, hi_l, sel);
		pairs[2 * y] = perm(h4, l4, 0x05010400u);
		pairs[2 * y + 1] = perm(h4, l4, 0x07030602u);
	}
	return !SIGNED || (e.base & 0xFFu) != 0x80u;
}

struct DecETC1 {
	static constexpr int kBlockBytes = 8, kPixelBytes = 4;
	template <bool CHECKED> static DH bool decode(uint2 blk, uint32_t mode_mask, uint32_t flags, uint32_t (&d)[16]) {
		return etc_colour<0, 0xFF000000u, CHECKED>(blk.x, blk.y, mode_mask, flags, d);
	}
};
struct DecETC2 {
	static constexpr int kBlockBytes = 8, kPixelBytes = 4;
	template <bool CHECKED> static DH bool decode(uint2 blk, uint32_t mode_mask, uint32_t flags, uint32_t (&d)[16]) {
		return etc_colour<1, 0xFF000000u, CHECKED>(blk.x, blk.y, mode_mask, flags, d);
	}
};
struct DecETC2Punchthrough {
	static constexpr int kBlockBytes = 8, kPixelBytes = 4;
	template <bool CHECKED> static DH bool decode(uint2 blk, uint32_t mode_mask, uint32_t flags, uint32_t (&d)[16]) {
		return etc_colour<2, 0xFF000000u, CHECKED>(blk.x, blk.y, mode_mask, flags, d);
	}
};
struct DecETC2EAC {
	static DH void prepare() { eac_prepare(); }
	static constexpr int kBlockBytes = 16, kPixelBytes = 4;
	// decompress-eac.c:54-86: colour = ETC2 on bytes 8-15, alpha = EAC on bytes 0-7
	template <bool CHECKED> static DH bool decode(uint4 blk, uint32_t mode_mask, uint32_t flags, uint32_t (&d)[16]) {
		if (!etc_colour<1, 0u, CHECKED>(blk.z, blk.w, mode_mask, flags, d)) return false;
		if (CHECKED && (flags & kFlagEncode) && ((blk.x >> 12) & 0xFu) == 0) return false;	// multiplier 0 (:62-64)
		eac_alpha_overlay(blk.x, blk.y, d);
		return true;
	}
};
struct DecEACR11 {
	static DH void prepare() { eac_prepare(); }
	static constexpr int kBlockBytes = 8, kPixelBytes = 2, kNative = kNatR16;
	template <bool CHECKED> static DH bool decode(uint2 blk, uint32_t, uint32_t, uint32_t (&d)[8]) {
		return eac11_channel<false>(blk.x, blk.y, d);
	}
};
struct DecEACSignedR11 {
	static DH void prepare() { eac_prepare(); }
	static constexpr int kBlockBytes = 8, kPixelBytes = 2, kNative = kNatSignedR16;
	template <bool CHECKED> static DH bool decode(uint2 blk, uint32_t, uint32_t, uint32_t (&d)[8]) {
		return eac11_channel<true>(blk.x, blk.y, d);
	}
};
template <bool SIGNED> struct DecEACRG11T {
	static DH void prepare() { eac_prepare(); }
	static constexpr int kBlockBytes = 16, kPixelBytes = 4, kNative = SIGNED ? kNatSignedRG16 : kNatRG16;
	// decompress-eac.c:144-157, 217-231: texel = R16 | G16 << 16
	template <bool CHECKED> static DH bool decode(uint4 blk, uint32_t, uint32_t, uint32_t (&d)[16]) {
		uint32_t r[8], g[8];
		const bool ok_r = eac11_channel<SIGNED>(blk.x, blk.y, r);
		const bool ok_g = eac11_channel<SIGNED>(blk.z, blk.w, g);
#pragma unroll
		for (int k = 0; k < 8; k++) {
			d[2 * k] = perm(g[k], r[k], 0x05040100u);
			d[2 * k + 1] = perm(g[k], r[k], 0x07060302u);
		}
		return ok_r && ok_g;
	}
};
using DecEACRG11 = DecEACRG11T<false>;
using DecEACSignedRG11 = DecEACRG11T<true>;

}  // namespace detexhip
