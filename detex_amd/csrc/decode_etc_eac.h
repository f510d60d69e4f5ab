// decode_etc_eac.h -- ETC1 / ETC2 / ETC2 punchthrough / ETC2+EAC and EAC R11/RG11 (+-signed), gfx950.
//
// One lane = one block.  The colour codec is restructured around per-sub-block 4-entry
// palettes of packed RGBA dwords: individual/differential blocks, T blocks and H blocks all
// reduce to "texel = palette[sub-block][2-bit selector]", so they share one branch-free texel
// loop (v_bfe_i32 lane masks + v_bfi_b32); only planar blocks take a separate path.  The
// selector planes are big-endian and texels are numbered column-major.
// Format tables live in __constant__ memory / literal operands (north_star).
#pragma once
#include "dev_common.h"
#include "decode_s3tc_rgtc.h"

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

DH uint32_t etc_entry(int32_t r, int32_t g, int32_t b, int32_t m, uint32_t alpha) {
	return pack_rgba(clamp255(r + m), clamp255(g + m), clamp255(b + m), 0u) | alpha;
}
DH uint32_t rep4(uint32_t v) { return v | (v << 4); }

// texel loop shared by every non-planar mode.  pal0 = sub-block 0 palette, pal1 = sub-block 1;
// flip selects the 2x4 / 4x2 split (decompress-etc.c:143-178).  Texel p is column-major
// (x = p>>2, y = p&3) and lands at row-major index y*4+x (decompress-etc.c:83).
DH void etc_texels(uint32_t word, bool flip, const uint32_t (&pal0)[4], const uint32_t (&pal1)[4], uint32_t (&d)[16]) {
	uint32_t palb[4], palc[4];	// quadrant (x<2,y>=2) and (x>=2,y<2)
#pragma unroll
	for (int k = 0; k < 4; k++) {
		palb[k] = flip ? pal1[k] : pal0[k];
		palc[k] = flip ? pal0[k] : pal1[k];
	}
#pragma unroll
	for (int p = 0; p < 16; p++) {
		const int x = p >> 2, y = p & 3;
		const uint32_t m_lo = bit_to_mask(word, p), m_hi = bit_to_mask(word, 16 + p);
		const uint32_t(&q)[4] = x < 2 ? (y < 2 ? pal0 : palb) : (y < 2 ? palc : pal1);
		d[y * 4 + x] = select4(m_lo, m_hi, q[0], q[1], q[2], q[3]);
	}
}

// KIND: 0 = ETC1, 1 = ETC2, 2 = ETC2 punchthrough.  ALPHA = alpha bits OR-ed into opaque texels
// (0xFF000000, or 0 when an EAC alpha plane is merged afterwards).
// decompress-etc.c:89-180 (ETC1), :202-367 (ETC2), :472-717 (punchthrough).
template <int KIND, uint32_t ALPHA, bool CHECKED>
DH bool etc_colour(uint32_t w0, uint32_t w1, uint32_t mode_mask, uint32_t flags, uint32_t (&d)[16]) {
	const uint32_t b0 = w0 & 0xFFu, b1 = (w0 >> 8) & 0xFFu, b2 = (w0 >> 16) & 0xFFu, b3 = w0 >> 24;
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
	if (mode_planar) {
		// :287-317: O, H, V in 6-7-6, MSB-replicated to 8 bits; texel = clamp((x(H-O) + y(V-O) + 4O + 2) >> 2)
		const uint32_t b4 = word >> 24, b5 = (word >> 16) & 0xFFu, b6 = (word >> 8) & 0xFFu, b7 = word & 0xFFu;
		int32_t o[3], h[3], v[3];
		o[0] = (b0 & 0x7Eu) >> 1;
		o[1] = ((b0 & 1u) << 6) | ((b1 & 0x7Eu) >> 1);
		o[2] = ((b1 & 1u) << 5) | (b2 & 0x18u) | ((b2 & 3u) << 1) | (b3 >> 7);
		h[0] = ((b3 & 0x7Cu) >> 1) | (b3 & 1u);
		h[1] = b4 >> 1;
		h[2] = ((b4 & 1u) << 5) | (b5 >> 3);
		v[0] = ((b5 & 7u) << 3) | (b6 >> 5);
		v[1] = ((b6 & 0x1Fu) << 2) | (b7 >> 6);
		v[2] = b7 & 0x3Fu;
#pragma unroll
		for (int k = 0; k < 3; k++) {
			const int s = (k == 1) ? 1 : 2, t = (k == 1) ? 6 : 4;
			o[k] = (o[k] << s) | (o[k] >> t);
			h[k] = (h[k] << s) | (h[k] >> t);
			v[k] = (v[k] << s) | (v[k] >> t);
			h[k] -= o[k];
			v[k] -= o[k];
			o[k] = 4 * o[k] + 2;
		}
		// clamp255(s >> 2) is written as clampi(s, 0, 1023) >> 2 (identical for every int s).
		// Besides saving nothing or costing nothing, this form matters: for the pair pattern
		// clamp(a >> 2) | clamp(b >> 2) << 8, hipcc 7.2 selects gfx950's v_ashr_pk_u8_i32 and ORs the
		// result as if it were zero-extended, but on MI355X its bits [31:16] are not zero -- the
		// blue byte came out corrupted (found by the parity tests; see DESIGN.md "toolchain notes").
#pragma unroll
		for (int y = 0; y < 4; y++)
#pragma unroll
			for (int x = 0; x < 4; x++)
				d[y * 4 + x] = pack_rgba((uint32_t)clampi(x * h[0] + y * v[0] + o[0], 0, 1023) >> 2,
					(uint32_t)clampi(x * h[1] + y * v[1] + o[1], 0, 1023) >> 2,
					(uint32_t)clampi(x * h[2] + y * v[2] + o[2], 0, 1023) >> 2, 0u) | ALPHA;
		return true;
	}
	uint32_t pal0[4], pal1[4];
	bool flip = (b3 & 1u) != 0;
	if (mode_t || mode_h) {
		// :202-285 / :565-649: two RGB444 base colours +- distance -> four paint colours
		int32_t c1r, c1g, c1b, c2r, c2g, c2b, dist;
		if (mode_t) {
			c1r = rep4(((b0 & 0x18u) >> 1) | (b0 & 3u)); c1g = rep4(b1 >> 4); c1b = rep4(b1 & 0xFu);
			c2r = rep4(b2 >> 4); c2g = rep4(b2 & 0xFu); c2b = rep4(b3 >> 4);
			dist = perm(ETC_DIST_HI, ETC_DIST_LO, ((b3 & 0xCu) >> 1) | (b3 & 1u)) & 0xFFu;
			pal0[0] = pack_rgba(c1r, c1g, c1b, 0u) | ALPHA;
			pal0[1] = etc_entry(c2r, c2g, c2b, dist, ALPHA);
			pal0[2] = pack_rgba(c2r, c2g, c2b, 0u) | ALPHA;
			pal0[3] = etc_entry(c2r, c2g, c2b, -dist, ALPHA);
		} else {
			c1r = rep4((b0 & 0x78u) >> 3);
			c1g = rep4(((b0 & 7u) << 1) | ((b1 & 0x10u) >> 4));
			c1b = rep4((b1 & 8u) | ((b1 & 3u) << 1) | (b2 >> 7));
			c2r = rep4((b2 & 0x78u) >> 3);
			c2g = rep4(((b2 & 7u) << 1) | (b3 >> 7));
			c2b = rep4((b3 & 0x78u) >> 3);
			const uint32_t v1 = (c1r << 16) + (c1g << 8) + c1b, v2 = (c2r << 16) + (c2g << 8) + c2b;
			dist = perm(ETC_DIST_HI, ETC_DIST_LO, (b3 & 4u) | ((b3 & 1u) << 1) | (v1 >= v2 ? 1u : 0u)) & 0xFFu;
			pal0[0] = etc_entry(c1r, c1g, c1b, dist, ALPHA);
			pal0[1] = etc_entry(c1r, c1g, c1b, -dist, ALPHA);
			pal0[2] = etc_entry(c2r, c2g, c2b, dist, ALPHA);
			pal0[3] = etc_entry(c2r, c2g, c2b, -dist, ALPHA);
		}
		if (!opaque) pal0[2] = 0u;		// punchthrough: selector 2 is fully transparent black (:483-499)
#pragma unroll
		for (int k = 0; k < 4; k++) pal1[k] = pal0[k];
		flip = false;
	} else {
		// individual (4+4 bits, replicated) or differential (5 bits + signed 3-bit delta) bases
		int32_t base0[3], base1[3];
		const uint32_t byte[3] = { b0, b1, b2 };
		const int32_t sum[3] = { sr, sg, sb };
#pragma unroll
		for (int k = 0; k < 3; k++) {
			const int32_t i0 = (byte[k] & 0xF0u) | (byte[k] >> 4), i1 = rep4(byte[k] & 0xFu);
			const int32_t d0 = (byte[k] & 0xF8u) | (byte[k] >> 5), d1 = (sum[k] << 3) | (sum[k] >> 2);
			base0[k] = individual ? i0 : d0;
			base1[k] = individual ? i1 : d1;
		}
		const uint32_t t0 = (b3 >> 5) & 7u, t1 = (b3 >> 2) & 7u;
		const int32_t s0 = opaque ? (int32_t)(perm(ETC_SMALL_HI, ETC_SMALL_LO, t0) & 0xFFu) : 0;
		const int32_t s1 = opaque ? (int32_t)(perm(ETC_SMALL_HI, ETC_SMALL_LO, t1) & 0xFFu) : 0;
		const int32_t l0 = perm(ETC_LARGE_HI, ETC_LARGE_LO, t0) & 0xFFu, l1 = perm(ETC_LARGE_HI, ETC_LARGE_LO, t1) & 0xFFu;
		pal0[0] = etc_entry(base0[0], base0[1], base0[2], s0, ALPHA);
		pal0[1] = etc_entry(base0[0], base0[1], base0[2], l0, ALPHA);
		pal0[2] = etc_entry(base0[0], base0[1], base0[2], -s0, ALPHA);
		pal0[3] = etc_entry(base0[0], base0[1], base0[2], -l0, ALPHA);
		pal1[0] = etc_entry(base1[0], base1[1], base1[2], s1, ALPHA);
		pal1[1] = etc_entry(base1[0], base1[1], base1[2], l1, ALPHA);
		pal1[2] = etc_entry(base1[0], base1[1], base1[2], -s1, ALPHA);
		pal1[3] = etc_entry(base1[0], base1[1], base1[2], -l1, ALPHA);
		if (!opaque) { pal0[2] = 0u; pal1[2] = 0u; }
	}
	etc_texels(word, flip, pal0, pal1, d);
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
	e.row = kEacMagnitudes[(hi >> 16) & 0xFu];
	e.sel_a = ((hi & 0xFFFFu) << 8) | (lo >> 24);	// texels 0-7, texel 0 in bits 23:21
	e.sel_b = lo & 0xFFFFFFu;			// texels 8-15
	return e;
}
DH uint32_t eac_selector(const EacWord &e, int p) { return ubfe(p < 8 ? e.sel_a : e.sel_b, 21 - 3 * (p & 7), 3); }
// modifier of selector s: s<4 -> -m[s], s>=4 -> m[s&3]-1
DH int32_t eac_modifier(uint32_t row, int s) {
	const int32_t m = (int32_t)ubfe(row, 4 * (s & 3), 4);
	return (s & 4) ? m - 1 : -m;
}

// ETC2_EAC alpha plane (decompress-eac.c:54-86): alpha = clamp255(base + modifier * multiplier),
// multiplier 0 allowed (A-8).  Builds the 8-entry alpha table once, then one v_perm per texel.
DH void eac_alpha_overlay(uint32_t w0, uint32_t w1, uint32_t (&d)[16]) {
	const EacWord e = eac_word(w0, w1);
	uint32_t lo = 0, hi = 0;
#pragma unroll
	for (int s = 0; s < 4; s++) {
		lo |= clamp255((int32_t)e.base + eac_modifier(e.row, s) * (int32_t)e.mult) << (8 * s);
		hi |= clamp255((int32_t)e.base + eac_modifier(e.row, s + 4) * (int32_t)e.mult) << (8 * s);
	}
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
	const int32_t mult8 = e.mult ? (int32_t)e.mult * 8 : 1;
	const int32_t base = SIGNED ? (int32_t)(int8_t)e.base * 8 : (int32_t)e.base * 8 + 4;
	uint32_t lo_l = 0, lo_h = 0, hi_l = 0, hi_h = 0;	// low-byte / high-byte tables of the 8 possible values
#pragma unroll
	for (int s = 0; s < 8; s++) {
		const int32_t raw = base + eac_modifier(e.row, s) * mult8;
		uint32_t v16;
		if (SIGNED) {
			const int32_t v = clampi(raw, -1023, 1023);
			const int32_t m = v < 0 ? -v : v;
			const int32_t wide = (m << 5) | (m >> 5);
			v16 = (uint32_t)(v < 0 ? -wide : wide) & 0xFFFFu;
		} else {
			const uint32_t v = (uint32_t)clampi(raw, 0, 2047);
			v16 = (v << 5) | (v >> 6);
		}
		if (s < 4) { lo_l |= (v16 & 0xFFu) << (8 * s); hi_l |= (v16 >> 8) << (8 * s); }
		else { lo_h |= (v16 & 0xFFu) << (8 * (s - 4)); hi_h |= (v16 >> 8) << (8 * (s - 4)); }
	}
#pragma unroll
	for (int y = 0; y < 4; y++) {
		// row y holds column-major texels p = 4x + y
		const uint32_t sel = eac_selector(e, y) | (eac_selector(e, 4 + y) << 8) | (eac_selector(e, 8 + y) << 16) |
			(eac_selector(e, 12 + y) << 24);
		const uint32_t l4 = perm(lo_h, lo_l, sel), h4 = perm(hi_h, hi_l, sel);
		pairs[2 * y] = perm(h4, l4, 0x05010400u);
		pairs[2 * y + 1] = perm(h4, l4, 0x07030602u);
	}
	return !SIGNED || (int8_t)e.base != -128;
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
	static constexpr int kBlockBytes = 8, kPixelBytes = 2;
	template <bool CHECKED> static DH bool decode(uint2 blk, uint32_t, uint32_t, uint32_t (&d)[8]) {
		return eac11_channel<false>(blk.x, blk.y, d);
	}
};
struct DecEACSignedR11 {
	static constexpr int kBlockBytes = 8, kPixelBytes = 2;
	template <bool CHECKED> static DH bool decode(uint2 blk, uint32_t, uint32_t, uint32_t (&d)[8]) {
		return eac11_channel<true>(blk.x, blk.y, d);
	}
};
template <bool SIGNED> struct DecEACRG11T {
	static constexpr int kBlockBytes = 16, kPixelBytes = 4;
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
