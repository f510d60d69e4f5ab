/*
 * oracle/detex_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see detex_oracle.h).
 *
 * A plain-C (C99, libc only) restatement of the algorithms of the hglm/detex v0.1.2 block
 * decoders, written from the format definitions and the reference's observable behaviour,
 * including the reference's quirks.  Every function cites the
 * reference lines whose behaviour it restates.  It deliberately does NOT share code with
 * detex_amd/csrc (the HIP kernels are a second, independent implementation), so that
 * kernel-vs-oracle agreement is evidence and not a tautology.
 *
 * Quirks reproduced (all verified against oracle/_ref by tests/test_oracle_pin.py):
 *   A-2  BC7 mode 6: the second P-bit (block bit 64) is never read -> endpoint 1 P-bit = 0
 *        (decompress-bptc.c:142-146 reads both P-bits from data0 >> 63).
 *   A-3  BC6H mode 12: b0[11] (block bit 63) is dropped with the canonical gcc -O2+ build
 *        (UB shift in bits.h:29-31 hit at decompress-bptc-float.c:462).
 *   A-4  BC1-3: RGB565 expanded by plain shifts, no low-bit replication (decompress-bc.c:34-39).
 */
#include <string.h>
#include "detex_oracle.h"

/* Reference quirks A-2 / A-3 are reproduced by default; orc_set_quirks(0) gives the
 * specification's behaviour instead -- the checker for libdetexhip's detexhipSetQuirks().  Not thread-safe: tests set it
 * around single-threaded calls only. */
static unsigned orc_quirks = ORC_QUIRK_BC7_MODE6_PBIT | ORC_QUIRK_BC6H_MODE12_BIT63;
void orc_set_quirks(unsigned mask) { orc_quirks = mask; }
#include "bptc_partitions.inc"

/* flag / mask values: detex.h:383-411 */
#define MASK_ETC_INDIVIDUAL   0x01u
#define MASK_ETC_DIFFERENTIAL 0x02u
#define MASK_ETC_T            0x04u
#define MASK_ETC_H            0x08u
#define MASK_ETC_PLANAR       0x10u
#define FLAG_ENCODE           0x1u
#define FLAG_OPAQUE_ONLY      0x2u
#define FLAG_NON_OPAQUE_ONLY  0x4u

static uint32_t rd32le(const uint8_t *p) {
	return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
static uint64_t rd64le(const uint8_t *p) { return (uint64_t)rd32le(p) | ((uint64_t)rd32le(p + 4) << 32); }
static uint32_t rd32be(const uint8_t *p) {
	return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3];
}
static uint64_t rd64be(const uint8_t *p) { return ((uint64_t)rd32be(p) << 32) | (uint64_t)rd32be(p + 4); }
static void wr16le(uint8_t *p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); }
static void wr_rgba(uint8_t *p, int r, int g, int b, int a) {
	/* detex.h:1008-1011: R in byte 0 */
	p[0] = (uint8_t)r; p[1] = (uint8_t)g; p[2] = (uint8_t)b; p[3] = (uint8_t)a;
}
static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* ------------------------------------------------------------------------------------------
 * S3TC: BC1 / BC1A / BC2 / BC3              (decompress-bc.c)
 * ---------------------------------------------------------------------------------------- */

/* decompress-bc.c:34-53 (BC1), :156-161 (BC2/3 always four-colour). pal[i] = {r,g,b}. */
static void s3tc_palette(uint32_t colors, int four_colour, int pal[4][3]) {
	const uint32_t c0 = colors & 0xFFFF, c1 = colors >> 16;
	pal[0][0] = (int)((c0 >> 11) & 0x1F) << 3; pal[0][1] = (int)((c0 >> 5) & 0x3F) << 2; pal[0][2] = (int)(c0 & 0x1F) << 3;
	pal[1][0] = (int)((c1 >> 11) & 0x1F) << 3; pal[1][1] = (int)((c1 >> 5) & 0x3F) << 2; pal[1][2] = (int)(c1 & 0x1F) << 3;
	for (int k = 0; k < 3; k++) {
		if (four_colour) {
			pal[2][k] = (2 * pal[0][k] + pal[1][k]) / 3;
			pal[3][k] = (pal[0][k] + 2 * pal[1][k]) / 3;
		} else {
			pal[2][k] = (pal[0][k] + pal[1][k]) / 2;
			pal[3][k] = 0;
		}
	}
}

/* decompress-bc.c:23-61 (alpha always 0xFF) and :87-132 (BC1A: colour 3 of 3-colour mode has alpha 0) */
static int decode_bc1(const uint8_t *in, uint32_t flags, uint8_t *out, int with_alpha) {
	const uint32_t colors = rd32le(in);
	const int opaque = (colors & 0xFFFF) > (colors >> 16);
	if (with_alpha) {
		if (opaque && (flags & FLAG_NON_OPAQUE_ONLY)) return 0;
		if (!opaque && (flags & FLAG_OPAQUE_ONLY)) return 0;
	}
	int pal[4][3];
	s3tc_palette(colors, opaque, pal);
	const uint32_t idx = rd32le(in + 4);
	for (int i = 0; i < 16; i++) {
		const int s = (idx >> (2 * i)) & 3;
		const int a = (with_alpha && !opaque && s == 3) ? 0 : 0xFF;
		wr_rgba(out + 4 * i, pal[s][0], pal[s][1], pal[s][2], a);
	}
	return 1;
}

/* decompress-bc.c:136-171 */
static int decode_bc2(const uint8_t *in, uint32_t flags, uint8_t *out) {
	const uint32_t colors = rd32le(in + 8);
	if ((colors & 0xFFFF) <= (colors >> 16) && (flags & FLAG_ENCODE)) return 0;
	int pal[4][3];
	s3tc_palette(colors, 1, pal);
	const uint32_t idx = rd32le(in + 12);
	const uint64_t abits = rd64le(in);
	for (int i = 0; i < 16; i++) {
		const int s = (idx >> (2 * i)) & 3;
		const int a = (int)((abits >> (4 * i)) & 0xF) * 255 / 15;
		wr_rgba(out + 4 * i, pal[s][0], pal[s][1], pal[s][2], a);
	}
	return 1;
}

/* The 8-entry BC3-alpha / RGTC value ramp for unsigned endpoints.
 * decompress-bc.c:210-237, decompress-rgtc.c:33-55 (floor /7 and /5, see detex.h:954-976). */
static void ramp8_unsigned(int e0, int e1, int v[8]) {
	v[0] = e0; v[1] = e1;
	if (e0 > e1) {
		for (int k = 1; k <= 6; k++) v[1 + k] = ((7 - k) * e0 + k * e1) / 7;
	} else {
		for (int k = 1; k <= 4; k++) v[1 + k] = ((5 - k) * e0 + k * e1) / 5;
		v[6] = 0; v[7] = 0xFF;
	}
}

/* decompress-bc.c:175-240 */
static int decode_bc3(const uint8_t *in, uint32_t flags, uint8_t *out) {
	const int a0 = in[0], a1 = in[1];
	if (a0 > a1 && (flags & FLAG_OPAQUE_ONLY)) return 0;
	const uint32_t colors = rd32le(in + 8);
	if ((colors & 0xFFFF) <= (colors >> 16) && (flags & FLAG_ENCODE)) return 0;
	int pal[4][3], av[8];
	s3tc_palette(colors, 1, pal);
	ramp8_unsigned(a0, a1, av);
	const uint32_t idx = rd32le(in + 12);
	const uint64_t abits = rd64le(in) >> 16;
	for (int i = 0; i < 16; i++) {
		const int s = (idx >> (2 * i)) & 3;
		wr_rgba(out + 4 * i, pal[s][0], pal[s][1], pal[s][2], av[(abits >> (3 * i)) & 7]);
	}
	return 1;
}

/* ------------------------------------------------------------------------------------------
 * RGTC1/2 (BC4/BC5), unsigned and signed    (decompress-rgtc.c)
 * ---------------------------------------------------------------------------------------- */

/* decompress-rgtc.c:26-60. stride/offset in bytes of the 8-bit output channel. */
static void rgtc_channel_unsigned(const uint8_t *in, uint8_t *out, int stride, int offset) {
	int v[8];
	ramp8_unsigned(in[0], in[1], v);
	const uint64_t bits = rd64le(in) >> 16;
	for (int i = 0; i < 16; i++)
		out[i * stride + offset] = (uint8_t)v[(bits >> (3 * i)) & 7];
}

/* decompress-rgtc.c:84-130. Output is a 16-bit signed channel. C '/' truncates toward zero,
 * which is what detexDivideMinus895To895By7 / Minus639To639By5 compute (detex.h:966-982). */
static int rgtc_channel_signed(const uint8_t *in, uint8_t *out, int stride, int offset) {
	int e0 = (int8_t)in[0], e1 = (int8_t)in[1];
	if (e0 == -127 && e1 == -128) return 0;		/* :90-92 */
	if (e0 == -128) e0 = -127;
	if (e1 == -128) e1 = -127;
	int v[8];
	v[0] = e0; v[1] = e1;
	if (e0 > e1) {
		for (int k = 1; k <= 6; k++) v[1 + k] = ((7 - k) * e0 + k * e1) / 7;
	} else {
		for (int k = 1; k <= 4; k++) v[1 + k] = ((5 - k) * e0 + k * e1) / 5;
		v[6] = -127; v[7] = 127;
	}
	const uint64_t bits = rd64le(in) >> 16;
	for (int i = 0; i < 16; i++) {
		const int r = v[(bits >> (3 * i)) & 7];
		wr16le(out + i * stride + offset, (uint32_t)((r + 127) * 65535 / 254 - 32768));	/* :125-126 */
	}
	return 1;
}

/* ------------------------------------------------------------------------------------------
 * ETC1 / ETC2 / ETC2 punchthrough           (decompress-etc.c)
 * ---------------------------------------------------------------------------------------- */

/* ETC1 intensity modifiers, {small, large} per table codeword (format spec; the reference
 * holds them as signed rows at decompress-etc.c:25-34: {a, b, -a, -b}). */
static const int etc_modifier[8][2] = {
	{ 2, 8 }, { 5, 17 }, { 9, 29 }, { 13, 42 }, { 18, 60 }, { 24, 80 }, { 33, 106 }, { 47, 183 }
};
/* ETC2 T/H distances (decompress-etc.c:200) */
static const int etc_distance[8] = { 3, 6, 11, 16, 23, 32, 41, 64 };

/* 2-bit selector of pixel p (column-major pixel numbering p = x*4 + y): LSB plane in bits 0-15,
 * MSB plane in bits 16-31 of the big-endian index word (decompress-etc.c:75-76). */
static int etc_selector(uint32_t word, int p) { return (int)(((word >> p) & 1) | (((word >> (16 + p)) & 1) << 1)); }
/* pixel p is column-major; the output is row-major (decompress-etc.c:83) */
static int etc_out_index(int p) { return (p & 3) * 4 + (p >> 2); }

static int sext3(int v) { return (v & 4) ? v - 8 : v; }

/* 5-bit base + 3-bit signed delta, both still scaled by 8 (decompress-etc.c:105-110, 337-342).
 * "overflow" (out of 0..31 before scaling) is what the reference tests with `& 0xFF07`. */
static int etc_diff_sum(uint8_t byte) { return (byte & 0xF8) + 8 * sext3(byte & 7); }
static int etc_overflows(int v) { return v < 0 || v > 255; }

/* Individual + differential sub-block decode shared by ETC1, ETC2 and the punchthrough
 * differential mode (decompress-etc.c:89-180 and :503-563).
 *  punch = 1: non-opaque punchthrough block: selector 0/2 modifier is 0 and selector 2 gives
 *  a fully zero pixel (decompress-etc.c:472-499).  Caller has already excluded overflow. */
static void etc_subblocks(const uint8_t *in, uint8_t *out, int differential, int punch) {
	int base[2][3];
	for (int k = 0; k < 3; k++) {
		if (differential) {
			int c0 = in[k] & 0xF8;
			int c1 = etc_diff_sum(in[k]);
			base[0][k] = c0 | (c0 >> 5);
			base[1][k] = c1 | ((c1 & 0xE0) >> 5);
		} else {
			base[0][k] = (in[k] & 0xF0) | (in[k] >> 4);
			base[1][k] = (in[k] & 0x0F) | ((in[k] & 0x0F) << 4);
		}
	}
	const int table[2] = { (in[3] >> 5) & 7, (in[3] >> 2) & 7 };
	const int flip = in[3] & 1;
	const uint32_t word = rd32be(in + 4);
	for (int p = 0; p < 16; p++) {
		const int x = p >> 2, y = p & 3;
		const int sb = flip ? (y >= 2) : (x >= 2);		/* :143-178 */
		const int sel = etc_selector(word, p);
		int mod = etc_modifier[table[sb]][sel & 1];
		if (punch && !(sel & 1)) mod = 0;
		if (sel & 2) mod = -mod;
		uint8_t *o = out + 4 * etc_out_index(p);
		if (punch && sel == 2)
			wr_rgba(o, 0, 0, 0, 0);
		else
			wr_rgba(o, clampi(base[sb][0] + mod, 0, 255), clampi(base[sb][1] + mod, 0, 255),
				clampi(base[sb][2] + mod, 0, 255), 0xFF);
	}
}

/* decompress-etc.c:89-180 */
static int decode_etc1(const uint8_t *in, uint32_t mode_mask, uint8_t *out) {
	const int differential = in[3] & 2;
	if (differential ? !(mode_mask & MASK_ETC_DIFFERENTIAL) : !(mode_mask & MASK_ETC_INDIVIDUAL)) return 0;
	if (differential)
		for (int k = 0; k < 3; k++)
			if (etc_overflows(etc_diff_sum(in[k]))) return 0;	/* :111-122 */
	etc_subblocks(in, out, differential, 0);
	return 1;
}

static int rep4(int v) { return v | (v << 4); }

/* T and H modes, decompress-etc.c:202-285 (opaque) and :565-649 (punchthrough mask). */
static void etc2_th(const uint8_t *in, uint8_t *out, int h_mode, int punch) {
	int c1[3], c2[3], paint[4][3];
	if (!h_mode) {
		c1[0] = rep4(((in[0] & 0x18) >> 1) | (in[0] & 3));
		c1[1] = rep4(in[1] >> 4);
		c1[2] = rep4(in[1] & 0xF);
		c2[0] = rep4(in[2] >> 4);
		c2[1] = rep4(in[2] & 0xF);
		c2[2] = rep4(in[3] >> 4);
		const int d = etc_distance[((in[3] & 0x0C) >> 1) | (in[3] & 1)];
		for (int k = 0; k < 3; k++) {
			paint[0][k] = c1[k];
			paint[1][k] = clampi(c2[k] + d, 0, 255);
			paint[2][k] = c2[k];
			paint[3][k] = clampi(c2[k] - d, 0, 255);
		}
	} else {
		c1[0] = rep4((in[0] & 0x78) >> 3);
		c1[1] = rep4(((in[0] & 7) << 1) | ((in[1] & 0x10) >> 4));
		c1[2] = rep4((in[1] & 8) | ((in[1] & 3) << 1) | (in[2] >> 7));
		c2[0] = rep4((in[2] & 0x78) >> 3);
		c2[1] = rep4(((in[2] & 7) << 1) | (in[3] >> 7));
		c2[2] = rep4((in[3] & 0x78) >> 3);
		const int v1 = (c1[0] << 16) + (c1[1] << 8) + c1[2];
		const int v2 = (c2[0] << 16) + (c2[1] << 8) + c2[2];
		const int d = etc_distance[(in[3] & 4) | ((in[3] & 1) << 1) | (v1 >= v2)];
		for (int k = 0; k < 3; k++) {
			paint[0][k] = clampi(c1[k] + d, 0, 255);
			paint[1][k] = clampi(c1[k] - d, 0, 255);
			paint[2][k] = clampi(c2[k] + d, 0, 255);
			paint[3][k] = clampi(c2[k] - d, 0, 255);
		}
	}
	const uint32_t word = rd32be(in + 4);
	for (int p = 0; p < 16; p++) {
		const int sel = etc_selector(word, p);
		uint8_t *o = out + 4 * etc_out_index(p);
		if (punch && sel == 2)
			wr_rgba(o, 0, 0, 0, 0);
		else
			wr_rgba(o, paint[sel][0], paint[sel][1], paint[sel][2], 0xFF);
	}
}

/* decompress-etc.c:287-317 */
static void etc2_planar(const uint8_t *in, uint8_t *out) {
	int o[3], h[3], v[3];
	o[0] = (in[0] & 0x7E) >> 1;
	o[1] = ((in[0] & 1) << 6) | ((in[1] & 0x7E) >> 1);
	o[2] = ((in[1] & 1) << 5) | (in[2] & 0x18) | ((in[2] & 3) << 1) | (in[3] >> 7);
	h[0] = ((in[3] & 0x7C) >> 1) | (in[3] & 1);
	h[1] = in[4] >> 1;
	h[2] = ((in[4] & 1) << 5) | (in[5] >> 3);
	v[0] = ((in[5] & 7) << 3) | (in[6] >> 5);
	v[1] = ((in[6] & 0x1F) << 2) | (in[7] >> 6);
	v[2] = in[7] & 0x3F;
	/* 6-7-6 -> 8 bits by MSB replication */
	o[0] = (o[0] << 2) | (o[0] >> 4); o[1] = (o[1] << 1) | (o[1] >> 6); o[2] = (o[2] << 2) | (o[2] >> 4);
	h[0] = (h[0] << 2) | (h[0] >> 4); h[1] = (h[1] << 1) | (h[1] >> 6); h[2] = (h[2] << 2) | (h[2] >> 4);
	v[0] = (v[0] << 2) | (v[0] >> 4); v[1] = (v[1] << 1) | (v[1] >> 6); v[2] = (v[2] << 2) | (v[2] >> 4);
	for (int y = 0; y < 4; y++)
		for (int x = 0; x < 4; x++) {
			int c[3];
			for (int k = 0; k < 3; k++) {
				/* arithmetic shift of a possibly negative sum, as in the reference (:311-313) */
				int s = x * (h[k] - o[k]) + y * (v[k] - o[k]) + 4 * o[k] + 2;
				s = (s >= 0) ? (s >> 2) : -((-s + 3) >> 2);
				c[k] = clampi(s, 0, 255);
			}
			wr_rgba(out + 4 * (y * 4 + x), c[0], c[1], c[2], 0xFF);
		}
}

/* 0 individual, 1 differential, 2 T, 3 H, 4 planar  (decompress-etc.c:370-395) */
static int etc2_mode(const uint8_t *in, int has_individual) {
	if (has_individual && !(in[3] & 2)) return 0;
	if (etc_overflows(etc_diff_sum(in[0]))) return 2;
	if (etc_overflows(etc_diff_sum(in[1]))) return 3;
	if (etc_overflows(etc_diff_sum(in[2]))) return 4;
	return 1;
}

/* decompress-etc.c:321-367 */
static int decode_etc2(const uint8_t *in, uint32_t mode_mask, uint8_t *out) {
	if (!(in[3] & 2)) return decode_etc1(in, mode_mask, out);
	if ((mode_mask & ~MASK_ETC_INDIVIDUAL) == 0) return 0;
	switch (etc2_mode(in, 1)) {
	case 2: if (!(mode_mask & MASK_ETC_T)) return 0; etc2_th(in, out, 0, 0); return 1;
	case 3: if (!(mode_mask & MASK_ETC_H)) return 0; etc2_th(in, out, 1, 0); return 1;
	case 4: if (!(mode_mask & MASK_ETC_PLANAR)) return 0; etc2_planar(in, out); return 1;
	default: return decode_etc1(in, mode_mask, out);
	}
}

/* decompress-etc.c:653-717 */
static int decode_etc2_punchthrough(const uint8_t *in, uint32_t mode_mask, uint32_t flags, uint8_t *out) {
	const int opaque = in[3] & 2;
	if (opaque && (flags & FLAG_NON_OPAQUE_ONLY)) return 0;
	if (!opaque && (flags & FLAG_OPAQUE_ONLY)) return 0;
	switch (etc2_mode(in, 0)) {
	case 2: if (!(mode_mask & MASK_ETC_T)) return 0; etc2_th(in, out, 0, !opaque); return 1;
	case 3: if (!(mode_mask & MASK_ETC_H)) return 0; etc2_th(in, out, 1, !opaque); return 1;
	case 4:
		if (!(mode_mask & MASK_ETC_PLANAR)) return 0;
		if (flags & FLAG_NON_OPAQUE_ONLY) return 0;
		etc2_planar(in, out);
		return 1;
	default:
		if (opaque) return decode_etc1(in, mode_mask, out);	/* differential bit == opaque bit */
		if (!(mode_mask & MASK_ETC_DIFFERENTIAL)) return 0;
		etc_subblocks(in, out, 1, 1);
		return 1;
	}
}

/* ------------------------------------------------------------------------------------------
 * EAC: ETC2_EAC alpha, R11 / RG11 unsigned and signed      (decompress-eac.c)
 * ---------------------------------------------------------------------------------------- */

/* EAC modifier table (format spec; reference row order is {negatives..., positives...} at
 * decompress-eac.c:21-38).  Held here as the four magnitudes m0..m3 per table: entries
 * 0..3 are -m[k]-1+... -- see eac_modifier() below: idx 0..3 -> -(m[k]), idx 4..7 -> m[k]-1. */
static const unsigned char eac_magnitude[16][4] = {
	{ 3, 6, 9, 15 }, { 3, 7, 10, 13 }, { 2, 5, 8, 13 }, { 2, 4, 6, 13 },
	{ 3, 6, 8, 12 }, { 3, 7, 9, 11 }, { 4, 7, 8, 11 }, { 3, 5, 8, 11 },
	{ 2, 6, 8, 10 }, { 2, 5, 8, 10 }, { 2, 4, 8, 10 }, { 2, 5, 7, 10 },
	{ 3, 4, 7, 10 }, { 1, 2, 3, 10 }, { 4, 6, 8, 9 }, { 3, 5, 7, 9 }
};
static int eac_modifier(int table, int idx) {
	const int m = eac_magnitude[table][idx & 3];
	return (idx & 4) ? m - 1 : -m;
}

/* decompress-eac.c:54-86: ETC2 colour from bytes 8..15, then the 8-bit alpha overlay. */
static int decode_etc2_eac(const uint8_t *in, uint32_t mode_mask, uint32_t flags, uint8_t *out) {
	if (!decode_etc2(in + 8, mode_mask, out)) return 0;
	const int base = in[0], table = in[1] & 0xF, mult = in[1] >> 4;
	if (mult == 0 && (flags & FLAG_ENCODE)) return 0;
	const uint64_t bits = rd64be(in) & 0xFFFFFFFFFFFFull;
	for (int p = 0; p < 16; p++) {
		const int idx = (int)((bits >> (45 - 3 * p)) & 7);
		out[4 * etc_out_index(p) + 3] = (uint8_t)clampi(base + eac_modifier(table, idx) * mult, 0, 255);
	}
	return 1;
}

/* decompress-eac.c:111-128 */
static void eac11_unsigned(const uint8_t *in, uint8_t *out, int stride, int offset) {
	const uint64_t q = rd64be(in);
	const int base = (int)(q >> 56) * 8 + 4;
	const int table = (int)(q >> 48) & 0xF;
	int mult8 = ((int)(q >> 52) & 0xF) * 8;
	if (mult8 == 0) mult8 = 1;
	for (int p = 0; p < 16; p++) {
		const int idx = (int)((q >> (45 - 3 * p)) & 7);
		const uint32_t v = (uint32_t)clampi(base + eac_modifier(table, idx) * mult8, 0, 2047);
		wr16le(out + etc_out_index(p) * stride + offset, (v << 5) | (v >> 6));
	}
}

/* decompress-eac.c:159-201 */
static int eac11_signed(const uint8_t *in, uint8_t *out, int stride, int offset) {
	const uint64_t q = rd64be(in);
	const int base = (int8_t)(q >> 56);
	if (base == -128) return 0;				/* :183-185 */
	const int table = (int)(q >> 48) & 0xF;
	int mult8 = ((int)(q >> 52) & 0xF) * 8;
	if (mult8 == 0) mult8 = 1;
	for (int p = 0; p < 16; p++) {
		const int idx = (int)((q >> (45 - 3 * p)) & 7);
		const int v = clampi(base * 8 + eac_modifier(table, idx) * mult8, -1023, 1023);
		const int m = v < 0 ? -v : v;
		const int wide = (m << 5) | (m >> 5);		/* :159-165, sign-symmetric replication */
		wr16le(out + etc_out_index(p) * stride + offset, (uint32_t)(v < 0 ? -wide : wide));
	}
	return 1;
}

/* ------------------------------------------------------------------------------------------
 * 128-bit little-endian bit reader for BPTC / BPTC_FLOAT    (bits.h, bits.c)
 * ---------------------------------------------------------------------------------------- */
typedef struct { uint64_t lo, hi; int pos; } bitreader;

static uint32_t bit_at(const bitreader *b, int pos) {
	return (uint32_t)((pos < 64 ? b->lo >> pos : b->hi >> (pos - 64)) & 1);
}
static uint32_t take(bitreader *b, int n) {		/* LSB first, as bits.c:22-44 */
	uint32_t v = 0;
	for (int i = 0; i < n; i++) v |= bit_at(b, b->pos++) << i;
	return v;
}

static int bptc_weight(int index, int bits) {		/* bptc-tables.c aWeight2/3/4, closed form */
	const int d = (1 << bits) - 1;
	return (64 * index + d / 2) / d;
}

/* ------------------------------------------------------------------------------------------
 * BPTC (BC7)                               (decompress-bptc.c)
 * ---------------------------------------------------------------------------------------- */
/* per-mode layout, decompress-bptc.c:24-43,45-71,134,195-225,265-267 */
static const struct { int ns, pb, rb, isb, cb, ab, epb, spb, ib, ib2; } bc7_mode[8] = {
	/*      subsets part rot isel  col alpha  endpoint-pbit shared-pbit  idx idx2 */
	{ 3, 4, 0, 0, 4, 0, 1, 0, 3, 0 },
	{ 2, 6, 0, 0, 6, 0, 0, 1, 3, 0 },
	{ 3, 6, 0, 0, 5, 0, 0, 0, 2, 0 },
	{ 2, 6, 0, 0, 7, 0, 1, 0, 2, 0 },
	{ 1, 0, 2, 1, 5, 6, 0, 0, 2, 3 },
	{ 1, 0, 2, 0, 7, 8, 0, 0, 2, 2 },
	{ 1, 0, 0, 0, 7, 7, 1, 0, 4, 0 },
	{ 2, 6, 0, 0, 5, 5, 1, 0, 2, 0 },
};

static int bc7_block_mode(const uint8_t *in) {		/* decompress-bptc.c:229-237 */
	for (int m = 0; m < 8; m++)
		if (in[0] & (1 << m)) return m;
	return -1;
}

static int expand_to_8(int v, int prec) {		/* decompress-bptc.c:160-175 */
	v <<= 8 - prec;
	return (v | (v >> prec)) & 0xFF;
}

/* decompress-bptc.c:354-512 (generic path) and :271-350 (mode 1 fast path; same result) */
static int decode_bptc(const uint8_t *in, uint32_t mode_mask, uint32_t flags, uint8_t *out) {
	const int mode = bc7_block_mode(in);
	if (mode < 0) return 0;
	if (!(mode_mask & (1u << mode))) return 0;
	if (mode >= 4 && (flags & FLAG_OPAQUE_ONLY)) return 0;
	if (mode < 4 && (flags & FLAG_NON_OPAQUE_ONLY)) return 0;
	bitreader br = { rd64le(in), rd64le(in + 8), mode + 1 };
	const int ns = bc7_mode[mode].ns, cb = bc7_mode[mode].cb, ab = bc7_mode[mode].ab;
	const int part = (int)take(&br, bc7_mode[mode].pb);
	const int rotation = (int)take(&br, bc7_mode[mode].rb);
	const int isel = (int)take(&br, bc7_mode[mode].isb);
	int ep[6][4];					/* [subset*2 + endpoint][rgba] */
	for (int c = 0; c < 3; c++)
		for (int e = 0; e < 2 * ns; e++) ep[e][c] = (int)take(&br, cb);
	for (int e = 0; e < 2 * ns; e++) ep[e][3] = ab ? (int)take(&br, ab) : 0;
	int cprec = cb, aprec = ab;
	if (bc7_mode[mode].epb) {
		for (int e = 0; e < 2 * ns; e++) {
			/* QUIRK A-2: in mode 6 the reference takes both P-bits from (data0 >> 63), so the
			 * second one (block bit 64) reads as 0 (decompress-bptc.c:142-146). */
			int p = (mode == 6 && e == 1 && (orc_quirks & ORC_QUIRK_BC7_MODE6_PBIT)) ? 0 : (int)bit_at(&br, br.pos + e);
			for (int c = 0; c < 4; c++) ep[e][c] = (ep[e][c] << 1) | p;
		}
		br.pos += 2 * ns;
		cprec++; aprec++;
	} else if (bc7_mode[mode].spb) {
		for (int s = 0; s < ns; s++) {
			const int p = (int)take(&br, 1);
			for (int c = 0; c < 3; c++) {
				ep[2 * s][c] = (ep[2 * s][c] << 1) | p;
				ep[2 * s + 1][c] = (ep[2 * s + 1][c] << 1) | p;
			}
		}
		cprec++;
	}
	for (int e = 0; e < 2 * ns; e++) {
		for (int c = 0; c < 3; c++) ep[e][c] = expand_to_8(ep[e][c], cprec);
		ep[e][3] = (mode <= 3) ? 0xFF : expand_to_8(ep[e][3], aprec);
	}
	int subset[16], anchor[3] = { 0, 0, 0 };
	for (int i = 0; i < 16; i++)
		subset[i] = ns == 1 ? 0 : (ns == 2 ? orc_partition2[part][i] - '0' : orc_partition3[part][i] - '0');
	if (ns == 2) anchor[1] = orc_anchor2[part];
	if (ns == 3) { anchor[1] = orc_anchor3_1[part]; anchor[2] = orc_anchor3_2[part]; }
	const int ib = bc7_mode[mode].ib, ib2 = bc7_mode[mode].ib2;
	int idx1[16], idx2[16];
	for (int i = 0; i < 16; i++) idx1[i] = (int)take(&br, i == anchor[subset[i]] ? ib - 1 : ib);
	for (int i = 0; i < 16; i++) idx2[i] = ib2 ? (int)take(&br, i == 0 ? ib2 - 1 : ib2) : idx1[i];
	/* decompress-bptc.c:374-375, 452-480: with the index-selection bit set the colour uses the
	 * secondary (3-bit) indices and alpha the primary (2-bit) ones. */
	const int cbits = (ib2 && isel) ? ib2 : ib;
	const int abits = ib2 ? (isel ? ib : ib2) : ib;
	for (int i = 0; i < 16; i++) {
		const int *e0 = ep[2 * subset[i]], *e1 = ep[2 * subset[i] + 1];
		const int ci = (ib2 && isel) ? idx2[i] : idx1[i];
		const int ai = ib2 ? (isel ? idx1[i] : idx2[i]) : idx1[i];
		const int wc = bptc_weight(ci, cbits), wa = bptc_weight(ai, abits);
		int px[4];
		for (int c = 0; c < 3; c++) px[c] = ((64 - wc) * e0[c] + wc * e1[c] + 32) >> 6;
		px[3] = ((64 - wa) * e0[3] + wa * e1[3] + 32) >> 6;
		if (rotation) {					/* :497-508 */
			const int t = px[3];
			px[3] = px[rotation - 1];
			px[rotation - 1] = t;
		}
		wr_rgba(out + 4 * i, px[0], px[1], px[2], px[3]);
	}
	return 1;
}

/* ------------------------------------------------------------------------------------------
 * BPTC_FLOAT (BC6H), unsigned and signed   (decompress-bptc-float.c)
 * ---------------------------------------------------------------------------------------- */

/* Bit layouts in the notation of the BPTC spec, in stream order after the mode bits; the
 * reference carries the same strings as comments (decompress-bptc-float.c:130-131, ...).
 * "xN[a:b]" = component x of endpoint N, stream delivers bit b first, then towards a.
 * Modes 0-9: two subsets (5 partition bits follow at bit 77), modes 10-13: one subset. */
static const char *const bc6h_layout[14] = {
	"g2[4],b2[4],b3[4],r0[9:0],g0[9:0],b0[9:0],r1[4:0],g3[4],g2[3:0],g1[4:0],b3[0],g3[3:0],b1[4:0],b3[1],b2[3:0],r2[4:0],b3[2],r3[4:0],b3[3]",
	"g2[5],g3[4],g3[5],r0[6:0],b3[0],b3[1],b2[4],g0[6:0],b2[5],b3[2],g2[4],b0[6:0],b3[3],b3[5],b3[4],r1[5:0],g2[3:0],g1[5:0],g3[3:0],b1[5:0],b2[3:0],r2[5:0],r3[5:0]",
	"r0[9:0],g0[9:0],b0[9:0],r1[4:0],r0[10],g2[3:0],g1[3:0],g0[10],b3[0],g3[3:0],b1[3:0],b0[10],b3[1],b2[3:0],r2[4:0],b3[2],r3[4:0],b3[3]",
	"r0[9:0],g0[9:0],b0[9:0],r1[3:0],r0[10],g3[4],g2[3:0],g1[4:0],g0[10],g3[3:0],b1[3:0],b0[10],b3[1],b2[3:0],r2[3:0],b3[0],b3[2],r3[3:0],g2[4],b3[3]",
	"r0[9:0],g0[9:0],b0[9:0],r1[3:0],r0[10],b2[4],g2[3:0],g1[3:0],g0[10],b3[0],g3[3:0],b1[4:0],b0[10],b2[3:0],r2[3:0],b3[1],b3[2],r3[3:0],b3[4],b3[3]",
	"r0[8:0],b2[4],g0[8:0],g2[4],b0[8:0],b3[4],r1[4:0],g3[4],g2[3:0],g1[4:0],b3[0],g3[3:0],b1[4:0],b3[1],b2[3:0],r2[4:0],b3[2],r3[4:0],b3[3]",
	"r0[7:0],g3[4],b2[4],g0[7:0],b3[2],g2[4],b0[7:0],b3[3],b3[4],r1[5:0],g2[3:0],g1[4:0],b3[0],g3[3:0],b1[4:0],b3[1],b2[3:0],r2[5:0],r3[5:0]",
	"r0[7:0],b3[0],b2[4],g0[7:0],g2[5],g2[4],b0[7:0],g3[5],b3[4],r1[4:0],g3[4],g2[3:0],g1[5:0],g3[3:0],b1[4:0],b3[1],b2[3:0],r2[4:0],b3[2],r3[4:0],b3[3]",
	"r0[7:0],b3[1],b2[4],g0[7:0],b2[5],g2[4],b0[7:0],b3[5],b3[4],r1[4:0],g3[4],g2[3:0],g1[4:0],b3[0],g3[3:0],b1[5:0],b2[3:0],r2[4:0],b3[2],r3[4:0],b3[3]",
	"r0[5:0],g3[4],b3[0],b3[1],b2[4],g0[5:0],g2[5],b2[5],b3[2],g2[4],b0[5:0],g3[5],b3[3],b3[5],b3[4],r1[5:0],g2[3:0],g1[5:0],g3[3:0],b1[5:0],b2[3:0],r2[5:0],r3[5:0]",
	"r0[9:0],g0[9:0],b0[9:0],r1[9:0],g1[9:0],b1[9:0]",
	"r0[9:0],g0[9:0],b0[9:0],r1[8:0],r0[10],g1[8:0],g0[10],b1[8:0],b0[10]",
	"r0[9:0],g0[9:0],b0[9:0],r1[7:0],r0[10:11],g1[7:0],g0[10:11],b1[7:0],b0[10:11]",
	"r0[9:0],g0[9:0],b0[9:0],r1[3:0],r0[10:15],g1[3:0],g0[10:15],b1[3:0],b0[10:15]",
};
/* endpoint precision (decompress-bptc-float.c:42-43) and delta widths r,g,b (0 = not transformed) */
static const int bc6h_epb[14] = { 10, 7, 11, 11, 11, 9, 8, 8, 8, 6, 10, 11, 12, 16 };
static const int bc6h_delta[14][3] = {
	{ 5, 5, 5 }, { 6, 6, 6 }, { 5, 4, 4 }, { 4, 5, 4 }, { 4, 4, 5 }, { 5, 5, 5 }, { 6, 5, 5 },
	{ 5, 6, 5 }, { 5, 5, 6 }, { 0, 0, 0 }, { 0, 0, 0 }, { 9, 9, 9 }, { 8, 8, 8 }, { 4, 4, 4 }
};

/* decompress-bptc-float.c:23-33: 2-bit codes 00/01 are modes 0/1, otherwise a 5-bit code */
static int bc6h_block_mode(const uint8_t *in, int *mode_bits) {
	const int low2 = in[0] & 3, low5 = in[0] & 0x1F;
	if (low2 < 2) { *mode_bits = 2; return low2; }
	*mode_bits = 5;
	if (low2 == 2) return 2 + (low5 >> 2);		/* 00010 .. 11110 -> modes 2..9 */
	return (low5 >> 2) < 4 ? 10 + (low5 >> 2) : -1;	/* 00011,00111,01011,01111 -> 10..13 */
}

static int sign_extend(int v, int bits) {		/* decompress-bptc-float.c:88-95 */
	return (v & (1 << (bits - 1))) ? (int)((uint32_t)v | ~((1u << bits) - 1)) : v;
}

static void bc6h_scatter(const uint8_t *in, int mode, int mode_bits, int comp[3][4]) {
	bitreader br = { rd64le(in), rd64le(in + 8), mode_bits };
	memset(comp, 0, 12 * sizeof(int));
	for (const char *s = bc6h_layout[mode]; *s; ) {
		const int c = (*s == 'r') ? 0 : (*s == 'g') ? 1 : 2;
		const int e = s[1] - '0';
		int a = 0, b;
		s += 3;					/* past "xN[" */
		while (*s >= '0' && *s <= '9') a = a * 10 + (*s++ - '0');
		b = a;
		if (*s == ':') { s++; b = 0; while (*s >= '0' && *s <= '9') b = b * 10 + (*s++ - '0'); }
		s++;					/* past "]" */
		if (*s == ',') s++;
		for (int k = b; ; k += (a >= b ? 1 : -1)) {
			uint32_t bit = bit_at(&br, br.pos);
			/* QUIRK A-3: mode 12 loses b0[11] = block bit 63 (decompress-bptc-float.c:462,
			 * UB shift in bits.h:29-31 as compiled by gcc >= -O2). */
			if (mode == 12 && br.pos == 63 && (orc_quirks & ORC_QUIRK_BC6H_MODE12_BIT63)) bit = 0;
			br.pos++;
			comp[c][e] |= (int)(bit << k);
			if (k == a) break;
		}
	}
}

static int bc6h_unquantize_unsigned(int x, int mode) {	/* decompress-bptc-float.c:52-63 */
	const int epb = bc6h_epb[mode];
	x &= 0xFFFF;
	if (mode == 13) return x;
	if (x == 0) return 0;
	if (x == (1 << epb) - 1) return 0xFFFF;
	return ((x << 15) + 0x4000) >> (epb - 1);
}

static int bc6h_unquantize_signed(int x, int mode) {	/* decompress-bptc-float.c:65-86 */
	const int epb = bc6h_epb[mode];
	x = (int16_t)x;
	if (epb >= 16) return x;
	const int neg = x < 0;
	if (neg) x = -x;
	int unq;
	if (x == 0) unq = 0;
	else if (x >= (1 << (epb - 1)) - 1) unq = 0x7FFF;
	else unq = ((x << 15) + 0x4000) >> (epb - 1);
	return neg ? -unq : unq;
}

/* decompress-bptc-float.c:110-626 */
static int decode_bptc_float(const uint8_t *in, uint32_t mode_mask, int is_signed, uint8_t *out) {
	int mode_bits;
	const int mode = bc6h_block_mode(in, &mode_bits);
	if (mode < 0) return 0;
	if (!(mode_mask & (1u << mode))) return 0;
	int ep[3][4];					/* [rgb][endpoint 0..3] */
	bc6h_scatter(in, mode, mode_bits, ep);
	const int ns = mode >= 10 ? 1 : 2, epb = bc6h_epb[mode];
	bitreader br = { rd64le(in), rd64le(in + 8), ns == 2 ? 77 : 65 };
	const int part = ns == 2 ? (int)take(&br, 5) : 0;
	for (int c = 0; c < 3; c++) {
		if (is_signed) ep[c][0] = sign_extend(ep[c][0], epb);
		for (int e = 1; e < 2 * ns; e++) {
			if (bc6h_delta[mode][c]) {		/* :496-510 transformed endpoints */
				ep[c][e] = sign_extend(ep[c][e], bc6h_delta[mode][c]);
				ep[c][e] = (int)((uint32_t)(ep[c][0] + ep[c][e]) & ((1u << epb) - 1));
			}
			if (is_signed) ep[c][e] = sign_extend(ep[c][e], epb);
		}
		for (int e = 0; e < 2 * ns; e++)
			ep[c][e] = is_signed ? bc6h_unquantize_signed(ep[c][e], mode) : bc6h_unquantize_unsigned(ep[c][e], mode);
	}
	const int ibits = ns == 1 ? 4 : 3;		/* :548-550 */
	const int anchor1 = ns == 2 ? orc_anchor2[part] : 0;
	for (int i = 0; i < 16; i++) {
		const int s = ns == 2 ? orc_partition2[part][i] - '0' : 0;
		const int is_anchor = (i == (s ? anchor1 : 0));
		const int w = bptc_weight((int)take(&br, is_anchor ? ibits - 1 : ibits), ibits);
		for (int c = 0; c < 3; c++) {
			int v = ((64 - w) * ep[c][2 * s] + w * ep[c][2 * s + 1] + 32) >> 6;	/* :97-108 */
			uint32_t h;
			if (is_signed) {			/* :576-609 sign-magnitude half */
				const int neg = v < 0;
				const int m = ((neg ? -v : v) * 31) >> 5;
				h = (uint32_t)m | ((neg && m != 0) ? 0x8000u : 0u);
			} else {
				h = (uint32_t)(v * 31 / 64);		/* :613-621 */
			}
			wr16le(out + 8 * i + 2 * c, h);
		}
		wr16le(out + 8 * i + 6, 0);			/* X = 0 (FLOAT_RGBX16) */
	}
	return 1;
}

/* ------------------------------------------------------------------------------------------
 * Dispatch + texture drivers                (texture.c)
 * ---------------------------------------------------------------------------------------- */
static const unsigned char fmt_block_bytes[ORC_FORMAT_COUNT] = {
	0, 8, 8, 16, 16, 8, 8, 16, 16, 16, 16, 16, 8, 8, 8, 16, 8, 8, 16, 16 };
static const unsigned char fmt_pixel_bytes[ORC_FORMAT_COUNT] = {
	0, 4, 4, 4, 4, 1, 2, 2, 4, 8, 8, 4, 4, 4, 4, 4, 2, 2, 4, 4 };

int orc_block_bytes(int fmt) { return (fmt > 0 && fmt < ORC_FORMAT_COUNT) ? fmt_block_bytes[fmt] : 0; }
int orc_pixel_bytes(int fmt) { return (fmt > 0 && fmt < ORC_FORMAT_COUNT) ? fmt_pixel_bytes[fmt] : 0; }

/* texture.c:27-48 function table */
int orc_decode_block(int fmt, const uint8_t *in, uint32_t mode_mask, uint32_t flags, uint8_t *out) {
	switch (fmt) {
	case ORC_BC1: return decode_bc1(in, flags, out, 0);
	case ORC_BC1A: return decode_bc1(in, flags, out, 1);
	case ORC_BC2: return decode_bc2(in, flags, out);
	case ORC_BC3: return decode_bc3(in, flags, out);
	case ORC_RGTC1: rgtc_channel_unsigned(in, out, 1, 0); return 1;
	case ORC_RGTC2: rgtc_channel_unsigned(in, out, 2, 0); rgtc_channel_unsigned(in + 8, out, 2, 1); return 1;
	case ORC_SIGNED_RGTC1: return rgtc_channel_signed(in, out, 2, 0);
	case ORC_SIGNED_RGTC2: return rgtc_channel_signed(in, out, 4, 0) && rgtc_channel_signed(in + 8, out, 4, 2);
	case ORC_BPTC_FLOAT: return decode_bptc_float(in, mode_mask, 0, out);
	case ORC_BPTC_SIGNED_FLOAT: return decode_bptc_float(in, mode_mask, 1, out);
	case ORC_BPTC: return decode_bptc(in, mode_mask, flags, out);
	case ORC_ETC1: return decode_etc1(in, mode_mask, out);
	case ORC_ETC2: return decode_etc2(in, mode_mask, out);
	case ORC_ETC2_PUNCHTHROUGH: return decode_etc2_punchthrough(in, mode_mask, flags, out);
	case ORC_ETC2_EAC: return decode_etc2_eac(in, mode_mask, flags, out);
	case ORC_EAC_R11: eac11_unsigned(in, out, 2, 0); return 1;
	case ORC_EAC_RG11: eac11_unsigned(in, out, 4, 0); eac11_unsigned(in + 8, out, 4, 2); return 1;
	case ORC_EAC_SIGNED_R11: return eac11_signed(in, out, 2, 0);
	case ORC_EAC_SIGNED_RG11: return eac11_signed(in, out, 4, 0) && eac11_signed(in + 8, out, 4, 2);
	default: return 0;
	}
}

/* texture.c:105-145 */
int orc_decompress_linear(int fmt, const uint8_t *data, int width, int height,
		int width_in_blocks, int height_in_blocks, uint8_t *pixel_buffer) {
	const int bs = orc_block_bytes(fmt), px = orc_pixel_bytes(fmt);
	if (!bs) return 0;
	int ok = 1;
	uint8_t block[256];
	for (int by = 0; by < height_in_blocks; by++) {
		const int rows = (by * 4 + 3 >= height) ? height - by * 4 : 4;
		for (int bx = 0; bx < width_in_blocks; bx++, data += bs) {
			if (!orc_decode_block(fmt, data, 0xFFFFFFFFu, 0, block)) {
				ok = 0;
				memset(block, 0, (size_t)(16 * px));
			}
			const int cols = (bx * 4 + 3 >= width) ? width - bx * 4 : 4;
			for (int r = 0; r < rows; r++)
				if (cols > 0)
					memcpy(pixel_buffer + ((size_t)(by * 4 + r) * (size_t)width + (size_t)bx * 4) * (size_t)px,
						block + r * 4 * px, (size_t)(cols * px));
		}
	}
	return ok;
}

/* texture.c:77-98 */
int orc_decompress_tiled(int fmt, const uint8_t *data, int width_in_blocks, int height_in_blocks,
		uint8_t *pixel_buffer) {
	const int bs = orc_block_bytes(fmt), px = orc_pixel_bytes(fmt);
	if (!bs) return 0;
	int ok = 1;
	const long n = (long)width_in_blocks * height_in_blocks;
	for (long i = 0; i < n; i++, data += bs, pixel_buffer += 16 * px)
		if (!orc_decode_block(fmt, data, 0xFFFFFFFFu, 0, pixel_buffer)) {
			ok = 0;
			memset(pixel_buffer, 0, (size_t)(16 * px));
		}
	return ok;
}

int orc_block_mode(int fmt, const uint8_t *in) {
	int mb;
	switch (fmt) {
	case ORC_BC1: case ORC_BC1A: { uint32_t c = rd32le(in); return (c & 0xFFFF) > (c >> 16) ? 0 : 1; }
	case ORC_BC2: case ORC_BC3: { uint32_t c = rd32le(in + 8); return (c & 0xFFFF) > (c >> 16) ? 0 : 1; }
	case ORC_ETC1: return (in[3] & 2) ? 1 : 0;
	case ORC_ETC2: return etc2_mode(in, 1);
	case ORC_ETC2_PUNCHTHROUGH: return etc2_mode(in, 0);
	case ORC_ETC2_EAC: return etc2_mode(in + 8, 1);
	case ORC_BPTC: return bc7_block_mode(in);
	case ORC_BPTC_FLOAT: case ORC_BPTC_SIGNED_FLOAT: return bc6h_block_mode(in, &mb);
	default: return 0;
	}
}

/* batch helpers for the Python tests (no reference counterpart) */
void orc_block_modes(int fmt, const uint8_t *data, long n_blocks, int32_t *modes_out) {
	const int bs = orc_block_bytes(fmt);
	for (long i = 0; i < n_blocks; i++) modes_out[i] = orc_block_mode(fmt, data + i * bs);
}

void orc_decode_blocks(int fmt, const uint8_t *data, long n_blocks, uint32_t mode_mask, uint32_t flags,
		uint8_t *pixel_buffer, uint8_t *ok_out) {
	const int bs = orc_block_bytes(fmt), px = orc_pixel_bytes(fmt);
	for (long i = 0; i < n_blocks; i++)
		ok_out[i] = (uint8_t)orc_decode_block(fmt, data + i * bs, mode_mask, flags, pixel_buffer + i * 16 * px);
}

/* FNV-1a-64 over a byte range (digest helper for the full-size stream goldens) */
uint64_t orc_fnv1a64(const uint8_t *p, size_t n) {
	uint64_t h = 0xcbf29ce484222325ull;
	for (size_t i = 0; i < n; i++) h = (h ^ p[i]) * 0x100000001b3ull;
	return h;
}

/* ------------------------------------------------------------------------------------------
 * The pixel-format conversions the GPU path offers as in-kernel epilogues (SURVEY.md 8f-2 and the
 * targets the reference's callers request, validate.c:204-209 / detex-view.c:182 / detex-convert.c:283-284).
 * Each is the composition detexMatchConversion (convert.c:885-1063) picks for that (source, target):
 *   kind 1  R<->B swap of 8-bit RGBA, 4th component kept (convert.c:37-52)
 *   kind 2  RGBX8 -> RGB8 (convert.c:671-684)
 *   kind 3  R<->B swap of 16-bit RGBX (convert.c:54-70)
 *   kind 4/5/6  native R8 | RG8 | R16 | RG16 | SIGNED_R16 | SIGNED_RG16 | FLOAT_RGBX16 -> RGBX8 | BGRX8 | RGB8:
 *       signed 16 -> unsigned 16 by + 32768 (convert.c:158-181); 16 -> 8 by (x + 127) * 255 / 65535
 *       (convert.c:258-281, 299-313); half -> 16 by lrintf(clamp01(f) * 65535.0f + 0.5f) with FE_DOWNWARD
 *       (half-float.c:304-312); 1/2 components -> RGBX8 with 0 for the missing ones and 0xFF for X
 *       (convert.c:219-243); then kind 1 / kind 2 as above.
 * native_pf is the decoder's pixel format (detex.h:83-379).  Pinned to the compiled reference component by
 * component (all 65536 values) in tests/test_oracle_pin.py.
 * ---------------------------------------------------------------------------------------- */
#include <fenv.h>
#include <math.h>
static float orc_half_to_float(uint16_t h) {	/* exact (what half-float.c's table holds) */
	const uint32_t sign = h >> 15, exponent = (h >> 10) & 31u, mantissa = h & 1023u;
	float f;
	if (exponent == 31u) f = mantissa ? NAN : INFINITY;
	else if (exponent == 0u) f = ldexpf((float)mantissa, -24);
	else f = ldexpf((float)(mantissa + 1024u), (int)exponent - 25);
	return sign ? -f : f;
}
static uint32_t orc_16_to_8(uint32_t x) { return (x + 127u) * 255u / 65535u; }
static uint32_t orc_half_to_16(uint16_t h) {
	volatile float f = orc_half_to_float(h);	/* volatile: the arithmetic below must run under the rounding mode set here */
	const int saved = fegetround();
	fesetround(FE_DOWNWARD);
	/* NaN: the reference build (gcc -Ofast: comparisons assume no NaN) yields 1.0 for every NaN pattern -- pinned to it
	 * over all 65536 patterns; BC6H itself never decodes to Inf or NaN (largest half 0x7BFF) */
	volatile float c = f != f ? 1.0f : (f < 0.0f ? 0.0f : (f > 1.0f ? 1.0f : f));
	volatile float t = c * 65535.0f;
	t = t + 0.5f;
	const long u = lrintf(t);
	fesetround(saved);
	return (uint32_t)u & 0xFFFFu;
}
/* pixel i of `in` (native format) as R, G, B bytes */
static int orc_native_rgb(uint32_t native_pf, const uint8_t *in, long i, uint8_t rgb[3]) {
	uint16_t v[4];
	rgb[0] = rgb[1] = rgb[2] = 0;
	switch (native_pf) {
	case 0x0000: rgb[0] = in[i]; return 0;						/* R8 */
	case 0x0110: rgb[0] = in[2 * i]; rgb[1] = in[2 * i + 1]; return 0;		/* RG8 */
	case 0x0101: memcpy(v, in + 2 * i, 2); rgb[0] = (uint8_t)orc_16_to_8(v[0]); return 0;	/* R16 */
	case 0x1101: memcpy(v, in + 2 * i, 2); rgb[0] = (uint8_t)orc_16_to_8((uint16_t)(v[0] + 32768u)); return 0;	/* SIGNED_R16 */
	case 0x0311: memcpy(v, in + 4 * i, 4); rgb[0] = (uint8_t)orc_16_to_8(v[0]); rgb[1] = (uint8_t)orc_16_to_8(v[1]); return 0;	/* RG16 */
	case 0x1311: memcpy(v, in + 4 * i, 4); rgb[0] = (uint8_t)orc_16_to_8((uint16_t)(v[0] + 32768u));
		rgb[1] = (uint8_t)orc_16_to_8((uint16_t)(v[1] + 32768u)); return 0;		/* SIGNED_RG16 */
	case 0x2721: memcpy(v, in + 8 * i, 8);							/* FLOAT_RGBX16 */
		for (int c = 0; c < 3; c++) rgb[c] = (uint8_t)orc_16_to_8(orc_half_to_16(v[c]));
		return 0;
	default: return -1;
	}
}
long orc_convert_pixels(uint32_t native_pf, int kind, const uint8_t *in, long n_pixels, uint8_t *out) {
	for (long i = 0; i < n_pixels; i++) {
		uint8_t rgb[3];
		switch (kind) {
		case 1: out[4 * i] = in[4 * i + 2]; out[4 * i + 1] = in[4 * i + 1]; out[4 * i + 2] = in[4 * i]; out[4 * i + 3] = in[4 * i + 3]; break;
		case 2: out[3 * i] = in[4 * i]; out[3 * i + 1] = in[4 * i + 1]; out[3 * i + 2] = in[4 * i + 2]; break;
		case 3: memcpy(out + 8 * i, in + 8 * i + 4, 2); memcpy(out + 8 * i + 2, in + 8 * i + 2, 2);
			memcpy(out + 8 * i + 4, in + 8 * i, 2); memcpy(out + 8 * i + 6, in + 8 * i + 6, 2); break;
		case 4: if (orc_native_rgb(native_pf, in, i, rgb)) return -1;
			out[4 * i] = rgb[0]; out[4 * i + 1] = rgb[1]; out[4 * i + 2] = rgb[2]; out[4 * i + 3] = 0xFF; break;
		case 5: if (orc_native_rgb(native_pf, in, i, rgb)) return -1;
			out[4 * i] = rgb[2]; out[4 * i + 1] = rgb[1]; out[4 * i + 2] = rgb[0]; out[4 * i + 3] = 0xFF; break;
		case 6: if (orc_native_rgb(native_pf, in, i, rgb)) return -1;
			out[3 * i] = rgb[0]; out[3 * i + 1] = rgb[1]; out[3 * i + 2] = rgb[2]; break;
		default: return -1;
		}
	}
	return n_pixels * ((kind == 2 || kind == 6) ? 3 : (kind == 3 ? 8 : 4));
}
