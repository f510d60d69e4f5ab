// decode_s3tc_rgtc.h -- BC1 / BC1A / BC2 / BC3 and RGTC1/2 (BC4/BC5, unsigned + signed) for gfx950.
//
// One lane decodes one 4x4 block into registers, row-major, as dwords d[row*P + k]
// (P = bytes per pixel = dwords per 4-pixel row).  Behaviour follows the reference decoders
// cited per function (/root/reference); the code structure does not: palettes are built as
// packed RGBA dwords and texels are picked with lane masks (v_bfe_i32 + v_bitop3_b32) or byte
// permutes (v_perm_b32) instead of per-texel switch statements; divisions are multiply-shifts, except
// the signed RGTC ramp, whose entries are single lookups in compile-time tables (LDS copy per workgroup).
#pragma once
#include "dev_common.h"

namespace detexhip {

enum : uint32_t {
	kFlagEncode = 0x1, kFlagOpaqueOnly = 0x2, kFlagNonOpaqueOnly = 0x4,	// detex.h:397-411
};

// RGB565 pair -> four packed palette entries (alpha byte = 0xFF except entry 3 of the
// three-colour mode, which is `three_colour_p3`).  decompress-bc.c:34-53: 565 is expanded by
// plain shifts (no low-bit replication), thirds are floor((2a+b)/3), the midpoint floor((a+b)/2).
DH void s3tc_palette(uint32_t c, bool four_colour, uint32_t three_colour_p3, uint32_t (&p)[4]) {
	const uint32_t r0 = (c >> 8) & 0xF8u, g0 = (c >> 3) & 0xFCu, b0 = (c << 3) & 0xF8u;
	const uint32_t r1 = (c >> 24) & 0xF8u, g1 = (c >> 19) & 0xFCu, b1 = (c >> 13) & 0xF8u;
	p[0] = pack_rgba(r0, g0, b0, 0xFFu);
	p[1] = pack_rgba(r1, g1, b1, 0xFFu);
	const uint32_t t2 = pack_rgba(div3_u(2 * r0 + r1), div3_u(2 * g0 + g1), div3_u(2 * b0 + b1), 0xFFu);
	const uint32_t t3 = pack_rgba(div3_u(r0 + 2 * r1), div3_u(g0 + 2 * g1), div3_u(b0 + 2 * b1), 0xFFu);
	// per-byte floor average; the 0xFF alpha bytes average to 0xFF
	const uint32_t h2 = (p[0] & p[1]) + (((p[0] ^ p[1]) & 0xFEFEFEFEu) >> 1);
	p[2] = four_colour ? t2 : h2;
	p[3] = four_colour ? t3 : three_colour_p3;
}

// sixteen 2-bit LSB-first selectors -> sixteen packed texels (decompress-bc.c:54-59)
DH void s3tc_texels(uint32_t idx, const uint32_t (&p)[4], uint32_t (&d)[16]) {
#pragma unroll
	for (int i = 0; i < 16; i++)
		d[i] = select4(bit_to_mask(idx, 2 * i), bit_to_mask(idx, 2 * i + 1), p[0], p[1], p[2], p[3]);
}

// The 8-entry BC3-alpha / RGTC ramp as eight bytes {lo: entries 0-3, hi: entries 4-7}.
// decompress-bc.c:210-237, decompress-rgtc.c:33-55: e0 > e1 -> six floor(/7) interpolants,
// else four floor(/5) interpolants then 0 and 255.
DH void ramp8_unsigned(uint32_t e0, uint32_t e1, uint32_t &lo, uint32_t &hi) {
	const bool seven = e0 > e1;
	uint32_t v[8];
	v[0] = e0; v[1] = e1;
#pragma unroll
	for (int k = 1; k <= 6; k++) {
		const uint32_t q7 = div7_u((7 - k) * e0 + k * e1);
		const uint32_t q5 = k <= 4 ? div5_u((uint32_t)(5 - k) * e0 + k * e1) : (k == 5 ? 0u : 255u);
		v[1 + k] = seven ? q7 : q5;
	}
	lo = v[0] | (v[1] << 8) | (v[2] << 16) | (v[3] << 24);
	hi = v[4] | (v[5] << 8) | (v[6] << 16) | (v[7] << 24);
}

// 48 bits of 3-bit codes held as two 24-bit halves (texels 0-7, 8-15)
struct Codes48 { uint32_t a, b; };
DH Codes48 codes48_from_le(uint32_t w0, uint32_t w1) {	// w0,w1 = first 8 bytes LE; codes start at byte 2
	Codes48 c;
	c.a = (w0 >> 16) | ((w1 & 0xFFu) << 16);
	c.b = w1 >> 8;
	return c;
}

struct DecBC1 {
	static constexpr int kBlockBytes = 8, kPixelBytes = 4;
	// decompress-bc.c:23-61
	template <bool CHECKED> static DH bool decode(uint2 blk, uint32_t, uint32_t, uint32_t (&d)[16]) {
		uint32_t p[4];
		s3tc_palette(blk.x, (blk.x & 0xFFFFu) > (blk.x >> 16), 0xFF000000u, p);
		s3tc_texels(blk.y, p, d);
		return true;
	}
};

struct DecBC1A {
	static constexpr int kBlockBytes = 8, kPixelBytes = 4;
	// decompress-bc.c:87-132
	template <bool CHECKED> static DH bool decode(uint2 blk, uint32_t, uint32_t flags, uint32_t (&d)[16]) {
		const bool opaque = (blk.x & 0xFFFFu) > (blk.x >> 16);
		if (CHECKED) {
			if (opaque && (flags & kFlagNonOpaqueOnly)) return false;
			if (!opaque && (flags & kFlagOpaqueOnly)) return false;
		}
		uint32_t p[4];
		s3tc_palette(blk.x, opaque, 0u, p);
		s3tc_texels(blk.y, p, d);
		return true;
	}
};

struct DecBC2 {
	static constexpr int kBlockBytes = 16, kPixelBytes = 4;
	// decompress-bc.c:136-171: colour block in bytes 8-15 (always four-colour), 4-bit alpha in bytes 0-7
	template <bool CHECKED> static DH bool decode(uint4 blk, uint32_t, uint32_t flags, uint32_t (&d)[16]) {
		if (CHECKED && (blk.z & 0xFFFFu) <= (blk.z >> 16) && (flags & kFlagEncode)) return false;
		uint32_t p[4];
		s3tc_palette(blk.z, true, 0u, p);
		s3tc_texels(blk.w, p, d);
#pragma unroll
		for (int i = 0; i < 16; i++) {
			const uint32_t a4 = ubfe(i < 8 ? blk.x : blk.y, 4 * (i & 7), 4);
			d[i] = (d[i] & 0x00FFFFFFu) | ((a4 | (a4 << 4)) << 24);	// a4 * 255 / 15 == a4 * 17
		}
		return true;
	}
};

struct DecBC3 {
	static constexpr int kBlockBytes = 16, kPixelBytes = 4;
	// decompress-bc.c:175-240
	template <bool CHECKED> static DH bool decode(uint4 blk, uint32_t, uint32_t flags, uint32_t (&d)[16]) {
		const uint32_t a0 = blk.x & 0xFFu, a1 = (blk.x >> 8) & 0xFFu;
		if (CHECKED) {
			if (a0 > a1 && (flags & kFlagOpaqueOnly)) return false;
			if ((blk.z & 0xFFFFu) <= (blk.z >> 16) && (flags & kFlagEncode)) return false;
		}
		uint32_t p[4], lo, hi;
		s3tc_palette(blk.z, true, 0u, p);
		s3tc_texels(blk.w, p, d);
		ramp8_unsigned(a0, a1, lo, hi);
		const Codes48 c = codes48_from_le(blk.x, blk.y);
#pragma unroll
		for (int i = 0; i < 16; i++) {
			const uint32_t code = ubfe(i < 8 ? c.a : c.b, 3 * (i & 7), 3);
			d[i] = (d[i] & 0x00FFFFFFu) | (perm(hi, lo, code) << 24);
		}
		return true;
	}
};

// one unsigned RGTC channel -> four dwords of four 8-bit texels each (decompress-rgtc.c:26-60)
DH void rgtc_channel_u8(uint32_t w0, uint32_t w1, uint32_t (&rows)[4]) {
	uint32_t lo, hi;
	ramp8_unsigned(w0 & 0xFFu, (w0 >> 8) & 0xFFu, lo, hi);
	const Codes48 c = codes48_from_le(w0, w1);
#pragma unroll
	for (int r = 0; r < 4; r++)
		rows[r] = perm(hi, lo, spread3to8(ubfe(r < 2 ? c.a : c.b, 12 * (r & 1), 12)));
}

// one signed RGTC channel -> eight dwords of two 16-bit texels each (decompress-rgtc.c:84-130).
// A ramp entry is  map16(trunc(((d-k)*e0 + k*e1) / d)),  d = 7 or 5, with the reference's truncating division
// (detex.h:966-982) and map16(q) = (q+127)*65535/254 - 32768.  The numerator spans only [-127*d, 127*d], so each
// entry is ONE lookup in a table indexed by the biased numerator (3.5 + 2.5 KiB of 16-bit values, generated at
// compile time, copied to LDS once per workgroup): per entry one v_mad_u32_u24 for the byte offset and a
// ds_read_u16, instead of a division, a sign fix-up and the 16-bit map in VALU arithmetic (the reference itself
// divides through lookup tables, division-tables.c).  Values are kept with the sign bit flipped (unsigned order).
constexpr uint32_t rgtc_map16_flipped(int32_t q) {		// q in [-127, 127]
	const uint32_t n = (uint32_t)(q + 127);
	return (n * 258u + ((n * 387u + 6u) >> 15)) & 0xFFFFu;		// == map16(q) ^ 0x8000, tests/test_host_logic.py
}
struct alignas(16) RgtcSignedTables {
	uint16_t t7[1792];	// index x + 896, x = (7-k)*e0 + k*e1
	uint16_t t5[1280];	// index x + 640, x = (5-k)*e0 + k*e1
	uint16_t map[256];	// index q + 127
};
constexpr RgtcSignedTables rgtc_signed_tables() {
	RgtcSignedTables t = {};
	for (int y = 0; y < 1792; y++) { const int q = (y - 896) / 7; t.t7[y] = (uint16_t)rgtc_map16_flipped(q < -127 ? -127 : (q > 127 ? 127 : q)); }
	for (int y = 0; y < 1280; y++) { const int q = (y - 640) / 5; t.t5[y] = (uint16_t)rgtc_map16_flipped(q < -127 ? -127 : (q > 127 ? 127 : q)); }
	for (int n = 0; n < 256; n++) t.map[n] = (uint16_t)rgtc_map16_flipped(n > 254 ? 127 : n - 127);
	return t;
}
__constant__ RgtcSignedTables kRgtcSignedTables = rgtc_signed_tables();
static_assert(sizeof(RgtcSignedTables) % 16 == 0, "copied with 16-byte moves");
DH RgtcSignedTables &rgtc_signed_lds() { __shared__ RgtcSignedTables t; return t; }
DH void rgtc_signed_prepare() {
	const uint4 *src = reinterpret_cast<const uint4 *>(&kRgtcSignedTables);
	uint4 *dst = reinterpret_cast<uint4 *>(&rgtc_signed_lds());
	for (uint32_t k = threadIdx.x; k < sizeof(RgtcSignedTables) / 16u; k += 256u) dst[k] = src[k];
	__syncthreads();
}
DH const RgtcSignedTables &rgtc_signed() { return rgtc_signed_lds(); }
// 16-bit entry at a BYTE offset (offsets are built pre-doubled so that no shift is needed)
DH uint32_t u16_at(const uint16_t *table, uint32_t byte_offset) {
	return *reinterpret_cast<const uint16_t *>(reinterpret_cast<const char *>(table) + byte_offset);
}

DH bool rgtc_channel_s16(uint32_t w0, uint32_t w1, uint32_t (&pairs)[8]) {
	int32_t e0 = sbfe(w0, 0, 8), e1 = sbfe(w0, 8, 8);
	const bool valid = !(e0 == -127 && e1 == -128);		// :90-92
	e0 = max(e0, -127);
	e1 = max(e1, -127);
	const uint32_t seven = cond_to_mask(e0 > e1);
	const RgtcSignedTables &t = rgtc_signed();
	// byte offsets 2*(x + bias): base 2*d*e0 + 2*bias, step 2*(e1 - e0) per k -- all non-negative
	const uint32_t step2 = (uint32_t)(2 * (e1 - e0));
	const uint32_t base7 = (uint32_t)(14 * e0 + 2 * 896), base5 = (uint32_t)(10 * e0 + 2 * 640);
	uint32_t w[8];						// the eight ramp entries as unsigned 16-bit (sign bit still flipped)
	w[0] = u16_at(t.map, (uint32_t)(2 * e0 + 254));
	w[1] = u16_at(t.map, (uint32_t)(2 * e1 + 254));
#pragma unroll
	for (int k = 1; k <= 6; k++) {
		const uint32_t w7 = u16_at(t.t7, base7 + (uint32_t)k * step2);
		const uint32_t w5 = k <= 4 ? u16_at(t.t5, base5 + (uint32_t)k * step2)
			: (k == 5 ? rgtc_map16_flipped(-127) : rgtc_map16_flipped(127));	// entries 6, 7 of the five-ramp: -1.0 and 1.0
		w[1 + k] = bfi(seven, w7, w5);
	}
	// low-byte and high-byte tables of the eight entries for v_perm lookups
	const uint32_t p01 = perm(w[1], w[0], 0x05010400u), p23 = perm(w[3], w[2], 0x05010400u);	// {lo a, lo b, hi a, hi b}
	const uint32_t p45 = perm(w[5], w[4], 0x05010400u), p67 = perm(w[7], w[6], 0x05010400u);
	const uint32_t lo_l = perm(p23, p01, 0x05040100u), lo_h = perm(p67, p45, 0x05040100u);
	const uint32_t hi_l = perm(p23, p01, 0x07060302u) ^ 0x80808080u, hi_h = perm(p67, p45, 0x07060302u) ^ 0x80808080u;
	const Codes48 c = codes48_from_le(w0, w1);
#pragma unroll
	for (int r = 0; r < 4; r++) {
		const uint32_t sel = spread3to8(ubfe(r < 2 ? c.a : c.b, 12 * (r & 1), 12));
		const uint32_t l4 = perm(lo_h, lo_l, sel), h4 = perm(hi_h, hi_l, sel);
		pairs[2 * r] = perm(h4, l4, 0x05010400u);	// texels 0,1 as {lo,hi,lo,hi}
		pairs[2 * r + 1] = perm(h4, l4, 0x07030602u);	// texels 2,3
	}
	return valid;
}

struct DecRGTC1 {
	static constexpr int kBlockBytes = 8, kPixelBytes = 1, kNative = kNatR8;
	static constexpr int kLaneBlocks = Tune::kRgtc1LaneBlocks;	// blocks per lane in the linear kernel (kernels.h: decode_linear_grouped)
	template <bool CHECKED> static DH bool decode(uint2 blk, uint32_t, uint32_t, uint32_t (&d)[4]) {
		rgtc_channel_u8(blk.x, blk.y, d);
		return true;
	}
};

struct DecRGTC2 {
	static constexpr int kBlockBytes = 16, kPixelBytes = 2, kNative = kNatRG8;
	// decompress-rgtc.c:72-77: R from bytes 0-7, G from bytes 8-15, interleaved R,G
	template <bool CHECKED> static DH bool decode(uint4 blk, uint32_t, uint32_t, uint32_t (&d)[8]) {
		uint32_t r[4], g[4];
		rgtc_channel_u8(blk.x, blk.y, r);
		rgtc_channel_u8(blk.z, blk.w, g);
#pragma unroll
		for (int row = 0; row < 4; row++) {
			d[2 * row] = perm(g[row], r[row], 0x05010400u);
			d[2 * row + 1] = perm(g[row], r[row], 0x07030602u);
		}
		return true;
	}
};

struct DecSignedRGTC1 {
	static DH void prepare() { rgtc_signed_prepare(); }
	static constexpr int kBlockBytes = 8, kPixelBytes = 2, kNative = kNatSignedR16;
	static constexpr int kLaneBlocks = 2;
	template <bool CHECKED> static DH bool decode(uint2 blk, uint32_t, uint32_t, uint32_t (&d)[8]) {
		return rgtc_channel_s16(blk.x, blk.y, d);
	}
};

struct DecSignedRGTC2 {
	static DH void prepare() { rgtc_signed_prepare(); }
	static constexpr int kBlockBytes = 16, kPixelBytes = 4, kNative = kNatSignedRG16;
	static constexpr int kStorePolicy = 4;		// plain `nt` row stores: `sc1 nt` costs this kernel 6-9 % (kernels.h: StorePolicy)
	// decompress-rgtc.c:141-147: texel = R16 | G16 << 16
	template <bool CHECKED> static DH bool decode(uint4 blk, uint32_t, uint32_t, uint32_t (&d)[16]) {
		uint32_t r[8], g[8];
		const bool ok_r = rgtc_channel_s16(blk.x, blk.y, r);
		const bool ok_g = rgtc_channel_s16(blk.z, blk.w, g);
#pragma unroll
		for (int k = 0; k < 8; k++) {
			d[2 * k] = perm(g[k], r[k], 0x05040100u);
			d[2 * k + 1] = perm(g[k], r[k], 0x07060302u);
		}
		return ok_r && ok_g;
	}
};

}  // namespace detexhip
