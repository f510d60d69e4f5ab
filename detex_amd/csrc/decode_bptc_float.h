// decode_bptc_float.h -- BPTC_FLOAT (BC6H), unsigned and signed, 14 modes, one lane per block.
//
// Structure: the only genuinely mode-specific step is the scatter of the endpoint bits, so that
// alone is a 14-way switch whose cases are generated AT COMPILE TIME from the bit-layout strings
// of the BPTC specification (constexpr parser -> per-field v_bfe_u32 / v_alignbit_b32 with
// literal positions).  Everything after it -- sign extension, delta transform, unquantisation,
// partition/anchor lookup, index extraction, interpolation, half-float finish -- is one shared
// branch-free path driven by four per-lane parameters (endpoint bits, three delta widths).
// All arithmetic is integer; the output is raw IEEE half bit patterns with X = 0
// (DETEX_PIXEL_FORMAT_FLOAT_RGBX16 / SIGNED_FLOAT_RGBX16).
//
// Reference quirk reproduced: in mode 12 block bit 63 (b0[11]) reads as 0.
#pragma once
#include "dev_common.h"
#include "decode_bptc.h"

namespace detexhip {

// Bit layouts in the notation of the BPTC specification, in stream order after the mode bits
// ("xN[a:b]": component x of endpoint N; the stream delivers bit b first).  The reference carries
// the same spec strings as comments (decompress-bptc-float.c:130-131, 157-159, ...).
constexpr const char *kBc6hLayout[14] = {
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
// endpoint precision (decompress-bptc-float.c:42-43) and delta widths r,g,b; 0 = untransformed (:128-485)
constexpr int kBc6hEpb[14] = { 10, 7, 11, 11, 11, 9, 8, 8, 8, 6, 10, 11, 12, 16 };
constexpr int kBc6hDelta[14][3] = {
	{ 5, 5, 5 }, { 6, 6, 6 }, { 5, 4, 4 }, { 4, 5, 4 }, { 4, 4, 5 }, { 5, 5, 5 }, { 6, 5, 5 },
	{ 5, 6, 5 }, { 5, 5, 6 }, { 0, 0, 0 }, { 0, 0, 0 }, { 9, 9, 9 }, { 8, 8, 8 }, { 4, 4, 4 },
};

struct Bc6hSeg { int comp, ep, dst_lo, len, pos; bool rev; };
struct Bc6hLayout { Bc6hSeg seg[26]; int n; };

constexpr Bc6hLayout bc6h_parse(const char *s, int pos) {
	Bc6hLayout L{};
	while (*s) {
		Bc6hSeg g{};
		g.comp = (*s == 'r') ? 0 : ((*s == 'g') ? 1 : 2);
		g.ep = s[1] - '0';
		s += 3;
		int a = 0;
		while (*s >= '0' && *s <= '9') a = a * 10 + (*s++ - '0');
		int b = a;
		if (*s == ':') {
			++s;
			b = 0;
			while (*s >= '0' && *s <= '9') b = b * 10 + (*s++ - '0');
		}
		++s;				// ']'
		if (*s == ',') ++s;
		g.rev = a < b;			// stream delivers b first: descending when written "lo:hi"
		g.dst_lo = a < b ? a : b;
		g.len = (a < b ? b - a : a - b) + 1;
		g.pos = pos;
		pos += g.len;
		L.seg[L.n++] = g;
	}
	return L;
}

// compile-time-positioned field of the 128-bit block
template <int POS, int LEN> DH uint32_t field_at(const Bits128 &b) {
	constexpr int k = POS >> 5, s = POS & 31;
	if constexpr (LEN == 32) return s == 0 ? b.w[k] : __builtin_amdgcn_alignbit(b.w[k + 1], b.w[k], s);
	else if constexpr (s + LEN <= 32) return ubfe(b.w[k], s, LEN);
	else return ubfe(__builtin_amdgcn_alignbit(b.w[k + 1], b.w[k], s), 0, LEN);
}

template <int M, int S> DH void bc6h_scatter(const Bits128 &b, uint32_t (&ep)[3][4]) {
	constexpr Bc6hLayout L = bc6h_parse(kBc6hLayout[M], M < 2 ? 2 : 5);
	if constexpr (S < L.n) {
		constexpr Bc6hSeg g = L.seg[S];
		uint32_t v = field_at<g.pos, g.len>(b);
		if constexpr (g.rev) v = __brev(v) >> (32 - g.len);
		ep[g.comp][g.ep] |= v << g.dst_lo;
		bc6h_scatter<M, S + 1>(b, ep);
	}
}

struct Bc6hParams { uint32_t epb, dr, dg, db; };
template <int M> DH Bc6hParams bc6h_mode(Bits128 b, uint32_t (&ep)[3][4]) {
	if constexpr (M == 12) b.w[1] &= 0x7FFFFFFFu;	// QUIRK A-3: block bit 63 dropped (decompress-bptc-float.c:462)
	bc6h_scatter<M, 0>(b, ep);
	return Bc6hParams{ (uint32_t)kBc6hEpb[M], (uint32_t)kBc6hDelta[M][0], (uint32_t)kBc6hDelta[M][1], (uint32_t)kBc6hDelta[M][2] };
}

// decompress-bptc-float.c:52-63
DH int32_t bc6h_unquantize_unsigned(uint32_t x, uint32_t epb) {
	const uint32_t mid = ((x << 15) + 0x4000u) >> ((epb - 1u) & 31u);
	uint32_t u = x == 0u ? 0u : (x == (1u << epb) - 1u ? 0xFFFFu : mid);
	return (int32_t)(epb >= 16u ? x : u);
}
// decompress-bptc-float.c:65-86
DH int32_t bc6h_unquantize_signed(int32_t x, uint32_t epb) {
	const bool neg = x < 0;
	const uint32_t ax = (uint32_t)(neg ? -x : x);
	const uint32_t mid = ((ax << 15) + 0x4000u) >> ((epb - 1u) & 31u);
	const uint32_t u = ax == 0u ? 0u : (ax >= (1u << (epb - 1u)) - 1u ? 0x7FFFu : mid);
	const int32_t s = neg ? -(int32_t)u : (int32_t)u;
	return epb >= 16u ? x : s;
}

template <bool SIGNED> struct DecBPTCFloatT {
	static constexpr int kBlockBytes = 16, kPixelBytes = 8;

	// decompress-bptc-float.c:110-626
	template <bool CHECKED> static DH bool decode(uint4 blk, uint32_t mode_mask, uint32_t, uint32_t (&d)[32]) {
		const Bits128 b = { { blk.x, blk.y, blk.z, blk.w } };
		// :23-33: 2-bit codes 00/01 = modes 0/1, otherwise a 5-bit code; 10011,10111,11011,11111 reserved
		const uint32_t low2 = blk.x & 3u, low5 = blk.x & 0x1Fu;
		const uint32_t mode = low2 < 2u ? low2 : (low2 == 2u ? 2u + (low5 >> 2) : 10u + (low5 >> 2));
		if (mode > 13u) return false;
		if (CHECKED && !(mode_mask & (1u << mode))) return false;
		uint32_t ep[3][4] = {};
		Bc6hParams p;
		switch (mode) {
		case 0: p = bc6h_mode<0>(b, ep); break;
		case 1: p = bc6h_mode<1>(b, ep); break;
		case 2: p = bc6h_mode<2>(b, ep); break;
		case 3: p = bc6h_mode<3>(b, ep); break;
		case 4: p = bc6h_mode<4>(b, ep); break;
		case 5: p = bc6h_mode<5>(b, ep); break;
		case 6: p = bc6h_mode<6>(b, ep); break;
		case 7: p = bc6h_mode<7>(b, ep); break;
		case 8: p = bc6h_mode<8>(b, ep); break;
		case 9: p = bc6h_mode<9>(b, ep); break;
		case 10: p = bc6h_mode<10>(b, ep); break;
		case 11: p = bc6h_mode<11>(b, ep); break;
		case 12: p = bc6h_mode<12>(b, ep); break;
		default: p = bc6h_mode<13>(b, ep); break;
		}
		const bool two = mode < 10u;
		const uint32_t delta[3] = { p.dr, p.dg, p.db };
		int32_t q[3][4];
#pragma unroll
		for (int c = 0; c < 3; c++) {
			// :487-518 sign extension and delta transform, :520-533 unquantisation
			const int32_t e0 = SIGNED ? sbfe(ep[c][0], 0, p.epb) : (int32_t)ep[c][0];
#pragma unroll
			for (int e = 0; e < 4; e++) {
				int32_t v = e0;
				if (e > 0) {
					const uint32_t t = ubfe((uint32_t)(e0 + sbfe(ep[c][e], 0, delta[c])), 0, p.epb);
					const uint32_t raw = delta[c] ? t : ep[c][e];
					v = SIGNED ? sbfe(raw, 0, p.epb) : (int32_t)raw;
				}
				q[c][e] = SIGNED ? bc6h_unquantize_signed(v, p.epb) : bc6h_unquantize_unsigned((uint32_t)v, p.epb);
			}
		}
		// partition (5 bits at block bit 77), anchor, index stream (:535-564)
		const uint32_t part = two ? ubfe(blk.z, 13, 5) : 0u;
		const uint32_t pmask = two ? (uint32_t)kPartition1Bit[part] : 0u;
		const uint32_t amask = 1u | (two ? (1u << (kAnchorWords[part] & 0xFu)) : 0u);
		const uint32_t ibits = two ? 3u : 4u;
		// index stream consumed LSB-first as a {hi,lo} pair advanced by funnel shifts
		uint32_t lo = two ? field_at<82, 32>(b) : field_at<65, 32>(b);
		uint32_t hi = two ? (blk.w >> 18) : (blk.w >> 1);
		const WeightParams wp = weight_params(ibits);
#pragma unroll
		for (int i = 0; i < 16; i++) {
			const uint32_t width = ibits - ((amask >> i) & 1u);	// anchor texels store one bit less
			const int32_t w = (int32_t)weight_of(ubfe(lo, 0, width), wp);
			lo = __builtin_amdgcn_alignbit(hi, lo, width);
			hi >>= width;
			const uint32_t ms = bit_to_mask(pmask, i);
			uint32_t h[3];
#pragma unroll
			for (int c = 0; c < 3; c++) {
				const int32_t e0 = (int32_t)bfi(ms, (uint32_t)q[c][2], (uint32_t)q[c][0]);
				const int32_t e1 = (int32_t)bfi(ms, (uint32_t)q[c][3], (uint32_t)q[c][1]);
				const int32_t v = (__mul24(64 - w, e0) + __mul24(w, e1) + 32) >> 6;	// :97-108
				if (SIGNED) {				// :576-609 sign-magnitude half
					const bool neg = v < 0;
					const uint32_t m = (uint32_t)__mul24(neg ? -v : v, 31) >> 5;
					h[c] = m | ((neg && m != 0u) ? 0x8000u : 0u);
				} else {
					h[c] = (uint32_t)__mul24(v, 31) >> 6;	// :613-621 (v >= 0: /64 == >>6)
				}
			}
			d[2 * i] = (h[0] & 0xFFFFu) | (h[1] << 16);
			d[2 * i + 1] = h[2] & 0xFFFFu;			// X = 0
		}
		return true;
	}
};
using DecBPTCFloat = DecBPTCFloatT<false>;
using DecBPTCSignedFloat = DecBPTCFloatT<true>;

}  // namespace detexhip
