// decode_bptc_float.h -- BPTC_FLOAT (BC6H), unsigned and signed, 14 modes, one lane per block.
//
// Structure: the only genuinely mode-specific step is the scatter of the endpoint bits.  The bit-layout strings
// of the BPTC specification are parsed AT COMPILE TIME (constexpr) into (a) per-mode descriptor words for the
// default divergence-free scatter -- every main field sits at one of a few canonical positions with a per-mode
// width, the 15 "loose" bits are routed by 5-bit source indices -- and (b) a 14-way switch with literal
// positions, kept as the A/B alternative (SWITCH_SCATTER).  Everything after it -- sign extension, delta
// transform, unquantisation, partition/anchor lookup, index extraction (two 32-bit windows), interpolation
// from per-lane LDS rows of base/diff values, half-float finish (packed 16-bit lanes for the signed
// sign-magnitude form) -- is one shared branch-free path driven by four per-lane parameters.
// All arithmetic is integer; the output is raw IEEE half bit patterns with X = 0
// (DETEX_PIXEL_FORMAT_FLOAT_RGBX16 / SIGNED_FLOAT_RGBX16).
//
// Reference quirk reproduced: in mode 12 block bit 63 (b0[11]) reads as 0.
#pragma once
#include <stddef.h>
#include "dev_common.h"
#include "bptc_common.h"

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
template <int M> DH Bc6hParams bc6h_mode(Bits128 b, uint32_t flags, uint32_t (&ep)[3][4]) {
	// QUIRK A-3: block bit 63 dropped (decompress-bptc-float.c:462) unless the spec switch is set
	if constexpr (M == 12) b.w[1] &= (flags & kFlagSpecBc6hMode12Bit63) ? 0xFFFFFFFFu : 0x7FFFFFFFu;
	bc6h_scatter<M, 0>(b, ep);
	return Bc6hParams{ (uint32_t)kBc6hEpb[M], (uint32_t)kBc6hDelta[M][0], (uint32_t)kBc6hDelta[M][1], (uint32_t)kBc6hDelta[M][2] };
}

// ---- divergence-free scatter ------------------------------------------------------------------------
// The 14-way switch above costs ~55 VALU ops per mode PRESENT in the wave (all 14 on mixed content).
// Every layout, however, keeps its main fields at fixed positions -- r0@5 g0@15 b0@25 r1@35 g1@45 b1@55
// r2@65 r3@71 and g2[3:0]@41 g3[3:0]@51 b2[3:0]@61 -- and differs only in (a) the field widths and
// (b) where 15 "loose" bits (g2[4:5] g3[4:5] b2[4:5] b3[0:5] r0/g0/b0[10]) sit among 25 candidate
// positions; the one-subset modes store the high bits of r0/g0/b0 bit-reversed right after r1/g1/b1.
// So one straight-line path serves all modes: widths and loose-bit routes come from a 4-dword
// descriptor per mode in __constant__ memory, DERIVED AT COMPILE TIME from the spec strings above.
constexpr int kLooseSrc[25] = { 2, 3, 4, 11, 12, 13, 14, 21, 22, 23, 24, 31, 32, 33, 34, 39, 40, 49, 50, 59, 60, 69, 70, 75, 76 };
struct Bc6hModeWords { uint32_t a, b, c, d; };	// a: w0 | rw<<4 | gw<<8 | bw<<12 | epb<<16 | transformed<<21; b,c,d: 15 x 5-bit routes
// The routes again as RIGHT-SHIFT amounts, one byte per loose bit: slot s (destination bit kLooseDst[s] of its endpoint word) takes
// bit `route` of the 25-bit candidate pool.  Slots 0-11 shift (pool << 5), slots 12-14 (pool << 10), right by (route + 5 - dst)
// resp. route, which leaves the wanted bit AT its destination: one full-rate shift by a per-lane amount and one v_bitop3_b32 per
// loose bit, where extracting the 5-bit route, extracting the bit and shifting it into place took three half-rate operations.
// "No source" is amount 31: bit 31 of either shifted pool is a zero (pool bits >= 25 are), and it lands on bit 0, which no slot
// with a non-zero destination keeps and which is 0 anyway.
constexpr int kLooseDst[15] = { 4, 5, 4, 5, 4, 5, 0, 1, 2, 3, 4, 5, 10, 10, 10 };
struct alignas(16) Bc6hRouteBytes { uint8_t amount[16]; };
constexpr int bc6h_loose_slot(int comp, int ep, int bit) {
	if (comp == 1 && ep == 2 && (bit == 4 || bit == 5)) return bit - 4;		// g2[4], g2[5]
	if (comp == 1 && ep == 3 && (bit == 4 || bit == 5)) return 2 + bit - 4;		// g3[4], g3[5]
	if (comp == 2 && ep == 2 && (bit == 4 || bit == 5)) return 4 + bit - 4;		// b2[4], b2[5]
	if (comp == 2 && ep == 3 && bit < 6) return 6 + bit;				// b3[0..5]
	if (ep == 0 && bit == 10) return 12 + comp;					// r0[10], g0[10], b0[10]
	return -1;
}
struct Bc6hDerived { Bc6hModeWords w; Bc6hRouteBytes r; bool ok; };
constexpr Bc6hDerived bc6h_derive(int M) {
	const Bc6hLayout L = bc6h_parse(kBc6hLayout[M], M < 2 ? 2 : 5);
	constexpr int main_pos[3][4] = { { 5, 35, 65, 71 }, { 15, 45, 41, 51 }, { 25, 55, 61, -1 } };
	uint32_t w0 = 0, w1[3] = { 0, 0, 0 }, route[15] = { 31, 31, 31, 31, 31, 31, 31, 31, 31, 31, 31, 31, 31, 31, 31 };
	bool ok = true;
	for (int s = 0; s < L.n; s++) {
		const Bc6hSeg g = L.seg[s];
		if (g.dst_lo == 0 && g.len > 1) {			// a main field: must sit at its canonical position
			ok = ok && g.pos == main_pos[g.comp][g.ep] && !g.rev;
			if (g.ep == 0) { ok = ok && (w0 == 0 || w0 == (uint32_t)g.len); w0 = (uint32_t)g.len; }
			else if (g.ep == 1) w1[g.comp] = (uint32_t)g.len;
			else if (g.comp == 0) ok = ok && (uint32_t)g.len == w1[0];	// r2, r3 are as wide as r1
			else ok = ok && g.len == 4;					// g2/g3/b2 low nibbles
		} else if (M >= 10) {					// reversed high bits of r0/g0/b0 directly after r1/g1/b1
			ok = ok && g.ep == 0 && g.dst_lo == 10 && g.pos == main_pos[g.comp][1] + (int)w1[g.comp] && g.len == 10 - (int)w1[g.comp] &&
				(g.rev || g.len == 1);
		} else {						// a loose bit
			const int slot = bc6h_loose_slot(g.comp, g.ep, g.dst_lo);
			int idx = -1;
			for (int k = 0; k < 25; k++) if (kLooseSrc[k] == g.pos) idx = k;
			ok = ok && g.len == 1 && slot >= 0 && idx >= 0;
			if (slot >= 0 && idx >= 0) route[slot] = (uint32_t)idx;
		}
	}
	const bool transformed = kBc6hDelta[M][0] != 0;
	if (transformed) ok = ok && (uint32_t)kBc6hDelta[M][0] == w1[0] && (uint32_t)kBc6hDelta[M][1] == w1[1] && (uint32_t)kBc6hDelta[M][2] == w1[2];
	Bc6hModeWords w{};
	w.a = w0 | (w1[0] << 4) | (w1[1] << 8) | (w1[2] << 12) | ((uint32_t)kBc6hEpb[M] << 16) | ((transformed ? 1u : 0u) << 21);
	for (int k = 0; k < 6; k++) { w.b |= route[k] << (5 * k); w.c |= route[6 + k] << (5 * k); }
	for (int k = 0; k < 3; k++) w.d |= route[12 + k] << (5 * k);
	Bc6hRouteBytes r{};
	for (int k = 0; k < 15; k++) {
		const uint32_t none = 31u;
		// (slots 12-14: the pool is shifted by 10, so candidates 22-24 would fall off the top: no layout routes them there)
		if (k >= 12 && route[k] != none) ok = ok && route[k] <= 21u;
		r.amount[k] = (uint8_t)(route[k] == none ? none : (k < 12 ? route[k] + 5u - (uint32_t)kLooseDst[k] : route[k]));
	}
	r.amount[15] = 31;
	return Bc6hDerived{ w, r, ok };
}
#define BC6H_CHECK(M) static_assert(bc6h_derive(M).ok, "BC6H layout does not fit the canonical-position scheme")
BC6H_CHECK(0); BC6H_CHECK(1); BC6H_CHECK(2); BC6H_CHECK(3); BC6H_CHECK(4); BC6H_CHECK(5); BC6H_CHECK(6);
BC6H_CHECK(7); BC6H_CHECK(8); BC6H_CHECK(9); BC6H_CHECK(10); BC6H_CHECK(11); BC6H_CHECK(12); BC6H_CHECK(13);
#undef BC6H_CHECK
__constant__ Bc6hModeWords kBc6hModeWords[14] = {
	bc6h_derive(0).w, bc6h_derive(1).w, bc6h_derive(2).w, bc6h_derive(3).w, bc6h_derive(4).w, bc6h_derive(5).w, bc6h_derive(6).w,
	bc6h_derive(7).w, bc6h_derive(8).w, bc6h_derive(9).w, bc6h_derive(10).w, bc6h_derive(11).w, bc6h_derive(12).w, bc6h_derive(13).w,
};
__constant__ Bc6hRouteBytes kBc6hRouteBytes[14] = {
	bc6h_derive(0).r, bc6h_derive(1).r, bc6h_derive(2).r, bc6h_derive(3).r, bc6h_derive(4).r, bc6h_derive(5).r, bc6h_derive(6).r,
	bc6h_derive(7).r, bc6h_derive(8).r, bc6h_derive(9).r, bc6h_derive(10).r, bc6h_derive(11).r, bc6h_derive(12).r, bc6h_derive(13).r,
};
// index weights (bptc-tables.c aWeight3 / aWeight4) as bytes: entries 0-7 for 3-bit indices, 16-31 for 4-bit ones, so that a
// texel's table address is (window & index mask) | base -- one v_bitop3_b32 -- and the weight one ds_read_u8
struct alignas(32) Bc6hWeightBytes { uint8_t w[32]; };
constexpr uint32_t bc6h_weight_of(uint32_t bits, uint32_t i) { return (i * bptc_weight_mul(bits) + bptc_weight_add(bits)) >> 16; }
constexpr Bc6hWeightBytes bc6h_weight_bytes() {
	Bc6hWeightBytes t = {};
	for (uint32_t i = 0; i < 8u; i++) t.w[i] = (uint8_t)bc6h_weight_of(3, i);
	for (uint32_t i = 0; i < 16u; i++) t.w[16u + i] = (uint8_t)bc6h_weight_of(4, i);
	return t;
}
__constant__ Bc6hWeightBytes kBc6hWeightBytes = bc6h_weight_bytes();

// ---- partition / index-stream table -----------------------------------------------------------------------
// One entry per two-subset partition (5-bit partition number) plus entry 32 for the one-subset modes, everything the
// texel loop needs that depends only on (partition, subset count), derived at compile time:
//   pmask12   one-bit subset number per texel, shifted to bit 12 (texel i's bit at 12 + i): a texel's LDS row address
//             is one shift and one v_bitop3_b32
//   route     lo | hi << 5 | hshift << 10: where the second subset's anchor is missing its top index bit -- position of
//             the zero bit to insert into the first (texels 0-7) or the second (8-15) 32-bit index window, 31 = none
//             (windows of 3-bit indices are 24 bits wide; one-subset blocks shift a zero word: `ones`); hshift = where
//             the second window starts in the block's last dword (decompress-bptc-float.c:535-564)
//   himask0   texel 0's own missing bit (bit 2 or 3 of the first window)
//   woff/imask/ibits   index width and mask, and where the weights of that width start in the byte table above
struct alignas(16) Bc6hPartEntry { uint32_t pmask12, route, himask0, ones, woff, pad, imask, ibits; };
struct Bc6hPartTable { Bc6hPartEntry e[33]; };
constexpr Bc6hPartTable bc6h_part_table() {
	Bc6hPartTable t = {};
	for (uint32_t p = 0; p < 32u; p++) {
		const uint32_t a = kAnchorWordsCx[p] & 15u;
		const uint32_t lo = a < 8u ? 3u * a + 2u : 31u, hi = a >= 8u ? 3u * (a - 8u) + 2u : 31u;
		const uint32_t half = 24u - (a < 8u ? 2u : 1u);			// index bits the first window consumes
		t.e[p] = Bc6hPartEntry{ (uint32_t)kPartition1BitCx[p] << 12, lo | (hi << 5) | ((82u + half - 96u) << 10), 0xFFFFFFFFu << 2, 0xFFFFFFFFu,
			0u, 0u, 7u, 3u };
	}
	t.e[32] = Bc6hPartEntry{ 0u, 31u | (31u << 5) | (0u << 10), 0xFFFFFFFFu << 3, 0u, 16u, 0u, 15u, 4u };
	return t;
}
__constant__ Bc6hPartTable kBc6hPartTable = bc6h_part_table();

// ---- LDS: per-lane blend rows at an aligned base (row address = v_bitop3 of the pre-shifted partition word), the
// partition table and the mode words; workgroup copies made by prepare() (dev_common.h: prepare_tables)
// (storage: like decode_bptc.h's, the one part of this decoder that differs between the device and the emulation of tests/host_emul,
// where LDS byte addresses held in 32-bit registers do not exist: the #else branch keeps the same interface over plain arrays)
#if defined(__HIPCC__)
struct Bc6hLds {
	uint4 row_a[2][256];		// per lane and subset: base r, g, b, diff r		(subset stride 4096: address bit 12)
	uint2 row_b[2][256];		//                      diff g, b			(subset stride 2048: address bit 11)
	Bc6hPartEntry part[33];
	Bc6hModeWords modes[14];
	Bc6hWeightBytes weights;	// (32-byte aligned: 33 * 32 + 14 * 16 = 1280)
	Bc6hRouteBytes routes[14];
};
// the workgroup tables as one constant image: the copy is one 16-byte load and one ds_write_b128 for 96 threads
struct alignas(32) Bc6hTables { Bc6hPartEntry part[33]; Bc6hModeWords modes[14]; Bc6hWeightBytes weights; Bc6hRouteBytes routes[14]; };
static_assert(sizeof(Bc6hTables) == 33 * 32 + 14 * 16 + 32 + 14 * 16 && offsetof(Bc6hLds, modes) - offsetof(Bc6hLds, part) == 33 * 32 &&
	offsetof(Bc6hLds, weights) - offsetof(Bc6hLds, part) == offsetof(Bc6hTables, weights) &&
	offsetof(Bc6hLds, routes) - offsetof(Bc6hLds, part) == offsetof(Bc6hTables, routes), "image = LDS layout");
constexpr Bc6hTables bc6h_tables() {
	Bc6hTables t = {};
	const Bc6hPartTable p = bc6h_part_table();
	for (int k = 0; k < 33; k++) t.part[k] = p.e[k];
	for (int k = 0; k < 14; k++) { t.modes[k] = bc6h_derive(k).w; t.routes[k] = bc6h_derive(k).r; }
	t.weights = bc6h_weight_bytes();
	return t;
}
__constant__ Bc6hTables kBc6hTables = bc6h_tables();
DH Bc6hLds &bc6h_lds() { __shared__ __attribute__((aligned(8192))) Bc6hLds s; return s; }
DH void bc6h_prepare() {
	typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
	if (threadIdx.x < sizeof(Bc6hTables) / 16u)
		reinterpret_cast<u32x4 *>(bc6h_lds().part)[threadIdx.x] = reinterpret_cast<const u32x4 *>(&kBc6hTables)[threadIdx.x];
	__syncthreads();
}
DH Bc6hModeWords bc6h_mode_words(uint32_t mode) { return bc6h_lds().modes[mode]; }
typedef __attribute__((address_space(3))) uint8_t bc6h_lds_u8;
// shift amounts of a mode's loose bits: one address, fifteen ds_read_u8 with immediate offsets
struct Bc6hRoutes {
	uint32_t base;
	DH explicit Bc6hRoutes(uint32_t mode) : base((uint32_t)(uintptr_t)bc6h_lds().routes + mode * (uint32_t)sizeof(Bc6hRouteBytes)) {}
	template <int K> DH uint32_t amount() const { return ((const bc6h_lds_u8 *)(uintptr_t)base)[K]; }
};
// weight of the index in the low bits of `window`; `table` = bc6h_weight_table(woff), `imask` in a VGPR
DH uint32_t bc6h_weight_table(uint32_t woff) { return (uint32_t)(uintptr_t)bc6h_lds().weights.w + woff; }
DH uint32_t bc6h_weight(uint32_t table, uint32_t window, uint32_t imask) {
	return *(const bc6h_lds_u8 *)(uintptr_t)(uint32_t)__builtin_amdgcn_bitop3_b32(window, imask, table, 0xEA);
}
struct Bc6hLane {
	uint32_t base_a, base_b;
	typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
	typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
	typedef __attribute__((address_space(3))) u32x4 lds_u4;
	typedef __attribute__((address_space(3))) u32x2 lds_u2;
	uint32_t bit12, bit11;		// the two subset strides, held in VGPRs: see get()
	DH Bc6hLane() {
		Bc6hLds &s = bc6h_lds();
		base_a = (uint32_t)(uintptr_t)&s.row_a[0][threadIdx.x];
		base_b = (uint32_t)(uintptr_t)&s.row_b[0][threadIdx.x];
		bit12 = 0x1000u; bit11 = 0x800u;
		// v_bitop3_b32 is VOP3 (no literal operand on gfx950): left alone the compiler keeps the masks in SGPRs, and a
		// full-rate VALU op with an SGPR source issues at half rate (tools/ubench/valu_rates.hip: and_sgpr, bitop3_sgpr)
		if constexpr (Tune::kMasksInVgprs) pin_vgpr(bit12, bit11);
	}
	DH void put(int sub, uint4 a, uint2 b) const {
		((lds_u4 *)(uintptr_t)base_a)[sub * 256] = u32x4{ a.x, a.y, a.z, a.w };
		((lds_u2 *)(uintptr_t)base_b)[sub * 256] = u32x2{ b.x, b.y };
	}
	// sel12 / sel11: any words with the texel's subset bit at bit 12 / bit 11
	DH void get(uint32_t sel12, uint32_t sel11, uint4 &a, uint2 &b) const {
		const u32x4 va = *(const lds_u4 *)(uintptr_t)(uint32_t)__builtin_amdgcn_bitop3_b32(sel12, bit12, base_a, 0xEA);
		const u32x2 vb = *(const lds_u2 *)(uintptr_t)(uint32_t)__builtin_amdgcn_bitop3_b32(sel11, bit11, base_b, 0xEA);
		a = uint4{ va.x, va.y, va.z, va.w }; b = uint2{ vb.x, vb.y };
	}
	static DH const Bc6hPartEntry &part(uint32_t i) { return bc6h_lds().part[i]; }
};
#else
DH void bc6h_prepare() {}
DH Bc6hModeWords bc6h_mode_words(uint32_t mode) { return kBc6hModeWords[mode]; }
struct Bc6hRoutes {
	uint32_t mode;
	DH explicit Bc6hRoutes(uint32_t m) : mode(m) {}
	template <int K> DH uint32_t amount() const { return kBc6hRouteBytes[mode].amount[K]; }
};
DH uint32_t bc6h_weight_table(uint32_t woff) { return woff; }
DH uint32_t bc6h_weight(uint32_t table, uint32_t window, uint32_t imask) { return kBc6hWeightBytes.w[(window & imask) | table]; }
struct Bc6hLane {
	uint4 ra[2]; uint2 rb[2];
	DH void put(int sub, uint4 a, uint2 b) { ra[sub] = a; rb[sub] = b; }
	DH void get(uint32_t sel12, uint32_t sel11, uint4 &a, uint2 &b) const { a = ra[(sel12 >> 12) & 1u]; b = rb[(sel11 >> 11) & 1u]; }
	static DH const Bc6hPartEntry &part(uint32_t i) { return kBc6hPartTable.e[i]; }
};
#endif

DH Bc6hParams bc6h_scatter_generic(const Bits128 &b, uint32_t mode, uint32_t flags, uint32_t (&ep)[3][4]) {
	const Bc6hModeWords mw = bc6h_mode_words(mode);
	const uint32_t w0 = mw.a & 15u, rw = ubfe(mw.a, 4, 4), gw = ubfe(mw.a, 8, 4), bw = ubfe(mw.a, 12, 4);
	const bool transformed = (mw.a >> 21) & 1u, one = mode >= 10u;
	// main fields at their canonical positions, per-lane widths
	const uint32_t win35 = field_at<35, 32>(b), win45 = field_at<45, 32>(b);
	uint32_t win55 = field_at<55, 32>(b);
	// QUIRK A-3: block bit 63 (b0[11] of mode 12) reads 0 -- unless the (wave-uniform) spec switch is set
	win55 = mode == 12u ? (win55 & ~((flags & kFlagSpecBc6hMode12Bit63) ? 0u : 0x100u)) : win55;
	ep[0][0] = ubfe(field_at<5, 10>(b), 0, w0);
	ep[1][0] = ubfe(field_at<15, 10>(b), 0, w0);
	ep[2][0] = ubfe(field_at<25, 10>(b), 0, w0);
	ep[0][1] = ubfe(win35, 0, rw);
	ep[1][1] = ubfe(win45, 0, gw);
	ep[2][1] = ubfe(win55, 0, bw);
	ep[0][2] = ubfe(field_at<65, 6>(b), 0, rw);
	ep[0][3] = ubfe(field_at<71, 6>(b), 0, rw);
	// the 25 candidate positions of the loose bits, compacted into one word (bit 31 stays 0 = "no source")
	const uint32_t pool = field_at<2, 3>(b) | (field_at<11, 4>(b) << 3) | (field_at<21, 4>(b) << 7) | (field_at<31, 4>(b) << 11) |
		(field_at<39, 2>(b) << 15) | (field_at<49, 2>(b) << 17) | (field_at<59, 2>(b) << 19) | (field_at<69, 2>(b) << 21) |
		(field_at<75, 2>(b) << 23);
	// each loose bit: the shifted pool moved right by the mode's amount for that slot puts it AT its destination (Bc6hRouteBytes)
	const Bc6hRoutes routes(mode);
	const uint32_t pool5 = pool << 5, pool10 = pool << 10;
	uint32_t bit10 = 1u << 10;			// (not an inline constant: kept in a VGPR, see Bc6hLane::bit12)
	if constexpr (Tune::kMasksInVgprs) pin_vgpr(bit10);
	ep[1][2] = and_or(pool5 >> routes.amount<0>(), 16u, and_or(pool5 >> routes.amount<1>(), 32u, field_at<41, 4>(b)));
	ep[1][3] = and_or(pool5 >> routes.amount<2>(), 16u, and_or(pool5 >> routes.amount<3>(), 32u, field_at<51, 4>(b)));
	ep[2][2] = and_or(pool5 >> routes.amount<4>(), 16u, and_or(pool5 >> routes.amount<5>(), 32u, field_at<61, 4>(b)));
	ep[2][3] = and_or(pool5 >> routes.amount<6>(), 1u, and_or(pool5 >> routes.amount<7>(), 2u, and_or(pool5 >> routes.amount<8>(), 4u,
		and_or(pool5 >> routes.amount<9>(), 8u, and_or(pool5 >> routes.amount<10>(), 16u, (pool5 >> routes.amount<11>()) & 32u)))));
	// one-subset modes: r0/g0/b0[10..] follow r1/g1/b1 bit-reversed (decompress-bptc-float.c:441-485); they have no loose bits
	// (every amount is 31), and the two-subset modes mask the reversed field away
	const uint32_t one_mask = cond_to_mask(one);
	const uint32_t hr = (__brev(ubfe(win35, rw, 10u - rw)) >> ((22u + rw) & 31u)) & one_mask;
	const uint32_t hg = (__brev(ubfe(win45, gw, 10u - gw)) >> ((22u + gw) & 31u)) & one_mask;
	const uint32_t hb = (__brev(ubfe(win55, bw, 10u - bw)) >> ((22u + bw) & 31u)) & one_mask;
	ep[0][0] = and_or(pool10 >> routes.amount<12>(), bit10, ep[0][0]) | (hr << 10);
	ep[1][0] = and_or(pool10 >> routes.amount<13>(), bit10, ep[1][0]) | (hg << 10);
	ep[2][0] = and_or(pool10 >> routes.amount<14>(), bit10, ep[2][0]) | (hb << 10);
	return Bc6hParams{ ubfe(mw.a, 16, 5), transformed ? rw : 0u, transformed ? gw : 0u, transformed ? bw : 0u };
}

// decompress-bptc-float.c:52-63: U(0) = 0, U(2^epb - 1) = 0xFFFF, otherwise ((x << 15) + 0x4000) >> (epb - 1); epb = 16 passes the
// value through.  Both terms of the middle case are multiples of 2^(epb-1) (epb <= 15 there), so it is (2x + 1) << (15 - epb)
// = (x << sh) + c exactly, sh = 16 - epb, c = 2^(15-epb); the two special values differ from it by -c (at 0) and by + c - 1 (at
// the maximum, where x + 1 = 2^epb).  All of it as two multiply-adds, no compare and no select -- the literal form cost two
// compares and three v_cndmask_b32 per value, whose VOP2 encoding issues at a fraction of the rate (dev_common.h):
//   U(x) = (x << sh) + min(x, 1) * c + ((x + 1) >> epb) * (c - 1),   c = 0 = c - 1 for epb = 16
struct Bc6hUnsignedUnq { uint32_t epb, sh, c, cm; };	// per block
DH Bc6hUnsignedUnq bc6h_unsigned_unq(uint32_t epb) { return Bc6hUnsignedUnq{ epb, 16u - epb, 0x8000u >> epb, 0x7FFFu >> epb }; }
DH int32_t bc6h_unquantize_unsigned(uint32_t x, const Bc6hUnsignedUnq &k) {
	const uint32_t mid = DETEX_UMUL24(nonzero_as_one(x), (x << k.sh) + k.c);
	return (int32_t)(DETEX_UMUL24((x + 1u) >> k.epb, k.cm) + mid);
}
// decompress-bptc-float.c:65-86: sign(x) * U(|x|) with U(0) = 0, U(a) = 0x7FFF for a >= lim = 2^(epb-1) - 1, else
// ((a << 15) + 0x4000) >> (epb - 1).  Both terms of that middle case are multiples of 2^(epb-1) (epb <= 15 there), so it is
// (2a + 1) << (15 - epb) exactly, and with the sign:  (2x + sign(x)) << (15 - epb), sign(x) = v_med3_i32(x, -1, 1) -- which is
// also right at x = 0.  |x| >= lim as one unsigned compare of x + lim - 1 against 2 lim - 1.  The 16-bit mode (epb = 16:
// the value passes through, :66) is the same expression with sign() replaced by 0, one shift less and a limit no value
// reaches.  The result is delivered TIMES FOUR (two more shift positions), which is what the texel loop's rows want.
// Seven instructions where the literal form (absolute value, three compares, four selects, negate) took thirteen.
struct Bc6hSignedUnq { int32_t unit, neg_unit; uint32_t shift, lim_m1, lim2_m1; };	// per block: from epb
DH Bc6hSignedUnq bc6h_signed_unq(uint32_t epb) {
	const uint32_t lim = (1u << ((epb - 1u) & 31u)) - 1u;
	const bool pass = epb >= 16u;
	return Bc6hSignedUnq{ pass ? 0 : 1, pass ? 0 : -1, pass ? 1u : 17u - epb, pass ? 40000u : lim - 1u, pass ? 0x7FFFFFFFu : 2u * lim - 1u };
}
DH int32_t bc6h_clamp_sign(int32_t x, int32_t lo, int32_t hi) { return med3_i32(x, lo, hi); }	// (one v_med3_i32: gfx950_prims.h)
DH int32_t bc6h_unquantize_signed_x4(int32_t x, const Bc6hSignedUnq &k) {
	const int32_t sign = bc6h_clamp_sign(x, k.neg_unit, k.unit);
	// (both sides through opaque(): evaluated unconditionally and chosen by a select -- the compiler otherwise branches)
	const uint32_t mid = opaque((uint32_t)(2 * x + sign) << k.shift), top = opaque((uint32_t)__mul24(sign, 4 * 0x7FFF));
	return (int32_t)((uint32_t)x + k.lim_m1 >= k.lim2_m1 ? top : mid);
}

// both signed 16-bit lanes: two's complement -> sign-magnitude half of trunc(v * 31 / 32) (decompress-bptc-float.c:576-609)
// sign_bits = 0x80008000 in a VGPR (Bc6hLane::bit12 explains why)
DH uint32_t bc6h_sign_magnitude_pk(uint32_t p, uint32_t sign_bits) {
	const uint32_t a = pk_max16(p, pk_sub16(0u, p));				// |v| (0x8000 stays 0x8000: read as unsigned below)
	// (Packed operations here: one instruction less than the plain 32-bit form with its lane mask.  This kernel runs at the board's
	// power cap, where an instruction costs about the same energy whatever its issue class -- profiles/AB_RECORD.md -- so the count
	// is what matters.  The last two steps stay plain: with m <= 0x7C00 per lane nothing carries across the lane boundary.)
	const uint32_t m = pk_sub_u16(a, pk_lshr16(pk_add16(a, 0x001F001Fu), 5));	// |v| - ceil(|v| / 32) <= 0x7C00
	return m | and3(m + 0x7FFF7FFFu, p, sign_bits);
}

// SWITCH_SCATTER = true keeps the per-mode switch (cheaper when a whole wave shares one mode); the
// default is the divergence-free scatter (profiles/AB_RECORD.md has the measured A/B).
template <bool SIGNED, bool SWITCH_SCATTER = false> struct DecBPTCFloatT {
	static constexpr int kBlockBytes = 16, kPixelBytes = 8, kNative = SIGNED ? kNatOther : kNatFloatRGBX16;
	static constexpr int kWavesPerSimd = Tune::kBc6hWavesPerSimd;
	static constexpr bool kRowWise = true;
	// the linear kernel requests the block BEFORE the tables are copied into LDS (kernels.h: LoadBeforeTables): with the blocks coming out of
	// HBM the unsigned format gains 4 % (107.2 -> 102.8-103.4 us, 8192^2, two runs) and nothing changes where they come from the Infinity
	// Cache (86.2 / 86.0); the signed format loses 8 % to the same change (102.5 -> 111), as do the RGTC / EAC formats with tables
	static constexpr bool kLoadBeforeTables = !SIGNED;
	static DH void prepare() { bc6h_prepare(); }

	// decompress-bptc-float.c:110-626
	// A block that fails (reserved mode, or a mode outside mode_mask) is decoded as the ALL-ZERO block instead of leaving
	// early: that is a mode-0 block whose endpoints are all 0, and it decodes to sixteen zero pixels on this very path -- so
	// the caller's zero-fill of a failed block (32 moves that nearly every wave of a random stream executes: one block in
	// sixteen carries a reserved mode) is not needed (kernels.h: decode_word)
	static constexpr bool kZeroOnFailure = true;
	template <bool CHECKED> static DH bool decode(uint4 blk, uint32_t mode_mask, uint32_t flags, uint32_t (&d)[32]) {
		return decode_rows<CHECKED>(blk, mode_mask, flags, [&](int r, const uint32_t (&row)[8]) {
#pragma unroll
			for (int k = 0; k < 8; k++) d[8 * r + k] = row[k];
		});
	}
	// the same, handing each texel row (four pixels, eight dwords) to `sink(r, row)` as soon as it is complete: the linear
	// kernel exchanges and stores a row while the next ones are still being interpolated, and never holds all 32 result dwords
	template <bool CHECKED, class Sink> static DH bool decode_rows(uint4 blk, uint32_t mode_mask, uint32_t flags, Sink &&sink) {
		uint32_t d[8];			// the current texel row
		// :23-33: 2-bit codes 00/01 = modes 0/1, otherwise a 5-bit code; 10011,10111,11011,11111 reserved
		stage_priority<Tune::kBc6hPrio, 0>();
		// (as arithmetic: the compiler turns the conditional form into three exec-mask branches)  codes ..10 -> 2 + bits 2-4, ..11 -> 10 + bits 2-4
		const uint32_t low2 = blk.x & 3u;
		const uint32_t coded = bfi(bit_to_mask(blk.x, 1), 2u + ubfe(blk.x, 2, 3) + ((blk.x & 1u) << 3), low2);
		const bool valid = coded <= 13u && (!CHECKED || (mode_mask & (1u << (coded & 31u))) != 0u);
		const uint32_t keep = cond_to_mask(valid);
		blk.x &= keep; blk.y &= keep; blk.z &= keep; blk.w &= keep;
		const uint32_t mode = coded & keep;
		const Bits128 b = { { blk.x, blk.y, blk.z, blk.w } };
		uint32_t ep[3][4] = {};
		Bc6hParams p;
		if (!SWITCH_SCATTER) p = bc6h_scatter_generic(b, mode, flags, ep);
		else switch (mode) {
		case 0: p = bc6h_mode<0>(b, flags, ep); break;
		case 1: p = bc6h_mode<1>(b, flags, ep); break;
		case 2: p = bc6h_mode<2>(b, flags, ep); break;
		case 3: p = bc6h_mode<3>(b, flags, ep); break;
		case 4: p = bc6h_mode<4>(b, flags, ep); break;
		case 5: p = bc6h_mode<5>(b, flags, ep); break;
		case 6: p = bc6h_mode<6>(b, flags, ep); break;
		case 7: p = bc6h_mode<7>(b, flags, ep); break;
		case 8: p = bc6h_mode<8>(b, flags, ep); break;
		case 9: p = bc6h_mode<9>(b, flags, ep); break;
		case 10: p = bc6h_mode<10>(b, flags, ep); break;
		case 11: p = bc6h_mode<11>(b, flags, ep); break;
		case 12: p = bc6h_mode<12>(b, flags, ep); break;
		default: p = bc6h_mode<13>(b, flags, ep); break;
		}
		const bool two = mode < 10u;
		// a wave of one-subset blocks only (modes 10-13: what encoders emit for smooth HDR content) does not
		// transform / unquantise the second subset's endpoints (wave-uniform; the random stream never qualifies)
		const bool wave_two = __builtin_amdgcn_ballot_w64(two) != 0;
		const uint32_t delta[3] = { p.dr, p.dg, p.db };
		int32_t q[3][4];
		const Bc6hSignedUnq sk = bc6h_signed_unq(p.epb);
		const Bc6hUnsignedUnq uk = bc6h_unsigned_unq(p.epb);
		// :487-518 in one form for transformed and untransformed modes: value = ext_epb(base' + sext_w(field)) with (base', w) =
		// (base, delta width) or (0, the whole field: epb for the signed format, 31 bits for the unsigned one, whose fields are
		// less than 2^16) -- the wrap to epb bits and the sign / zero extension are one v_bfe
		int32_t base_t[3];
		uint32_t width_t[3];
#pragma unroll
		for (int c = 0; c < 3; c++) {
			base_t[c] = delta[c] ? (SIGNED ? sbfe(ep[c][0], 0, p.epb) : (int32_t)ep[c][0]) : 0;
			width_t[c] = delta[c] ? delta[c] : (SIGNED ? p.epb : 31u);
		}
		// :487-518 sign extension and delta transform, :520-533 unquantisation
		auto endpoint = [&](int c, int e) {
			const uint32_t sum = e == 0 ? ep[c][0] : (uint32_t)(base_t[c] + sbfe(ep[c][e], 0, width_t[c]));
			q[c][e] = SIGNED ? bc6h_unquantize_signed_x4(sbfe(sum, 0, p.epb), sk)			// signed: 4 * value
				: bc6h_unquantize_unsigned(e == 0 ? sum : ubfe(sum, 0, p.epb), uk);
		};
#pragma unroll
		for (int c = 0; c < 3; c++) { endpoint(c, 0); endpoint(c, 1); q[c][2] = 0; q[c][3] = 0; }
		if (wave_two) {
#pragma unroll
			for (int c = 0; c < 3; c++) { endpoint(c, 2); endpoint(c, 3); }
		}
		// partition (5 bits at block bit 77), anchors, index stream (:535-564)
		const Bc6hPartEntry &pe = Bc6hLane::part(two ? ubfe(blk.z, 13, 5) : 32u);
		// Index stream, LSB-first, read through two 32-bit windows: texels 0-7 consume 8*ibits - (anchors among them) <= 31
		// bits (texel 0 is always an anchor), texels 8-15 start where they ended, and that second window lies in the last
		// dword.  The anchors' absent top bits are then inserted as zeros (w + (w & himask): two full-rate ops each), so
		// every texel reads its index at a regular position: one `and`, one shift.
		const uint32_t route = pe.route, ones = pe.ones;		// shifts use the low 5 bits of their amount
		uint32_t win = two ? field_at<82, 32>(b) : field_at<65, 32>(b);
		uint32_t win_hi = blk.w >> (route >> 10);
		win += win & pe.himask0;
		win += win & (ones << (route & 31u));
		win_hi += win_hi & (ones << ((route >> 5) & 31u));
		const uint32_t ibits = pe.ibits, imask = pe.imask, wtable = bc6h_weight_table(pe.woff);
		// ((64-w)*e0 + w*e1 + 32) >> 6  ==  (64*e0 + 32 + w*(e1-e0)) >> 6  (:97-108): per subset keep
		// base = 64*e0 + 32 and diff = e1 - e0 (a texel channel is one v_mad_i32_i24 + one shift) in per-lane LDS
		// rows, fetched per texel by the partition bit
		Bc6hLane lane;
#pragma unroll
		for (int s = 0; s < 2; s++) {
			if (s == 1 && !wave_two) break;
			uint4 ra; uint2 rb;
			// signed: q[] holds 4 * value, so that the 16 result bits of (base + w * diff) >> 6 are bytes 1-2 of the sum
			// (|sum| < 2^23, 4 * |diff| < 2^19: still inside v_mad_i32_i24's operands) and one v_perm_b32 both drops the
			// six fraction bits and packs two channels -- no shifts in the texel loop
			// unsigned: base = (64 * e0 + 32) << 10 < 2^32, and the texel loop multiplies diff by the weight << 10, so that the
			// value (64 * e0 + 32 + w * diff) >> 6 < 2^16 is the HIGH HALF of the sum: the next multiply reads it there
			// (v_mul_u32_u24 with src0_sel:WORD_1), no shift
			constexpr int kScale = SIGNED ? 64 : 65536, kRound = SIGNED ? 128 : 32768;
			ra.x = (uint32_t)q[0][2 * s] * (uint32_t)kScale + (uint32_t)kRound;
			ra.y = (uint32_t)q[1][2 * s] * (uint32_t)kScale + (uint32_t)kRound;
			ra.z = (uint32_t)q[2][2 * s] * (uint32_t)kScale + (uint32_t)kRound;
			ra.w = (uint32_t)(q[0][2 * s + 1] - q[0][2 * s]);
			rb.x = (uint32_t)(q[1][2 * s + 1] - q[1][2 * s]);
			rb.y = (uint32_t)(q[2][2 * s + 1] - q[2][2 * s]);
			lane.put(s, ra, rb);
		}
		stage_priority<Tune::kBc6hPrio, 1>();
		const uint32_t p12 = pe.pmask12, p11 = p12 >> 1;
		uint32_t b_even = 0, sign_bits = 0x80008000u;
		if constexpr (SIGNED && Tune::kMasksInVgprs) pin_vgpr(sign_bits);
#pragma unroll
		for (int i = 0; i < 16; i++) {
			if (i == 8) { win = win_hi; stage_priority<Tune::kBc6hPrio, 2>(); }
			// the weight from the byte table in LDS (one v_bitop3_b32 for the address; the closed form cost an `and`, a multiply-add
			// and a shift, two of them half-rate); unsigned: weight << 10 (<= 2^16)
			const int32_t w = (int32_t)(bc6h_weight(wtable, win, imask) << (SIGNED ? 0 : 10));
			win >>= ibits;
			uint4 ra; uint2 rb;
			lane.get(p12 >> i, p11 >> i, ra, rb);
			const int32_t bs[3] = { (int32_t)ra.x, (int32_t)ra.y, (int32_t)ra.z }, df[3] = { (int32_t)ra.w, (int32_t)rb.x, (int32_t)rb.y };
			int32_t v[3];
#pragma unroll
			for (int c = 0; c < 3; c++) v[c] = (int32_t)((uint32_t)bs[c] + (uint32_t)__mul24(w, df[c]));	// unsigned: (value << 16) | fraction
			if (SIGNED) {
				// :576-609 sign-magnitude half: m = (|v|*31)>>5 = |v| - ceil(|v|/32), sign bit only if m != 0.  Every v fits a
				// signed 16-bit lane (|v| <= 0x8000, and 0x8000 still comes out right in the unsigned steps), so R,G of this
				// texel and B of two neighbouring texels are finished two per VGPR; bit 15 of m + 0x7FFF is set iff m != 0.
				// v[] are the sums times 4 here: value c = bits 8-23
				d[2 * (i & 3)] = bc6h_sign_magnitude_pk(perm((uint32_t)v[1], (uint32_t)v[0], 0x06050201u), sign_bits);
				if ((i & 1) == 0) {
					b_even = (uint32_t)v[2];
				} else {
					const uint32_t hb = bc6h_sign_magnitude_pk(perm((uint32_t)v[2], b_even, 0x06050201u), sign_bits);
					d[2 * (i & 3) - 1] = hb & 0xFFFFu;		// X = 0
					d[2 * (i & 3) + 1] = hb >> 16;
				}
			} else {
				// :613-621: half = value * 31 / 64 (value >= 0) = bytes 1-2 of value * 124 < 2^23; the v_perm that packs R and G
				// takes them from there, so only B needs a shift
				uint32_t h[3];
#pragma unroll
				for (int c = 0; c < 3; c++) h[c] = DETEX_UMUL24((uint32_t)v[c] >> 16, 124u);
				d[2 * (i & 3)] = perm(h[1], h[0], 0x06050201u);
				d[2 * (i & 3) + 1] = h[2] >> 8;			// X = 0
			}
			if ((i & 3) == 3) sink(i >> 2, d);
		}
		return valid;
	}
};
using DecBPTCFloat = DecBPTCFloatT<false>;
using DecBPTCSignedFloat = DecBPTCFloatT<true>;

}  // namespace detexhip
