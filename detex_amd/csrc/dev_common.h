// dev_common.h -- device-side helpers shared by the gfx950 block decoders.
//
// Everything here is integer bit manipulation mapped 1:1 onto CDNA4 VALU ops
// (v_bfe_u32/i32, v_bitop3_b32, v_perm_b32, v_mul_u32_u24, packed 16-bit adds, v_sat_pk_u8_i16): the
// decode path has no floating point and no MFMA work.  The exact-division
// helpers are DETEX_HD so that tests/test_host_logic.py::test_intmath can compile them for the host
// and prove them exhaustively equal to C integer division on the domains the reference's LUTs cover
// (division-tables.c: 0..767 /3, 0..1279 /5, 0..1791 /7); the signed forms and the closed-form
// weight / 16-bit-map helpers below them are the checked statements the decoders' faster biased
// variants are tested against.
#pragma once
#include <stdint.h>
#include "tune.h"

// Device builds take the gfx950 primitives (inline assembly, vector extensions) from gfx950_prims.h.  The GPU-less emulation of
// tests/host_emul defines the same names in plain C++ BEFORE it includes this file (hip_host_shim.h, which also defines that header's
// include guard); any other host translation unit (tests of the exact-division helpers) gets the DETEX_HD helpers only.  This is the
// one place the decoders' headers ask which of the three they are compiled for.
#if defined(__HIPCC__) || defined(DETEXHIP_GFX950_PRIMS_H)
#include "gfx950_prims.h"
#define DETEXHIP_DEVICE_CODE 1
#else
#define DETEX_HD static inline
#define DETEX_UMUL24(a, b) (((a) & 0xFFFFFFu) * ((b) & 0xFFFFFFu))
#endif

namespace detexhip {

// native pixel format of a decoder (Dec::kNative; decoders without the member decode to RGBA8 / RGBX8): selects the
// pixel-format epilogues a decoder can feed (kernels.h)
enum : int { kNatRGBA8 = 0, kNatR8, kNatRG8, kNatR16, kNatSignedR16, kNatRG16, kNatSignedRG16, kNatFloatRGBX16, kNatOther };

// ---- exact small-domain unsigned division by multiply-shift (products < 2^24 * 2^16) -----
DETEX_HD uint32_t div3_u(uint32_t x) { return (x * 43691u) >> 17; }  // exact for x < 98304
DETEX_HD uint32_t div5_u(uint32_t x) { return (x * 52429u) >> 18; }  // exact for x < 81920
DETEX_HD uint32_t div7_u(uint32_t x) { return (x * 9363u) >> 16; }   // exact for x < 2048*... (tested to 4095)
// truncating signed division (detex.h:966-982 semantics: sign(v) * (|v| / d))
DETEX_HD int32_t div7_s(int32_t v) { int32_t q = (int32_t)div7_u((uint32_t)(v < 0 ? -v : v)); return v < 0 ? -q : q; }
DETEX_HD int32_t div5_s(int32_t v) { int32_t q = (int32_t)div5_u((uint32_t)(v < 0 ? -v : v)); return v < 0 ? -q : q; }
// BPTC interpolation weight of an n-bit index: (64*i + (2^n-1)/2) / (2^n-1), n in {2,3,4}
// (bptc-tables.c aWeight2/3/4 in closed form; magic = ceil(65536/d))
DETEX_HD uint32_t bptc_weight(uint32_t index, uint32_t bits) {
	const uint32_t d = (1u << bits) - 1u;
	const uint32_t magic = bits == 2 ? 21846u : (bits == 3 ? 9363u : 4370u);
	return (((index << 6) + (d >> 1)) * magic) >> 16;
}
// The same weight in 16 bits, for two indices side by side in one register (v_pk_mad_u16): (index * m + 128) >> 8 with
// m = round(64 * 256 / (2^n - 1)); the product stays below 2^16 and the result equals aWeight2/3/4 for every index
// (tests/test_host_logic.py proves all 4 + 8 + 16 cases).
constexpr uint32_t bptc_weight16_mul(uint32_t bits) { return bits == 2 ? 5461u : (bits == 3 ? 2341u : 1092u); }

// signed-RGTC value map [-127,127] -> int16 (decompress-rgtc.c:125-126): (v+127)*65535/254 - 32768
// 16-bit component -> 8-bit as the reference converts it (convert.c:258-267, 299-313): (x + 127) * 255 / 65535
// = floor((x + 127) / 257) = ((x + 127) * 0xFF01) >> 24 for every x < 65536 (0xFF01 * 257 = 2^24 + 1 and the product stays
// below 2^32: one v_mad_u32_u24 and a shift; checked exhaustively in tests/test_host_logic.py)
DETEX_HD uint32_t component16_to_8(uint32_t x) { return ((x & 0xFFFFu) * 0xFF01u + 127u * 0xFF01u) >> 24; }	// (the mask lets the compiler pick the 24-bit multiply)

// division-free: 65535 = 254*258 + 3, so n*65535/254 = n*258 + floor(3n/254) with floor(3n/254) = (n>=85)+(n>=170)+(n>=254)
DETEX_HD uint32_t rgtc_signed_to_16(int32_t v) {
	const uint32_t n = (uint32_t)(v + 127);
	return (n * 258u + (n >= 85u ? 1u : 0u) + (n >= 170u ? 1u : 0u) + (n >= 254u ? 1u : 0u) - 32768u) & 0xFFFFu;
}

#if defined(DETEXHIP_DEVICE_CODE)
// ---- single-instruction bit-field idioms ---------------------------------------------------
DH uint32_t ubfe(uint32_t v, uint32_t off, uint32_t width) { return __builtin_amdgcn_ubfe(v, off, width); }
DH int32_t sbfe(uint32_t v, uint32_t off, uint32_t width) { return __builtin_amdgcn_sbfe((int32_t)v, off, width); }
// (opaque(), pin_vgpr(), bfi(), and_or(), or3(), and3(): gfx950_prims.h)
// all-ones if bit `bit` of v is set, else 0 (v_bfe_i32 of a 1-bit field)
DH uint32_t bit_to_mask(uint32_t v, uint32_t bit) { return opaque((uint32_t)__builtin_amdgcn_sbfe((int32_t)v, bit, 1u)); }
// all-ones if c, else 0: one v_cndmask to build the mask, then any number of v_bfi_b32 selects
DH uint32_t cond_to_mask(bool c) { return opaque(c ? 0xFFFFFFFFu : 0u); }
// byte permute: result byte i = byte sel[i] of the 8-byte pool {s0 (4..7), s1 (0..3)}; 0x0C -> 0x00
DH uint32_t perm(uint32_t s0, uint32_t s1, uint32_t sel) { return __builtin_amdgcn_perm(s0, s1, sel); }
DH int32_t clampi(int32_t v, int32_t lo, int32_t hi) { return min(max(v, lo), hi); }
DH uint32_t clamp255(int32_t v) { return (uint32_t)clampi(v, 0, 255); }
DH uint32_t pack_rgba(uint32_t r, uint32_t g, uint32_t b, uint32_t a) { return r | (g << 8) | (b << 16) | (a << 24); }
DH uint32_t bswap32(uint32_t v) { return perm(0u, v, 0x00010203u); }

// pack16(lo, hi) = lo[15:0] | hi[15:0] << 16 (one v_perm_b32)
DH uint32_t pack16(uint32_t lo, uint32_t hi) { return perm(hi, lo, 0x05040100u); }
// (the packed 16-bit operations pk_add16 ... pk_mad_u16, sat_u8_pk16: gfx950_prims.h)

// 4-way select of packed values by a 2-bit selector held as two lane masks
DH uint32_t select4(uint32_t m_lo, uint32_t m_hi, uint32_t p0, uint32_t p1, uint32_t p2, uint32_t p3) {
	return bfi(m_hi, bfi(m_lo, p3, p2), bfi(m_lo, p1, p0));
}

// spread four 3-bit codes (bits 0..11 of c) into the four bytes of a v_perm selector
DH uint32_t spread3to8(uint32_t c) {
	return (c & 0x7u) | ((c << 5) & 0x700u) | ((c << 10) & 0x70000u) | ((c << 15) & 0x7000000u);
}

// ---- 128-bit little-endian block viewed as four dwords -------------------------------------
struct Bits128 { uint32_t w[4]; };
// bits [pos, pos+32) of the block for a per-lane pos in 0..127 (bits beyond 127 read as 0).
// Straight-line: the dword pair is picked with two lane masks (v_bfe_i32) and five v_bfi_b32,
// then one v_alignbit_b32 -- no dynamic register indexing, no exec-mask branches.
DH uint32_t extract32(const Bits128 &b, uint32_t pos) {
	const uint32_t k0 = bit_to_mask(pos, 5), k1 = bit_to_mask(pos, 6);
	const uint32_t x_lo = bfi(k1, b.w[2], b.w[0]), x_mid = bfi(k1, b.w[3], b.w[1]), x_hi = b.w[2] & ~k1;
	return __builtin_amdgcn_alignbit(bfi(k0, x_hi, x_mid), bfi(k0, x_mid, x_lo), pos & 31u);
}

// ---- workgroup-shared lookup tables in LDS ---------------------------------------------------------
// A decoder that needs format tables declares `static DH void prepare()`: every kernel calls
// prepare_tables<Dec>() with all 256 threads before anything else (it ends in a workgroup barrier),
// which copies the tables from __constant__ memory into LDS once per workgroup.  A table lookup by a
// per-lane index is then one ds_read instead of a dependent global load queued behind the kernel's own
// streaming stores (the BPTC decoders chain three such lookups in front of all their arithmetic).
template <class D> DH auto call_prepare(int) -> decltype(D::prepare(), void()) { D::prepare(); }
template <class D> DH void call_prepare(long) {}
template <class D> DH void prepare_tables() { call_prepare<D>(0); }

// ---- per-lane private rows in LDS ------------------------------------------------------------
// ROWS values of T owned by each lane of a 256-thread workgroup, laid out [row][lane] so that any
// mix of per-lane row numbers is bank-conflict free.  Used where a decoder must pick one of a few
// per-block values by a per-lane, per-texel index: a dynamic register index would become a chain
// of v_bfi/v_cndmask (VALU, the scarce resource of the BPTC kernels), whereas ds_read_b128 with
// a computed address costs one address op and runs on the otherwise idle LDS pipe.  A lane only
// ever reads what it wrote itself: no barrier.  TAG keeps distinct tables distinct.
template <class T, int ROWS, int TAG> struct LaneRows {
	T *p;
	DH LaneRows() {
		__shared__ T mem[ROWS * 256];
		p = mem + threadIdx.x;
	}
	DH void put(int r, const T &v) { p[r * 256] = v; }
	DH T get(uint32_t r) const { return p[r * 256u]; }
};
#endif  // DETEXHIP_DEVICE_CODE

}  // namespace detexhip
