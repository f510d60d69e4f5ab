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

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define DETEX_HD __host__ __device__ __forceinline__
#define DH __device__ __forceinline__
#define DETEXHIP_DEVICE_CODE 1
#elif defined(DETEXHIP_HOST_EMULATION)
// test-only: tests/host_emul/hip_host_shim.h emulates the gfx950 builtins so the decoders below
// can be exercised by g++ in containers without a GPU (never part of libdetexhip.so)
#include "hip_host_shim.h"
#define DETEX_HD static inline
#define DETEXHIP_DEVICE_CODE 1
#else
#define DETEX_HD static inline
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
// 24-bit multiply (v_mul_u32_u24, full rate; a plain 32-bit '*' of unbounded operands is v_mul_lo_u32)
#if defined(__HIPCC__)
#define DETEX_UMUL24(a, b) __umul24((a), (b))
#else
#define DETEX_UMUL24(a, b) (((a) & 0xFFFFFFu) * ((b) & 0xFFFFFFu))
#endif
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
// Lane masks are made OPAQUE to the optimiser.  Left visible, hipcc proves a mask is 0 / ~0 and
// rewrites every (a & m) | (b & ~m) into v_cmp + v_cndmask_b32; runs of VOP2-encoded
// v_cndmask_b32 issue at ~23 cycles each on MI355X (tools/ubench/valu_rates.hip: 23.3 vs 4.5 for
// v_bfi_b32 and 4.4 for the VOP3 encoding) -- it made the *shorter* unsigned BC6H kernel 1.5x slower
// than the signed one.  The empty asm costs no instruction; selects then stay v_bfi_b32.
#if defined(__HIPCC__)
DH uint32_t opaque(uint32_t m) { asm("" : "+v"(m)); return m; }
#else
DH uint32_t opaque(uint32_t m) { return m; }
#endif
// all-ones if bit `bit` of v is set, else 0 (v_bfe_i32 of a 1-bit field)
DH uint32_t bit_to_mask(uint32_t v, uint32_t bit) { return opaque((uint32_t)__builtin_amdgcn_sbfe((int32_t)v, bit, 1u)); }
// all-ones if c, else 0: one v_cndmask to build the mask, then any number of v_bfi_b32 selects
DH uint32_t cond_to_mask(bool c) { return opaque(c ? 0xFFFFFFFFu : 0u); }
// (a & m) | (b & ~m).  Written as gfx950's v_bitop3_b32 (truth table 0xCA for m, a, b): it issues in 2.5
// cycles per wave64 like the plain VOP2 logic ops, against 4.5 for v_bfi_b32 / v_and_or_b32 / v_perm_b32
// (tools/ubench/valu_rates.hip; profiles/r01/valu_rates.txt).
#if defined(__HIPCC__)
DH uint32_t bfi(uint32_t m, uint32_t a, uint32_t b) { return __builtin_amdgcn_bitop3_b32(m, a, b, 0xCA); }
// (a & b) | c and a | b | c through the same instruction
DH uint32_t and_or(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0xEA); }
DH uint32_t or3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0xFE); }
DH uint32_t and3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x80); }
#else
DH uint32_t bfi(uint32_t m, uint32_t a, uint32_t b) { return (a & m) | (b & ~m); }
DH uint32_t and_or(uint32_t a, uint32_t b, uint32_t c) { return (a & b) | c; }
DH uint32_t or3(uint32_t a, uint32_t b, uint32_t c) { return a | b | c; }
DH uint32_t and3(uint32_t a, uint32_t b, uint32_t c) { return a & b & c; }
#endif
// byte permute: result byte i = byte sel[i] of the 8-byte pool {s0 (4..7), s1 (0..3)}; 0x0C -> 0x00
DH uint32_t perm(uint32_t s0, uint32_t s1, uint32_t sel) { return __builtin_amdgcn_perm(s0, s1, sel); }
// the high 16-bit half of v shifted left by S: ONE v_lshlrev_b32 with sub-dword source selection (SDWA src1_sel:WORD_1); the
// compiler's own form of (v >> 16) << S is a shift and a mask
template <int S> DH uint32_t high_half_shl(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
	uint32_t r;
	asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(r) : "n"(S), "v"(v));
	return r;
#else
	return (v >> 16) << S;
#endif
}
// min(v, 1) as ONE v_min_u32 (the compiler canonicalises it to compare + select)
DH uint32_t nonzero_as_one(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
	uint32_t r;
	asm("v_min_u32 %0, 1, %1" : "=v"(r) : "v"(v));
	return r;
#else
	return v ? 1u : 0u;
#endif
}
DH int32_t clampi(int32_t v, int32_t lo, int32_t hi) { return min(max(v, lo), hi); }
DH uint32_t clamp255(int32_t v) { return (uint32_t)clampi(v, 0, 255); }
DH uint32_t pack_rgba(uint32_t r, uint32_t g, uint32_t b, uint32_t a) { return r | (g << 8) | (b << 16) | (a << 24); }
DH uint32_t bswap32(uint32_t v) { return perm(0u, v, 0x00010203u); }

// ---- two signed 16-bit lanes per VGPR (v_pk_add_u16 / v_pk_sub_u16 / v_pk_ashrrev_i16) --------
// pack16(lo, hi) = lo[15:0] | hi[15:0] << 16 (one v_perm_b32)
DH uint32_t pack16(uint32_t lo, uint32_t hi) { return perm(hi, lo, 0x05040100u); }
#if defined(__HIPCC__)
typedef int16_t pk_i16 __attribute__((ext_vector_type(2)));
DH pk_i16 to_pk_i16(uint32_t v) { pk_i16 r; __builtin_memcpy(&r, &v, 4); return r; }
DH uint32_t of_pk_i16(pk_i16 v) { uint32_t r; __builtin_memcpy(&r, &v, 4); return r; }
DH uint32_t pk_add16(uint32_t a, uint32_t b) { return of_pk_i16(to_pk_i16(a) + to_pk_i16(b)); }
DH uint32_t pk_sub16(uint32_t a, uint32_t b) { return of_pk_i16(to_pk_i16(a) - to_pk_i16(b)); }
DH uint32_t pk_ashr16(uint32_t a, int s) { return of_pk_i16(to_pk_i16(a) >> (int16_t)s); }
DH uint32_t pk_mul16(uint32_t a, uint32_t b) { return of_pk_i16(to_pk_i16(a) * to_pk_i16(b)); }
DH uint32_t pk_max16(uint32_t a, uint32_t b) { return of_pk_i16(__builtin_elementwise_max(to_pk_i16(a), to_pk_i16(b))); }	// signed
DH uint32_t pk_min16(uint32_t a, uint32_t b) { return of_pk_i16(__builtin_elementwise_min(to_pk_i16(a), to_pk_i16(b))); }	// signed
DH uint32_t pk_lshl16(uint32_t a, int s) { return of_pk_i16(to_pk_i16(a) << (int16_t)s); }
typedef uint16_t pk_u16 __attribute__((ext_vector_type(2)));
DH uint32_t pk_lshr16(uint32_t a, int s) { pk_u16 v; __builtin_memcpy(&v, &a, 4); v = v >> (uint16_t)s; uint32_t r; __builtin_memcpy(&r, &v, 4); return r; }
// both signed 16-bit lanes clamped to 0..255: lane 0 -> byte 0, lane 1 -> byte 1.  Only bytes 0
// and 1 of the result may be used (callers gather them with v_perm_b32).
DH uint32_t sat_u8_pk16(uint32_t a) { uint32_t r; asm("v_sat_pk_u8_i16 %0, %1" : "=v"(r) : "v"(a)); return r; }
#else
DH uint32_t pk_add16(uint32_t a, uint32_t b) { return ((a + b) & 0xFFFFu) | (((a >> 16) + (b >> 16)) << 16); }
DH uint32_t pk_sub16(uint32_t a, uint32_t b) { return ((a - b) & 0xFFFFu) | (((a >> 16) - (b >> 16)) << 16); }
DH uint32_t pk_ashr16(uint32_t a, int s) {
	return ((uint32_t)((int32_t)(int16_t)(a & 0xFFFFu) >> s) & 0xFFFFu) | ((uint32_t)((int32_t)(int16_t)(a >> 16) >> s) << 16);
}
DH uint32_t pk_mul16(uint32_t a, uint32_t b) { return ((a * b) & 0xFFFFu) | (((a >> 16) * (b >> 16)) << 16); }
DH uint32_t pk_max16(uint32_t a, uint32_t b) {
	const int32_t al = (int16_t)(a & 0xFFFFu), ah = (int16_t)(a >> 16), bl = (int16_t)(b & 0xFFFFu), bh = (int16_t)(b >> 16);
	return ((uint32_t)max(al, bl) & 0xFFFFu) | ((uint32_t)max(ah, bh) << 16);
}
DH uint32_t pk_lshr16(uint32_t a, int s) { return ((a & 0xFFFFu) >> s) | (((a >> 16) >> s) << 16); }
DH uint32_t pk_min16(uint32_t a, uint32_t b) {
	const int32_t al = (int16_t)(a & 0xFFFFu), ah = (int16_t)(a >> 16), bl = (int16_t)(b & 0xFFFFu), bh = (int16_t)(b >> 16);
	return ((uint32_t)min(al, bl) & 0xFFFFu) | ((uint32_t)min(ah, bh) << 16);
}
DH uint32_t pk_lshl16(uint32_t a, int s) { return ((a << s) & 0xFFFFu) | ((((a >> 16) << s) & 0xFFFFu) << 16); }
DH uint32_t sat_u8_pk16(uint32_t a) {
	const int32_t lo = (int16_t)(a & 0xFFFFu), hi = (int16_t)(a >> 16);
	return (uint32_t)clampi(lo, 0, 255) | ((uint32_t)clampi(hi, 0, 255) << 8) | 0xDEAD0000u;	// poison the unspecified half
}
#endif

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

// Issue priority of this wave among the waves of its SIMD (s_setprio, 0..3; higher is served first).  STAGE is the
// position in the decode (0 = block arrived, 1 = endpoints done, 2 = second half of the texels), POLICY the Tune
// constant: 0 = leave the hardware's arbitration alone; 1 = priority rises with progress, so waves close to their
// stores finish first and completions -- and with them the store traffic -- spread out instead of arriving in
// generations; 2 = only the last stage is raised; 3 = the reverse of 1 (control experiment).
template <int POLICY, int STAGE> DH void stage_priority() {
#if defined(__HIP_DEVICE_COMPILE__)
	if constexpr (POLICY == 1) __builtin_amdgcn_s_setprio(STAGE + 1);
	else if constexpr (POLICY == 2) { if constexpr (STAGE == 2) __builtin_amdgcn_s_setprio(3); }
	else if constexpr (POLICY == 3) __builtin_amdgcn_s_setprio(2 - STAGE);
#endif
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
#if defined(__HIPCC__)
	T *p;
	DH LaneRows() {
		__shared__ T mem[ROWS * 256];
		p = mem + threadIdx.x;
	}
	DH void put(int r, const T &v) { p[r * 256] = v; }
	DH T get(uint32_t r) const { return p[r * 256u]; }
#else
	T mem[ROWS];
	DH LaneRows() {}
	DH void put(int r, const T &v) { mem[r] = v; }
	DH T get(uint32_t r) const { return mem[r]; }
#endif
};
#endif  // DETEXHIP_DEVICE_CODE

}  // namespace detexhip
