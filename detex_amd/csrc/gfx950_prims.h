// gfx950_prims.h -- every primitive of the decoders and kernels that is written in gfx950 inline assembly, in clang's vector
// extensions or against a compiler quirk, in ONE place.  Device code only; the decoders and kernels above it contain no
// `#if` for any of this.
//
// tests/host_emul/hip_host_shim.h defines the SAME names in plain C++ (and this file's include guard), so that g++ can compile
// the decoders in the GPU-less container and run their logic against the oracle (tests/test_host_logic.py): what those tests
// cannot cover is exactly this file -- the instruction selection itself -- which the -m gpu parity tests cover on the hardware.
#ifndef DETEXHIP_GFX950_PRIMS_H
#define DETEXHIP_GFX950_PRIMS_H
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DETEX_HD __host__ __device__ __forceinline__
#define DH __device__ __forceinline__
// 24-bit multiply (v_mul_u32_u24, full rate; a plain 32-bit '*' of unbounded operands is v_mul_lo_u32)
#define DETEX_UMUL24(a, b) __umul24((a), (b))

namespace detexhip {

// Values made OPAQUE to the optimiser (an empty asm: no instruction).  Lane masks: left visible, hipcc proves a mask is 0 / ~0 and
// rewrites every (a & m) | (b & ~m) into v_cmp + v_cndmask_b32; runs of VOP2-encoded v_cndmask_b32 issue at ~23 cycles each on
// MI355X (tools/ubench/valu_rates.hip: 23.3 vs 4.5 for v_bfi_b32 and 4.4 for the VOP3 encoding) -- it made the *shorter*
// unsigned BC6H kernel 1.5x slower than the signed one.  Selects then stay v_bfi_b32 / v_bitop3_b32.
DH uint32_t opaque(uint32_t m) { asm("" : "+v"(m)); return m; }
// Constants and loaded words PINNED into VGPRs at this point (asm volatile: also a scheduling point).  v_bitop3_b32 and the VOP3P
// instructions take no literal on gfx950: left alone the compiler keeps such constants in SGPRs, and a full-rate VALU operation
// with an SGPR source issues at half rate (valu_rates.hip: and_sgpr / bitop3_sgpr); a load whose words are pinned together stays
// ONE wide load and is waited for here.
template <class A> DH void pin_vgpr(A &a) { asm volatile("" : "+v"(a)); }
template <class A, class B> DH void pin_vgpr(A &a, B &b) { asm volatile("" : "+v"(a), "+v"(b)); }
template <class A, class B, class C, class D> DH void pin_vgpr(A &a, B &b, C &c, D &d) { asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
template <class A, class B, class C, class D, class E> DH void pin_vgpr(A &a, B &b, C &c, D &d, E &e) { asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e)); }

// (a & m) | (b & ~m) as v_bitop3_b32 (truth table 0xCA for m, a, b): 2.5 cycles per wave64 like the plain VOP2 logic
// operations, against 4.5 for v_bfi_b32 / v_and_or_b32 / v_perm_b32 (profiles/r01/valu_rates.txt)
DH uint32_t bfi(uint32_t m, uint32_t a, uint32_t b) { return __builtin_amdgcn_bitop3_b32(m, a, b, 0xCA); }
// (a & b) | c, a | b | c and a & b & c through the same instruction
DH uint32_t and_or(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0xEA); }
DH uint32_t or3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0xFE); }
DH uint32_t and3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x80); }

// the high 16-bit half of v shifted left by S: ONE v_lshlrev_b32 with sub-dword source selection (SDWA src1_sel:WORD_1); the
// compiler's own form of (v >> 16) << S is a shift and a mask
template <int S> DH uint32_t high_half_shl(uint32_t v) {
	uint32_t r;
	asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(r) : "n"(S), "v"(v));
	return r;
}
// min(v, 1) as ONE v_min_u32 (the compiler canonicalises it to compare + select)
DH uint32_t nonzero_as_one(uint32_t v) { uint32_t r; asm("v_min_u32 %0, 1, %1" : "=v"(r) : "v"(v)); return r; }
// clamp(x, lo, hi) as ONE v_med3_i32 (left to itself the compiler builds it from two compares and two selects)
DH int32_t med3_i32(int32_t x, int32_t lo, int32_t hi) { int32_t r; asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(lo), "v"(hi)); return r; }

// 8- and 16-byte vectors that move as ONE ds / global instruction (clang vector extensions)
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// ---- two 16-bit lanes per VGPR (VOP3P: v_pk_add_u16 / v_pk_sub_u16 / v_pk_ashrrev_i16 / v_pk_mad_u16 ...) --------------------------
typedef int16_t pk_i16 __attribute__((ext_vector_type(2)));
typedef uint16_t pk_u16 __attribute__((ext_vector_type(2)));
DH pk_i16 to_pk_i16(uint32_t v) { pk_i16 r; __builtin_memcpy(&r, &v, 4); return r; }
DH uint32_t of_pk_i16(pk_i16 v) { uint32_t r; __builtin_memcpy(&r, &v, 4); return r; }
DH pk_u16 to_pk_u16(uint32_t v) { pk_u16 r; __builtin_memcpy(&r, &v, 4); return r; }
DH uint32_t of_pk_u16(pk_u16 v) { uint32_t r; __builtin_memcpy(&r, &v, 4); return r; }
DH uint32_t pk_add16(uint32_t a, uint32_t b) { return of_pk_i16(to_pk_i16(a) + to_pk_i16(b)); }
DH uint32_t pk_sub16(uint32_t a, uint32_t b) { return of_pk_i16(to_pk_i16(a) - to_pk_i16(b)); }
DH uint32_t pk_ashr16(uint32_t a, int s) { return of_pk_i16(to_pk_i16(a) >> (int16_t)s); }
DH uint32_t pk_mul16(uint32_t a, uint32_t b) { return of_pk_i16(to_pk_i16(a) * to_pk_i16(b)); }
DH uint32_t pk_max16(uint32_t a, uint32_t b) { return of_pk_i16(__builtin_elementwise_max(to_pk_i16(a), to_pk_i16(b))); }	// signed
DH uint32_t pk_min16(uint32_t a, uint32_t b) { return of_pk_i16(__builtin_elementwise_min(to_pk_i16(a), to_pk_i16(b))); }	// signed
DH uint32_t pk_lshl16(uint32_t a, int s) { return of_pk_i16(to_pk_i16(a) << (int16_t)s); }
DH uint32_t pk_lshr16(uint32_t a, int s) { return of_pk_u16(to_pk_u16(a) >> (uint16_t)s); }
// both signed 16-bit lanes clamped to 0..255: lane 0 -> byte 0, lane 1 -> byte 1.  Only bytes 0
// and 1 of the result may be used (callers gather them with v_perm_b32).
DH uint32_t sat_u8_pk16(uint32_t a) { uint32_t r; asm("v_sat_pk_u8_i16 %0, %1" : "=v"(r) : "v"(a)); return r; }
// a * b + c and a - b per lane, lanes wrap mod 2^16
DH uint32_t pk_mad_u16(uint32_t a, uint32_t b, uint32_t c) { return of_pk_u16(to_pk_u16(a) * to_pk_u16(b) + to_pk_u16(c)); }
DH uint32_t pk_sub_u16(uint32_t a, uint32_t b) { return of_pk_u16(to_pk_u16(a) - to_pk_u16(b)); }
// both 16-bit lanes of a multiplied by the HIGH / the LOW half of b (VOP3P op_sel broadcast: no v_perm_b32 needed to
// duplicate a weight into both lanes), plus c
DH uint32_t pk_mad_u16_bhi(uint32_t a, uint32_t b, uint32_t c) {
	uint32_t r;
	asm("v_pk_mad_u16 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(r) : "v"(a), "v"(b), "v"(c));
	return r;
}
DH uint32_t pk_mad_u16_blo(uint32_t a, uint32_t b, uint32_t c) {
	uint32_t r;
	asm("v_pk_mad_u16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(b), "v"(c));
	return r;
}
// per-lane shift amounts (< 16) in the 16-bit lanes of s: v_pk_lshlrev_b16 / v_pk_lshrrev_b16 (vector shifts rather
// than inline asm, so that compile-time-constant amounts become inline operands instead of VGPRs)
DH uint32_t pk_lshl_v(uint32_t s, uint32_t a) { return of_pk_u16(to_pk_u16(a) << to_pk_u16(s)); }
DH uint32_t pk_lshr_v(uint32_t s, uint32_t a) { return of_pk_u16(to_pk_u16(a) >> to_pk_u16(s)); }

// Issue priority of this wave among the waves of its SIMD (s_setprio, 0..3; higher is served first).  STAGE is the
// position in the decode (0 = block arrived, 1 = endpoints done, 2 = second half of the texels), POLICY the Tune
// constant: 0 = leave the hardware's arbitration alone; 1 = priority rises with progress, so waves close to their
// stores finish first and completions -- and with them the store traffic -- spread out instead of arriving in
// generations; 2 = only the last stage is raised; 3 = the reverse of 1 (control experiment).
template <int POLICY, int STAGE> DH void stage_priority() {
	if constexpr (POLICY == 1) __builtin_amdgcn_s_setprio(STAGE + 1);
	else if constexpr (POLICY == 2) { if constexpr (STAGE == 2) __builtin_amdgcn_s_setprio(3); }
	else if constexpr (POLICY == 3) __builtin_amdgcn_s_setprio(2 - STAGE);
}
// a pause of N * 64 cycles (measurement builds: between a wave's row stores)
template <int N> DH void sleep_cycles64() { if constexpr (N > 0) __builtin_amdgcn_s_sleep(N); }

// A streaming store of 4 / 8 / 12 / 16 bytes with cache policy POLICY: bit 0 = sc0, bit 1 = sc1, bit 2 = nt (gfx940+; sc0 / sc1 are
// the coherence scope, nt the non-temporal hint).  __builtin_nontemporal_store emits `nt` alone (4).  Measured: without nt the decode
// kernels lose a quarter (write-allocate in L2 beside the block stream); `sc1 nt` (6) beats plain `nt` by 1-3 % (RGTC1: 15 %) for the
// kernels with 32-bit and narrower pixels at 8192^2 and 16384^2 and loses 2.7 % for the 64-bit pixels of BC6H.
// (The compiler has no way to emit these policies, so the instruction is inline asm -- and inline asm is opaque to the hazard
// recognizer: a VMEM store of more than 8 bytes reads its data registers up to two cycles AFTER it issues, and a VALU instruction
// must not overwrite them in that window (gfx940: two wait states).  The register allocator reuses a row's registers for the next
// row's address at once: without the s_nop behind the store a tenth of the first texel rows of BC7 blocks carried eight bytes of
// pointer, differently on every run; tests/test_gpu_host_multi.py caught it.)
template <int POLICY, int DWORDS, class V, class P> DH void store_with_policy(V v, P *p) {
	static_assert(DWORDS >= 1 && DWORDS <= 4, "dword .. dwordx4");
	if constexpr (POLICY == 4) __builtin_nontemporal_store(v, p);
	else if constexpr (POLICY == 0) *p = v;
	else {
#define DETEXHIP_STORE(BITS) \
		if constexpr (DWORDS == 1) asm volatile("global_store_dword %0, %1, off " BITS :: "v"(p), "v"(v) : "memory"); \
		else if constexpr (DWORDS == 2) asm volatile("global_store_dwordx2 %0, %1, off " BITS :: "v"(p), "v"(v) : "memory"); \
		else if constexpr (DWORDS == 3) asm volatile("global_store_dwordx3 %0, %1, off " BITS "\n\ts_nop 1" :: "v"(p), "v"(v) : "memory"); \
		else asm volatile("global_store_dwordx4 %0, %1, off " BITS "\n\ts_nop 1" :: "v"(p), "v"(v) : "memory")
		if constexpr (POLICY == 1) { DETEXHIP_STORE("sc0"); }
		else if constexpr (POLICY == 2) { DETEXHIP_STORE("sc1"); }
		else if constexpr (POLICY == 3) { DETEXHIP_STORE("sc0 sc1"); }
		else if constexpr (POLICY == 5) { DETEXHIP_STORE("sc0 nt"); }
		else if constexpr (POLICY == 6) { DETEXHIP_STORE("sc1 nt"); }
		else { DETEXHIP_STORE("sc0 sc1 nt"); }
#undef DETEXHIP_STORE
	}
}

// MEASUREMENT BUILDS ONLY (Tune::kLoadPolicy): the block load with an explicit cache policy (bits as for the stores), waited for at once
template <int POLICY> DH void load_with_policy(const uint2 *p, uint2 &blk) {
	u32x2 b;
#define DETEXHIP_LOAD(BITS) asm volatile("global_load_dwordx2 %0, %1, off " BITS "\n\ts_waitcnt vmcnt(0)" : "=&v"(b) : "v"(p) : "memory")
	if constexpr (POLICY == 1) { DETEXHIP_LOAD("sc0"); } else if constexpr (POLICY == 2) { DETEXHIP_LOAD("sc1"); } else if constexpr (POLICY == 3) { DETEXHIP_LOAD("sc0 sc1"); }
	else if constexpr (POLICY == 4) { DETEXHIP_LOAD("nt"); } else if constexpr (POLICY == 5) { DETEXHIP_LOAD("sc0 nt"); } else if constexpr (POLICY == 6) { DETEXHIP_LOAD("sc1 nt"); }
	else { DETEXHIP_LOAD("sc0 sc1 nt"); }
#undef DETEXHIP_LOAD
	blk = uint2{ b.x, b.y };
}
template <int POLICY> DH void load_with_policy(const uint4 *p, uint4 &blk) {
	u32x4 b;
#define DETEXHIP_LOAD(BITS) asm volatile("global_load_dwordx4 %0, %1, off " BITS "\n\ts_waitcnt vmcnt(0)" : "=&v"(b) : "v"(p) : "memory")
	if constexpr (POLICY == 1) { DETEXHIP_LOAD("sc0"); } else if constexpr (POLICY == 2) { DETEXHIP_LOAD("sc1"); } else if constexpr (POLICY == 3) { DETEXHIP_LOAD("sc0 sc1"); }
	else if constexpr (POLICY == 4) { DETEXHIP_LOAD("nt"); } else if constexpr (POLICY == 5) { DETEXHIP_LOAD("sc0 nt"); } else if constexpr (POLICY == 6) { DETEXHIP_LOAD("sc1 nt"); }
	else { DETEXHIP_LOAD("sc0 sc1 nt"); }
#undef DETEXHIP_LOAD
	blk = uint4{ b.x, b.y, b.z, b.w };
}

// A block load with a second, never-used load of the same width issued right behind it (a read-ahead: `ahead` is only wanted in the
// caches).  Loads return in order, so the wait is for all but the last one: the block is there, the read-ahead still travelling.  The
// compiler does not know about that outstanding load: the caller keeps `sink` alive to the end of the kernel (keep_alive) so that its
// registers are not handed to anything else before the data lands; s_endpgm waits for it.
template <class Word> struct ReadAheadSink;
template <> struct ReadAheadSink<uint4> { u32x4 v; };
template <> struct ReadAheadSink<uint2> { u32x2 v; };
DH void load_with_read_ahead(const uint4 *block, const uint4 *ahead, uint4 &blk, ReadAheadSink<uint4> &sink) {
	u32x4 b;
	asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %3, off\n\ts_waitcnt vmcnt(1)" : "=&v"(b), "=&v"(sink.v) : "v"(block), "v"(ahead) : "memory");
	blk = uint4{ b.x, b.y, b.z, b.w };
}
DH void load_with_read_ahead(const uint2 *block, const uint2 *ahead, uint2 &blk, ReadAheadSink<uint2> &sink) {
	u32x2 b;
	asm volatile("global_load_dwordx2 %0, %2, off\n\tglobal_load_dwordx2 %1, %3, off\n\ts_waitcnt vmcnt(1)" : "=&v"(b), "=&v"(sink.v) : "v"(block), "v"(ahead) : "memory");
	blk = uint2{ b.x, b.y };
}
template <class Word> DH void keep_alive(const ReadAheadSink<Word> &s) { asm volatile("" :: "v"(s.v)); }

// Four 16-byte loads that go to memory whatever the caches hold (sc0 sc1: system scope), issued back to back and waited for once:
// the resident kernel's poll of its request line in pinned host memory (kernels_resident.h) -- one round trip across the link.
DH void load_system_4x16(const void *p, u32x4 &a, u32x4 &b, u32x4 &c, u32x4 &d) {
	asm volatile("global_load_dwordx4 %0, %4, off sc0 sc1\n\t"
		"global_load_dwordx4 %1, %4, off offset:16 sc0 sc1\n\t"
		"global_load_dwordx4 %2, %4, off offset:32 sc0 sc1\n\t"
		"global_load_dwordx4 %3, %4, off offset:48 sc0 sc1\n\t"
		"s_waitcnt vmcnt(0)"
		: "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(p) : "memory");
}
// The same poll with two more 16-byte loads from a second address (the lane's tagged block chunks), all six in flight together.
DH void load_system_4x16_and_2x16(const void *p, const void *q, u32x4 &a, u32x4 &b, u32x4 &c, u32x4 &d, u32x4 &e, u32x4 &f) {
	asm volatile("global_load_dwordx4 %0, %6, off sc0 sc1\n\t"
		"global_load_dwordx4 %1, %6, off offset:16 sc0 sc1\n\t"
		"global_load_dwordx4 %2, %6, off offset:32 sc0 sc1\n\t"
		"global_load_dwordx4 %3, %6, off offset:48 sc0 sc1\n\t"
		"global_load_dwordx4 %4, %7, off sc0 sc1\n\t"
		"global_load_dwordx4 %5, %7, off offset:16 sc0 sc1\n\t"
		"s_waitcnt vmcnt(0)"
		: "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d), "=&v"(e), "=&v"(f) : "v"(p), "v"(q) : "memory");
}
DH void load_system_2x16(const void *q, u32x4 &e, u32x4 &f) {
	asm volatile("global_load_dwordx4 %0, %2, off sc0 sc1\n\t"
		"global_load_dwordx4 %1, %2, off offset:16 sc0 sc1\n\t"
		"s_waitcnt vmcnt(0)"
		: "=&v"(e), "=&v"(f) : "v"(q) : "memory");
}

}  // namespace detexhip
#endif
