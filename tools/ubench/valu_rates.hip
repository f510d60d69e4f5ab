// tools/ubench/valu_rates.hip -- measures the issue cost of the integer VALU ops the decoders are
// built from (cycles per wave64 instruction per SIMD, 8 waves/SIMD resident), to price kernels
// against the 4-cycle baseline seen in profiles/r01.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rates tools/ubench/valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <chrono>
#include <cstring>
#include <cstdlib>
#define REP16(x) x x x x x x x x x x x x x x x x
#define KERNEL(NAME, ASM)                                                                     \
	__global__ __launch_bounds__(256) void k_##NAME(uint32_t *out, uint32_t seed, int iters) {  \
		uint32_t a = threadIdx.x ^ seed, b = a * 3u + 1u, c = a + 7u, d = a ^ 0x55u;            \
		uint32_t s1 = seed | 1u, s2 = seed + 3u;                                               \
		for (int i = 0; i < iters; i++) {                                                       \
			REP16(asm volatile(ASM : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(s1), "v"(s2));)    \
		}                                                                                       \
		out[blockIdx.x * 256 + threadIdx.x] = a ^ b ^ c ^ d;                                    \
	}
// each asm = 4 independent instructions (one per chain)
KERNEL(v_add_u32, "v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %5\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %5")
KERNEL(v_mul_u32_u24, "v_mul_u32_u24 %0, %0, %4\n v_mul_u32_u24 %1, %1, %5\n v_mul_u32_u24 %2, %2, %4\n v_mul_u32_u24 %3, %3, %5")
KERNEL(v_mul_i32_i24, "v_mul_i32_i24 %0, %0, %4\n v_mul_i32_i24 %1, %1, %5\n v_mul_i32_i24 %2, %2, %4\n v_mul_i32_i24 %3, %3, %5")
KERNEL(v_mad_i32_i24, "v_mad_i32_i24 %0, %0, %4, %5\n v_mad_i32_i24 %1, %1, %5, %4\n v_mad_i32_i24 %2, %2, %4, %5\n v_mad_i32_i24 %3, %3, %5, %4")
KERNEL(v_mad_u32_u24, "v_mad_u32_u24 %0, %0, %4, %5\n v_mad_u32_u24 %1, %1, %5, %4\n v_mad_u32_u24 %2, %2, %4, %5\n v_mad_u32_u24 %3, %3, %5, %4")
KERNEL(v_mul_lo_u32, "v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %5\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %5")
KERNEL(v_mul_hi_u32, "v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %5\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %5")
KERNEL(v_bfe_u32, "v_bfe_u32 %0, %0, %4, %5\n v_bfe_u32 %1, %1, %5, %4\n v_bfe_u32 %2, %2, %4, %5\n v_bfe_u32 %3, %3, %5, %4")
KERNEL(v_bfi_b32, "v_bfi_b32 %0, %0, %4, %5\n v_bfi_b32 %1, %1, %5, %4\n v_bfi_b32 %2, %2, %4, %5\n v_bfi_b32 %3, %3, %5, %4")
KERNEL(v_perm_b32, "v_perm_b32 %0, %0, %4, %5\n v_perm_b32 %1, %1, %5, %4\n v_perm_b32 %2, %2, %4, %5\n v_perm_b32 %3, %3, %5, %4")
KERNEL(v_alignbit_b32, "v_alignbit_b32 %0, %0, %4, %5\n v_alignbit_b32 %1, %1, %5, %4\n v_alignbit_b32 %2, %2, %4, %5\n v_alignbit_b32 %3, %3, %5, %4")
KERNEL(v_lshl_or_b32, "v_lshl_or_b32 %0, %0, %4, %5\n v_lshl_or_b32 %1, %1, %5, %4\n v_lshl_or_b32 %2, %2, %4, %5\n v_lshl_or_b32 %3, %3, %5, %4")
KERNEL(v_and_or_b32, "v_and_or_b32 %0, %0, %4, %5\n v_and_or_b32 %1, %1, %5, %4\n v_and_or_b32 %2, %2, %4, %5\n v_and_or_b32 %3, %3, %5, %4")
KERNEL(v_med3_i32, "v_med3_i32 %0, %0, %4, %5\n v_med3_i32 %1, %1, %5, %4\n v_med3_i32 %2, %2, %4, %5\n v_med3_i32 %3, %3, %5, %4")
KERNEL(v_lshrrev_b32, "v_lshrrev_b32 %0, %4, %0\n v_lshrrev_b32 %1, %5, %1\n v_lshrrev_b32 %2, %4, %2\n v_lshrrev_b32 %3, %5, %3")
KERNEL(v_pk_mad_u16, "v_pk_mad_u16 %0, %0, %4, %5\n v_pk_mad_u16 %1, %1, %5, %4\n v_pk_mad_u16 %2, %2, %4, %5\n v_pk_mad_u16 %3, %3, %5, %4")
KERNEL(v_cndmask_b32, "v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %5, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %5, vcc")
KERNEL(v_cmp_cnd, "v_cmp_lt_u32 vcc, %0, %4\n v_cndmask_b32 %1, %1, %5, vcc\n v_cmp_lt_u32 vcc, %2, %5\n v_cndmask_b32 %3, %3, %4, vcc")
KERNEL(v_bfrev_b32, "v_bfrev_b32 %0, %0\n v_bfrev_b32 %1, %1\n v_bfrev_b32 %2, %2\n v_bfrev_b32 %3, %3")
KERNEL(cnd_sgpr_mask, "v_cndmask_b32 %0, %0, %4, s[20:21]\n v_cndmask_b32 %1, %1, %5, s[20:21]\n v_cndmask_b32 %2, %2, %4, s[20:21]\n v_cndmask_b32 %3, %3, %5, s[20:21]")
KERNEL(cnd_add_1to1, "v_cndmask_b32 %0, %0, %4, vcc\n v_add_u32 %1, %1, %5\n v_cndmask_b32 %2, %2, %4, vcc\n v_add_u32 %3, %3, %5")
KERNEL(cnd_bfe_1to1, "v_cndmask_b32 %0, %0, %4, vcc\n v_bfe_u32 %1, %1, %5, %4\n v_cndmask_b32 %2, %2, %4, vcc\n v_bfe_u32 %3, %3, %5, %4")
KERNEL(cnd_1_in_4, "v_cndmask_b32 %0, %0, %4, vcc\n v_bfe_u32 %1, %1, %5, %4\n v_add_u32 %2, %2, %4\n v_bfe_u32 %3, %3, %5, %4")
KERNEL(cmp_then_3cnd, "v_cmp_lt_u32 vcc, %0, %4\n v_cndmask_b32 %1, %1, %5, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %5, vcc")
KERNEL(cnd_e64_vcc, "v_cndmask_b32_e64 %0, %0, %4, vcc\n v_cndmask_b32_e64 %1, %1, %5, vcc\n v_cndmask_b32_e64 %2, %2, %4, vcc\n v_cndmask_b32_e64 %3, %3, %5, vcc")
KERNEL(bfe_i32_mask, "v_bfe_i32 %0, %0, %4, 1\n v_bfe_i32 %1, %1, %5, 1\n v_bfe_i32 %2, %2, %4, 1\n v_bfe_i32 %3, %3, %5, 1")
// dependency distance: 1 = every instruction consumes the previous one's result, 2 = two interleaved chains
KERNEL(dep1_add, "v_add_u32 %0, %0, %4\n v_add_u32 %0, %0, %5\n v_add_u32 %0, %0, %4\n v_add_u32 %0, %0, %5")
KERNEL(dep2_add, "v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %5\n v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %5")
KERNEL(dep1_bfe, "v_bfe_u32 %0, %0, %4, %5\n v_bfe_u32 %0, %0, %5, %4\n v_bfe_u32 %0, %0, %4, %5\n v_bfe_u32 %0, %0, %5, %4")
KERNEL(dep2_bfe, "v_bfe_u32 %0, %0, %4, %5\n v_bfe_u32 %1, %1, %5, %4\n v_bfe_u32 %0, %0, %4, %5\n v_bfe_u32 %1, %1, %5, %4")
KERNEL(dep1_mix, "v_mad_i32_i24 %0, %0, %4, %5\n v_lshrrev_b32 %0, 6, %0\n v_mul_i32_i24 %0, 31, %0\n v_lshrrev_b32 %0, 6, %0")
KERNEL(dep2_mix, "v_mad_i32_i24 %0, %0, %4, %5\n v_mad_i32_i24 %1, %1, %4, %5\n v_lshrrev_b32 %0, 6, %0\n v_lshrrev_b32 %1, 6, %1")
// encoding size vs operand count: the same add as VOP2 (4 bytes), VOP3 (8 bytes), VOP2 + 32-bit literal (8 bytes)
KERNEL(add_e64, "v_add_u32_e64 %0, %0, %4\n v_add_u32_e64 %1, %1, %5\n v_add_u32_e64 %2, %2, %4\n v_add_u32_e64 %3, %3, %5")
KERNEL(and_literal, "v_and_b32 %0, 0x12345678, %0\n v_and_b32 %1, 0x0f0f0f0f, %1\n v_and_b32 %2, 0x12345678, %2\n v_and_b32 %3, 0x0f0f0f0f, %3")
KERNEL(and_inline, "v_and_b32 %0, 15, %0\n v_and_b32 %1, 7, %1\n v_and_b32 %2, 15, %2\n v_and_b32 %3, 7, %3")
KERNEL(and_sgpr, "v_and_b32 %0, s20, %0\n v_and_b32 %1, s21, %1\n v_and_b32 %2, s20, %2\n v_and_b32 %3, s21, %3")
KERNEL(bfe_inline, "v_bfe_u32 %0, %0, 4, 8\n v_bfe_u32 %1, %1, 3, 9\n v_bfe_u32 %2, %2, 4, 8\n v_bfe_u32 %3, %3, 3, 9")
KERNEL(bfe_sgpr, "v_bfe_u32 %0, %0, s20, 8\n v_bfe_u32 %1, %1, s21, 9\n v_bfe_u32 %2, %2, s20, 8\n v_bfe_u32 %3, %3, s21, 9")
KERNEL(perm_sgpr_sel, "v_perm_b32 %0, %0, %4, s20\n v_perm_b32 %1, %1, %5, s21\n v_perm_b32 %2, %2, %4, s20\n v_perm_b32 %3, %3, %5, s21")
KERNEL(v_mov, "v_mov_b32 %0, %4\n v_mov_b32 %1, %5\n v_mov_b32 %2, %4\n v_mov_b32 %3, %5")
KERNEL(v_sat_pk, "v_sat_pk_u8_i16 %0, %0\n v_sat_pk_u8_i16 %1, %1\n v_sat_pk_u8_i16 %2, %2\n v_sat_pk_u8_i16 %3, %3")
KERNEL(v_pk_add_u16, "v_pk_add_u16 %0, %0, %4\n v_pk_add_u16 %1, %1, %5\n v_pk_add_u16 %2, %2, %4\n v_pk_add_u16 %3, %3, %5")
KERNEL(v_pk_ashr, "v_pk_ashrrev_i16 %0, 2, %0 op_sel_hi:[0,1]\n v_pk_ashrrev_i16 %1, 2, %1 op_sel_hi:[0,1]\n v_pk_ashrrev_i16 %2, 2, %2 op_sel_hi:[0,1]\n v_pk_ashrrev_i16 %3, 2, %3 op_sel_hi:[0,1]")
KERNEL(v_lshl_add, "v_lshl_add_u32 %0, %0, 2, %4\n v_lshl_add_u32 %1, %1, 3, %5\n v_lshl_add_u32 %2, %2, 2, %4\n v_lshl_add_u32 %3, %3, 3, %5")
KERNEL(v_add3, "v_add3_u32 %0, %0, %4, %5\n v_add3_u32 %1, %1, %5, %4\n v_add3_u32 %2, %2, %4, %5\n v_add3_u32 %3, %3, %5, %4")
KERNEL(v_xor, "v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %5\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %5")
KERNEL(v_sub, "v_sub_u32 %0, %0, %4\n v_sub_u32 %1, %1, %5\n v_sub_u32 %2, %2, %4\n v_sub_u32 %3, %3, %5")
KERNEL(v_lshlrev, "v_lshlrev_b32 %0, 3, %0\n v_lshlrev_b32 %1, 5, %1\n v_lshlrev_b32 %2, 3, %2\n v_lshlrev_b32 %3, 5, %3")
KERNEL(v_min_max, "v_max_i32 %0, %0, %4\n v_min_i32 %1, %1, %5\n v_max_i32 %2, %2, %4\n v_min_i32 %3, %3, %5")
KERNEL(v_ffbl, "v_ffbl_b32 %0, %0\n v_ffbl_b32 %1, %1\n v_ffbl_b32 %2, %2\n v_ffbl_b32 %3, %3")
// mixes: do VOP2 and VOP3 overlap, does SALU work ride along for free
KERNEL(mix_add_bfe, "v_add_u32 %0, %0, %4\n v_bfe_u32 %1, %1, %5, %4\n v_add_u32 %2, %2, %4\n v_bfe_u32 %3, %3, %5, %4")
KERNEL(mix_valu_salu, "v_bfe_u32 %0, %0, %4, %5\n s_add_u32 s20, s20, 1\n v_bfe_u32 %1, %1, %5, %4\n s_and_b32 s21, s21, s20\n v_bfe_u32 %2, %2, %4, %5\n s_add_u32 s20, s20, 3\n v_bfe_u32 %3, %3, %5, %4\n s_lshl_b32 s21, s21, 1")
KERNEL(mix_add_salu, "v_add_u32 %0, %0, %4\n s_add_u32 s20, s20, 1\n v_add_u32 %1, %1, %5\n s_and_b32 s21, s21, s20\n v_add_u32 %2, %2, %4\n s_add_u32 s20, s20, 3\n v_add_u32 %3, %3, %5\n s_lshl_b32 s21, s21, 1")
KERNEL(lshl_vgpr, "v_lshlrev_b32 %0, %4, %0\n v_lshlrev_b32 %1, %5, %1\n v_lshlrev_b32 %2, %4, %2\n v_lshlrev_b32 %3, %5, %3")
KERNEL(lshr_inline, "v_lshrrev_b32 %0, 3, %0\n v_lshrrev_b32 %1, 5, %1\n v_lshrrev_b32 %2, 3, %2\n v_lshrrev_b32 %3, 5, %3")
KERNEL(ashr_inline, "v_ashrrev_i32 %0, 3, %0\n v_ashrrev_i32 %1, 5, %1\n v_ashrrev_i32 %2, 3, %2\n v_ashrrev_i32 %3, 5, %3")
KERNEL(v_or, "v_or_b32 %0, %0, %4\n v_or_b32 %1, %1, %5\n v_or_b32 %2, %2, %4\n v_or_b32 %3, %3, %5")
KERNEL(v_not, "v_not_b32 %0, %0\n v_not_b32 %1, %1\n v_not_b32 %2, %2\n v_not_b32 %3, %3")
KERNEL(v_bitop3, "v_bitop3_b32 %0, %0, %4, %5 bitop3:0xe0\n v_bitop3_b32 %1, %1, %5, %4 bitop3:0xe0\n v_bitop3_b32 %2, %2, %4, %5 bitop3:0xe0\n v_bitop3_b32 %3, %3, %5, %4 bitop3:0xe0")
KERNEL(v_bcnt, "v_bcnt_u32_b32 %0, %0, %4\n v_bcnt_u32_b32 %1, %1, %5\n v_bcnt_u32_b32 %2, %2, %4\n v_bcnt_u32_b32 %3, %3, %5")
KERNEL(and_sdwa, "v_and_b32_sdwa %0, %0, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_and_b32_sdwa %1, %1, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_and_b32_sdwa %2, %2, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_and_b32_sdwa %3, %3, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD")
KERNEL(add_sdwa, "v_add_u32_sdwa %0, %0, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_add_u32_sdwa %1, %1, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_add_u32_sdwa %2, %2, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_add_u32_sdwa %3, %3, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD")
KERNEL(mov_sdwa, "v_mov_b32_sdwa %0, %4 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0\n v_mov_b32_sdwa %1, %5 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0\n v_mov_b32_sdwa %2, %4 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0\n v_mov_b32_sdwa %3, %5 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0")
KERNEL(add_lit, "v_add_u32 %0, 0x12345, %0\n v_add_u32 %1, 0x54321, %1\n v_add_u32 %2, 0x12345, %2\n v_add_u32 %3, 0x54321, %3")
KERNEL(max_u32, "v_max_u32 %0, %0, %4\n v_max_u32 %1, %1, %5\n v_max_u32 %2, %2, %4\n v_max_u32 %3, %3, %5")
KERNEL(cmp_only, "v_cmp_lt_u32 vcc, %0, %4\n v_cmp_lt_u32 s[20:21], %1, %5\n v_cmp_lt_u32 vcc, %2, %4\n v_cmp_lt_u32 s[20:21], %3, %5")
KERNEL(pk_mul_lo, "v_pk_mul_lo_u16 %0, %0, %4\n v_pk_mul_lo_u16 %1, %1, %5\n v_pk_mul_lo_u16 %2, %2, %4\n v_pk_mul_lo_u16 %3, %3, %5")
KERNEL(pk_min_max, "v_pk_max_i16 %0, %0, %4\n v_pk_min_i16 %1, %1, %5\n v_pk_max_i16 %2, %2, %4\n v_pk_min_i16 %3, %3, %5")
KERNEL(v_lshrrev_b64, "v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %5\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %5")
// fast-class ops with one SGPR source (round 2: the compiler keeps VOP3 constants such as bitop3's mask in SGPRs)
KERNEL(bitop3_sgpr, "v_bitop3_b32 %0, %0, s20, %5 bitop3:0xea\n v_bitop3_b32 %1, %1, s21, %4 bitop3:0xea\n v_bitop3_b32 %2, %2, s20, %5 bitop3:0xea\n v_bitop3_b32 %3, %3, s21, %4 bitop3:0xea")
KERNEL(bitop3_inline, "v_bitop3_b32 %0, %0, 16, %5 bitop3:0xea\n v_bitop3_b32 %1, %1, 32, %4 bitop3:0xea\n v_bitop3_b32 %2, %2, 16, %5 bitop3:0xea\n v_bitop3_b32 %3, %3, 32, %4 bitop3:0xea")
KERNEL(add_sgpr, "v_add_u32 %0, s20, %0\n v_add_u32 %1, s21, %1\n v_add_u32 %2, s20, %2\n v_add_u32 %3, s21, %3")
KERNEL(xor_sgpr, "v_xor_b32 %0, s20, %0\n v_xor_b32 %1, s21, %1\n v_xor_b32 %2, s20, %2\n v_xor_b32 %3, s21, %3")
KERNEL(lshr_sgpr_amt, "v_lshrrev_b32 %0, s20, %0\n v_lshrrev_b32 %1, s21, %1\n v_lshrrev_b32 %2, s20, %2\n v_lshrrev_b32 %3, s21, %3")
KERNEL(lshr_sgpr_val, "v_lshrrev_b32 %0, %4, s20\n v_lshrrev_b32 %1, %5, s21\n v_lshrrev_b32 %2, %4, s20\n v_lshrrev_b32 %3, %5, s21")
KERNEL(mov_sgpr, "v_mov_b32 %0, s20\n v_mov_b32 %1, s21\n v_mov_b32 %2, s20\n v_mov_b32 %3, s21")
KERNEL(sub_sgpr, "v_sub_u32 %0, s20, %0\n v_sub_u32 %1, s21, %1\n v_sub_u32 %2, s20, %2\n v_sub_u32 %3, s21, %3")
KERNEL(and_vcc_lo, "v_and_b32 %0, vcc_lo, %0\n v_and_b32 %1, vcc_hi, %1\n v_and_b32 %2, vcc_lo, %2\n v_and_b32 %3, vcc_hi, %3")
KERNEL(dpp_mov, "v_mov_b32_dpp %0, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
KERNEL(lshl_inline, "v_lshlrev_b32 %0, 2, %0\n v_lshlrev_b32 %1, 3, %1\n v_lshlrev_b32 %2, 2, %2\n v_lshlrev_b32 %3, 3, %3")
KERNEL(add_self, "v_add_u32 %0, %0, %0\n v_add_u32 %1, %1, %1\n v_add_u32 %2, %2, %2\n v_add_u32 %3, %3, %3")

// a 16-byte LDS read per instruction at a computed (conflict-free) address, four independent chains (what the BPTC decoders' subset rows cost)
__global__ __launch_bounds__(256) void k_ds_read_b128(uint32_t *out, uint32_t seed, int iters) {
	__shared__ uint4 rows[4 * 256];
	for (int k = threadIdx.x; k < 4 * 256; k += 256) rows[k] = uint4{ seed + k, seed ^ k, seed * k, seed - k };
	__syncthreads();
	uint32_t a = threadIdx.x, acc = 0;
	for (int i = 0; i < iters; i++) {
#pragma unroll
		for (int r = 0; r < 64; r++) {
			const uint4 v = rows[((a + r) & 3u) * 256u + threadIdx.x];
			acc += v.x ^ v.w;
			a += v.y & 1u;
		}
	}
	out[blockIdx.x * 256 + threadIdx.x] = acc ^ a;
}

// LDS instructions with next to no VALU around them: sixteen reads (or writes) at immediate offsets from one per-lane base, one xor each
#define LDS_KERNEL(NAME, ASM, STRIDE)                                                                                   \
	__global__ __launch_bounds__(256) void k_##NAME(uint32_t *out, uint32_t seed, int iters) {                          \
		__shared__ uint4 rows[16 * 256];                                                                                 \
		for (int k = threadIdx.x; k < 16 * 256; k += 256) rows[k] = uint4{ seed + k, seed ^ k, seed * k, seed - k };      \
		__syncthreads();                                                                                                 \
		const uint32_t base = (uint32_t)(uintptr_t)rows + threadIdx.x * STRIDE;                                          \
		u4 acc = { seed, 1, 2, 3 };                                                                                      \
		for (int i = 0; i < iters; i++) {                                                                                \
			REP16(asm volatile(ASM : "+v"(acc) : "v"(base) : "memory");)                                                  \
		}                                                                                                                \
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                              \
		out[blockIdx.x * 256 + threadIdx.x] = acc.x ^ acc.y ^ acc.z ^ acc.w;                                             \
	}
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
LDS_KERNEL(lds_read_b128, "ds_read_b128 %0, %1 offset:4096\n s_waitcnt lgkmcnt(0)", 16)
LDS_KERNEL(lds_read_b128_x4, "ds_read_b128 %0, %1 offset:4096\n ds_read_b128 %0, %1 offset:8192\n ds_read_b128 %0, %1 offset:12288\n ds_read_b128 %0, %1 offset:16384\n s_waitcnt lgkmcnt(0)", 16)
LDS_KERNEL(lds_write_b128_x4, "ds_write_b128 %1, %0 offset:4096\n ds_write_b128 %1, %0 offset:8192\n ds_write_b128 %1, %0 offset:12288\n ds_write_b128 %1, %0 offset:16384\n s_waitcnt lgkmcnt(0)", 16)

struct Entry { const char *name; void (*fn)(uint32_t *, uint32_t, int); };
#define E(NAME) { #NAME, k_##NAME }
int main(int argc, char **argv) {
	Entry table[] = { E(v_add_u32), E(v_mul_u32_u24), E(v_mul_i32_i24), E(v_mad_i32_i24), E(v_mad_u32_u24), E(v_mul_lo_u32), E(v_mul_hi_u32),
		E(v_bfe_u32), E(v_bfi_b32), E(v_perm_b32), E(v_alignbit_b32), E(v_lshl_or_b32), E(v_and_or_b32), E(v_med3_i32), E(v_lshrrev_b32),
		E(v_pk_mad_u16), E(v_cndmask_b32), E(v_cmp_cnd), E(v_bfrev_b32), E(cnd_sgpr_mask), E(cnd_add_1to1), E(cnd_bfe_1to1), E(cnd_1_in_4), E(cmp_then_3cnd), E(cnd_e64_vcc), E(bfe_i32_mask), E(dep1_add), E(dep2_add), E(dep1_bfe), E(dep2_bfe), E(dep1_mix), E(dep2_mix),
		E(add_e64), E(and_literal), E(and_inline), E(and_sgpr), E(bfe_inline), E(bfe_sgpr), E(perm_sgpr_sel), E(v_mov), E(v_sat_pk), E(v_pk_add_u16),
		E(v_pk_ashr), E(v_lshl_add), E(v_add3), E(v_xor), E(v_sub), E(v_lshlrev), E(v_min_max), E(v_ffbl), E(mix_add_bfe), E(mix_valu_salu), E(mix_add_salu),
		E(lshl_vgpr), E(lshr_inline), E(ashr_inline), E(v_or), E(v_not), E(v_bitop3), E(v_bcnt), E(and_sdwa), E(add_sdwa), E(mov_sdwa), E(add_lit), E(max_u32),
		E(cmp_only), E(pk_mul_lo), E(pk_min_max),
		E(bitop3_sgpr), E(bitop3_inline), E(add_sgpr), E(xor_sgpr), E(lshr_sgpr_amt), E(lshr_sgpr_val), E(mov_sgpr), E(sub_sgpr), E(and_vcc_lo), E(dpp_mov), E(lshl_inline), E(add_self), E(ds_read_b128), E(lds_read_b128), E(lds_read_b128_x4), E(lds_write_b128_x4) };
	hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
	const int cus = prop.multiProcessorCount, blocks = cus * 8, iters = 512;	// 8 blocks x 4 waves = 32 waves/CU = 8 per SIMD
	const double clk = prop.clockRate * 1e3;
	uint32_t *out; hipMalloc(&out, blocks * 256 * 4);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	if (argc >= 4 && !strcmp(argv[1], "loop")) {
		// valu_rates loop NAME SECONDS: the one kernel back to back for that long (tools/gpu_power_classes.py samples clock and power meanwhile)
		for (const Entry &t : table) {
			if (strcmp(t.name, argv[2])) continue;
			const double seconds = atof(argv[3]);
			const auto t0 = std::chrono::steady_clock::now();
			long launches = 0;
			while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
				for (int k = 0; k < 8; k++) hipLaunchKernelGGL(t.fn, dim3(blocks), dim3(256), 0, 0, out, 12345u, 4096);
				hipDeviceSynchronize();
				launches += 8;
			}
			const double elapsed = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
			printf("loop %s launches %ld seconds %.3f wave_instructions_per_s %.4g\n", t.name, launches, elapsed, (double)launches * blocks * 4 * 4096.0 * 64 / elapsed);
			return 0;
		}
		printf("loop: no kernel named %s\n", argv[2]);
		return 1;
	}
	printf("%d CUs, %.0f MHz nominal; cycles per wave64 instruction per SIMD (8 waves/SIMD):\n", cus, clk / 1e6);
	for (const Entry &t : table) {
		hipLaunchKernelGGL(t.fn, dim3(blocks), dim3(256), 0, 0, out, 12345u, 8);
		hipEventRecord(e0);
		hipLaunchKernelGGL(t.fn, dim3(blocks), dim3(256), 0, 0, out, 12345u, iters);
		hipEventRecord(e1); hipEventSynchronize(e1);
		float ms; hipEventElapsedTime(&ms, e0, e1);
		const double insts_per_simd = 8.0 * iters * 16 * 4;	// waves/SIMD x iters x 16 asm x 4 instructions
		printf("  %-16s %7.2f cycles\n", t.name, ms * 1e-3 * clk / insts_per_simd);
	}
	return 0;
}
