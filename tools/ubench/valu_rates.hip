// tools/ubench/valu_rates.hip -- measures the issue cost of the integer VALU ops the decoders are
// built from (cycles per wave64 instruction per SIMD, 8 waves/SIMD resident), to price kernels
// against the 4-cycle baseline seen in profiles/r01.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rates tools/ubench/valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
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
KERNEL(v_lshrrev_b64, "v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %5\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %5")

struct Entry { const char *name; void (*fn)(uint32_t *, uint32_t, int); };
#define E(NAME) { #NAME, k_##NAME }
int main() {
	Entry table[] = { E(v_add_u32), E(v_mul_u32_u24), E(v_mul_i32_i24), E(v_mad_i32_i24), E(v_mad_u32_u24), E(v_mul_lo_u32), E(v_mul_hi_u32),
		E(v_bfe_u32), E(v_bfi_b32), E(v_perm_b32), E(v_alignbit_b32), E(v_lshl_or_b32), E(v_and_or_b32), E(v_med3_i32), E(v_lshrrev_b32),
		E(v_pk_mad_u16), E(v_cndmask_b32), E(v_cmp_cnd), E(v_bfrev_b32), E(cnd_sgpr_mask), E(cnd_add_1to1), E(cnd_bfe_1to1), E(cnd_1_in_4), E(cmp_then_3cnd), E(cnd_e64_vcc), E(bfe_i32_mask), E(dep1_add), E(dep2_add), E(dep1_bfe), E(dep2_bfe), E(dep1_mix), E(dep2_mix) };
	hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
	const int cus = prop.multiProcessorCount, blocks = cus * 8, iters = 512;	// 8 blocks x 4 waves = 32 waves/CU = 8 per SIMD
	const double clk = prop.clockRate * 1e3;
	uint32_t *out; hipMalloc(&out, blocks * 256 * 4);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
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
