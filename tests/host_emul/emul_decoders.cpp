// tests/host_emul/emul_decoders.cpp -- TEST-ONLY harness: compiles the DEVICE decoders of
// detex_amd/csrc with g++ (gfx950 builtins emulated by hip_host_shim.h) and exposes a batch
// entry so tests/test_host_emulation.py can compare the device decode logic with the oracle in
// this GPU-less container.  Not part of the product; libdetexhip.so never contains this.
#include <string.h>
#include <type_traits>
#include "hip_host_shim.h"		// FIRST: the emulated launch environment and gfx950_prims.h in plain C++
#include "dev_common.h"
#include "decode_s3tc_rgtc.h"
#include "decode_etc_eac.h"
#include "decode_bptc.h"
#include "decode_bptc_float.h"

using namespace detexhip;

template <int BYTES> struct Word;
template <> struct Word<8> { typedef uint2 type; };
template <> struct Word<16> { typedef uint4 type; };

// decoders that promise sixteen zero pixels for a failed block (Dec::kZeroOnFailure) are taken at their word here, so the
// emulation tests check the promise
template <class Dec, class = void> struct ZeroOnFailure { static constexpr bool value = false; };
template <class Dec> struct ZeroOnFailure<Dec, typename std::enable_if<Dec::kZeroOnFailure>::type> { static constexpr bool value = true; };

template <class Dec> static void run(const uint8_t *in, long n, uint32_t mode_mask, uint32_t flags, int checked, uint8_t *out, uint8_t *ok) {
	constexpr int P = Dec::kPixelBytes;
	// the workgroup's table copies, made by the decoder's own prepare() as all 256 threads of a workgroup would
	for (unsigned t = 0; t < 256u; t++) { threadIdx.x = t; prepare_tables<Dec>(); }
	threadIdx.x = 0;
	for (long i = 0; i < n; i++) {
		typename Word<Dec::kBlockBytes>::type blk;
		memcpy(&blk, in + i * Dec::kBlockBytes, Dec::kBlockBytes);
		uint32_t d[4 * P];
		memset(d, 0xA5, sizeof d);
		const bool r = checked ? Dec::template decode<true>(blk, mode_mask, flags, d) : Dec::template decode<false>(blk, mode_mask, flags, d);
		if (!r && !ZeroOnFailure<Dec>::value) memset(d, 0, sizeof d);
		memcpy(out + i * 16 * P, d, 16 * P);
		ok[i] = r;
	}
}

// what the other lanes of the emulated wave answer to a ballot (0 = nothing, ~0 = yes to everything: hip_host_shim.h)
extern "C" void emul_set_other_lanes_vote(unsigned long long v) { emul_other_lanes_vote = v; }

extern "C" int emul_decode_blocks(int fmt, const uint8_t *in, long n, uint32_t mode_mask, uint32_t flags, int checked, uint8_t *out, uint8_t *ok) {
	switch (fmt) {
	case 1: run<DecBC1>(in, n, mode_mask, flags, checked, out, ok); break;
	case 2: run<DecBC1A>(in, n, mode_mask, flags, checked, out, ok); break;
	case 3: run<DecBC2>(in, n, mode_mask, flags, checked, out, ok); break;
	case 4: run<DecBC3>(in, n, mode_mask, flags, checked, out, ok); break;
	case 5: run<DecRGTC1>(in, n, mode_mask, flags, checked, out, ok); break;
	case 6: run<DecSignedRGTC1>(in, n, mode_mask, flags, checked, out, ok); break;
	case 7: run<DecRGTC2>(in, n, mode_mask, flags, checked, out, ok); break;
	case 8: run<DecSignedRGTC2>(in, n, mode_mask, flags, checked, out, ok); break;
	case 9: run<DecBPTCFloat>(in, n, mode_mask, flags, checked, out, ok); break;
	case 109: run<DecBPTCFloatT<false, true>>(in, n, mode_mask, flags, checked, out, ok); break;
	case 110: run<DecBPTCFloatT<true, true>>(in, n, mode_mask, flags, checked, out, ok); break;
	case 10: run<DecBPTCSignedFloat>(in, n, mode_mask, flags, checked, out, ok); break;
	case 11: run<DecBPTC>(in, n, mode_mask, flags, checked, out, ok); break;
	case 111: run<DecBPTCPlain>(in, n, mode_mask, flags, checked, out, ok); break;
	case 12: run<DecETC1>(in, n, mode_mask, flags, checked, out, ok); break;
	case 13: run<DecETC2>(in, n, mode_mask, flags, checked, out, ok); break;
	case 14: run<DecETC2Punchthrough>(in, n, mode_mask, flags, checked, out, ok); break;
	case 15: run<DecETC2EAC>(in, n, mode_mask, flags, checked, out, ok); break;
	case 16: run<DecEACR11>(in, n, mode_mask, flags, checked, out, ok); break;
	case 17: run<DecEACSignedR11>(in, n, mode_mask, flags, checked, out, ok); break;
	case 18: run<DecEACRG11>(in, n, mode_mask, flags, checked, out, ok); break;
	case 19: run<DecEACSignedRG11>(in, n, mode_mask, flags, checked, out, ok); break;
	default: return 1;
	}
	return 0;
}
