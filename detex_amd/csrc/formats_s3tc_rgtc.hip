// formats_s3tc_rgtc.hip -- kernels and launchers of BC1 / BC1A / BC2 / BC3 (decompress-bc.c) and RGTC1 / RGTC2 +- signed
// (decompress-rgtc.c): format indices 1-8 of the table the reference keeps in texture.c:27-48.
#include "decode_s3tc_rgtc.h"
#include "launchers.h"

namespace detexhip {

// (resident workgroups per CU of the linear kernels, round 6: the caps of rounds 3-5 were swept on ONE input decoded again and again, whose
// blocks are re-read from the memory-side Infinity Cache.  With the blocks coming out of HBM -- R different inputs in turn,
// tools/gpu_rotating.py, profiles/r06/rotating_wg_sweep* -- a cap of five costs the kernels with 8-byte blocks and the 2 x 16-bit formats
// 10-15 % (BC1 51.0 -> 45.9 us uncapped, BC1A 50.9 -> 45.9, SIGNED_RGTC2 62.7 -> 54.2; 16384^2 BC1 199 -> 177) for 0-2 % on the repeated
// input (BC1 8192^2 41.4 / 41.5, 16384^2 161 / 164): a read that takes three times as long needs the waves the cap removes.  Those run
// uncapped now; BC2 / BC3 (16-byte blocks) keep five: best on both sides (BC3 16384^2 195 against 203).  The block-major driver likewise
// (profiles/r06/rotating_wg_sweep_tiled.txt: BC1 53.9 -> 46.6 us with the blocks out of HBM and 42.0 -> 41.3 on the repeated input).)
// (a function-local table: a namespace-scope const object would also be emitted into the device code object, where the launchers do not exist)
const FormatEntry *formats_s3tc_rgtc() {
	static const FormatEntry rows[8] = {
		FMT_RA(BC1, DecBC1, kClassS3TC, 0, 0), FMT_RA(BC1A, DecBC1A, kClassS3TC, 0, 0), FMT(BC2, DecBC2, kClassS3TCat8, 5, 5), FMT(BC3, DecBC3, kClassS3TCat8, 5, 5),
		FMT_L(RGTC1, DecRGTC1, kClassNone, 0, 0, 5), FMT(SIGNED_RGTC1, DecSignedRGTC1, kClassNone, 0, 0), FMT(RGTC2, DecRGTC2, kClassNone, 6, 0),
		FMT(SIGNED_RGTC2, DecSignedRGTC2, kClassNone, 0, 0),
	};
	return rows;
}

}  // namespace detexhip
