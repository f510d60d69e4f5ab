// formats_etc_eac.hip -- kernels and launchers of ETC1 / ETC2 / ETC2 punchthrough (decompress-etc.c) and ETC2_EAC, EAC R11 / RG11
// +- signed (decompress-eac.c): format indices 12-19 of the table the reference keeps in texture.c:27-48.
#include "decode_etc_eac.h"
#include "launchers.h"

namespace detexhip {

// (resident workgroups per CU, re-swept with the `sc1 nt` row stores: EAC_R11 / EAC_SIGNED_R11 run best uncapped, ETC2 at six; the
// block-major driver of the ETC family at seven.  Round 6, blocks coming out of HBM instead of the Infinity Cache -- formats_s3tc_rgtc.hip
// has the story: ETC1 52.3 -> 46.4 us uncapped (42.1 -> 43.4 on the repeated input), EAC_RG11 57.5 -> 51.4, EAC_SIGNED_RG11 58.4 -> 51.6:
// those three run uncapped now; ETC2 / punchthrough keep six, equal on both sides.)
// (a function-local table: a namespace-scope const object would also be emitted into the device code object, where the launchers do not exist)
const FormatEntry *formats_etc_eac() {
	static const FormatEntry rows[8] = {
		FMT(ETC1, DecETC1, kClassETC1, 0, 7), FMT(ETC2, DecETC2, kClassETC2, 6, 7), FMT(ETC2_PUNCHTHROUGH, DecETC2Punchthrough, kClassETC2PT, 6, 7),
		FMT(ETC2_EAC, DecETC2EAC, kClassETC2at8, 0, 7),
		FMT(EAC_R11, DecEACR11, kClassNone, 0, 0), FMT(EAC_SIGNED_R11, DecEACSignedR11, kClassNone, 0, 0), FMT(EAC_RG11, DecEACRG11, kClassNone, 0, 0),
		FMT(EAC_SIGNED_RG11, DecEACSignedRG11, kClassNone, 0, 0),
	};
	return rows;
}

}  // namespace detexhip
