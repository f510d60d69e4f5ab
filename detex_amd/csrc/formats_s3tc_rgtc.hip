// formats_s3tc_rgtc.hip -- kernels and launchers of BC1 / BC1A / BC2 / BC3 (decompress-bc.c) and RGTC1 / RGTC2 +- signed
// (decompress-rgtc.c): format indices 1-8 of the table the reference keeps in texture.c:27-48.
#include "decode_s3tc_rgtc.h"
#include "launchers.h"

namespace detexhip {

// (a function-local table: a namespace-scope const object would also be emitted into the device code object, where the launchers do not exist)
const FormatEntry *formats_s3tc_rgtc() {
	static const FormatEntry rows[8] = {
		FMT(BC1, DecBC1, kClassS3TC, 5, 5), FMT(BC1A, DecBC1A, kClassS3TC, 5, 5), FMT(BC2, DecBC2, kClassS3TCat8, 5, 5), FMT(BC3, DecBC3, kClassS3TCat8, 5, 5),
		FMT_L(RGTC1, DecRGTC1, kClassNone, 0, 0, 5), FMT(SIGNED_RGTC1, DecSignedRGTC1, kClassNone, 0, 0), FMT(RGTC2, DecRGTC2, kClassNone, 6, 0),
		FMT(SIGNED_RGTC2, DecSignedRGTC2, kClassNone, 5, 5),
	};
	return rows;
}

}  // namespace detexhip
