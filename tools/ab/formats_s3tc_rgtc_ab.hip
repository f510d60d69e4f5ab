// measurement build only (make lib-ab): the product's BC1-3 / RGTC table with the A/B linear launcher (ab_dispatch.h)
#include "decode_s3tc_rgtc.h"
#include "ab_dispatch.h"
#include "variant_tile4x4.h"		// variant 1: BC1 in 4x4-block wave tiles
#include "formats_s3tc_rgtc.hip"
