// measurement build only (make lib-ab): the product's ETC / EAC table with the A/B linear launcher (ab_dispatch.h)
#include "decode_etc_eac.h"
#include "ab_dispatch.h"
#include "formats_etc_eac.hip"
