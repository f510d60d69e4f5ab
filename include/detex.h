/*
 * include/detex.h -- compatibility header of libdetexhip for the block-decode path of
 * hglm/detex v0.1.2.
 *
 * Freshly written mirror of the INTERFACE FACTS of the reference's detex.h (names, enum
 * values, struct layout, signatures) for the entry points this library implements on the GPU.
 * A client compiled against the reference's own detex.h links against libdetexhip.so
 * unchanged (tests/test_abi.py compiles one against /root/reference/detex.h when present);
 * this header exists so that the repo is self-contained on machines without the reference.
 *
 * Each declaration cites the reference interface it replaces (file:line in /root/reference).
 * Everything of the reference API that is NOT listed here (DDS / raw / PNG I/O, KTX saving,
 * detexConvertPixels as a stand-alone call, HDR, encoder helpers) is out of scope (SURVEY.md
 * section 8) and is not exported.  Target pixel formats accepted by the decode entry points:
 * the native one, RGBX8 <-> RGBA8, and -- converted inside the kernel with detexConvertPixels'
 * semantics (convert.c:37-70, 671-684) -- BGRA8, BGRX8, RGB8 and FLOAT_BGRX16 (constants below).
 */
#ifndef DETEXHIP_COMPAT_DETEX_H
#define DETEXHIP_COMPAT_DETEX_H

#include <stdbool.h>
#include <stdint.h>
#include <stdlib.h>

#ifdef __cplusplus
extern "C" {
#endif

/* exported with default visibility from libdetexhip.so (built with -fvisibility=hidden) */
#define DETEX_API __attribute__((visibility("default")))

#define DETEX_MAX_BLOCK_SIZE 256	/* detex.h:78 -- bytes of the largest decoded 4x4 block */

/* ---- pixel formats (detex.h:83-379); bit layout: 0x0F00 = bytes per pixel - 1 ---------- */
enum {
	DETEX_PIXEL_FORMAT_R8 = 0x0000,
	DETEX_PIXEL_FORMAT_RG8 = 0x0110,
	DETEX_PIXEL_FORMAT_R16 = 0x0101,
	DETEX_PIXEL_FORMAT_SIGNED_R16 = 0x1101,
	DETEX_PIXEL_FORMAT_RG16 = 0x0311,
	DETEX_PIXEL_FORMAT_SIGNED_RG16 = 0x1311,
	DETEX_PIXEL_FORMAT_RGBX8 = 0x0320,
	DETEX_PIXEL_FORMAT_RGBA8 = 0x0334,
	DETEX_PIXEL_FORMAT_RGB8 = 0x0220,		/* in-kernel epilogue targets */
	DETEX_PIXEL_FORMAT_BGRX8 = 0x0328,
	DETEX_PIXEL_FORMAT_BGRA8 = 0x033C,
	DETEX_PIXEL_FORMAT_FLOAT_BGRX16 = 0x2729,
	DETEX_PIXEL_FORMAT_FLOAT_RGBX16 = 0x2721,
	DETEX_PIXEL_FORMAT_SIGNED_FLOAT_RGBX16 = 0x3721,
};

/* ---- compressed texture formats (detex.h:577-727) --------------------------------------
 * texture_format = (compressed format index << 24) | 128-bit-block flag | native pixel format */
enum {
	DETEX_TEXTURE_FORMAT_PIXEL_FORMAT_MASK = 0x0000FFFF,
	DETEX_TEXTURE_FORMAT_128BIT_BLOCK_BIT = 0x00800000,
	DETEX_TEXTURE_FORMAT_BC1 = 0x01000320,
	DETEX_TEXTURE_FORMAT_BC1A = 0x02000334,
	DETEX_TEXTURE_FORMAT_BC2 = 0x03800334,
	DETEX_TEXTURE_FORMAT_BC3 = 0x04800334,
	DETEX_TEXTURE_FORMAT_RGTC1 = 0x05000000,
	DETEX_TEXTURE_FORMAT_SIGNED_RGTC1 = 0x06001101,
	DETEX_TEXTURE_FORMAT_RGTC2 = 0x07800110,
	DETEX_TEXTURE_FORMAT_SIGNED_RGTC2 = 0x08801311,
	DETEX_TEXTURE_FORMAT_BPTC_FLOAT = 0x09802721,
	DETEX_TEXTURE_FORMAT_BPTC_SIGNED_FLOAT = 0x0A803721,
	DETEX_TEXTURE_FORMAT_BPTC = 0x0B800334,
	DETEX_TEXTURE_FORMAT_ETC1 = 0x0C000320,
	DETEX_TEXTURE_FORMAT_ETC2 = 0x0D000320,
	DETEX_TEXTURE_FORMAT_ETC2_PUNCHTHROUGH = 0x0E000334,
	DETEX_TEXTURE_FORMAT_ETC2_EAC = 0x0F800334,
	DETEX_TEXTURE_FORMAT_EAC_R11 = 0x10000101,
	DETEX_TEXTURE_FORMAT_EAC_SIGNED_R11 = 0x11001101,
	DETEX_TEXTURE_FORMAT_EAC_RG11 = 0x12800311,
	DETEX_TEXTURE_FORMAT_EAC_SIGNED_RG11 = 0x13801311,
};

/* ---- mode masks (detex.h:383-395) and decompression flags (detex.h:397-411) ------------ */
enum {
	DETEX_MODE_MASK_ETC_INDIVIDUAL = 0x1,
	DETEX_MODE_MASK_ETC_DIFFERENTIAL = 0x2,
	DETEX_MODE_MASK_ETC_T = 0x4,
	DETEX_MODE_MASK_ETC_H = 0x8,
	DETEX_MODE_MASK_ETC_PLANAR = 0x10,
	DETEX_MODE_MASK_ALL_MODES_ETC1 = 0x3,
	DETEX_MODE_MASK_ALL_MODES_ETC2 = 0x1F,
	DETEX_MODE_MASK_ALL_MODES_ETC2_PUNCHTHROUGH = 0x1E,
	DETEX_MODE_MASK_ALL_MODES_BPTC = 0xFF,
	DETEX_MODE_MASK_ALL_MODES_BPTC_FLOAT = 0x3FFF,
	DETEX_MODE_MASK_ALL = 0xFFFFFFFF,
};
enum {
	DETEX_DECOMPRESS_FLAG_ENCODE = 0x1,
	DETEX_DECOMPRESS_FLAG_OPAQUE_ONLY = 0x2,
	DETEX_DECOMPRESS_FLAG_NON_OPAQUE_ONLY = 0x4,
};

/* ---- detexTexture (detex.h:729-736): 32 bytes on LP64, data at offset 8 ---------------- */
typedef struct {
	uint32_t format;
	uint8_t *data;
	int width;
	int height;
	int width_in_blocks;
	int height_in_blocks;
} detexTexture;

/* ---- size helpers (detex.h:879-881, 913-930) ------------------------------------------- */
static inline int detexGetPixelSize(uint32_t pixel_format) { return 1 + (int)((pixel_format & 0xF00) >> 8); }
static inline uint32_t detexGetCompressedFormat(uint32_t texture_format) { return texture_format >> 24; }
static inline uint32_t detexGetCompressedBlockSize(uint32_t texture_format) {
	return 8 + ((texture_format & DETEX_TEXTURE_FORMAT_128BIT_BLOCK_BIT) >> 20);
}
static inline uint32_t detexFormatIsCompressed(uint32_t texture_format) { return (texture_format >> 24) != 0; }
/* (detex.h:89, 906-909: bit 2 of a pixel or texture format = it has an alpha component; validate.c:202 picks BGRA8 or BGRX8 by it) */
static inline uint32_t detexFormatHasAlpha(uint32_t format) { return (format & 0x4u) != 0; }
static inline uint32_t detexGetPixelFormat(uint32_t texture_format) {
	return texture_format & DETEX_TEXTURE_FORMAT_PIXEL_FORMAT_MASK;
}

/* ---- the 19 per-block decoders (detex.h:435-531) ---------------------------------------
 * bool f(const uint8_t *bitstring, uint32_t mode_mask, uint32_t flags, uint8_t *pixel_buffer)
 * One 8/16-byte block -> 16 pixels of the format's native pixel size, row-major.  Returns
 * false for a mode excluded by mode_mask, an opaque/non-opaque/encode filter in flags, or an
 * invalid block; pixel_buffer is then left untouched.
 * libdetexhip runs these on the GPU as well: the block travels as a kernel argument, the pixels come back through a pinned
 * exchange buffer whose completion word the caller polls (7 us), and from the second call in a row of one format on the
 * request goes to a kernel that is already resident (5-6 us).  The reference needs 0.03-0.1 us: a client that decodes many
 * blocks calls detexhipDecompressBlocks (detexhip.h) ONCE instead of looping here. */
#define DETEXHIP_DECLARE_BLOCK_FN(NAME) \
	DETEX_API bool detexDecompressBlock##NAME(const uint8_t *bitstring, uint32_t mode_mask, uint32_t flags, \
		uint8_t *pixel_buffer);
DETEXHIP_DECLARE_BLOCK_FN(BC1)			/* decompress-bc.c:23 */
DETEXHIP_DECLARE_BLOCK_FN(BC1A)			/* decompress-bc.c:87 */
DETEXHIP_DECLARE_BLOCK_FN(BC2)			/* decompress-bc.c:136 */
DETEXHIP_DECLARE_BLOCK_FN(BC3)			/* decompress-bc.c:175 */
DETEXHIP_DECLARE_BLOCK_FN(RGTC1)		/* decompress-rgtc.c:64 */
DETEXHIP_DECLARE_BLOCK_FN(SIGNED_RGTC1)		/* decompress-rgtc.c:134 */
DETEXHIP_DECLARE_BLOCK_FN(RGTC2)		/* decompress-rgtc.c:72 */
DETEXHIP_DECLARE_BLOCK_FN(SIGNED_RGTC2)		/* decompress-rgtc.c:141 */
DETEXHIP_DECLARE_BLOCK_FN(BPTC_FLOAT)		/* decompress-bptc-float.c:631 */
DETEXHIP_DECLARE_BLOCK_FN(BPTC_SIGNED_FLOAT)	/* decompress-bptc-float.c:640 */
DETEXHIP_DECLARE_BLOCK_FN(BPTC)			/* decompress-bptc.c:354 */
DETEXHIP_DECLARE_BLOCK_FN(ETC1)			/* decompress-etc.c:89 */
DETEXHIP_DECLARE_BLOCK_FN(ETC2)			/* decompress-etc.c:321 */
DETEXHIP_DECLARE_BLOCK_FN(ETC2_PUNCHTHROUGH)	/* decompress-etc.c:653 */
DETEXHIP_DECLARE_BLOCK_FN(ETC2_EAC)		/* decompress-eac.c:54 */
DETEXHIP_DECLARE_BLOCK_FN(EAC_R11)		/* decompress-eac.c:132 */
DETEXHIP_DECLARE_BLOCK_FN(EAC_SIGNED_R11)	/* decompress-eac.c:206 */
DETEXHIP_DECLARE_BLOCK_FN(EAC_RG11)		/* decompress-eac.c:144 */
DETEXHIP_DECLARE_BLOCK_FN(EAC_SIGNED_RG11)	/* decompress-eac.c:217 */
#undef DETEXHIP_DECLARE_BLOCK_FN

/* ---- generic block + whole-texture drivers (detex.h:747-765, texture.c:55-145) ---------
 * pixel_format: the format's native pixel format; for RGBX8 / RGBA8 natives either of the two (the
 * reference's no-op conversion edge, convert.c:768-769); and the targets the reference's callers ask for,
 * converted inside the decode kernel with the exact result of detexConvertPixels' path (convert.c:885-1063) --
 * BGRA8 / BGRX8 (validate.c:204-209, detex-view.c:182), RGB8 (detex-convert.c:283-284), RGBA8 / RGBX8 for the
 * one- and two-component and half-float formats, FLOAT_BGRX16 for BPTC_FLOAT -- for every format the reference
 * itself can convert to them (all but BPTC_SIGNED_FLOAT; the signed 16-bit formats have no path to BGRA8).  (FLOAT_RGB16 for BPTC_FLOAT
 * is not among them: the reference's converter for that edge, convert.c:704-718, writes through an uninitialised pointer.)
 * Any other target is outside the block-decode path: the call returns false with an error message and the
 * texture drivers zero-fill the output.
 * Threads: re-entrant, per-thread device state (detexhip.h: detexhipReleaseThreadResources).  A texture driver call with at least 32 MiB of
 * compressed blocks runs ONE helper thread of the library's own for its duration (it uploads the blocks band by band while the calling
 * thread decodes and downloads: the link is full duplex; DETEXHIP_HOST_DUPLEX=0 in the environment turns that off); nothing of it outlives the call. */
DETEX_API bool detexDecompressBlock(const uint8_t *bitstring, uint32_t texture_format, uint32_t mode_mask,
	uint32_t flags, uint8_t *pixel_buffer, uint32_t pixel_format);	/* texture.c:55 */
DETEX_API bool detexDecompressTextureTiled(const detexTexture *texture, uint8_t *pixel_buffer,
	uint32_t pixel_format);						/* texture.c:77 */
DETEX_API bool detexDecompressTextureLinear(const detexTexture *texture, uint8_t *pixel_buffer,
	uint32_t pixel_format);						/* texture.c:105 */

/* ---- KTX1 loader for block-compressed payloads (detex.h:826-836, ktx.c:36-189) -----------
 * enough of the reference's loader that its own call sequence
 * (detexLoadKTXFile -> detexDecompressTextureLinear, validate.c:135,208) runs against this
 * library.  Textures and their data are malloc'ed; the caller frees them (as in the reference).
 * Uncompressed payloads and the other containers (DDS/raw/PNG) are not provided. */
DETEX_API bool detexLoadKTXFileWithMipmaps(const char *filename, int max_mipmaps, detexTexture ***textures_out,
	int *nu_levels_out);							/* ktx.c:36 */
DETEX_API bool detexLoadKTXFile(const char *filename, detexTexture **texture_out);	/* ktx.c:180 */

/* ---- error convention (detex.h:806, misc.c:73-94): thread-local last-error string,
 * NULL until the first error on the calling thread, overwritten by each later error. */
DETEX_API const char *detexGetErrorMessage(void);
DETEX_API void detexSetErrorMessage(const char *format, ...);			/* misc.h:20 */

/* ---- data symbols the reference's detex.h inlines refer to (detex.h:933,954,960,974) ----
 * Exported so that clients built against the reference header still link.  Values: plain
 * clamp / truncating integer division. */
#ifndef DETEXHIP_BUILDING_LIBRARY
DETEX_API extern const uint8_t detex_clamp0to255_table[767];
DETEX_API extern const uint8_t detex_division_by_3_table[768];
DETEX_API extern const uint8_t detex_division_by_7_table[1792];
DETEX_API extern const uint8_t detex_division_by_5_table[1280];
#endif

#ifdef __cplusplus
}
#endif
#endif
