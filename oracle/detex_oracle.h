/*
 * oracle/detex_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the hglm/detex block-decode hot path, used only as the
 * checker: tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * Nothing under detex_amd/ or include/ may include, link or call anything under oracle/.
 *
 * Parity status: PINNED.  tests/test_oracle_pin.py holds this restatement bit-identical to
 * the compiled reference itself (oracle/_ref/libdetex_ref.so, built by oracle/Makefile from
 * /root/reference with the reference's own flags) on the 17 bundled test-texture-*.ktx
 * fixtures, mode-forced vectors for every mode / invalid-block class of every format, clipped
 * texture sizes and millions of random blocks per format; the committed goldens under
 * tests/golden/ were produced by the compiled reference (tools/make_goldens.py).
 *
 * Format indices are the reference's compressed-format indices (texture.c:27-48,
 * detex.h "compressed texture format index" = texture_format >> 24).
 */
#ifndef DETEX_ORACLE_H
#define DETEX_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
	ORC_BC1 = 1, ORC_BC1A, ORC_BC2, ORC_BC3, ORC_RGTC1, ORC_SIGNED_RGTC1, ORC_RGTC2,
	ORC_SIGNED_RGTC2, ORC_BPTC_FLOAT, ORC_BPTC_SIGNED_FLOAT, ORC_BPTC, ORC_ETC1, ORC_ETC2,
	ORC_ETC2_PUNCHTHROUGH, ORC_ETC2_EAC, ORC_EAC_R11, ORC_EAC_SIGNED_R11, ORC_EAC_RG11,
	ORC_EAC_SIGNED_RG11, ORC_FORMAT_COUNT
};

/* compressed block size (8/16) and native pixel size in bytes for a format index; 0 if invalid */
/* quirk switch: bit 0 = A-2 (BC7 mode 6 second P-bit reads 0), bit 1 = A-3 (BC6H mode 12 drops block bit 63); default both */
#define ORC_QUIRK_BC7_MODE6_PBIT 1u
#define ORC_QUIRK_BC6H_MODE12_BIT63 2u
void orc_set_quirks(unsigned mask);
int orc_block_bytes(int fmt);
int orc_pixel_bytes(int fmt);

/* One 4x4 block -> 16 native pixels, row-major (detex.h:435-531 contract). Returns 1/0. */
int orc_decode_block(int fmt, const uint8_t *bitstring, uint32_t mode_mask, uint32_t flags,
	uint8_t *pixel_buffer);

/* Whole texture, native pixel format, row-major with pitch width*px, clipped to width/height,
 * invalid blocks zero-filled, returns 0 if any block failed (texture.c:105-145). */
int orc_decompress_linear(int fmt, const uint8_t *data, int width, int height,
	int width_in_blocks, int height_in_blocks, uint8_t *pixel_buffer);

/* Whole texture, block-major output, no clipping (texture.c:77-98). */
int orc_decompress_tiled(int fmt, const uint8_t *data, int width_in_blocks,
	int height_in_blocks, uint8_t *pixel_buffer);

/* Mode classifier used to build mode-forced streams (mirrors detexGetMode*:
 * decompress-bc.c:63-69, decompress-etc.c:183-190,370-395,721-742, decompress-bptc.c:603-610,
 * decompress-bptc-float.c:647-658). Returns -1 for reserved BPTC/BPTC_FLOAT codes. */
int orc_block_mode(int fmt, const uint8_t *bitstring);

/* batch forms for the tests: modes of n blocks; per-block decode with per-block ok flags
 * (failed blocks are left as the decoder left them, exactly like the per-block API). */
uint64_t orc_fnv1a64(const uint8_t *p, size_t n);
/* pixel conversions offered as GPU epilogues: kind 1 = swap R/B of 32-bit pixels, 2 = RGBX8 -> RGB8,
 * 3 = swap R/B of 64-bit (4 x 16) pixels; returns bytes written (convert.c:37-70, 671-684) */
long orc_convert_pixels(uint32_t native_pixel_format, int kind, const uint8_t *in, long n_pixels, uint8_t *out);
void orc_block_modes(int fmt, const uint8_t *data, long n_blocks, int32_t *modes_out);
void orc_decode_blocks(int fmt, const uint8_t *data, long n_blocks, uint32_t mode_mask, uint32_t flags,
	uint8_t *pixel_buffer, uint8_t *ok_out);

#ifdef __cplusplus
}
#endif
#endif
