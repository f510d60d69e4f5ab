// ktx_loader.cpp -- detexLoadKTXFile / detexLoadKTXFileWithMipmaps for block-compressed KTX1 files
// (SURVEY.md section 8f-1).
//
// Lets the reference's own call sequence run end to end against libdetexhip.so:
//     detexLoadKTXFile(name, &tex);  detexDecompressTextureLinear(tex, pixels, BGRA8);   (validate.c:135,208)
// Scope: KTX 1.1 files of either endianness whose glInternalFormat is one of the 19 block formats
// of this library, key/value data skipped, mip levels with their 4-byte padding -- the behaviour
// of ktx.c:36-189 for those files, same ownership (texture structs and data are malloc'ed, the
// caller frees them) and the same error texts.  Uncompressed KTX payloads and every other
// container (DDS, raw, PNG) stay with the reference library: out of scope, not a decode path.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define DETEXHIP_BUILDING_LIBRARY 1
#include "../../include/detex.h"

namespace {

const unsigned char kKtxId[12] = { 0xAB, 0x4B, 0x54, 0x58, 0x20, 0x31, 0x31, 0xBB, 0x0D, 0x0A, 0x1A, 0x0A };

// glInternalFormat -> texture format (file-info.c:75-97, the compressed rows)
struct GlFormat { uint32_t gl; uint32_t texture_format; };
const GlFormat kGlFormats[] = {
	{ 0x83F0, DETEX_TEXTURE_FORMAT_BC1 }, { 0x83F1, DETEX_TEXTURE_FORMAT_BC1A }, { 0x83F2, DETEX_TEXTURE_FORMAT_BC2 },
	{ 0x83F3, DETEX_TEXTURE_FORMAT_BC3 }, { 0x8DBB, DETEX_TEXTURE_FORMAT_RGTC1 }, { 0x8DBC, DETEX_TEXTURE_FORMAT_SIGNED_RGTC1 },
	{ 0x8DBD, DETEX_TEXTURE_FORMAT_RGTC2 }, { 0x8DBE, DETEX_TEXTURE_FORMAT_SIGNED_RGTC2 }, { 0x8E8C, DETEX_TEXTURE_FORMAT_BPTC },
	{ 0x8E8F, DETEX_TEXTURE_FORMAT_BPTC_FLOAT }, { 0x8E8E, DETEX_TEXTURE_FORMAT_BPTC_SIGNED_FLOAT },
	{ 0x8D64, DETEX_TEXTURE_FORMAT_ETC1 }, { 0x9274, DETEX_TEXTURE_FORMAT_ETC2 }, { 0x9275, DETEX_TEXTURE_FORMAT_ETC2_PUNCHTHROUGH },
	{ 0x9278, DETEX_TEXTURE_FORMAT_ETC2_EAC }, { 0x9270, DETEX_TEXTURE_FORMAT_EAC_R11 }, { 0x9271, DETEX_TEXTURE_FORMAT_EAC_SIGNED_R11 },
	{ 0x9272, DETEX_TEXTURE_FORMAT_EAC_RG11 }, { 0x9273, DETEX_TEXTURE_FORMAT_EAC_SIGNED_RG11 },
};

uint32_t bswap(uint32_t v) { return (v >> 24) | ((v >> 8) & 0xFF00u) | ((v << 8) & 0xFF0000u) | (v << 24); }

void free_levels(detexTexture **textures, int n) {
	for (int j = 0; j < n; j++) {
		if (textures[j]) free(textures[j]->data);
		free(textures[j]);
	}
	free(textures);
}

}  // namespace

// ktx.c:36-176
extern "C" bool detexLoadKTXFileWithMipmaps(const char *filename, int max_mipmaps, detexTexture ***textures_out,
		int *nu_levels_out) {
	FILE *f = fopen(filename, "rb");
	if (!f) { detexSetErrorMessage("detexLoadKTXFileWithMipmaps: Could not open file %s", filename); return false; }
	uint32_t header[16];
	auto fail_read = [&]() { detexSetErrorMessage("detexLoadKTXFileWithMipmaps: Error reading file %s", filename); fclose(f); return false; };
	if (fread(header, 1, 64, f) != 64) return fail_read();
	if (memcmp(header, kKtxId, 12) != 0) {
		detexSetErrorMessage("detexLoadKTXFileWithMipmaps: Couldn't find KTX signature");
		fclose(f);
		return false;
	}
	const bool swapped = header[3] == 0x01020304u;
	if (swapped) for (int i = 3; i < 16; i++) header[i] = bswap(header[i]);
	const uint32_t gl_internal = header[7];
	uint32_t texture_format = 0;
	for (const GlFormat &g : kGlFormats) if (g.gl == gl_internal) texture_format = g.texture_format;
	if (!texture_format) {
		detexSetErrorMessage("detexLoadKTXFileWithMipmaps: Unsupported format in .ktx file (glInternalFormat = 0x%04X)"
			" -- libdetexhip loads block-compressed payloads only", gl_internal);
		fclose(f);
		return false;
	}
	const size_t block_bytes = detexGetCompressedBlockSize(texture_format);
	// The header is file content: dimensions are range-checked before any arithmetic (the decode entry points take at
	// most 32768 x 32768; the reference reads them unchecked, ktx.c:73-75) and every size is computed in 64 bits.
	const uint32_t w32 = header[9], h32 = header[10] == 0 ? 1u : header[10];
	if (w32 == 0 || w32 > 32768u || h32 > 32768u) {
		detexSetErrorMessage("detexLoadKTXFileWithMipmaps: Error loading file %s: texture size %ux%u is outside 1..32768", filename,
			(unsigned)w32, (unsigned)h32);
		fclose(f);
		return false;
	}
	int width = (int)w32, height = (int)h32;
	int levels = (int)header[14] < 1 || header[14] > 32u ? (header[14] > 32u ? 32 : 1) : (int)header[14];
	if (levels > max_mipmaps) levels = max_mipmaps;
	if (levels < 1) { fclose(f); detexSetErrorMessage("detexLoadKTXFileWithMipmaps: max_mipmaps must be at least 1"); return false; }
	if (header[15] > 0) {							// key/value data (:99-107): must really be there
		const long here = ftell(f);
		if (here < 0 || fseek(f, 0, SEEK_END) != 0) return fail_read();
		const long end = ftell(f);
		if (end < 0 || (uint64_t)header[15] > (uint64_t)(end - here) || fseek(f, here + (long)header[15], SEEK_SET) != 0) return fail_read();
	}
	detexTexture **textures = static_cast<detexTexture **>(calloc((size_t)levels, sizeof(detexTexture *)));
	if (!textures) return fail_read();
	for (int i = 0; i < levels; i++) {
		uint32_t image_size;
		if (fread(&image_size, 1, 4, f) != 4) { free_levels(textures, levels); return fail_read(); }
		if (swapped) image_size = bswap(image_size);
		const int wb = (width + 3) / 4, hb = (height + 3) / 4;
		const uint64_t need = (uint64_t)wb * (uint64_t)hb * (uint64_t)block_bytes;
		if ((uint64_t)image_size != need) {
			detexSetErrorMessage("detexLoadKTXFileWithMipmaps: Error loading file %s: Image size field of mipmap level %d "
				"does not match (%d vs %d)", filename, i, (int)image_size, (int)need);
			free_levels(textures, levels);
			fclose(f);
			return false;
		}
		detexTexture *t = static_cast<detexTexture *>(calloc(1, sizeof(detexTexture)));
		if (!t) { free_levels(textures, levels); return fail_read(); }
		textures[i] = t;
		t->format = texture_format;
		t->data = static_cast<uint8_t *>(malloc(need ? (size_t)need : 1));
		if (!t->data) { free_levels(textures, levels); return fail_read(); }
		t->width = width; t->height = height; t->width_in_blocks = wb; t->height_in_blocks = hb;
		if (fread(t->data, 1, (size_t)need, f) != (size_t)need) { free_levels(textures, levels); return fail_read(); }
		width >>= 1; height >>= 1;			// next level, rounding down (:160-163)
		if (i + 1 < levels) {				// mipPadding (:166-175)
			const long pad = 3 - (long)((image_size + 3) % 4);
			char padding[4];
			if (pad > 0 && fread(padding, 1, (size_t)pad, f) != (size_t)pad) { free_levels(textures, levels); return fail_read(); }
		}
	}
	fclose(f);
	*nu_levels_out = levels;
	*textures_out = textures;
	return true;
}

// ktx.c:180-189
extern "C" bool detexLoadKTXFile(const char *filename, detexTexture **texture_out) {
	int levels = 0;
	detexTexture **textures = nullptr;
	if (!detexLoadKTXFileWithMipmaps(filename, 1, &textures, &levels)) return false;
	*texture_out = textures[0];
	free(textures);
	return true;
}
