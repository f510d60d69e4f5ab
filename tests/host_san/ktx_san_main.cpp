// tests/host_san/ktx_san_main.cpp -- TEST-ONLY: the KTX loader of libdetexhip (detex_amd/csrc/ktx_loader.cpp, the one piece of the
// library that parses file content) compiled with AddressSanitizer + UndefinedBehaviorSanitizer and fed a corpus of hostile
// files by tests/test_sanitized_host.py.  Usage: ktx_san FILE...   prints one line per file; any sanitizer report aborts with a
// non-zero exit code.  Not part of the product.
#include <cstdarg>
#include <initializer_list>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../detex_amd/csrc/ktx_loader.cpp"

static char g_message[512];
extern "C" void detexSetErrorMessage(const char *format, ...) {
	va_list args;
	va_start(args, format);
	vsnprintf(g_message, sizeof g_message, format, args);
	va_end(args);
}

int main(int argc, char **argv) {
	int loaded = 0, refused = 0;
	for (int i = 1; i < argc; i++) {
		for (int max_levels : { 1, 3, 16, 40, 0, -5 }) {
			detexTexture **textures = nullptr;
			int levels = 0;
			g_message[0] = 0;
			if (detexLoadKTXFileWithMipmaps(argv[i], max_levels, &textures, &levels)) {
				// touch every byte the loader claims to have produced, then give it back the way a caller would
				unsigned long sum = 0;
				for (int l = 0; l < levels; l++) {
					const detexTexture *t = textures[l];
					const size_t n = (size_t)t->width_in_blocks * (size_t)t->height_in_blocks * (size_t)detexGetCompressedBlockSize(t->format);
					for (size_t k = 0; k < n; k++) sum += t->data[k];
					if (t->width < 1 || t->height < 1 || t->width > 32768 || t->height > 32768) { printf("bad geometry accepted: %s\n", argv[i]); return 3; }
				}
				for (int l = 0; l < levels; l++) { free(textures[l]->data); free(textures[l]); }
				free(textures);
				loaded++;
				(void)sum;
			} else {
				if (!g_message[0]) { printf("refused without a message: %s\n", argv[i]); return 4; }
				refused++;
			}
		}
		detexTexture *one = nullptr;
		if (detexLoadKTXFile(argv[i], &one)) { free(one->data); free(one); }
	}
	printf("ktx_san: %d loads, %d refusals, no sanitizer report\n", loaded, refused);
	return 0;
}
