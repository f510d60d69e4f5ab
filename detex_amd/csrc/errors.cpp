// errors.cpp -- the reference's error convention (misc.c:73-94): a thread-local, malloc'ed message that every error replaces --
// and the data symbols the reference header's inline helpers refer to (detex.h:933,954,960,974), so that clients compiled against
// the reference's detex.h link.
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

static thread_local char *t_error_message = nullptr;

extern "C" __attribute__((visibility("default"))) void detexSetErrorMessage(const char *format, ...) {
	va_list args;
	va_start(args, format);
	char *message = nullptr;
	if (vasprintf(&message, format, args) < 0) message = strdup("detexSetErrorMessage: vasprintf returned error");
	va_end(args);
	free(t_error_message);
	t_error_message = message;
}

extern "C" __attribute__((visibility("default"))) const char *detexGetErrorMessage(void) { return t_error_message; }

// generated at compile time, value = clamp / truncating division (never used by this library)
namespace {
template <int N> struct ByteTable { uint8_t v[N]; };
template <int N, int D> constexpr ByteTable<N> make_division_table() {
	ByteTable<N> t{};
	for (int i = 0; i < N; i++) t.v[i] = (uint8_t)(i / D);
	return t;
}
constexpr ByteTable<767> make_clamp_table() {
	ByteTable<767> t{};
	for (int i = 0; i < 767; i++) t.v[i] = (uint8_t)(i < 255 ? 0 : (i > 510 ? 255 : i - 255));
	return t;
}
}  // namespace
extern "C" {
__attribute__((visibility("default"))) extern const ByteTable<767> detex_clamp0to255_table_storage __asm__("detex_clamp0to255_table");
__attribute__((visibility("default"))) extern const ByteTable<768> detex_division_by_3_table_storage __asm__("detex_division_by_3_table");
__attribute__((visibility("default"))) extern const ByteTable<1792> detex_division_by_7_table_storage __asm__("detex_division_by_7_table");
__attribute__((visibility("default"))) extern const ByteTable<1280> detex_division_by_5_table_storage __asm__("detex_division_by_5_table");
const ByteTable<767> detex_clamp0to255_table_storage = make_clamp_table();
const ByteTable<768> detex_division_by_3_table_storage = make_division_table<768, 3>();
const ByteTable<1792> detex_division_by_7_table_storage = make_division_table<1792, 7>();
const ByteTable<1280> detex_division_by_5_table_storage = make_division_table<1280, 5>();
}
