# tests/host_san/san.mk -- the instrumentation flags of the sanitizer builds (included by the root Makefile where it exists).
# Listed in .gpurunignore: the GPU pool runs no sanitizer builds, so neither this file nor the binaries made with it travel there;
# the instrumented programs run in the build container (tests/test_sanitized_host.py, no device needed).
SANFLAGS := -O1 -g -fsanitize=address,undefined -fno-gpu-sanitize -fno-sanitize-recover=all
KTX_SAN_FLAGS := -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -fno-omit-frame-pointer -Wall
# the KTX loader alone under g++'s sanitizers, fed a hostile corpus
ktx-san: tests/host_san/ktx_san
tests/host_san/ktx_san: tests/host_san/ktx_san_main.cpp detex_amd/csrc/ktx_loader.cpp include/detex.h
	g++ $(KTX_SAN_FLAGS) -o $@ $<
