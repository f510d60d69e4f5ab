# Top-level build: the product library (hipcc, gfx950) and the test-only oracle.
#   make -j         -> detex_amd/lib/libdetexhip.so  +  oracle/ checkers
#   make lib        -> only the product library
#   make ubench     -> tools/ubench/valu_rates (instruction-rate micro-benchmark) and tools/ubench/libhbmref.so (HBM fill / copy
#                      reference kernels of bench.py); measurement tools, not product
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
CSRC  := detex_amd/csrc
LIB   := detex_amd/lib/libdetexhip.so
# (under tests/: build/ does not travel to the GPU box -- .gpurunignore -- and tests/test_ab_variants.py loads this build itself)
LIB_AB := tests/ab_build/libdetexhip_ab.so
HDRS  := $(wildcard $(CSRC)/*.h) $(CSRC)/bptc_tables.inc include/detex.h include/detexhip.h

all: lib oracle ubench c-client
lib: $(LIB)
ubench: tools/ubench/valu_rates tools/ubench/host_latency tools/ubench/host_midsize tools/ubench/big_footprint hbmref
# (links the product library: the large-footprint sweep goes through the device entry itself)
tools/ubench/big_footprint: tools/ubench/big_footprint.hip include/detexhip.h $(LIB)
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -o $@ $< -Ldetex_amd/lib -ldetexhip -Wl,-rpath,'$$ORIGIN/../../detex_amd/lib'
tools/ubench/host_midsize: tools/ubench/host_midsize.hip
	$(HIPCC) --offload-arch=$(ARCH) -O2 -std=c++17 -pthread -o $@ $<
tools/ubench/host_latency: tools/ubench/host_latency.hip
	$(HIPCC) --offload-arch=$(ARCH) -O2 -std=c++17 -o $@ $<
tools/ubench/valu_rates: tools/ubench/valu_rates.hip
	$(HIPCC) --offload-arch=$(ARCH) -O3 -o $@ $<
# HBM fill / copy reference kernels bench.py times beside the decode kernel (measurement tooling, not product)
hbmref: tools/ubench/libhbmref.so
tools/ubench/libhbmref.so: tools/ubench/hbm_ref.hip
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -shared -fvisibility=hidden -o $@ $<

# The library's translation units (detex_amd/csrc/host_internal.h lists what each one holds).  Only the .hip files contain device code:
# one per format family plus the histogram kernels, compiled in parallel (make -j); the .cpp files are host code built by the same driver.
SRCS_HIP := formats_s3tc_rgtc formats_etc_eac formats_bptc formats_bptc_float histogram
SRCS_CPP := errors device_tier host_tier host_resident multi_device ktx_loader
OBJDIR   := build/obj
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function -Wno-pass-failed $(EXTRA_HIPFLAGS)
OBJS     := $(addprefix $(OBJDIR)/,$(addsuffix .o,$(SRCS_HIP) $(SRCS_CPP)))

$(OBJDIR)/%.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c -o $@ $<
$(OBJDIR)/%.o: $(CSRC)/%.cpp $(HDRS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c -o $@ $<
# (measurement build only: the translation units of tools/ab include the product's headers and format tables, never the other way round)
$(OBJDIR)/%_ab.o: tools/ab/%_ab.hip $(HDRS) $(wildcard tools/ab/*.h) $(wildcard $(CSRC)/formats_*.hip)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -I$(CSRC) -c -o $@ $<
$(LIB): $(OBJS)
	@mkdir -p $(dir $(LIB))
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)

# measurement build with the rejected A/B kernels (profiles/AB_RECORD.md; DETEXHIP_LIB=$(LIB_AB) bench.py --variant N): the product's
# host objects and histogram kernels + tools/ab's format-table translation units in place of the product's
SRCS_HIP_AB := formats_s3tc_rgtc_ab formats_etc_eac_ab formats_bptc_ab formats_bptc_float_ab histogram
lib-ab:
	$(MAKE) lib LIB=$(LIB_AB) OBJDIR=build/obj_ab SRCS_HIP="$(SRCS_HIP_AB)"

# the library's host code under AddressSanitizer + UndefinedBehaviorSanitizer with mains that call every entry point with hostile arguments /
# end threads and the process with resident kernels alive (tests/test_sanitized_host.py).  CONTAINER ONLY: the GPU pool runs no sanitizer
# builds, so the instrumentation flags live in tests/host_san/san.mk, which .gpurunignore keeps off the GPU box together with the
# instrumented binaries; host code only is instrumented.  The GPU box runs the uninstrumented builds of the same programs (host-plain).
-include tests/host_san/san.mk
need-san:
	@test -n "$(SANFLAGS)" || { echo "tests/host_san/san.mk is absent (GPU box): instrumented builds are made in the container only"; exit 1; }
api-san: need-san
	$(MAKE) tests/host_san/api_san OBJDIR=build/obj_san EXTRA_HIPFLAGS="$(SANFLAGS)"
$(OBJDIR)/api_san_main.o: tests/host_san/api_san_main.cpp include/detex.h include/detexhip.h
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c -o $@ $<
tests/host_san/api_san: $(OBJDIR)/api_san_main.o $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) $(EXTRA_HIPFLAGS) -o $@ $^
# thread / process / dlclose teardown of the host tier's per-thread state (tests/host_san/teardown_san_main.cpp): the test main linked with
# the instrumented objects, and the same objects as a shared library for its dlclose mode
teardown-san: need-san
	$(MAKE) tests/host_san/teardown_san tests/host_san/libdetexhip_san.so OBJDIR=build/obj_san EXTRA_HIPFLAGS="$(SANFLAGS)"
$(OBJDIR)/teardown_san_main.o: tests/host_san/teardown_san_main.cpp include/detex.h include/detexhip.h
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -DTEARDOWN_SAN_LINKED -c -o $@ $<
tests/host_san/teardown_san: $(OBJDIR)/teardown_san_main.o $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) $(EXTRA_HIPFLAGS) -pthread -o $@ $^ -ldl
tests/host_san/libdetexhip_san.so: $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) $(EXTRA_HIPFLAGS) -shared -fPIC -o $@ $^
# the same two programs WITHOUT instrumentation, linked against the product library like any client: what the GPU box runs
# (tests/test_gpu_host_programs.py); its dlclose mode opens detex_amd/lib/libdetexhip.so itself
PLAIN := tests/host_san/api_plain tests/host_san/teardown_plain
host-plain: $(PLAIN)
tests/host_san/api_plain: tests/host_san/api_san_main.cpp include/detex.h include/detexhip.h $(LIB)
	g++ -std=c++17 -O1 -g -Wall -o $@ $< -Ldetex_amd/lib -ldetexhip -Wl,-rpath,'$$ORIGIN/../../detex_amd/lib'
tests/host_san/teardown_plain: tests/host_san/teardown_san_main.cpp include/detex.h include/detexhip.h $(LIB)
	g++ -std=c++17 -O1 -g -Wall -DTEARDOWN_SAN_LINKED -pthread -o $@ $< -Ldetex_amd/lib -ldetexhip -Wl,-rpath,'$$ORIGIN/../../detex_amd/lib' -ldl

# a plain C client of the drop-in boundary (tests/c_client/detex_client.c; tests/test_c_client.py runs it on the GPU box): gcc, this
# repository's detex.h, -ldetexhip with an rpath to the library; where the reference's sources are present (build container) a second
# binary from the SAME source compiled against the REFERENCE's own detex.h -- the client a libdetex user would already have
REFHDR ?= /root/reference
CLIENT := tests/c_client/detex_client
c-client: $(CLIENT) $(if $(wildcard $(REFHDR)/detex.h),$(CLIENT)_refhdr) $(if $(wildcard oracle/_ref/libdetex_ref.so),$(CLIENT)_reflib)
# the same program over the COMPILED REFERENCE (oracle/_ref, where it has been built): bench.py's like-for-like CPU figure for the small calls
$(CLIENT)_reflib: $(CLIENT).c include/detex.h oracle/_ref/libdetex_ref.so
	gcc -std=c99 -O2 -Wall -Wextra -Iinclude -o $@ $< -Loracle/_ref -ldetex_ref -Wl,-rpath,'$$ORIGIN/../../oracle/_ref'
$(CLIENT): $(CLIENT).c include/detex.h include/detexhip.h $(LIB)
	gcc -std=c99 -O2 -Wall -Wextra -DWITH_DETEXHIP -Iinclude -o $@ $< -Ldetex_amd/lib -ldetexhip -Wl,-rpath,'$$ORIGIN/../../detex_amd/lib'
$(CLIENT)_refhdr: $(CLIENT).c $(REFHDR)/detex.h include/detexhip.h $(LIB)
	gcc -std=c99 -D_POSIX_C_SOURCE=200809L -O2 -Wall -DWITH_DETEXHIP -I$(REFHDR) -Iinclude -o $@ $< -Ldetex_amd/lib -ldetexhip -Wl,-rpath,'$$ORIGIN/../../detex_amd/lib'

oracle:
	$(MAKE) -C oracle all

clean:
	rm -rf $(LIB) $(LIB_AB) build/obj build/obj_ab build/obj_san tests/host_san/api_san tests/host_san/teardown_san tests/host_san/libdetexhip_san.so tests/host_san/ktx_san $(PLAIN) $(CLIENT) $(CLIENT)_refhdr $(CLIENT)_reflib tools/ubench/valu_rates tools/ubench/big_footprint tools/ubench/libhbmref.so
	$(MAKE) -C oracle clean
.PHONY: all lib lib-ab need-san api-san teardown-san ktx-san host-plain c-client oracle ubench hbmref clean
