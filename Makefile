# Top-level build: the product library (hipcc, gfx950) and the test-only oracle.
#   make            -> detex_amd/lib/libdetexhip.so  +  oracle/ checkers
#   make lib        -> only the product library
#   make ubench     -> tools/ubench/valu_rates (instruction-rate micro-benchmark) and tools/ubench/libhbmref.so (HBM fill / copy
#                      reference kernels of bench.py); measurement tools, not product
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
CSRC  := detex_amd/csrc
LIB   := detex_amd/lib/libdetexhip.so
LIB_AB := build/explib/libdetexhip_ab.so
HDRS  := $(wildcard $(CSRC)/*.h) $(wildcard $(CSRC)/ab/*.h) $(CSRC)/bptc_tables.inc include/detex.h include/detexhip.h

all: lib oracle ubench
lib: $(LIB)
ubench: tools/ubench/valu_rates hbmref
tools/ubench/valu_rates: tools/ubench/valu_rates.hip
	$(HIPCC) --offload-arch=$(ARCH) -O3 -o $@ $<
# HBM fill / copy reference kernels bench.py times beside the decode kernel (measurement tooling, not product)
hbmref: tools/ubench/libhbmref.so
tools/ubench/libhbmref.so: tools/ubench/hbm_ref.hip
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -shared -fvisibility=hidden -o $@ $<

$(LIB): $(CSRC)/detexhip.hip $(CSRC)/ktx_loader.cpp $(HDRS)
	@mkdir -p detex_amd/lib
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -shared -fvisibility=hidden \
		-Wall -Wno-unused-function -o $@ $(CSRC)/detexhip.hip $(CSRC)/ktx_loader.cpp

# measurement build with the rejected A/B kernels of DESIGN.md section 5 (DETEXHIP_LIB=$(LIB_AB) bench.py --variant N)
lib-ab: $(LIB_AB)
$(LIB_AB): $(CSRC)/detexhip.hip $(CSRC)/ktx_loader.cpp $(HDRS)
	@mkdir -p build/explib
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -shared -fvisibility=hidden -DDETEXHIP_AB_VARIANTS \
		-Wall -Wno-unused-function -o $@ $(CSRC)/detexhip.hip $(CSRC)/ktx_loader.cpp

oracle:
	$(MAKE) -C oracle all

clean:
	rm -f $(LIB) $(LIB_AB) tools/ubench/valu_rates tools/ubench/libhbmref.so
	$(MAKE) -C oracle clean
.PHONY: all lib lib-ab oracle ubench hbmref clean
