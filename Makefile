# Top-level build: the product library (hipcc, gfx950) and the test-only oracle.
#   make            -> detex_amd/lib/libdetexhip.so  +  oracle/ checkers
#   make lib        -> only the product library
#   make ubench     -> tools/ubench/valu_rates (instruction-rate micro-benchmark; measurement tool, not product)
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
CSRC  := detex_amd/csrc
LIB   := detex_amd/lib/libdetexhip.so
HDRS  := $(wildcard $(CSRC)/*.h) $(CSRC)/bptc_tables.inc include/detex.h include/detexhip.h

all: lib oracle ubench
lib: $(LIB)
ubench: tools/ubench/valu_rates
tools/ubench/valu_rates: tools/ubench/valu_rates.hip
	$(HIPCC) --offload-arch=$(ARCH) -O3 -o $@ $<

$(LIB): $(CSRC)/detexhip.hip $(CSRC)/ktx_loader.cpp $(HDRS)
	@mkdir -p detex_amd/lib
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -shared -fvisibility=hidden \
		-Wall -Wno-unused-function -o $@ $(CSRC)/detexhip.hip $(CSRC)/ktx_loader.cpp

oracle:
	$(MAKE) -C oracle all

clean:
	rm -f $(LIB) tools/ubench/valu_rates
	$(MAKE) -C oracle clean
.PHONY: all lib oracle ubench clean
