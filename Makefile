# Top-level build: the product library (HIP, gfx950 only) and the CPU checker.
#   make lib      -> radix_sorting_amd/librsx.so
#   make oracle   -> oracle/liboracle.so (+ oracle/_ref/* where /root/reference exists)
HIPCC   ?= /opt/rocm/bin/hipcc
ARCH    ?= gfx950
HIPFLAGS = --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Wall -Wno-unused-function
CSRC     = radix_sorting_amd/csrc

all: lib oracle

lib: radix_sorting_amd/librsx.so

radix_sorting_amd/librsx.so: $(CSRC)/rsx.hip $(CSRC)/rsx_kernels.hpp include/rsx.h
	$(HIPCC) $(HIPFLAGS) -shared $(CSRC)/rsx.hip -o $@

oracle:
	$(MAKE) -C oracle

clean:
	rm -f radix_sorting_amd/librsx.so
	$(MAKE) -C oracle clean

.PHONY: all lib oracle clean
