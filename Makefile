# Top-level build: the product library (HIP, gfx950 only) and the CPU checker.
#   make lib      -> radix_sorting_amd/librsx.so
#   make oracle   -> oracle/liboracle.so (+ oracle/_ref/* where /root/reference exists)
HIPCC   ?= /opt/rocm/bin/hipcc
ARCH    ?= gfx950
HIPFLAGS = --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Wall -Wno-unused-function
CSRC     = radix_sorting_amd/csrc

all: lib oracle cpp cli

lib: radix_sorting_amd/librsx.so

radix_sorting_amd/librsx.so: $(CSRC)/rsx.hip $(CSRC)/rsx_kernels.hpp $(CSRC)/rsx_hist.hpp $(CSRC)/rsx_scatter2.hpp $(CSRC)/rsx_small.hpp $(CSRC)/rsx_hybrid.hpp $(CSRC)/rsx_leaf16.hpp $(CSRC)/rsx_pass16.hpp $(CSRC)/rsx_pass32.hpp $(CSRC)/rsx_multi_state.hpp $(CSRC)/rsx_multi_entry.hpp $(CSRC)/rsx_records.hpp $(CSRC)/rsx_logroute.hpp $(CSRC)/rsx_pass2w.hpp $(CSRC)/rsx_pass64.hpp $(CSRC)/rsx_leafc.hpp include/rsx.h
	$(HIPCC) $(HIPFLAGS) -shared $(CSRC)/rsx.hip -o $@

oracle: lib
	$(MAKE) -C oracle

# C++ check of the template surface in include/ (needs a GPU to run: tests/test_gpu_cpp.py)
cpp: tests/cpp/dropin_check

tests/cpp/dropin_check: tests/cpp/dropin_check.cpp include/radix_sort.hpp include/radix_sort_rank.hpp include/radix_sort_basic_kdf.hpp include/rsx.h radix_sorting_amd/librsx.so
	g++ -std=gnu++17 -O2 -Wall -Iinclude tests/cpp/dropin_check.cpp -Lradix_sorting_amd -lrsx \
	-Wl,-rpath,'$$ORIGIN/../../radix_sorting_amd' -Wl,-rpath-link,/opt/rocm/lib -pthread -o $@

# counterparts of the reference's `radix` and `radix_bench` commands on this repo's headers (tools/radix.cpp, tools/radix_bench.cpp)
cli: tools/radix tools/radix_bench

tools/radix: tools/radix.cpp include/radix_sort.hpp include/radix_sort_basic_kdf.hpp include/rsx.h radix_sorting_amd/librsx.so
	g++ -std=gnu++17 -O2 -Wall -Iinclude -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ tools/radix.cpp -Lradix_sorting_amd -lrsx \
	-L/opt/rocm/lib -lamdhip64 -Wl,-rpath,'$$ORIGIN/../radix_sorting_amd' -Wl,-rpath,/opt/rocm/lib -o $@

tools/radix_bench: tools/radix_bench.cpp include/radix_sort.hpp include/radix_sort_rank.hpp include/radix_sort_basic_kdf.hpp include/rsx.h radix_sorting_amd/librsx.so
	g++ -std=gnu++17 -O2 -Wall -Iinclude -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ tools/radix_bench.cpp -Lradix_sorting_amd -lrsx \
	-L/opt/rocm/lib -lamdhip64 -Wl,-rpath,'$$ORIGIN/../radix_sorting_amd' -Wl,-rpath,/opt/rocm/lib -o $@

clean:
	rm -f radix_sorting_amd/librsx.so tests/cpp/dropin_check tools/radix tools/radix_bench
	$(MAKE) -C oracle clean

.PHONY: all lib oracle cpp cli clean
