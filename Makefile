# Build the MI355X OFFT library (HIP kernels + C host) and the test oracle.
#   make            -> offt_amd/liboffthip.so, oracle/liboracle.so
#   make harness    -> bin/run-fft (C harness, links the HIP runtime explicitly)
ROCM      ?= /opt/rocm
HIPCC     ?= $(ROCM)/bin/hipcc
CC        ?= gcc
ARCH      ?= gfx950
CSRC      := offt_amd/csrc
BUILD     := build
HIPFLAGS  := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -I$(CSRC) -Iinclude -I$(BUILD)
CFLAGS    := -std=gnu11 -O2 -g -Wall -Wextra -fPIC -I$(ROCM)/include -I$(CSRC) -Iinclude

all: offt_amd/liboffthip.so oracle/liboracle.so tests/liboffthip_test.so tools/liboffthip_diag.so

$(BUILD):
	mkdir -p $(BUILD)

# the kernel instantiations are grouped into several translation units so that `make -j` compiles them in parallel
HIPSRC    := offt_reg_pow2_f32_big offt_kernels offt_reg_pow2_f64 offt_reg_pow2_f64_1024 offt_reg_pow2_f64_anysplit offt_reg_pow2_f32 offt_reg_pow2_f32_anysplit offt_reg_pow2_f32_pair offt_reg_pow2_tw4 offt_reg_mixed_f64_a offt_reg_mixed_f64_b \
             offt_reg_mixed_f64_c offt_reg_mixed_f64_d offt_reg_mixed_f64_e offt_reg_mixed_f32_a offt_reg_mixed_f32_b offt_reg_bluestein offt_reg_bluestein_f32
HIPOBJ    := $(HIPSRC:%=$(BUILD)/%.o)
$(BUILD)/%.o: $(CSRC)/%.hip $(CSRC)/offt_panel.hpp $(CSRC)/offt_bluestein.hpp $(CSRC)/offt_hipk.h $(CSRC)/offt_w32_consts.h $(CSRC)/offt_wr_consts.h | $(BUILD)
	$(HIPCC) $(HIPFLAGS) -Rpass-analysis=kernel-resource-usage -c $< -o $@ 2> $(BUILD)/$*.resource_usage.txt || (cat $(BUILD)/$*.resource_usage.txt; false)

# the device part of offt_panel.hpp as a string inside the library: plan-time specialisation through hipRTC
$(BUILD)/offt_rtc_source.inc: $(CSRC)/offt_panel.hpp $(CSRC)/offt_hipk.h $(CSRC)/offt_w32_consts.h $(CSRC)/offt_wr_consts.h tools/gen_rtc_source.py | $(BUILD)
	python3 tools/gen_rtc_source.py $@
$(BUILD)/offt_kernels.o: $(BUILD)/offt_rtc_source.inc

$(BUILD)/offt_host.o: $(CSRC)/offt_host.c $(CSRC)/offt_hipk.h $(CSRC)/offt_backend.h include/offt.h include/offt_hip.h | $(BUILD)
	$(CC) $(CFLAGS) -c $< -o $@

# The HIP runtime is deliberately NOT a DT_NEEDED entry: the hosting process
# decides which libamdhip64 is in use (PyTorch bundles its own); C programs link
# -lamdhip64 themselves.
offt_amd/liboffthip.so: $(HIPOBJ) $(BUILD)/offt_host.o
	g++ -shared -o $@ $^ -Wl,--allow-shlib-undefined -Wl,-Bsymbolic -ldl -lm -lpthread

# TEST build of the same library: identical kernel objects, the host compiled with -DOFFT_TEST_SEAMS, which adds the two
# test-only entry points of offt_backend.h (CPU descriptor interpreter for the host-logic tests, host-staged transport
# for several ranks on one GPU).  The product library above does not contain them.
$(BUILD)/offt_host_test.o: $(CSRC)/offt_host.c $(CSRC)/offt_hipk.h $(CSRC)/offt_backend.h include/offt.h include/offt_hip.h | $(BUILD)
	$(CC) $(CFLAGS) -DOFFT_TEST_SEAMS -c $< -o $@
tests/liboffthip_test.so: $(HIPOBJ) $(BUILD)/offt_host_test.o
	g++ -shared -o $@ $^ -Wl,--allow-shlib-undefined -Wl,-Bsymbolic -ldl -lm -lpthread

# DIAGNOSTICS build for launchers (bench.py's exchange-only / compute-only split of a multi-rank run): the product plus
# offt_hip_set_debug_skip, which makes an execute leave out its passes or its exchanges.  Never the measured library.
$(BUILD)/offt_host_diag.o: $(CSRC)/offt_host.c $(CSRC)/offt_hipk.h $(CSRC)/offt_backend.h include/offt.h include/offt_hip.h | $(BUILD)
	$(CC) $(CFLAGS) -DOFFT_BENCH_DIAGNOSTICS -c $< -o $@
tools/liboffthip_diag.so: $(HIPOBJ) $(BUILD)/offt_host_diag.o
	g++ -shared -o $@ $^ -Wl,--allow-shlib-undefined -Wl,-Bsymbolic -ldl -lm -lpthread

oracle/liboracle.so: oracle/oracle_fft.c oracle/oracle_offt.c oracle/oracle.h
	$(CC) -std=gnu11 -O3 -fopenmp -fPIC -shared -Ioracle -o $@ oracle/oracle_fft.c oracle/oracle_offt.c -lm

clean:
	rm -rf $(BUILD) offt_amd/liboffthip.so tests/liboffthip_test.so tools/liboffthip_diag.so oracle/liboracle.so bin

.PHONY: all clean

# test-only CPU interpreter of pass descriptors (never linked into the product)
tests/libcpubackend.so: tests/cpu_backend.c oracle/oracle_fft.c oracle/oracle.h offt_amd/csrc/offt_backend.h
	$(CC) -std=gnu11 -O2 -fPIC -shared -Ioracle -Ioffamd -I$(CSRC) -o $@ tests/cpu_backend.c oracle/oracle_fft.c -lm

# run-fft-compatible C harness.  MPI=1 builds the multi-rank variant
# against an MPI found at MPI_PREFIX (e.g. /opt/conda; linked by file name, not -L: a conda lib directory also holds an
# older libstdc++ that must not shadow the system one the HIP runtime needs).
MPI_PREFIX ?= /opt/conda
harness: bin/run-fft
bin/run-fft: harness/run-fft.c offt_amd/liboffthip.so
	mkdir -p bin
ifeq ($(MPI),1)
	$(CC) -std=gnu11 -O2 -Wall -DOFFT_HARNESS_MPI -Iinclude -I$(MPI_PREFIX)/include -o $@ harness/run-fft.c -Lofft_amd -loffthip -L$(ROCM)/lib -lamdhip64 $(MPI_PREFIX)/lib/libmpi.so -lm -Wl,-rpath-link,/usr/lib/x86_64-linux-gnu -Wl,-rpath,'$$ORIGIN/../offt_amd' -Wl,-rpath,$(ROCM)/lib -Wl,-rpath,$(MPI_PREFIX)/lib
else
	$(CC) -std=gnu11 -O2 -Wall -Iinclude -o $@ harness/run-fft.c -Lofft_amd -loffthip -L$(ROCM)/lib -lamdhip64 -lm -Wl,-rpath,'$$ORIGIN/../offt_amd' -Wl,-rpath,$(ROCM)/lib
endif

# host logic under AddressSanitizer + UBSan on the CPU test backend (GPU ASan is not available on the pool)
asan-test: $(HIPOBJ)
	mkdir -p $(BUILD)/asan
	$(CC) -std=gnu11 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -fPIC -DOFFT_TEST_SEAMS -I$(ROCM)/include -I$(CSRC) -Iinclude -c $(CSRC)/offt_host.c -o $(BUILD)/asan/offt_host.o
	g++ -shared -fsanitize=address,undefined -o $(BUILD)/asan/liboffthip.so $(HIPOBJ) $(BUILD)/asan/offt_host.o -Wl,--allow-shlib-undefined -ldl -lm -lpthread
	$(CC) -std=gnu11 -O1 -g -fsanitize=address,undefined -fPIC -shared -Ioracle -I$(CSRC) -o $(BUILD)/asan/libcpubackend.so tests/cpu_backend.c oracle/oracle_fft.c -lm
	cp tests/libcpubackend.so $(BUILD)/asan/libcpubackend.so.orig
	cp $(BUILD)/asan/libcpubackend.so tests/libcpubackend.so
	LD_PRELOAD="$$($(CC) -print-file-name=libasan.so) $$($(CC) -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0 \
	  OFFT_AMD_TEST_LIB=$(CURDIR)/$(BUILD)/asan/liboffthip.so python -m pytest tests/test_host_logic.py tests/test_p2p_world.py -x -q; \
	  rc=$$?; cp $(BUILD)/asan/libcpubackend.so.orig tests/libcpubackend.so; exit $$rc
