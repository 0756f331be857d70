# Builds the product library (libeddsa_amd/libeddsa_amd.so: C host code + HIP kernels for gfx950)
# and the test-only oracle (oracle/).  No GPU is needed to build: hipcc cross-compiles.
ROCM    ?= /opt/rocm
HIPCC   ?= $(ROCM)/bin/hipcc
CC      ?= gcc
ARCH    ?= gfx950
CSRC    := libeddsa_amd/csrc
BUILD   := build
LIB     := libeddsa_amd/libeddsa_amd.so

# -amdgpu-dpp-combine=false: the combiner's v_subrev_u32_dpp computes dpp(src1) - src0 on this hardware instead of
# src1 - dpp(src0) (tools/microbench/dpp_subrev.hip); the lane exchanges of quad_lanes.h stay separate v_mov_b32_dpp
HIPFLAGS := -O3 --offload-arch=$(ARCH) -fPIC -fvisibility=hidden -DEDDSA_BUILD -Iinclude -I$(CSRC) -mllvm -amdgpu-dpp-combine=false
CFLAGS   := -std=c11 -O2 -fPIC -fvisibility=hidden -Wall -Wextra -DEDDSA_BUILD -Iinclude -I$(CSRC) -I$(ROCM)/include

PROBE   := libeddsa_amd/libeddsa_amd_probe.so
# The same objects plus the test and measurement hooks of include/eddsa_amd_debug.h (-DEDDSA_AMD_DEBUG_BUILD compiles them into
# the two host files; the kernels are the SAME object files): what the tests, bench.py and tools/ load.  $(LIB) exports the
# reference's 13 names and the batched set of include/eddsa_amd.h, nothing else (tests/test_abi_and_host.py).
DEBUGLIB := libeddsa_amd/libeddsa_amd_debug.so

all: $(LIB) $(DEBUGLIB) $(PROBE) oracle

$(BUILD)/kernels.o: $(CSRC)/kernels.hip $(wildcard $(CSRC)/*.h)
	@mkdir -p $(BUILD)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(BUILD)/%.o: $(CSRC)/%.c $(CSRC)/engine.h $(CSRC)/eddsa_kernels.h include/eddsa.h include/eddsa_amd.h
	@mkdir -p $(BUILD)
	$(CC) $(CFLAGS) -c $< -o $@

$(BUILD)/%.dbg.o: $(CSRC)/%.c $(CSRC)/engine.h $(CSRC)/eddsa_kernels.h include/eddsa.h include/eddsa_amd.h include/eddsa_amd_debug.h
	@mkdir -p $(BUILD)
	$(CC) $(CFLAGS) -DEDDSA_AMD_DEBUG_BUILD -c $< -o $@

$(DEBUGLIB): $(BUILD)/kernels.o $(BUILD)/rlc.o $(BUILD)/eddsa_amd.dbg.o $(BUILD)/host_pipe.dbg.o
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) -Wl,-soname,libeddsa_amd_debug.so -o $@ $^ -lpthread -ldl

# SONAME = the reference's (libeddsa.so.0, reference lib/CMakeLists.txt:43-44): a program linked against
# the reference loads this library through the libeddsa.so.0 link without being relinked.
$(LIB): $(BUILD)/kernels.o $(BUILD)/rlc.o $(BUILD)/eddsa_amd.o $(BUILD)/host_pipe.o
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) -Wl,-soname,libeddsa.so.0 -o $@ $^ -lpthread -ldl
	ln -sf libeddsa_amd.so libeddsa_amd/libeddsa.so.0

$(BUILD)/rlc.o: $(CSRC)/rlc.hip $(wildcard $(CSRC)/*.h)
	@mkdir -p $(BUILD)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

# The layer probes (include/eddsa_amd_probe.h): test infrastructure, a library of its own - nothing of it is linked into $(LIB)
probe: $(PROBE)
$(PROBE): $(CSRC)/probe.hip $(wildcard $(CSRC)/*.h) include/eddsa_amd_probe.h
	@mkdir -p $(BUILD)
	$(HIPCC) -O3 --offload-arch=$(ARCH) -fPIC -fvisibility=hidden -DEDDSA_PROBE_BUILD -Iinclude -I$(CSRC) -mllvm -amdgpu-dpp-combine=false -shared $< -o $@

oracle:
	$(MAKE) -C oracle all

microbench:
	$(HIPCC) -O3 --offload-arch=$(ARCH) tools/microbench/valu_rates.hip -o tools/microbench/valu_rates.bin
	$(HIPCC) -O3 --offload-arch=$(ARCH) -I$(CSRC) tools/microbench/fe_rates.hip -o tools/microbench/fe_rates.bin

clean:
	rm -rf $(BUILD) $(LIB) $(DEBUGLIB) $(PROBE) libeddsa_amd/libeddsa.so.0
	$(MAKE) -C oracle clean

.PHONY: all oracle probe microbench clean
