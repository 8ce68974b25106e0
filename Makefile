# Builds libpll_amd/libpll_amd.so: C host library + HIP kernels for gfx950.
#   make            product library
#   make oracle     CPU checkers under oracle/ (test infrastructure)
#   make all        both
HIPCC    ?= /opt/rocm/bin/hipcc
CC       ?= gcc
ARCH     ?= gfx950
OUT      := libpll_amd/libpll_amd.so
BUILD    := build

HOST_SRC := $(wildcard libpll_amd/csrc/host/*.c)
HIP_SRC  := $(wildcard libpll_amd/csrc/hip/*.hip)
HOST_OBJ := $(patsubst libpll_amd/csrc/host/%.c,$(BUILD)/host_%.o,$(HOST_SRC))
HIP_OBJ  := $(patsubst libpll_amd/csrc/hip/%.hip,$(BUILD)/hip_%.o,$(HIP_SRC))

# -ffp-contract=off everywhere: fused multiply-adds appear only where the
# source spells fma(); bit-exact parity with the reference depends on it.
CFLAGS   := -std=gnu11 -O2 -fPIC -g -Wall -Wextra -ffp-contract=off -fvisibility=hidden \
            -Iinclude -Ilibpll_amd/csrc/host
HIPFLAGS := --offload-arch=$(ARCH) -O3 -fPIC -ffp-contract=off -fvisibility=hidden \
            -Wall -Iinclude -Ilibpll_amd/csrc/hip

.PHONY: lib oracle all clean
lib: $(OUT)
all: lib oracle

$(BUILD):
	mkdir -p $@

$(BUILD)/host_%.o: libpll_amd/csrc/host/%.c include/pll_amd.h include/pllhip.h libpll_amd/csrc/host/internal.h | $(BUILD)
	$(CC) $(CFLAGS) -c $< -o $@

$(BUILD)/hip_%.o: libpll_amd/csrc/hip/%.hip include/pllhip.h $(wildcard libpll_amd/csrc/hip/*.hpp) | $(BUILD)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(OUT): $(HOST_OBJ) $(HIP_OBJ)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -Wl,-Bsymbolic -Wl,-soname,libpll_amd.so -Wl,-rpath,/opt/rocm/lib \
	    -o $@ $(HOST_OBJ) $(HIP_OBJ) -lm -ldl

oracle:
	$(MAKE) -C oracle all

clean:
	rm -rf $(BUILD) $(OUT)
	$(MAKE) -C oracle clean
