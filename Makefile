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
            -Wall -Iinclude -Ilibpll_amd/csrc/hip $(EXTRA_HIPFLAGS)

.PHONY: lib oracle all clean asan
lib: $(OUT)
all: lib oracle

$(BUILD):
	mkdir -p $@

$(BUILD)/host_%.o: libpll_amd/csrc/host/%.c include/pll_amd.h include/pllhip.h libpll_amd/csrc/host/internal.h | $(BUILD)
	$(CC) $(CFLAGS) -c $< -o $@

$(BUILD)/hip_%.o: libpll_amd/csrc/hip/%.hip include/pllhip.h $(wildcard libpll_amd/csrc/hip/*.hpp) $(wildcard libpll_amd/csrc/hip/*.inc) | $(BUILD)
	$(HIPCC) $(HIPFLAGS) $(HIPFLAGS_$*) -c $< -o $@

# partials_aa_fused.hip keeps its values in the accumulation registers a0..a109 behind the compiler's
# back (inline assembly): the matrix-core instructions must then keep THEIR accumulators in ordinary
# registers, or the compiler parks them in a0.. between two assembly blocks (seen), and nothing may be
# spilled there either.  Clobber lists do not guarantee that, so the build itself checks: the object is
# compiled with -save-temps and tools/check_agprs.py reads the assembly of THIS compilation (this ARCH, these
# flags, this compiler) -- an instruction outside the slot assembly that names a0..a109 fails the build.
HIPFLAGS_partials_aa_fused := -mllvm -amdgpu-mfma-vgpr-form -mllvm -amdgpu-spill-vgpr-to-agpr=0

$(BUILD)/hip_partials_aa_fused.o: libpll_amd/csrc/hip/partials_aa_fused.hip include/pllhip.h $(wildcard libpll_amd/csrc/hip/*.hpp) $(wildcard libpll_amd/csrc/hip/*.inc) tools/check_agprs.py | $(BUILD)
	rm -rf $(BUILD)/aa_fused_tmp && mkdir -p $(BUILD)/aa_fused_tmp
	$(HIPCC) $(HIPFLAGS) $(HIPFLAGS_partials_aa_fused) -save-temps=obj -c $< -o $(BUILD)/aa_fused_tmp/hip_partials_aa_fused.o
	python3 tools/check_agprs.py $(AGPR_CHECK_FLAGS) $(BUILD)/aa_fused_tmp/partials_aa_fused-hip-amdgcn-amd-amdhsa-$(ARCH).s
	mv $(BUILD)/aa_fused_tmp/hip_partials_aa_fused.o $@

$(OUT): $(HOST_OBJ) $(HIP_OBJ)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -Wl,-Bsymbolic -Wl,-soname,libpll_amd.so -Wl,-rpath,/opt/rocm/lib \
	    -o $@ $(HOST_OBJ) $(HIP_OBJ) -lm -ldl

oracle:
	$(MAKE) -C oracle all

# CPU-side sanitizer build (the reference's analogue: test/eval_valgrind.sh): the host C
# layer and the oracle compiled with AddressSanitizer + UBSan, linked with the ordinary HIP
# objects, and the whole non-GPU test suite run against them (host logic, the FASTA/PHYLIP
# corpus, compression, tree builders, the planner's dry run).  CPU build only -- sanitizers
# are not run on the GPU box.
ASAN_DIR   := $(BUILD)/asan
ASAN_FLAGS := -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -O1
ASAN_OBJ   := $(patsubst libpll_amd/csrc/host/%.c,$(ASAN_DIR)/host_%.o,$(HOST_SRC))

$(ASAN_DIR):
	mkdir -p $@

$(ASAN_DIR)/host_%.o: libpll_amd/csrc/host/%.c include/pll_amd.h include/pllhip.h libpll_amd/csrc/host/internal.h | $(ASAN_DIR)
	gcc $(filter-out -O2,$(CFLAGS)) $(ASAN_FLAGS) -c $< -o $@

$(ASAN_DIR)/libpll_amd.so: $(ASAN_OBJ) $(HIP_OBJ)
	g++ -shared -fPIC $(ASAN_FLAGS) -Wl,-Bsymbolic -Wl,-rpath,/opt/rocm/lib -o $@ $(ASAN_OBJ) $(HIP_OBJ) \
	    -L/opt/rocm/lib -lamdhip64 -lm -ldl

$(ASAN_DIR)/liboracle.so: $(wildcard oracle/*.c) oracle/oracle.h | $(ASAN_DIR)
	gcc -std=c99 -fPIC -g -Wall -Wextra -ffp-contract=off -mfma -mavx2 -D_GNU_SOURCE $(ASAN_FLAGS) \
	    -shared -Wl,-Bsymbolic -o $@ $(wildcard oracle/*.c) -lm

asan: $(ASAN_DIR)/libpll_amd.so $(ASAN_DIR)/liboracle.so
	LD_PRELOAD="$$(gcc -print-file-name=libasan.so) $$(gcc -print-file-name=libubsan.so)" \
	ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:allocator_may_return_null=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
	PLL_AMD_LIB=$(abspath $(ASAN_DIR)/libpll_amd.so) PLL_ORACLE_LIB=$(abspath $(ASAN_DIR)/liboracle.so) \
	python3 -m pytest tests -x -q -m "not gpu" -p no:cacheprovider

clean:
	rm -rf $(BUILD) $(OUT)
	$(MAKE) -C oracle clean
