# Top-level build: the gfx950 product library, the C++ host, and (test infrastructure) the CPU oracle.
#
#   make            -> rbrt_amd/lib/librbrt_hip.so  (HIP kernels + C ABI)   [hipcc, gfx950 only]
#                      rbrt_amd/lib/librbrt_host.so, rbrt_amd/bin/rbrt      [C++ host: CLI/YAML/OBJ/PNG]
#                      oracle/librbrt_oracle.so                            [CPU checker, tests only]
HIPCC ?= /opt/rocm/bin/hipcc
CXX ?= g++
ARCH ?= gfx950

# -ffp-contract=off is load-bearing: hipcc's default (fast-honor-pragmas) would fuse mul+add into FMA
# and the radiance would no longer match the reference's unfused f32 arithmetic (vec3_avx.rs:18-21).
# -fno-slp-vectorize: left on, the SLP vectoriser packs pairs of scalar f32 mul/add into v_pk_mul_f32 / v_pk_add_f32, which
# issue at half the rate of the plain forms on gfx950 and need register shuffles around them (profiles/r03_valu_rate.txt;
# measured on the frame: -1 % isolated launch, -1.9 % per pipelined step, same image).
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize \
            -Wall -Wextra -Wno-unused-parameter
CSRC := rbrt_amd/csrc
LIBDIR := rbrt_amd/lib
BINDIR := rbrt_amd/bin

all: $(LIBDIR)/librbrt_hip.so host oracle

$(LIBDIR)/librbrt_hip.so: $(CSRC)/kernels.hip $(CSRC)/megakernel.inl $(CSRC)/api.cpp $(CSRC)/bvh.cpp $(CSRC)/bvh.h \
                          $(CSRC)/bvh_device.hip $(CSRC)/bvh_device.h \
                          $(CSRC)/device_types.h include/rbrt_hip.h include/rbrt_hip_debug.h
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(CSRC)/kernels.hip $(CSRC)/bvh_device.hip $(CSRC)/api.cpp $(CSRC)/bvh.cpp

# Analysis build (not the product): every region marker of the megakernel stamps s_memtime; tools/region_profile.py reads it.
timers: $(LIBDIR)/librbrt_hip_timers.so
$(LIBDIR)/librbrt_hip_timers.so: $(LIBDIR)/librbrt_hip.so
	$(HIPCC) $(HIPFLAGS) -DRBRT_REGION_TIMERS=1 -shared -o $@ $(CSRC)/kernels.hip $(CSRC)/bvh_device.hip $(CSRC)/api.cpp $(CSRC)/bvh.cpp

host:
	@if [ -f rbrt_amd/host/Makefile ]; then $(MAKE) -C rbrt_amd/host; fi

oracle:
	$(MAKE) -C oracle

# CPU-side C++ under AddressSanitizer + UBSan / ThreadSanitizer (tests/cpp/host_selftest.cpp)
asan tsan:
	$(MAKE) -C tests/cpp $@

clean:
	rm -rf $(LIBDIR) $(BINDIR)
	$(MAKE) -C oracle clean

# the flags, for the tools that build variants of the library (tools/build_variant.sh, tools/kres.py)
print-hipflags:
	@echo $(HIPFLAGS)

.PHONY: all host oracle clean asan tsan timers print-hipflags
