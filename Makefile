# Builds the in-tree native artefacts:
#   tungsten_amd/lib/libtungsten_hip.so   product: C++11 host side + HIP kernels + extern "C" shim (gfx950)
#   tungsten_amd/lib/tungsten_hip         product: CLI (same role as the reference's `tungsten` binary)
#   oracle/liboracle.so                   TEST INFRASTRUCTURE: CPU restatement (never linked by the product)
#   oracle/libm_host.so                   TEST INFRASTRUCTURE: csrc/hip/pt_libm.h compiled for the host (checked against the host libm)
#   oracle/_ref/*                         TEST INFRASTRUCTURE: the reference itself (only where /root/reference exists)
HIPCC    ?= hipcc
CC       ?= gcc
ARCH     ?= gfx950
LIBDIR   := tungsten_amd/lib
# PROFILE=1: the development build with k_shade's section timers (-DPT_PROFILE) as libtungsten_hip_prof.so next to the product
# library (TUNGSTEN_AMD_LIB=... python tools/sweep.py ...); objects of its own
# VARIANT=name VARFLAGS="-D..." : an experiment's build of the device code next to the product library, libtungsten_hip_<name>.so with objects
# of its own (A/B runs within one GPU session: TUNGSTEN_AMD_LIB=tungsten_amd/lib/libtungsten_hip_<name>.so python bench.py ...)
OBJDIR   := $(if $(PROFILE),build/obj_prof,$(if $(VARIANT),build/obj_$(VARIANT),build/obj))
LIBNAME  := $(if $(PROFILE),libtungsten_hip_prof.so,$(if $(VARIANT),libtungsten_hip_$(VARIANT).so,libtungsten_hip.so))
HOSTSRC  := $(wildcard tungsten_amd/csrc/host/*.cpp)
HOSTLIB  := $(filter-out tungsten_amd/csrc/host/main.cpp,$(HOSTSRC))
HOSTOBJ  := $(patsubst tungsten_amd/csrc/host/%.cpp,$(OBJDIR)/host_%.o,$(HOSTLIB))
# the shim + one translation unit per family of k_shade instantiations (they compile in parallel under make -j)
HIPSRC   := $(wildcard tungsten_amd/csrc/hip/*.hip)
HIPOBJ   := $(patsubst tungsten_amd/csrc/hip/%.hip,$(OBJDIR)/%.o,$(HIPSRC))
HIPHDR   := $(wildcard tungsten_amd/csrc/hip/*.h) include/tungsten_hip.h
# -ffp-contract=off: no FMA contraction, so device arithmetic rounds like the CPU reference/oracle
# (DESIGN.md "Numerics"); TG_FAST=1 allows contraction.
FPFLAGS  := $(if $(TG_FAST),-ffp-contract=fast,-ffp-contract=off)
# (-ffp-contract=off: the host restates float32 arithmetic of the reference whose bits matter -- the instance tree and Embree's top-level tree,
# bounds, CDF tables --, so no flag a packager adds, -march=native say, may fuse a multiply with an add)
HOSTFLAGS:= -std=c++11 -O2 -fPIC -ffp-contract=off -Wall -Wextra -Wno-unused-parameter
# -fno-slp-vectorize (round 5): the SLP vectoriser packs the kernels' float arithmetic into v_pk_mul / v_pk_add -- 3.7 cycles per pair on gfx950
# against 2 x 2.0 for the scalar pair, plus the v_mov that put operands side by side -- and the packing costs registers: the shadow walk 119 -> 98
# VGPRs (five waves per SIMD instead of four), k_shade's conductor-family variant 43 -> 0 spilled registers.  Same IEEE operations either way.
# Metric's workload +3.5 %, mesh1m +2 %, instances10k +1.8 %, Cornell box +0.9 % (profiles/r5_ab_no_slp.txt).  SLP=1 gives the vectoriser back.
HIPFLAGS := --offload-arch=$(ARCH) $(if $(HIPOPT),$(HIPOPT),-O3) -std=c++17 -fPIC $(FPFLAGS) $(if $(SLP),,-fno-slp-vectorize) -Wno-unused-result $(if $(PROFILE),-DPT_PROFILE,) $(VARFLAGS)

all: $(LIBDIR)/$(LIBNAME) $(if $(PROFILE)$(VARIANT),,$(LIBDIR)/tungsten_hip oracle/liboracle.so oracle/libm_host.so)

$(OBJDIR)/host_%.o: tungsten_amd/csrc/host/%.cpp $(wildcard tungsten_amd/csrc/host/*.hpp) include/tungsten_hip.h include/tungsten_host.h
	@mkdir -p $(OBJDIR)
	g++ $(HOSTFLAGS) -c $< -o $@

# shade_simple.hip -- the Cornell box's one-launch kernel (k_shade<MASK_LEAN, ., FUSE_TRACE | FUSE_SHADOW | FUSE_LOOP>) and the class-0 shading variants -- is scheduled with
# the AMDGPU back end's max-ilp strategy: Cornell box 2 551 -> 2 600 Msamples/s (+1.9 %, two alternations, images identical), the metric's workload and mesh1m level
# (profiles/r6_ab_compiler_options.txt; on the traversal and the other shading translation units the strategy changes nothing, round 5 and the same file).  SCHED= gives the default back.
SCHED ?= max-ilp
$(OBJDIR)/shade_simple.o: HIPFLAGS += $(if $(SCHED),-mllvm -amdgpu-sched-strategy=$(SCHED),)
# tail.hip -- k_tail, one latency-bound launch per part at the end of every pass (a sixth of a 16-spp pass of the as-shipped materialtest) -- is built -Os: 264 -> 234 VGPRs,
# as shipped 743 -> 750 Msamples/s in three alternations (profiles/r6_ab_compiler_options.txt, session 55); the throughput kernels keep -O3.  TAILOPT=-O3 gives it back.
TAILOPT ?= -Os
$(OBJDIR)/tail.o: HIPFLAGS := $(if $(HIPOPT),$(HIPFLAGS),$(subst -O3,$(TAILOPT),$(HIPFLAGS)))
# tungsten_hip.hip and walk_shadow.hip -- the closest-hit and shadow walks and the small kernels around them -- are built -Os as well: the metric's workload 1 167 -> 1 175 Msamples/s
# (+0.7 %, better in seven of seven alternations), mesh1m +0.5 %, instances10k level (profiles/r6_ab_compiler_options.txt, sessions 57 / 58; -O2 had read level to +0.6 % in session 49).
# The shading units keep -O3 (-Os: level or slower).  WALKOPT=-O3 gives it back.
# walk_shadow.hip alone reads another 0.3 % at -O1 (its launches 603 -> 596 us; the metric better in five of five alternations, session 62; mesh1m level): SHADOWOPT.
WALKOPT ?= -Os
SHADOWOPT ?= -O1
$(OBJDIR)/tungsten_hip.o: HIPFLAGS := $(if $(HIPOPT),$(HIPFLAGS),$(subst -O3,$(WALKOPT),$(HIPFLAGS)))
$(OBJDIR)/walk_shadow.o: HIPFLAGS := $(if $(HIPOPT),$(HIPFLAGS),$(subst -O3,$(SHADOWOPT),$(HIPFLAGS)))

$(OBJDIR)/%.o: tungsten_amd/csrc/hip/%.hip $(HIPHDR)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIBDIR)/$(LIBNAME): $(HOSTOBJ) $(HIPOBJ)
	@mkdir -p $(LIBDIR)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC $^ -o $@ -lpthread -ldl

$(LIBDIR)/tungsten_hip: tungsten_amd/csrc/host/main.cpp $(LIBDIR)/libtungsten_hip.so
	g++ $(HOSTFLAGS) $< -o $@ -L$(LIBDIR) -ltungsten_hip -Wl,-rpath,'$$ORIGIN' -lpthread

oracle/liboracle.so: oracle/oracle.c include/tungsten_hip.h
	$(CC) -std=c99 -O2 -ffp-contract=off -fopenmp -fPIC -shared $< -o $@ -lm

oracle/libm_host.so: oracle/libm_host.cpp tungsten_amd/csrc/hip/pt_libm.h
	$(CXX) -std=c++17 -O2 -ffp-contract=off -mfma -fopenmp -fPIC -shared $< -o $@ -lm

# the reference itself, only where its sources are mounted
ref:
	@if [ -d /root/reference/src ]; then $(MAKE) -f oracle/Makefile.ref -j$$(nproc) all; else echo "no /root/reference: keeping prebuilt oracle/_ref"; fi

clean:
	rm -rf build $(LIBDIR) oracle/liboracle.so oracle/libm_host.so

.PHONY: all ref clean
