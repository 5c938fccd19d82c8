# Builds the three shared libraries in-tree:
#   pathtracer_amd/libptcore.so  HIP kernels + C ABI (gfx950 only)
#   pathtracer_amd/libpthost.so  C++ host scene layer (links libptcore)
#   oracle/liboracle.so          CPU oracle, test infrastructure only
HIPCC ?= /opt/rocm/bin/hipcc
CXX ?= g++
ARCH ?= gfx950
# FMA contraction off + correctly rounded f32 divide/sqrt: first-hit ids are compared in bits
HIPFLAGS = -O3 --offload-arch=$(ARCH) -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fPIC -shared -std=c++17 -Wall
HOSTFLAGS = -O2 -ffp-contract=off -fno-fast-math -fPIC -shared -std=c++17 -Wall -Wextra -pthread

CORE_SRC = pathtracer_amd/csrc/ptcore.hip pathtracer_amd/csrc/bvh_build.hip pathtracer_amd/csrc/ptgroup.hip
CORE_DEP = $(wildcard pathtracer_amd/csrc/*.h) include/ptcore.h include/ptscene_src.h
HOST_DEP = pathtracer_amd/host/pthost.hpp include/pthost.h include/ptcore.h include/ptscene_src.h

all: pathtracer_amd/libptcore.so pathtracer_amd/libpthost.so pathtracer_amd/ptrender oracle

pathtracer_amd/libptcore.so: $(CORE_SRC) $(CORE_DEP)
	$(HIPCC) $(HIPFLAGS) -o $@ $(CORE_SRC) -ldl

pathtracer_amd/libpthost.so: pathtracer_amd/host/pthost.cpp pathtracer_amd/host/objload.cpp pathtracer_amd/host/imgio.cpp $(HOST_DEP) pathtracer_amd/libptcore.so
	$(CXX) $(HOSTFLAGS) -o $@ pathtracer_amd/host/pthost.cpp pathtracer_amd/host/objload.cpp pathtracer_amd/host/imgio.cpp -Lpathtracer_amd -lptcore -lz -Wl,-rpath,'$$ORIGIN'

pathtracer_amd/ptrender: pathtracer_amd/host/ptrender.cpp $(HOST_DEP) pathtracer_amd/libpthost.so
	$(CXX) -O2 -ffp-contract=off -std=c++17 -Wall -o $@ -Ipathtracer_amd/host pathtracer_amd/host/ptrender.cpp -Lpathtracer_amd -lpthost -lptcore -Wl,-rpath,'$$ORIGIN'

oracle:
	$(MAKE) -C oracle

clean:
	rm -f pathtracer_amd/libptcore.so pathtracer_amd/libpthost.so pathtracer_amd/ptrender
	$(MAKE) -C oracle clean

.PHONY: all oracle clean

# The same library with every scene reference checked before use (PTCORE_LIB=... selects it in the Python binding).
pathtracer_amd/libptcore_check.so: $(CORE_SRC) $(CORE_DEP)
	$(HIPCC) $(HIPFLAGS) -DPT_BOUNDS_CHECK -o $@ $(CORE_SRC) -ldl
check-lib: pathtracer_amd/libptcore_check.so

# The same library with the lane-occupancy counters of the walking kernel (PTCORE_LIB=... python tools/walk_stats.py knob 768).
pathtracer_amd/libptcore_stats.so: $(CORE_SRC) $(CORE_DEP)
	$(HIPCC) $(HIPFLAGS) -DPT_WALK_STATS -o $@ $(CORE_SRC) -ldl
stats-lib: pathtracer_amd/libptcore_stats.so

# Register / LDS / scratch use of every kernel (no output file).
resources:
	$(HIPCC) $(filter-out -shared -fPIC,$(HIPFLAGS)) --cuda-device-only -c -Rpass-analysis=kernel-resource-usage -o /dev/null pathtracer_amd/csrc/ptcore.hip 2>&1 | grep -E "Function Name|VGPRs:|SGPRs:|ScratchSize|Occupancy|LDS Size" | sed 's/.*remark: [^ ]* //' | paste - - - - - -
.PHONY: resources check-lib stats-lib

# Microbenchmarks behind design decisions (run on the GPU box).
tools/micro/atomic_rate: tools/micro/atomic_rate.hip
	$(HIPCC) -O3 --offload-arch=$(ARCH) -o $@ $<
tools/micro/valu_rate: tools/micro/valu_rate.hip
	$(HIPCC) -O2 -w --offload-arch=$(ARCH) -o $@ $<
