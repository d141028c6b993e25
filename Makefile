# Builds the product library (libhrt.so: HIP kernels for gfx950 + host C ABI) and the CPU oracle.
#   make            -> nvidia-optix-ray-tracer_amd/lib/libhrt.so, oracle/liboracle.so, tools
#   make lib        -> the product library only
#   make oracle     -> the test oracle only
HIPCC   ?= /opt/rocm/bin/hipcc
CXX     ?= g++
PKG     := nvidia-optix-ray-tracer_amd
CSRC    := $(PKG)/csrc
LIBDIR  := $(PKG)/lib
ARCH    := gfx950

# -ffp-contract=off: arithmetic that feeds control flow must round exactly like the oracle.
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Iinclude -I$(CSRC) -Wall -Wno-unused-function -Wno-pass-failed $(EXTRA_HIPFLAGS)
CXXFLAGS := -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -I$(CSRC) -Wall -pthread

all: lib oracle tools

lib: $(LIBDIR)/libhrt.so $(LIBDIR)/libhrt_io.so

$(LIBDIR)/kernels.o: $(CSRC)/kernels.hip $(CSRC)/device_types.h $(CSRC)/srgb_pow.h $(CSRC)/trav_common.h
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIBDIR)/fused.o: $(CSRC)/fused.hip $(CSRC)/device_types.h $(CSRC)/trav_common.h $(CSRC)/trav_lean.h
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIBDIR)/fused_queue.o: $(CSRC)/fused_queue.hip $(CSRC)/device_types.h $(CSRC)/trav_common.h $(CSRC)/trav_lean.h
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIBDIR)/build.o: $(CSRC)/build.hip $(CSRC)/build.h $(CSRC)/build_dev.h $(CSRC)/bvh8.h $(CSRC)/bvh8_geom.h
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIBDIR)/build_split.o: $(CSRC)/build_split.hip $(CSRC)/build.h $(CSRC)/build_dev.h $(CSRC)/bvh8.h $(CSRC)/bvh8_geom.h
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIBDIR)/refit.o: $(CSRC)/refit.hip $(CSRC)/device_types.h $(CSRC)/bvh8_geom.h
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIBDIR)/pose.o: $(CSRC)/pose.hip $(CSRC)/device_types.h $(CSRC)/cr_trig.h $(CSRC)/srgb_pow.h
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

API_DEPS := $(CSRC)/knobs.def $(CSRC)/build.h $(CSRC)/hrt_internal.hpp $(CSRC)/device_types.h $(CSRC)/bvh8.h $(CSRC)/bvh8_geom.h include/hrt.h include/hrt_params.h

$(LIBDIR)/hrt_accel.o: $(CSRC)/hrt_accel.cpp $(API_DEPS)
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIBDIR)/hrt_api.o: $(CSRC)/hrt_api.cpp $(API_DEPS)
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIBDIR)/bvh8_build.o: $(CSRC)/bvh8_build.cpp $(CSRC)/bvh8.h $(CSRC)/bvh8_geom.h
	@mkdir -p $(LIBDIR)
	$(CXX) $(CXXFLAGS) -c $< -o $@

$(LIBDIR)/bvh8_host_api.o: $(CSRC)/bvh8_host_api.cpp $(CSRC)/bvh8.h include/hrt.h include/hrt_params.h
	@mkdir -p $(LIBDIR)
	$(CXX) $(CXXFLAGS) -c $< -o $@

$(LIBDIR)/libhrt.so: $(LIBDIR)/kernels.o $(LIBDIR)/fused.o $(LIBDIR)/fused_queue.o $(LIBDIR)/build.o $(LIBDIR)/build_split.o $(LIBDIR)/refit.o $(LIBDIR)/pose.o $(LIBDIR)/hrt_api.o $(LIBDIR)/hrt_accel.o $(LIBDIR)/bvh8_build.o $(LIBDIR)/bvh8_host_api.o
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $^ -pthread
	@cat $(sort $(wildcard $(CSRC)/*.hip $(CSRC)/*.h $(CSRC)/*.hpp $(CSRC)/*.cpp include/*.h)) | sha256sum | cut -c1-16 > $(LIBDIR)/BUILD_ID

# host-side readers of the reference's input formats (include/hrt_io.h): plain C++, no GPU
$(LIBDIR)/libhrt_io.so: $(CSRC)/host/scene_io.cpp $(CSRC)/host/json_min.hpp $(CSRC)/cr_trig.h $(CSRC)/srgb_pow.h include/hrt_io.h include/hrt_params.h
	@mkdir -p $(LIBDIR)
	$(CXX) $(CXXFLAGS) -shared -o $@ $<

oracle:
	$(MAKE) -C oracle

# hrt_render is the multi-GPU host (csrc/host/multi_gpu.hpp calls RCCL directly): built only where RCCL is installed, so that
# the default target works on a single-GPU host without it
ROCM_PATH ?= /opt/rocm
ifneq ($(wildcard $(ROCM_PATH)/include/rccl/rccl.h),)
MULTI_GPU_TOOL := $(LIBDIR)/hrt_render
else
MULTI_GPU_TOOL :=
$(info RCCL not found under $(ROCM_PATH): skipping hrt_render (the multi-GPU driver))
endif
tools: $(MULTI_GPU_TOOL) $(LIBDIR)/hrt_time_render $(LIBDIR)/hrt_mesh_render

TOOL_HDRS := include/hrt.h include/hrt_params.h include/hrt_io.h
$(LIBDIR)/hrt_render: $(CSRC)/host/hrt_render.cpp $(CSRC)/host/renderer_host.hpp $(CSRC)/host/multi_gpu.hpp $(TOOL_HDRS) $(LIBDIR)/libhrt.so
	$(HIPCC) --offload-arch=$(ARCH) -O2 -std=c++17 -Iinclude -I$(CSRC) -I$(ROCM_PATH)/include -o $@ $< -L$(LIBDIR) -lhrt -L$(ROCM_PATH)/lib -lrccl -pthread -Wl,-rpath,'$$ORIGIN' -Wl,-rpath,$(ROCM_PATH)/lib

$(LIBDIR)/hrt_time_render: $(CSRC)/host/hrt_time_render.cpp $(CSRC)/host/renderer_host.hpp $(TOOL_HDRS) $(LIBDIR)/libhrt.so $(LIBDIR)/libhrt_io.so
	$(HIPCC) --offload-arch=$(ARCH) -O2 -std=c++17 -Iinclude -I$(CSRC) -o $@ $< -L$(LIBDIR) -lhrt -lhrt_io -Wl,-rpath,'$$ORIGIN'

$(LIBDIR)/hrt_mesh_render: $(CSRC)/host/hrt_mesh_render.cpp $(CSRC)/host/renderer_host.hpp $(TOOL_HDRS) $(LIBDIR)/libhrt.so $(LIBDIR)/libhrt_io.so
	$(HIPCC) --offload-arch=$(ARCH) -O2 -std=c++17 -Iinclude -I$(CSRC) -o $@ $< -L$(LIBDIR) -lhrt -lhrt_io -pthread -Wl,-rpath,'$$ORIGIN'

clean:
	rm -rf $(LIBDIR)
	$(MAKE) -C oracle clean

.PHONY: all lib oracle tools clean

# instrumented build for tools/lane_stats.py: lane-utilisation counters compiled into the path kernels
stats: $(LIBDIR)/libhrt_stats.so
$(LIBDIR)/libhrt_stats.so: $(LIBDIR)/fused_queue.o $(CSRC)/kernels.hip $(CSRC)/fused.hip $(CSRC)/trav_common.h $(CSRC)/trav_lean.h $(LIBDIR)/libhrt.so
	$(HIPCC) $(HIPFLAGS) -DHRT_LANE_STATS -c $(CSRC)/kernels.hip -o $(LIBDIR)/kernels_stats.o
	$(HIPCC) $(HIPFLAGS) -DHRT_LANE_STATS -c $(CSRC)/fused.hip -o $(LIBDIR)/fused_stats.o
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(LIBDIR)/kernels_stats.o $(LIBDIR)/fused_stats.o $(LIBDIR)/fused_queue.o $(LIBDIR)/build.o $(LIBDIR)/build_split.o $(LIBDIR)/refit.o $(LIBDIR)/pose.o $(LIBDIR)/hrt_api.o $(LIBDIR)/hrt_accel.o $(LIBDIR)/bvh8_build.o $(LIBDIR)/bvh8_host_api.o -pthread

# ---- sanitizer build (CPU only; GPU AddressSanitizer is not available and is never attempted): the host-side code that takes
# untrusted files or builds trees on the host -- the format readers, the host BVH8 builder, the test oracle -- compiled with
# AddressSanitizer + UndefinedBehaviorSanitizer into $(LIBDIR)/asan/, and `make asan-test` runs the CPU tests that exercise
# them (truncated / garbage inputs included: tests/test_fuzz_inputs_cpu.py) against those libraries.
ASAN_DIR   := $(LIBDIR)/asan
ASAN_FLAGS := -O1 -g -std=c++17 -fPIC -ffp-contract=off -fsanitize=address,undefined -fno-omit-frame-pointer -fno-sanitize-recover=undefined -Iinclude -I$(CSRC) -Wall -pthread
asan: $(ASAN_DIR)/libhrt_io.so $(ASAN_DIR)/libhrt_host_bvh.so $(ASAN_DIR)/liboracle.so
$(ASAN_DIR)/libhrt_io.so: $(CSRC)/host/scene_io.cpp $(CSRC)/host/json_min.hpp $(CSRC)/cr_trig.h $(CSRC)/srgb_pow.h include/hrt_io.h include/hrt_params.h
	@mkdir -p $(ASAN_DIR)
	$(CXX) $(ASAN_FLAGS) -shared -o $@ $<
$(ASAN_DIR)/libhrt_host_bvh.so: $(CSRC)/bvh8_build.cpp $(CSRC)/bvh8_host_api.cpp $(CSRC)/bvh8.h $(CSRC)/bvh8_geom.h include/hrt.h
	@mkdir -p $(ASAN_DIR)
	$(CXX) $(ASAN_FLAGS) -DHRT_HOST_ONLY -shared -o $@ $(CSRC)/bvh8_build.cpp $(CSRC)/bvh8_host_api.cpp
$(ASAN_DIR)/liboracle.so: oracle/oracle.c
	@mkdir -p $(ASAN_DIR)
	gcc -O1 -g -std=c11 -fPIC -shared -fopenmp -ffp-contract=off -fsanitize=address,undefined -fno-omit-frame-pointer -fno-sanitize-recover=undefined -o $@ $< -lquadmath -lm
asan-test: asan
	LD_PRELOAD="$$(gcc -print-file-name=libasan.so) $$(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1 \
	HRT_IO_LIB=$(CURDIR)/$(ASAN_DIR)/libhrt_io.so HRT_HOST_BVH_LIB=$(CURDIR)/$(ASAN_DIR)/libhrt_host_bvh.so HRT_ORACLE_LIB=$(CURDIR)/$(ASAN_DIR)/liboracle.so \
	python3 -m pytest tests/test_fuzz_inputs_cpu.py tests/test_io_cpu.py tests/test_oracle_cpu.py tests/test_host_cpu.py -x -q -p no:cacheprovider -k "not pow_pin and not cpp_driver and not exports_every and not no_cpu_fallback and not sbt_header and not sanitizer_job"
.PHONY: asan asan-test
