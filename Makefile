# Builds the in-tree artefacts: the render-path library (HIP, gfx950), the `run` CLI and the
# CPU oracle used by the tests.  hipcc cross-compiles gfx950 without a GPU present.
HIPCC    ?= /opt/rocm/bin/hipcc
CXX      ?= g++
ARCH     ?= gfx950
CSRC     := raytracer_amd/csrc
# -ffp-contract=off everywhere: the reference arithmetic has no fused multiply-adds and the
# accept/reject decisions of the traversal are compared bit for bit.
HOSTFLAGS := -std=c++17 -O2 -ffp-contract=off -fPIC -pthread -Wall -Wno-unused-function -Iinclude
HIPFLAGS  := -std=c++17 -O3 -ffp-contract=off -fPIC --offload-arch=$(ARCH) -Wall -Wno-unused-function -Iinclude

LIB := raytracer_amd/librt_amd.so
RUN := raytracer_amd/run
# The same library with the leaf loop of long-lived waves (four triangle records in flight, tools/gen_trace_asm.py) forced on
# every leaf of every wave: production takes that loop for few leaves of few waves, the parity tests run against this build
# too (tests/test_stress_variant.py) so that every scene walks through it.
STRESS := raytracer_amd/librt_amd_stress.so

all: $(LIB) $(RUN) $(STRESS) raytracer_amd/finish_check oracle

$(CSRC)/rt_host.o: $(CSRC)/rt_host.cpp $(CSRC)/rt_internal.h include/rt_amd.h
	$(CXX) $(HOSTFLAGS) -c $< -o $@
$(CSRC)/rt_image.o: $(CSRC)/rt_image.cpp $(CSRC)/rt_internal.h include/rt_amd.h
	$(CXX) $(HOSTFLAGS) -c $< -o $@

HIPOBJS := $(CSRC)/rt_device.o $(CSRC)/rt_kernel_lane.o $(CSRC)/rt_kernel_wave.o $(CSRC)/rt_queue.o $(CSRC)/rt_gather.o

$(CSRC)/%.o: $(CSRC)/%.hip $(CSRC)/rt_kernels.h $(CSRC)/rt_internal.h $(CSRC)/rt_math.h $(CSRC)/rt_trace_asm.inc include/rt_amd.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIB): $(CSRC)/rt_host.o $(CSRC)/rt_image.o $(HIPOBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -pthread -o $@ $^ -lz -ldl

# (the kernel source includes "rt_trace_asm.inc" from its own directory: the stress build compiles a copy of the source beside a block of its own)
build/stress/rt_trace_asm.inc: tools/gen_trace_asm.py
	mkdir -p build/stress
	RT_GEN_BATCH_MIN=1 RT_GEN_BATCH_WORK=0 python3 tools/gen_trace_asm.py $@ > /dev/null
build/stress/rt_kernel_wave.hip: $(CSRC)/rt_kernel_wave.hip
	mkdir -p build/stress
	cp $< $@
build/stress/rt_kernel_wave.o: build/stress/rt_kernel_wave.hip build/stress/rt_trace_asm.inc $(CSRC)/rt_kernels.h $(CSRC)/rt_internal.h $(CSRC)/rt_math.h include/rt_amd.h
	$(HIPCC) $(HIPFLAGS) -I$(CSRC) -c $< -o $@
$(STRESS): $(CSRC)/rt_host.o $(CSRC)/rt_image.o $(filter-out $(CSRC)/rt_kernel_wave.o,$(HIPOBJS)) build/stress/rt_kernel_wave.o
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -pthread -o $@ $^ -lz -ldl

# GPU check of the assembly block's decision routine for triangle candidates against the compiled form (tests/test_stress_variant.py runs it)
raytracer_amd/finish_check: tools/gen_finish_check.py tools/gen_trace_asm.py
	python3 tools/gen_finish_check.py > /dev/null
	$(HIPCC) -O3 -ffp-contract=off --offload-arch=$(ARCH) build/finish_check.hip -o $@

$(RUN): $(CSRC)/run_main.cpp $(LIB) include/rt_amd.h
	$(CXX) $(HOSTFLAGS) $< -o $@ -Lraytracer_amd -lrt_amd -Wl,-rpath,'$$ORIGIN'

oracle:
	$(MAKE) -C oracle

clean:
	rm -f $(CSRC)/*.o $(LIB) $(RUN) $(STRESS) raytracer_amd/finish_check
	rm -rf build/stress
	$(MAKE) -C oracle clean

.PHONY: all oracle clean
