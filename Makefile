# Top-level build: the gfx950 engine (libm3hip.so), the synthetic-MC generator
# (libm3synth.so) and the test-only CPU oracle (oracle/).  No cmake needed.
HIPCC ?= /opt/rocm/bin/hipcc
CC ?= gcc
ARCH ?= gfx950
LIBDIR = mach3_amd/lib
CSRC = mach3_amd/csrc

# -ffp-contract=off : only the explicit fmaf() calls fuse (bit-parity with the reference order)
# -munsafe-fp-atomics: hardware f64 atomic add (global_atomic_add_f64 / ds_add_f64), no CAS loop
HIPFLAGS = --offload-arch=$(ARCH) -std=c++17 -O3 -fPIC -ffp-contract=off -munsafe-fp-atomics \
           -Wall -Wextra -Wno-unused-parameter -Iinclude

all: $(LIBDIR)/libm3hip.so $(LIBDIR)/libm3synth.so oracle

# the engine: one object per translation unit (make -j builds them side by side; the kernels take the longest), one link
ENGINE_TUS = m3hip_api m3hip_upload m3hip_sample m3hip_step m3hip_comm m3hip_par m3hip_kernels
ENGINE_SRC = $(ENGINE_TUS:%=$(CSRC)/%.hip) $(CSRC)/m3hip_layout.cpp
ENGINE_HDR = $(CSRC)/m3hip_ctx.h $(CSRC)/m3hip_kernels.h $(CSRC)/m3hip_layout.h include/m3hip.h
ENGINE_OBJ = $(ENGINE_TUS:%=build/obj/%.o) build/obj/m3hip_layout.o
build/obj/%.o: $(CSRC)/%.hip $(ENGINE_HDR)
	@mkdir -p build/obj
	$(HIPCC) $(HIPFLAGS) -c -o $@ $<
build/obj/m3hip_layout.o: $(CSRC)/m3hip_layout.cpp $(CSRC)/m3hip_layout.h
	@mkdir -p build/obj
	$(HIPCC) $(HIPFLAGS) -x hip -c -o $@ $<
$(LIBDIR)/libm3hip.so: $(ENGINE_OBJ)
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(ENGINE_OBJ) -ldl -lpthread

$(LIBDIR)/libm3synth.so: mach3_amd/synth/m3synth.c mach3_amd/synth/m3synth.h
	@mkdir -p $(LIBDIR)
	$(CC) -O3 -fPIC -shared -fopenmp -ffp-contract=off -Wall -Wextra -o $@ mach3_amd/synth/m3synth.c -lm

oracle:
	$(MAKE) -C oracle

# diagnostic build of the engine with phase stamps in the brick kernel (scripts/brick_stamps.py)
stamps: build/libm3hip_stamps.so
build/libm3hip_stamps.so: $(ENGINE_SRC) $(ENGINE_HDR)
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -DM3_BRICK_STAMPS -shared -o $@ $(ENGINE_SRC) -ldl -lpthread

# diagnostic build: the brick kernel without arithmetic and LDS traffic (loads, waits and barriers only) -- the memory side's share
hollow: build/libm3hip_hollow.so
build/libm3hip_hollow.so: $(ENGINE_SRC) $(ENGINE_HDR)
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -DM3_BRICK_HOLLOW=$(or $(HOLLOW),1) -shared -o $@ $(ENGINE_SRC) -ldl -lpthread

# experiment build: make variant NAME=foo XDEF="-DM3X_FOO" -> build/libm3hip_foo.so (load with M3HIP_LIB; scripts/gexp.py)
variant:
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) $(XDEF) -shared -o build/libm3hip_$(NAME).so $(ENGINE_SRC) -ldl -lpthread

isa: $(CSRC)/m3hip_kernels.hip
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) $(XDEF) -c --cuda-device-only -save-temps=obj -Rpass-analysis=kernel-resource-usage \
	    -o build/m3hip_kernels.o $(CSRC)/m3hip_kernels.hip

clean:
	rm -rf $(LIBDIR)/*.so build/*
	$(MAKE) -C oracle clean

.PHONY: all oracle clean isa stamps hollow variant

# C++ host driver over the facade headers (include/mach3_amd/): DragRace + a short MR2T2 chain
examples: build/mr2t2_dragrace build/likelihood_fit_nm build/mr2t2_sharded build/two_samples_one_fitter build/facade_check build/c5_step_loop build/libm3facade_test.so build/libm3seam2_test.so

# C shim over the header-only facade for the CPU tests (links libm3hip.so only for the symbols the headers reference)
build/libm3facade_test.so: tests/cpp/facade_shim.cpp include/mach3_amd/*.h include/m3hip.h $(LIBDIR)/libm3hip.so
	@mkdir -p build
	g++ -std=c++17 -O2 -fPIC -shared -fopenmp -ffp-contract=off -Wall -Wextra -Iinclude -o $@ tests/cpp/facade_shim.cpp -L$(LIBDIR) -lm3hip \
	    -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,'$$ORIGIN/../$(LIBDIR)' -Wl,-rpath,/opt/rocm/lib

# the reference's own SMonolithGPU call sites against the facade (tests/test_cpp_facade.py)
build/libm3seam2_test.so: tests/cpp/seam2_callsites.cpp include/mach3_amd/*.h include/m3hip.h $(LIBDIR)/libm3hip.so
	@mkdir -p build
	g++ -std=c++17 -O2 -fPIC -shared -ffp-contract=off -Wall -Wextra -Iinclude -o $@ tests/cpp/seam2_callsites.cpp -L$(LIBDIR) -lm3hip \
	    -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,'$$ORIGIN/../$(LIBDIR)' -Wl,-rpath,/opt/rocm/lib

build/facade_check: examples/facade_check.cpp include/mach3_amd/*.h include/m3hip.h $(LIBDIR)/libm3hip.so
	@mkdir -p build
	g++ -std=c++17 -O2 -Wall -Wextra -ffp-contract=off -Iinclude -o $@ examples/facade_check.cpp -L$(LIBDIR) -lm3hip \
	    -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,'$$ORIGIN/../mach3_amd/lib' -Wl,-rpath,/opt/rocm/lib

# BASELINE configs[4]: the adaptive-MCMC step loop at 1 040 parameters, host phases timed beside the GPU step
build/c5_step_loop: examples/c5_step_loop.cpp include/mach3_amd/*.h include/m3hip.h $(LIBDIR)/libm3hip.so $(LIBDIR)/libm3synth.so
	@mkdir -p build
	g++ -std=c++17 -O3 -march=x86-64-v3 -fopenmp -Wall -Wextra -Iinclude -o $@ examples/c5_step_loop.cpp -L$(LIBDIR) -lm3hip -lm3synth \
	    -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,'$$ORIGIN/../$(LIBDIR)' -Wl,-rpath,/opt/rocm/lib

build/mr2t2_dragrace: examples/mr2t2_dragrace.cpp include/mach3_amd/*.h include/m3hip.h $(LIBDIR)/libm3hip.so $(LIBDIR)/libm3synth.so
	@mkdir -p build
	g++ -std=c++17 -O2 -Wall -Wextra -Iinclude -o $@ examples/mr2t2_dragrace.cpp -L$(LIBDIR) -lm3hip -lm3synth \
	    -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,'$$ORIGIN/../$(LIBDIR)' -Wl,-rpath,/opt/rocm/lib

# the minimisers' caller: LikelihoodFit::CalcChi2 under a Nelder-Mead simplex, an Asimov fit that must return its truth
build/likelihood_fit_nm: examples/likelihood_fit_nm.cpp include/mach3_amd/*.h include/m3hip.h $(LIBDIR)/libm3hip.so $(LIBDIR)/libm3synth.so
	@mkdir -p build
	g++ -std=c++17 -O2 -Wall -Wextra -Iinclude -o $@ examples/likelihood_fit_nm.cpp -L$(LIBDIR) -lm3hip -lm3synth \
	    -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,'$$ORIGIN/../$(LIBDIR)' -Wl,-rpath,/opt/rocm/lib

# one sample over several GPUs through the facade: N processes, one per GPU, the same seeded chain on every rank
build/mr2t2_sharded: examples/mr2t2_sharded.cpp include/mach3_amd/*.h include/m3hip.h $(LIBDIR)/libm3hip.so $(LIBDIR)/libm3synth.so
	@mkdir -p build
	g++ -std=c++17 -O2 -Wall -Wextra -Iinclude -o $@ examples/mr2t2_sharded.cpp -L$(LIBDIR) -lm3hip -lm3synth \
	    -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,'$$ORIGIN/../$(LIBDIR)' -Wl,-rpath,/opt/rocm/lib

# two live sample handlers (a brick sample and a double-build binned-spline sample) in one fitter
build/two_samples_one_fitter: examples/two_samples_one_fitter.cpp include/mach3_amd/*.h include/m3hip.h $(LIBDIR)/libm3hip.so $(LIBDIR)/libm3synth.so
	@mkdir -p build
	g++ -std=c++17 -O2 -Wall -Wextra -Iinclude -o $@ examples/two_samples_one_fitter.cpp -L$(LIBDIR) -lm3hip -lm3synth \
	    -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,'$$ORIGIN/../$(LIBDIR)' -Wl,-rpath,/opt/rocm/lib

.PHONY: examples

# calibration programs behind the ceilings quoted in DESIGN.md / profiles/README.md (run on an MI355X)
TOOLS = build/sector_fetch build/hbm_read_bw build/hbm_chunk_bw build/hbm_run_bw build/hbm_group_bw build/fetch_size_widths build/hbm_width_bw build/mall_probe build/step_floor
tools: $(TOOLS)
build/%: tools/%.hip
	@mkdir -p build
	$(HIPCC) --offload-arch=$(ARCH) -O3 -o $@ $<
.PHONY: tools
