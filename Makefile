# Build of the MI355X-native multigrid library (gfx950 only) and of the CPU oracle.
HIPCC      ?= /opt/rocm/bin/hipcc
ARCH       ?= gfx950
CSRC       := dealii_multigrid_amd/csrc
LIBDIR     := dealii_multigrid_amd/lib
DEBUGFLAGS ?=
HIPFLAGS   := $(DEBUGFLAGS) --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-function -Iinclude
LIB        := $(LIBDIR)/libmgamd.so
HDRS       := $(wildcard $(CSRC)/*.hpp) include/mgamd.h include/mgamd_dev.h

BIN        := dealii_multigrid_amd/bin/multigrid_throughput

all: $(LIB) $(BIN) oracle

$(LIBDIR)/runtime.o: $(CSRC)/runtime.hip $(HDRS)
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

# the operator kernels: one translation unit per (number type, degree), see csrc/apply_inst.hip
APPLY_OBJS := $(foreach t,f64 f32,$(foreach p,1 2 3 4,$(LIBDIR)/apply_$(t)_p$(p).o))
$(LIBDIR)/apply_f64_p%.o: $(CSRC)/apply_inst.hip $(HDRS)
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -DMGAMD_INST_T=double -DMGAMD_INST_P=$* -c $< -o $@
$(LIBDIR)/apply_f32_p%.o: $(CSRC)/apply_inst.hip $(HDRS)
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -DMGAMD_INST_T=float -DMGAMD_INST_P=$* -c $< -o $@

$(LIBDIR)/c_api_device.o: $(CSRC)/c_api_device.hip $(HDRS)
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIBDIR)/c_api_host.o: $(CSRC)/c_api_host.cpp $(HDRS)
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -x hip -c $< -o $@

$(LIB): $(LIBDIR)/runtime.o $(APPLY_OBJS) $(LIBDIR)/c_api_device.o $(LIBDIR)/c_api_host.o
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $^ -L/opt/rocm/lib -lrccl

$(BIN): dealii_multigrid_amd/harness/multigrid_throughput.cpp $(CSRC)/mgamd.hpp include/mgamd.h $(LIB)
	@mkdir -p dealii_multigrid_amd/bin
	g++ -O2 -std=c++17 -Wall -Iinclude $< -o $@ -L$(LIBDIR) -lmgamd -Wl,-rpath,'$$ORIGIN/../lib'

oracle:
	$(MAKE) -C oracle

clean:
	rm -rf $(LIBDIR)/*.o $(LIB) oracle/_build

.PHONY: all oracle clean

# development build with in-kernel stamps / ablations (tools/stamps.py): MGAMD_LIBRARY=dealii_multigrid_amd/lib_debug/libmgamd.so
debug:
	$(MAKE) LIBDIR=dealii_multigrid_amd/lib_debug DEBUGFLAGS=-DMGAMD_KERNEL_DEBUG dealii_multigrid_amd/lib_debug/libmgamd.so
.PHONY: debug

# micro-benchmark behind DESIGN.md's "MFMA only where it pays" statement (tools/mfma_probe.hip)
tools/bin/mfma_probe: tools/mfma_probe.hip $(HDRS)
	@mkdir -p tools/bin
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -I$(CSRC) -Iinclude $< -o $@
