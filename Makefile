# Build of the MI355X-native multigrid library (gfx950 only) and of the CPU oracle.
HIPCC      ?= /opt/rocm/bin/hipcc
ARCH       ?= gfx950
CSRC       := dealii_multigrid_amd/csrc
LIBDIR     := dealii_multigrid_amd/lib
DEBUGFLAGS ?=
HIPFLAGS   := $(DEBUGFLAGS) --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-function -Iinclude
LIB        := $(LIBDIR)/libmgamd.so
HDRS       := $(wildcard $(CSRC)/*.hpp) include/mgamd.h

all: $(LIB) oracle

$(LIBDIR)/runtime.o: $(CSRC)/runtime.hip $(HDRS)
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIBDIR)/c_api_device.o: $(CSRC)/c_api_device.hip $(HDRS)
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIBDIR)/c_api_host.o: $(CSRC)/c_api_host.cpp $(HDRS)
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -x hip -c $< -o $@

$(LIB): $(LIBDIR)/runtime.o $(LIBDIR)/c_api_device.o $(LIBDIR)/c_api_host.o
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $^

oracle:
	$(MAKE) -C oracle

clean:
	rm -rf $(LIBDIR)/*.o $(LIB) oracle/_build

.PHONY: all oracle clean
