# Builds libdehalo.so (HIP, gfx950 only) and the oracle (test infrastructure).
HIPCC ?= /opt/rocm/bin/hipcc
PKG := delay-encryption-in-halo2_amd
CSRC := $(PKG)/csrc
HIPFLAGS ?= -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function -ffp-contract=off
LIB := $(PKG)/libdehalo.so

all: $(LIB) oracle host_example

$(LIB): $(CSRC)/capi.hip $(CSRC)/fp.cuh $(CSRC)/ec.cuh $(CSRC)/ntt.cuh $(CSRC)/msm.cuh $(CSRC)/field_constants.h include/dehalo.h
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(CSRC)/capi.hip -Wl,-rpath,/opt/rocm/lib

oracle:
	$(MAKE) -C oracle liboracle.so

host_example: $(LIB) $(PKG)/host/halo2_backend.hpp $(PKG)/host/example.cpp
	g++ -std=c++17 -O1 -Wall -o $(PKG)/host/example $(PKG)/host/example.cpp -L$(PKG) -ldehalo -Wl,-rpath,'$$ORIGIN/..' -Wl,-rpath,/opt/rocm/lib

clean:
	rm -f $(LIB); $(MAKE) -C oracle clean
.PHONY: all oracle clean host_example
