# Builds dedflow_amd/libdedflow.so: hand-written gfx950 HIP kernels (csrc/) + the
# C host layer behind the reference's object API (host/).  In-tree artefact: the .so
# is git-ignored but travels to the GPU box with the gpurun snapshot.
HIPCC   ?= /opt/rocm/bin/hipcc
CC      ?= gcc
ARCH    ?= gfx950
ROCM    ?= /opt/rocm
HIPFLAGS = -O3 --offload-arch=$(ARCH) -fPIC -std=c++17 -Iinclude
CFLAGS   = -O2 -std=gnu99 -fPIC -Wall -Wno-unused-function -D__HIP_PLATFORM_AMD__ -Iinclude -I$(ROCM)/include -Idedflow_amd/host

KSRC = $(wildcard dedflow_amd/csrc/*.hip)
HSRC = $(wildcard dedflow_amd/host/*.c)
KOBJ = $(KSRC:.hip=.o)
HOBJ = $(HSRC:.c=.o)
LIB  = dedflow_amd/libdedflow.so

all: $(LIB)

dedflow_amd/csrc/%.o: dedflow_amd/csrc/%.hip dedflow_amd/csrc/dfl_common.hpp include/dedflow_kernels.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

dedflow_amd/host/%.o: dedflow_amd/host/%.c include/dedflow.h include/dedflow_kernels.h dedflow_amd/host/host_private.h
	$(CC) $(CFLAGS) -c $< -o $@

$(LIB): $(KOBJ) $(HOBJ)
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) -o $@ $^

oracle:
	$(MAKE) -C oracle

clean:
	rm -f $(KOBJ) $(HOBJ) $(LIB)

.PHONY: all oracle clean
