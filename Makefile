# Builds dedflow_amd/libdedflow.so: hand-written gfx950 HIP kernels (csrc/) + the
# C host layer behind the reference's object API (host/).  In-tree artefact: the .so
# is git-ignored but travels to the GPU box with the gpurun snapshot.
HIPCC   ?= /opt/rocm/bin/hipcc
CC      ?= gcc
ARCH    ?= gfx950
ROCM    ?= /opt/rocm
HIPFLAGS = -O3 --offload-arch=$(ARCH) -fPIC -std=c++17 -Iinclude
CFLAGS   = -O2 -std=gnu99 -fPIC -fopenmp -Wall -Wno-unused-function -D__HIP_PLATFORM_AMD__ -Iinclude -I$(ROCM)/include -Idedflow_amd/host

KSRC = $(wildcard dedflow_amd/csrc/*.hip)
HSRC = $(wildcard dedflow_amd/host/*.c)
KOBJ = $(KSRC:.hip=.o)
HOBJ = $(HSRC:.c=.o)
LIB  = dedflow_amd/libdedflow.so

H5LIB  = dedflow_amd/libdedflow_h5.so
HDF5   ?= /opt/conda

all: $(LIB) $(H5LIB)

# HDF5 mesh / solution formats (reference schema) in their own library: the core library has no
# HDF5 dependency.  HDF5 1.10.6 C library of the image.
$(H5LIB): dedflow_amd/h5/h5io.c include/dedflow.h $(LIB)
	@if [ -f $(HDF5)/include/hdf5.h ]; then \
	  $(CC) $(CFLAGS) -I$(HDF5)/include -shared $< -o $@ -Ldedflow_amd -ldedflow -L$(HDF5)/lib -lhdf5 \
	    -Wl,-rpath,$(HDF5)/lib -Wl,-rpath,'$$ORIGIN'; \
	else echo "HDF5 headers not found under $(HDF5): skipping $@"; fi

dedflow_amd/csrc/%.o: dedflow_amd/csrc/%.hip dedflow_amd/csrc/dfl_common.hpp dedflow_amd/csrc/asm_device.hpp include/dedflow_kernels.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

dedflow_amd/host/%.o: dedflow_amd/host/%.c include/dedflow.h include/dedflow_kernels.h dedflow_amd/host/host_private.h dedflow_amd/host/solver_private.h dedflow_amd/host/rcb.h
	$(CC) $(CFLAGS) -c $< -o $@

$(LIB): $(KOBJ) $(HOBJ)
	$(HIPCC) -shared -fPIC -fopenmp --offload-arch=$(ARCH) -o $@ $^

oracle:
	$(MAKE) -C oracle

# CPU sanitizer build: the C host layer under AddressSanitizer + UBSan, linked with the same device objects (GPU ASan is
# not available on this pool; sanitizers run on the CPU-only tests: tools/run_cpu_sanitized.sh)
SANLIB  = dedflow_amd/libdedflow_asan.so
SANOBJ  = $(HSRC:.c=.san.o)
dedflow_amd/host/%.san.o: dedflow_amd/host/%.c include/dedflow.h include/dedflow_kernels.h dedflow_amd/host/host_private.h dedflow_amd/host/solver_private.h dedflow_amd/host/rcb.h
	$(CC) $(CFLAGS) -O1 -g -fno-omit-frame-pointer -fsanitize=address,undefined -c $< -o $@
$(SANLIB): $(KOBJ) $(SANOBJ)
	$(HIPCC) -shared -fPIC -fopenmp --offload-arch=$(ARCH) -fsanitize=address,undefined -o $@ $^ || \
	$(CC) -shared -fPIC -fopenmp -fsanitize=address,undefined -o $@ $^ -L$(ROCM)/lib -lamdhip64 -lstdc++ -Wl,-rpath,$(ROCM)/lib
asan: $(SANLIB)
	$(MAKE) -C oracle asan

clean:
	rm -f $(KOBJ) $(HOBJ) $(LIB) $(SANOBJ) $(SANLIB)

.PHONY: all oracle clean asan
