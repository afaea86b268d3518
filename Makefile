# Top-level build: the product library (HIP, gfx950), the C host tools, and the checker.
#   make lib     -> som_lvq_pak_amd/libsomhip.so   (hipcc cross-compiles without a GPU)
#   make tools   -> som_lvq_pak_amd/host/bin/{vsom,lvqtrain,qerror,accuracy,vcal,...}
#   make oracle  -> oracle/liboracle.so (+ oracle/_ref when /root/reference is present)
HIPCC    ?= /opt/rocm/bin/hipcc
ARCH     ?= gfx950
# -fno-slp-vectorize: hipcc otherwise packs the scan's scalar fp32 ops into v_pk_* with
# SGPR shuffles; measured 17 % slower on MI355X (profiles/r01_notes.md)
EXTRA    ?=
HIPFLAGS  = --offload-arch=$(ARCH) -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -Wall -Wno-unused-result $(EXTRA)
CSRC      = som_lvq_pak_amd/csrc
LIB       = som_lvq_pak_amd/libsomhip.so

.PHONY: all lib tools oracle clean isa
all: lib oracle tools

lib: $(LIB)
KHDR      = $(CSRC)/kernels.hpp $(wildcard $(CSRC)/kernels/*.hpp) $(wildcard $(CSRC)/host_*.inc)
# librccl.so is opened at run time by the first somhip_comm_create (host_comm.inc), never linked: -ldl only
$(LIB): $(CSRC)/somhip.hip $(KHDR) $(CSRC)/schedule.hpp include/somhip.h Makefile
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(CSRC)/somhip.hip -ldl

# device ISA of the kernels, for the no-FMA check (tests/test_build.py) and for reading
isa: $(CSRC)/somhip.hip $(KHDR)
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) --cuda-device-only -S -o build/somhip_gfx950.s $(CSRC)/somhip.hip \
	    -Rpass-analysis=kernel-resource-usage 2> build/resource_usage.txt || true

tools:
	@if [ -f som_lvq_pak_amd/host/Makefile ]; then $(MAKE) -s -C som_lvq_pak_amd/host; fi

oracle:
	$(MAKE) -s -C oracle oracle
	$(MAKE) -s -C oracle -j4 ref

clean:
	rm -rf build $(LIB)
	$(MAKE) -s -C oracle clean
