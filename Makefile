# Builds the gfx950 shared library in-tree (the .so travels to the GPU box with the snapshot).
HIPCC    ?= /opt/rocm/bin/hipcc
ARCH     ?= gfx950
CSRC     := endodav_amd/csrc
OUT      := endodav_amd/lib/libendodav_hip.so
SRCS     := $(CSRC)/gemm.hip $(CSRC)/gemm_dma.hip $(CSRC)/gemm_x6.hip $(CSRC)/conv_dma.hip $(CSRC)/attn_spatial.hip $(CSRC)/attn_spatial_bwd.hip $(CSRC)/norms.hip $(CSRC)/temporal.hip \
            $(CSRC)/resample.hip $(CSRC)/prep.hip $(CSRC)/bwd.hip $(CSRC)/wgrad.hip $(CSRC)/loss.hip $(CSRC)/loss_trainer.hip $(CSRC)/engine.hip $(CSRC)/api.hip
OBJS     := $(SRCS:$(CSRC)/%.hip=build/%.o)
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++20 -fPIC -Wall -Wno-unused-function -fno-gpu-rdc

all: $(OUT)

# attention forward: no NaN can occur (scores are finite or -inf, never inf - inf), and without the flag every MFMA result entering an fmaxf is
# first canonicalised by an extra v_max_f32 (VALU instructions cost matrix-pipe time)
build/attn_spatial.o: HIPFLAGS += -fno-honor-nans

build/%.o: $(CSRC)/%.hip $(CSRC)/ops.hpp $(CSRC)/common.hpp $(CSRC)/gemm_common.hpp $(CSRC)/loss_common.hpp include/endodav_hip.h
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(OUT): $(OBJS)
	@mkdir -p endodav_amd/lib
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)

clean:
	rm -rf build $(OUT)
.PHONY: all clean
