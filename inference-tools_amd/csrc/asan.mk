# Included by Makefile when present (kept off the GPU boxes by .gpurunignore).
# Host-side sanitizer build (CPU box only - GPU AddressSanitizer is not available on the pool): AddressSanitizer +
# UndefinedBehaviorSanitizer on the host code of every translation unit, device code untouched (-fno-gpu-sanitize).
# tests/test_sanitizers_cpu.py loads it into a child interpreter (LD_PRELOAD of the sanitizer runtime) and drives
# what works without a device: argument validation of every entry point, gpmi_flow_task_lists, handle creation failing.
ASAN_OUT   := $(ROOT)/build_asan/libgpmi_asan.so
ASAN_FLAGS := --offload-arch=$(ARCH) -O1 -g -std=c++17 -fPIC -fsanitize=address,undefined -fno-gpu-sanitize -shared-libsan \
              -fno-omit-frame-pointer -fno-sanitize-recover=undefined $(INC)
ASAN_OBJS  := $(patsubst %.hip,$(ROOT)/build_asan/%.o,$(SRCS))

$(ROOT)/build_asan/%.o: $(ROOT)/%.hip $(ROOT)/gpmi_internal.h $(ROOT)/gemm_tiles.h $(ROOT)/kmath.h $(ROOT)/factor16_steps.h $(ROOT)/potrf_diag.h $(ROOT)/api_internal.h $(ROOT)/../../include/gpmi.h
	@mkdir -p $(ROOT)/build_asan
	$(HIPCC) $(ASAN_FLAGS) -c $< -o $@

$(ASAN_OUT): $(ASAN_OBJS) $(ROOT)/gpmi.map
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -fsanitize=address,undefined -shared-libsan -Wl,--version-script=$(ROOT)/gpmi.map -o $@ $(ASAN_OBJS) -ldl

asan: $(ASAN_OUT)

