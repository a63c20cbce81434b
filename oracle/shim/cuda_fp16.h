// oracle/shim/cuda_fp16.h -- TEST INFRASTRUCTURE ONLY.  Host stand-in so that the reference's two
// host-only ".cu" files (core/device.cu, memory/streamer.cu) compile with -D__CUDACC__ under g++.
// Only the storage type is needed: `using float16_t = half` (reference src/core/types.h:12-17).
#pragma once
#include <cstdint>
struct half { uint16_t bits; };
static_assert(sizeof(half) == 2, "half must be 2 bytes");
