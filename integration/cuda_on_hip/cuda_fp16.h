// integration/cuda_on_hip/cuda_fp16.h -- REFERENCE-SIDE BINDING: the reference's host code only needs the NAME `half`
// as a 16-bit storage type (src/core/types.h:12-17; all half arithmetic lives in the kernels this build replaces).
#pragma once
#include <cstdint>
struct half { uint16_t bits; };
