// common.hip.h -- shared device helpers for the gfx950 (CDNA4, wave64) kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/ntk.h"

#define NTK_WAVE 64

#define NTK_HIP_TRY(expr)                                   \
    do {                                                    \
        hipError_t e__ = (expr);                            \
        if (e__ != hipSuccess) return NTK_E_LAUNCH;         \
    } while (0)

namespace ntk {

hipStream_t resolve_stream(void* s);   // runtime.cpp: NULL -> compute stream
int         last_launch_status();      // hipGetLastError -> NTK_OK / NTK_E_LAUNCH

// ---- fp16 <-> fp32: v_cvt_f32_f16 / v_cvt_f16_f32 (RNE, denormals preserved: hipcc default mode) ----
__device__ __forceinline__ float h2f(uint16_t bits) {
    _Float16 h;
    __builtin_memcpy(&h, &bits, 2);
    return (float)h;
}
__device__ __forceinline__ uint16_t f2h(float f) {
    _Float16 h = (_Float16)f;
    uint16_t bits;
    __builtin_memcpy(&bits, &h, 2);
    return bits;
}

// ---- wave64 reductions: every lane ends with the full result ---------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// Block-wide sum / max through LDS scratch `red` (>= 16 floats). All threads must call; result to all.
__device__ __forceinline__ float block_sum(float v, float* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    __syncthreads();  // protect `red` against a previous use
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float t = 0.0f;
    for (int i = 0; i < nw; ++i) t += red[i];  // fixed order: deterministic
    return t;
}
__device__ __forceinline__ float block_max(float v, float* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_max(v);
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float t = red[0];
    for (int i = 1; i < nw; ++i) t = fmaxf(t, red[i]);
    return t;
}

// ---- byte-granular LDS reads: the staged weight stream keeps GGUF's 2-byte alignment -----------------
// 4 bytes at any even offset: two aligned dwords + v_alignbyte_b32
__device__ __forceinline__ uint32_t lds_u32_at(const uint8_t* base, int off) {
    const int a = off & ~3;
    const uint32_t lo = *reinterpret_cast<const uint32_t*>(base + a);
    const uint32_t hi = *reinterpret_cast<const uint32_t*>(base + a + 4);
    return __builtin_amdgcn_alignbyte(hi, lo, (uint32_t)(off & 3));
}
// N consecutive dwords starting at any even offset (N+1 aligned reads)
template <int N>
__device__ __forceinline__ void lds_read_dwords(uint32_t (&dst)[N], const uint8_t* base, int off) {
    const int a = off & ~3;
    const uint32_t sh = (uint32_t)(off & 3);
    uint32_t w[N + 1];
#pragma unroll
    for (int i = 0; i <= N; ++i) w[i] = *reinterpret_cast<const uint32_t*>(base + a + 4 * i);
#pragma unroll
    for (int i = 0; i < N; ++i) dst[i] = __builtin_amdgcn_alignbyte(w[i + 1], w[i], sh);
}
__device__ __forceinline__ uint16_t lds_u16_at(const uint8_t* base, int off) {
    return *reinterpret_cast<const uint16_t*>(base + off);
}

// byte k (0..3) of a dword as float: unsigned / signed
__device__ __forceinline__ float ub2f(uint32_t w, int k) { return (float)((w >> (8 * k)) & 0xFFu); }
__device__ __forceinline__ float sb2f(uint32_t w, int k) { return (float)(int)(int8_t)(w >> (8 * k)); }

// K-quant 6-bit (scale, min) pair j of the 12 packed bytes held in three dwords (types.h:112-117)
__device__ __forceinline__ void kq_scale_min(uint32_t s0, uint32_t s1, uint32_t s2, int j, float& sc, float& mn) {
    const uint64_t lo = (uint64_t)s0 | ((uint64_t)s1 << 32);  // bytes 0..7
    auto byte = [&](int n) -> uint32_t {
        return n < 8 ? (uint32_t)((lo >> (8 * n)) & 0xFFu) : ((s2 >> (8 * (n - 8))) & 0xFFu);
    };
    uint32_t s, m;
    if (j < 4) {
        s = byte(j) & 63u;
        m = byte(j + 4) & 63u;
    } else {
        s = (byte(j + 4) & 0xFu) | ((byte(j - 4) >> 6) << 4);
        m = (byte(j + 4) >> 4) | ((byte(j) >> 6) << 4);
    }
    sc = (float)s;
    mn = (float)m;
}

}  // namespace ntk
