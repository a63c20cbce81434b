// valu_rate.hip -- issue rate of the VALU instructions a K-quant decoder could be built from (MI355X, one wave per SIMD x 4 ...).
// hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int OP>
__global__ void rate_kernel(int* out, int iters, int seed) {
    int a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    const int b = 0x01020304 + seed, c = 0x7f7f7f7f ^ seed;
    for (int i = 0; i < iters; ++i) {
#define STEP(x)                                                                                                         \
        if (OP == 0) asm volatile("v_dot4_i32_i8 %0, %1, %2, %0" : "+v"(x) : "v"(b), "v"(c));                            \
        else if (OP == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(*(long long*)&x##p) : "v"(bb), "v"(cc));     \
        else if (OP == 2) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(x) : "v"(b));                                    \
        else if (OP == 3) asm volatile("v_and_b32 %0, %1, %0" : "+v"(x) : "v"(c));                                       \
        else if (OP == 4) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(x) : "v"(b), "v"(c));                       \
        else if (OP == 5) asm volatile("v_perm_b32 %0, %1, %0, %2" : "+v"(x) : "v"(b), "v"(c));
        long long a0p = a0, a1p = a1, a2p = a2, a3p = a3, a4p = a4, a5p = a5, a6p = a6, a7p = a7;
        const long long bb = ((long long)b << 32) | (unsigned)c, cc = bb ^ 0x0101010101010101ll;
        STEP(a0) STEP(a1) STEP(a2) STEP(a3) STEP(a4) STEP(a5) STEP(a6) STEP(a7)
        if (OP == 1) { a0 += (int)a0p; a1 += (int)a1p; a2 += (int)a2p; a3 += (int)a3p; a4 += (int)a4p; a5 += (int)a5p; a6 += (int)a6p; a7 += (int)a7p; }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int OP>
int run(const char* name, int* d) {
    const int iters = 20000, grid = 256 * 8, block = 256;   // 8 workgroups of 4 waves per CU: 8 waves per SIMD
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(rate_kernel<OP>, dim3(grid), dim3(block), 0, 0, d, 100, 1);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(rate_kernel<OP>, dim3(grid), dim3(block), 0, 0, d, iters, 1);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double insts = (double)grid * (block / 64) * iters * 8;          // wave instructions
    const double per_simd_cycle = insts / (1024.0 * ms * 1e-3 * 2.4e9);     // 1024 SIMDs at 2.4 GHz
    printf("%-22s %8.3f ms  %.3f wave-instructions per SIMD cycle (1 / %.2f cycles)\n", name, ms, per_simd_cycle, 1.0 / per_simd_cycle);
    return 0;
}

int main() {
    int* d;
    CHECK(hipMalloc(&d, 256 * 8 * 256 * sizeof(int)));
    run<0>("v_dot4_i32_i8", d);
    run<4>("v_dot4_u32_u8", d);
    run<1>("v_pk_fma_f32", d);
    run<2>("v_cvt_f32_ubyte1", d);
    run<3>("v_and_b32", d);
    run<5>("v_perm_b32", d);
    return 0;
}
