// launch_floor.hip -- what does a dependent chain of short kernels cost on MI355X, by launch geometry?
// (tuning aid; not part of the product).  hipcc --offload-arch=gfx950 -O3 -o launch_floor launch_floor.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

extern __shared__ float smem[];
__global__ void k_empty(float* y) { if (threadIdx.x == 1023 && blockIdx.x == 99999) y[0] = smem[0]; }
// every workgroup reads the same 16 KB vector (the GEMV activation), one float4 per thread for nx4 passes, then one store
__global__ void k_readx(const float* __restrict__ x, float* __restrict__ y, int n4) {
    float acc = 0.f;
    for (int i = threadIdx.x; i < n4; i += blockDim.x) { float4 v = reinterpret_cast<const float4*>(x)[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 123.456f) y[blockIdx.x] = acc;
}
// stream: every wave reads `rows` rows of `row_bytes` (16 B per lane per load), plus the x read when n4 > 0, writes 1 float per row
__global__ void k_stream(const u32x4* __restrict__ W, const float* __restrict__ x, float* __restrict__ y, int n4, int rows_total, int row16) {
    float acc = 0.f;
    for (int i = threadIdx.x; i < n4; i += blockDim.x) { float4 v = reinterpret_cast<const float4*>(x)[i]; acc += v.x; }
    const int lane = threadIdx.x & 63, wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), nw = gridDim.x * (blockDim.x >> 6);
    for (int r = wave; r < rows_total; r += nw) {
        unsigned s = 0;
        for (int j = lane; j < row16; j += 64) { u32x4 v = __builtin_nontemporal_load(W + (size_t)r * row16 + j); s += v.x ^ v.y ^ v.z ^ v.w; }
        for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o, 64);
        if (lane == 0) y[r] = acc + (float)s;
    }
}

template <class F> static float time_chain(hipStream_t st, int chain, int reps, F launch) {
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    for (int i = 0; i < chain; ++i) launch(i);
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipGraphLaunch(ge, st);
    hipStreamSynchronize(st);
    hipEventRecord(a, st);
    for (int i = 0; i < reps; ++i) hipGraphLaunch(ge, st);
    hipEventRecord(b, st);
    hipStreamSynchronize(st);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
    return ms * 1e3f / (reps * chain);
}

int main() {
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const size_t WBYTES = (size_t)2 << 30;   // 2 GiB pool: successive launches touch different matrices (nothing cache resident)
    u32x4* W; float *x, *y;
    CK(hipMalloc(&W, WBYTES)); CK(hipMalloc(&x, 1 << 20)); CK(hipMalloc(&y, 4 << 20));
    CK(hipMemset(W, 1, WBYTES)); CK(hipMemset(x, 0, 1 << 20));
    CK(hipFuncSetAttribute((const void*)k_empty, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int chain = 64, reps = 20;
    printf("%-58s %8s\n", "dependent chain of 64 launches, hipGraph replay", "us/launch");
    struct G { int grid, thr, lds; } geos[] = {{512, 512, 0}, {512, 512, 42 * 1024}, {256, 512, 42 * 1024}, {256, 1024, 42 * 1024}, {1024, 256, 21 * 1024},
                                              {2048, 128, 0}, {256, 256, 0}, {64, 256, 0}, {1, 64, 0}};
    for (auto g : geos) {
        float t = time_chain(st, chain, reps, [&](int) { hipLaunchKernelGGL(k_empty, dim3(g.grid), dim3(g.thr), g.lds, st, y); });
        printf("empty       grid %5d x %4d thr, lds %6d B             %8.2f\n", g.grid, g.thr, g.lds, t);
    }
    for (auto g : geos) {
        if (g.grid < 64) continue;
        float t = time_chain(st, chain, reps, [&](int) { hipLaunchKernelGGL(k_readx, dim3(g.grid), dim3(g.thr), g.lds, st, x, y, 1024); });
        printf("read x 16KB grid %5d x %4d thr, lds %6d B             %8.2f\n", g.grid, g.thr, g.lds, t);
    }
    struct M { const char* name; int rows, row_bytes; } mats[] = {{"kv 1024x4352", 1024, 4352}, {"o 4096x4352", 4096, 4352}, {"qkv 6144x4352", 6144, 4352},
                                                                   {"down 4096x15232", 4096, 15232}, {"gate|up 28672x4352", 28672, 4352}};
    for (auto m : mats) {
        const size_t mb = (size_t)m.rows * m.row_bytes;
        const int nmat = (int)(WBYTES / mb);
        for (auto g : {G{512, 512, 42 * 1024}, G{256, 1024, 42 * 1024}, G{1024, 256, 21 * 1024}, G{2048, 256, 0}}) {
            for (int withx = 0; withx < 2; ++withx) {
                float t = time_chain(st, chain, reps, [&](int i) {
                    hipLaunchKernelGGL(k_stream, dim3(g.grid), dim3(g.thr), g.lds, st, W + (size_t)(i % nmat) * (mb / 16), x, y, withx ? 1024 : 0, m.rows, m.row_bytes / 16);
                });
                printf("stream %-20s grid %5d x %4d lds %6d x=%d  %8.2f us  %7.1f GB/s\n", m.name, g.grid, g.thr, g.lds, withx, t, mb / t * 1e-3);
            }
        }
    }
    return 0;
}
