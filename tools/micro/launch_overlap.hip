// launch_overlap.hip -- can consecutive dependent streaming kernels overlap (next kernel's launch + first weight fetch
// behind the previous kernel's tail) if the dependency is carried by an in-kernel arrival counter instead of the stream?
// (tuning aid for DESIGN.md "Next" item 1; not part of the product)
//   chain: kernel i reads the 4096-float vector x_i, streams its matrix, writes x_{i+1}[r] = f(row r, x_i) for r < 4096.
//   serial : one stream, ordinary dependent launches.
//   overlap: kernels alternate between two streams (i on stream i%2: i+2 still follows i in stream order); kernel i
//            prefetches its first row, then waits until counter[i-1] == workgroups(i-1), reads x_i with sc1 loads.
//            Producers store x with sc1 (write-through) stores, drain them, then bump the counter (agent scope).
// hipcc --offload-arch=gfx950 -O3 -o launch_overlap launch_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float ld_sc1(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// rows x row16*16 bytes; every wave owns rows wave, wave + nwaves, ...; y[r % 4096] gets a value depending on row bytes and x
template <bool FLAGS>
__global__ __launch_bounds__(512) void k_layer(const u32x4* __restrict__ W, const float* x, float* y, int rows, int row16,
                                               unsigned* wait_ctr, unsigned wait_target, unsigned* done_ctr, int* err, unsigned* started_ctr) {
    __shared__ float xs[4096];
    if (FLAGS && started_ctr && threadIdx.x == 0) __hip_atomic_fetch_add(started_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int lane = threadIdx.x & 63, wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), nw = gridDim.x * (blockDim.x >> 6);
    // first row in flight before anything else
    u32x4 pf[5];
    int r = wave;
    const int nl = (row16 + 63) / 64;
    if (r < rows) for (int j = 0; j < 5; ++j) if (j < nl) pf[j] = __builtin_nontemporal_load(W + (size_t)r * row16 + min(lane + 64 * j, row16 - 1));
    if (FLAGS && wait_ctr) {
        if (threadIdx.x == 0) {
            int it = 0;
            while (__hip_atomic_load(wait_ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < wait_target) {
                __builtin_amdgcn_s_sleep(2);
                if (++it > (1 << 16)) { *err = 1; break; }   // bounded: never hang the GPU
            }
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) xs[i] = FLAGS ? ld_sc1(x + i) : x[i];
    __syncthreads();
    for (; r < rows; r += nw) {
        unsigned s = 0;
        for (int j = 0; j < 5; ++j) if (j < nl) s += pf[j].x ^ pf[j].y ^ pf[j].z ^ pf[j].w;
        const int rn = r + nw;
        if (rn < rows) for (int j = 0; j < 5; ++j) if (j < nl) pf[j] = __builtin_nontemporal_load(W + (size_t)rn * row16 + min(lane + 64 * j, row16 - 1));
        float acc = (float)(s & 0xFF) * xs[(r + lane) & 4095];
        for (int o = 32; o; o >>= 1) acc += __shfl_xor(acc, o, 64);
        if (lane == 0 && r < 4096) { const float v = acc * 1e-4f + 1.0f; if (FLAGS) st_sc1(y + r, v); else y[r] = v; }
    }
    if (FLAGS && done_ctr) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(done_ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// head of the second stream: lets its first kernel become eligible only once kernel 0 is fully resident (otherwise both
// start together and the waiting kernel can take every workgroup slot)
__global__ void k_gate(unsigned* started_ctr, unsigned target, int* err) {
    int it = 0;
    while (__hip_atomic_load(started_ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(2);
        if (++it > (1 << 16)) { *err = 2; break; }
    }
}

int main() {
    hipStream_t s0, s1; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    const size_t WBYTES = (size_t)3 << 30;
    u32x4* W; float* xbuf; unsigned* ctr; int* err;
    CK(hipMalloc(&W, WBYTES)); CK(hipMalloc(&xbuf, 2 * 4096 * 4)); CK(hipMalloc(&ctr, 4096)); CK(hipMalloc(&err, 4));
    CK(hipMemset(W, 3, WBYTES)); CK(hipMemset(err, 0, 4));
    std::vector<float> ones(4096, 1.0f);
    struct M { int rows, rb; } layer[5] = {{6144, 4352}, {4096, 4352}, {28672, 4352}, {4096, 15232}, {1024, 4352}};   // qkv, o, gate|up, down, (kv-size filler)
    const int L = 4, NLAY = 12, chain = L * NLAY, grid = 512;
    size_t off[64]; size_t o = 0;
    for (int i = 0; i < chain; ++i) { off[i] = o; o += (size_t)layer[i % L].rows * layer[i % L].rb; }
    if (o > WBYTES) { printf("pool too small\n"); return 1; }
    size_t total_bytes = o;
    for (int mode = 0; mode < 2; ++mode) {
        // eager launches (a captured two-stream graph may be linearised onto one queue, where a kernel waiting for a LATER
        // kernel's counter never finishes): two real streams, the host enqueues faster than the GPU drains
        hipEvent_t a, b, fork, join; CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); CK(hipEventCreate(&fork)); CK(hipEventCreate(&join));
        float best = 1e9f; double sum = 0;
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipMemcpy(xbuf, ones.data(), 4096 * 4, hipMemcpyHostToDevice));
            CK(hipMemset(ctr, 0, 4096));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(a, s0));
            if (mode == 1) { CK(hipEventRecord(fork, s0)); CK(hipStreamWaitEvent(s1, fork, 0)); }
            for (int i = 0; i < chain; ++i) {
                const M m = layer[i % L];
                const float* xin = xbuf + (i & 1) * 4096; float* yout = xbuf + ((i + 1) & 1) * 4096;
                hipStream_t st = mode == 1 ? ((i & 1) ? s1 : s0) : s0;
                if (mode == 0) hipLaunchKernelGGL(k_layer<false>, dim3(grid), dim3(512), 0, st, W + off[i] / 16, xin, yout, m.rows, m.rb / 16, nullptr, 0u, nullptr, err, nullptr);
                else {
                    if (i == 1) hipLaunchKernelGGL(k_gate, dim3(1), dim3(1), 0, s1, ctr + 512, (unsigned)grid, err);
                    hipLaunchKernelGGL(k_layer<true>, dim3(grid), dim3(512), 0, st, W + off[i] / 16, xin, yout, m.rows, m.rb / 16, i ? ctr + (i - 1) : nullptr, (unsigned)grid, ctr + i, err, i == 0 ? ctr + 512 : nullptr);
                }
            }
            if (mode == 1) { CK(hipEventRecord(join, s1)); CK(hipStreamWaitEvent(s0, join, 0)); }
            CK(hipEventRecord(b, s0)); CK(hipStreamSynchronize(s0)); CK(hipStreamSynchronize(s1));
            float ms = 0; CK(hipEventElapsedTime(&ms, a, b)); if (rep) best = ms < best ? ms : best;
            std::vector<float> out(4096); CK(hipMemcpy(out.data(), xbuf + (chain & 1) * 4096, 4096 * 4, hipMemcpyDeviceToHost));
            sum = 0; for (int k = 0; k < 4096; ++k) sum += out[k] * (1 + (k % 7));
        }
        int herr = 0; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
        printf("%-8s chain of %d launches (12 x qkv,o,gate|up,down; %.1f MB): %8.1f us = %6.2f us/launch, %6.1f GB/s, checksum %.6f, spin-timeout %d\n",
               mode ? "overlap" : "serial", chain, total_bytes / 1e6, best * 1e3, best * 1e3 / chain, total_bytes / best / 1e6, sum, herr);
    }
    return 0;
}
