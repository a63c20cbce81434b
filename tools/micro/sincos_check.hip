// sincos_check.hip -- is sincosf(x) bit-identical to (sinf(x), cosf(x)) on gfx950 for the RoPE angles (pos * inv_freq)?
// hipcc --offload-arch=gfx950 -O3 -o sincos_check sincos_check.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const float* freq, int nf, int npos, unsigned long long* diff) {
    const int pos = blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= npos) return;
    unsigned long long d = 0;
    for (int i = 0; i < nf; ++i) {
        const float a = pos * freq[i] * 1.0f;
        float s2, c2;
        sincosf(a, &s2, &c2);
        const float s1 = sinf(a), c1 = cosf(a);
        d += (__float_as_uint(s1) != __float_as_uint(s2)) + (__float_as_uint(c1) != __float_as_uint(c2));
    }
    if (d) atomicAdd(diff, d);
}
int main() {
    const int nf = 128, npos = 1 << 17;
    float h[nf];
    for (int i = 0; i < 64; ++i) h[i] = 1.0f / powf(500000.0f, (2.0f * i) / 128.0f);      // Llama-3 theta, head_dim 128
    for (int i = 0; i < 64; ++i) h[64 + i] = 1.0f / powf(10000.0f, (2.0f * i) / 128.0f);  // Llama-2 theta
    float* f; unsigned long long* d; unsigned long long hd = 0;
    hipMalloc(&f, sizeof(h)); hipMalloc(&d, 8); hipMemcpy(f, h, sizeof(h), hipMemcpyHostToDevice); hipMemset(d, 0, 8);
    hipLaunchKernelGGL(k, dim3(npos / 256), dim3(256), 0, 0, f, nf, npos, d);
    hipMemcpy(&hd, d, 8, hipMemcpyDeviceToHost);
    printf("sincosf vs sinf/cosf over %d positions x %d frequencies: %llu differing results\n", npos, nf, hd);
    return 0;
}
