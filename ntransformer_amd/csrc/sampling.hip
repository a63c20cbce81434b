// sampling.hip -- the reference's token sampler on the device (second half).
//
// Replaces, for temperature > 0 and 0 < top_k <= 64, the per-token host work of the reference's decode loop
// (reference src/inference/engine.cpp:113-119: a blocking 513 KB logits download, then Sampler::apply_repeat_penalty and
// Sampler::sample, reference src/inference/sampler.cpp:30-117, whose std::partial_sort walks all 128 256 candidates).
// Same arithmetic, step for step: the repeat penalty (one application per occurrence in the window, in order), logit /
// temperature, the top-k candidates in descending order, expf(l - max) summed and normalised SEQUENTIALLY in that order by
// one thread (the reference's float summation order), the top-p cut and renormalisation, and the walk of the cumulative
// distribution against a uniform draw.  The draw itself stays on the host (`r`, from the engine's std::mt19937, one per
// token exactly as Sampler::sample consumes it), so the sampled stream is the reference's for the same seed.
//
// Top-k of 128 256 logits: workgroup b sorts its 2048-element chunk in LDS (bitonic, 64-bit keys = orderable value bits :
// inverted index, so ties resolve to the lower token id) and emits its best 64; one workgroup then sorts the
// <= 64 x 63 survivors.  Two launches (+ the penalty), ~0.5 MB of L2-resident reads.
#include "common.hip.h"
#include <cfloat>

namespace ntk {

constexpr int SK_CHUNK = 2048;    // logits per first-stage workgroup
constexpr int SK_KEEP = 64;       // survivors per chunk = the largest supported top_k
constexpr int SK_MAXCAND = 4096;  // second stage sorts up to this many survivors (64 chunks -> vocabularies up to 131 072)

__device__ __forceinline__ unsigned long long sk_key(float v, int idx) {   // larger key = larger value, then smaller index
    uint32_t b = __float_as_uint(v);
    b = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
    return ((unsigned long long)b << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)idx);
}
__device__ __forceinline__ float sk_value(unsigned long long k) {
    uint32_t b = (uint32_t)(k >> 32);
    b = (b & 0x80000000u) ? (b & 0x7FFFFFFFu) : ~b;
    return __uint_as_float(b);
}
__device__ __forceinline__ int sk_index(unsigned long long k) { return (int)(0xFFFFFFFFu - (uint32_t)k); }

// descending bitonic sort of n (power of two) keys in LDS by the whole workgroup
__device__ void sk_sort_desc(unsigned long long* keys, int n) {
    for (int size = 2; size <= n; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int t = threadIdx.x; t < n / 2; t += blockDim.x) {
                const int lo = 2 * t - (t & (stride - 1));      // index with bit `stride` clear
                const int hi = lo + stride;
                const bool desc = (lo & size) == 0;
                const unsigned long long a = keys[lo], b = keys[hi];
                if ((a < b) == desc) { keys[lo] = b; keys[hi] = a; }
            }
        }
    }
    __syncthreads();
}

// reference sampler.cpp:30-45, one thread: the window is walked in order, a token that occurs twice is penalised twice
__global__ void sample_penalty_kernel(float* __restrict__ logits, int n, const int* __restrict__ recent, int n_recent, float penalty) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    for (int i = 0; i < n_recent; ++i) {
        const int t = recent[i];
        if (t < 0 || t >= n) continue;
        const float v = logits[t];
        logits[t] = v > 0.0f ? v / penalty : v * penalty;
    }
}

__global__ __launch_bounds__(1024) void sample_topk_stage1(const float* __restrict__ logits, int n, float temperature,
                                                           unsigned long long* __restrict__ cand) {
    __shared__ unsigned long long keys[SK_CHUNK];
    const int base = blockIdx.x * SK_CHUNK;
    for (int i = threadIdx.x; i < SK_CHUNK; i += blockDim.x) {
        const int idx = base + i;
        // candidates_[i] = {logits[i] / temperature, i}  (sampler.cpp:56-58); padding sorts last
        keys[i] = idx < n ? sk_key(logits[idx] / temperature, idx) : 0ull;
    }
    sk_sort_desc(keys, SK_CHUNK);
    for (int i = threadIdx.x; i < SK_KEEP; i += blockDim.x) cand[(size_t)blockIdx.x * SK_KEEP + i] = keys[i];
}

__global__ __launch_bounds__(1024) void sample_topk_stage2(const unsigned long long* __restrict__ cand, int n_cand, int top_k,
                                                           float top_p, float r, int* __restrict__ d_out, int* __restrict__ h_mirror) {
    __shared__ unsigned long long keys[SK_MAXCAND];
    int npad = 64;
    while (npad < n_cand) npad <<= 1;
    for (int i = threadIdx.x; i < npad; i += blockDim.x) keys[i] = i < n_cand ? cand[i] : 0ull;
    sk_sort_desc(keys, npad);
    if (threadIdx.x != 0) return;
    // sampler.cpp:73-116 on the k best, one thread, the reference's order of float operations
    int k = top_k;
    while (k > 0 && keys[k - 1] == 0ull) --k;   // vocabulary smaller than top_k
    float p[SK_KEEP];
    const float mx = sk_value(keys[0]);
    float sum = 0.0f;
    for (int i = 0; i < k; ++i) {
        // expf correctly rounded (evaluated in double, rounded once): what the host's libm returns for the reference
        p[i] = (float)exp((double)(sk_value(keys[i]) - mx));
        sum += p[i];
    }
    for (int i = 0; i < k; ++i) p[i] /= sum;
    if (top_p < 1.0f && top_p > 0.0f) {
        float cum = 0.0f;
        int cutoff = k;
        for (int i = 0; i < k; ++i) {
            cum += p[i];
            if (cum >= top_p) { cutoff = i + 1; break; }
        }
        k = cutoff;
        sum = 0.0f;
        for (int i = 0; i < k; ++i) sum += p[i];
        for (int i = 0; i < k; ++i) p[i] /= sum;
    }
    int pick = sk_index(keys[k - 1]);   // fallback (sampler.cpp:115)
    float cum = 0.0f;
    for (int i = 0; i < k; ++i) {
        cum += p[i];
        if (r <= cum) { pick = sk_index(keys[i]); break; }
    }
    *d_out = pick;
    if (h_mirror) *h_mirror = pick;
}

}  // namespace ntk

extern "C" {

using namespace ntk;

size_t ntk_sample_scratch_bytes(int n) {
    const int chunks = (n + SK_CHUNK - 1) / SK_CHUNK;
    return (size_t)chunks * SK_KEEP * sizeof(unsigned long long) + 256;
}

int ntk_sample_top_k(float* logits, int n, const int* d_recent, int n_recent, float repeat_penalty, float temperature, int top_k,
                     float top_p, float r, int* d_out_token, int* h_mirror, void* scratch, void* stream) {
    if (!logits || !d_out_token || !scratch) return NTK_E_NULL;
    if (n <= 0 || n_recent < 0 || (n_recent > 0 && !d_recent)) return NTK_E_SHAPE;
    if (!(temperature > 0.0f) || top_k <= 0 || top_k > SK_KEEP) return NTK_E_SHAPE;   // greedy: ntk_argmax; wider top-k: host sampler
    const int chunks = (n + SK_CHUNK - 1) / SK_CHUNK;
    if (chunks * SK_KEEP > SK_MAXCAND) return NTK_E_SHAPE;                             // vocabularies beyond 131 072
    hipStream_t st = resolve_stream(stream);
    if (repeat_penalty > 1.0f && n_recent > 0)
        hipLaunchKernelGGL(sample_penalty_kernel, dim3(1), dim3(64), 0, st, logits, n, d_recent, n_recent, repeat_penalty);
    unsigned long long* cand = static_cast<unsigned long long*>(scratch);
    hipLaunchKernelGGL(sample_topk_stage1, dim3(chunks), dim3(1024), 0, st, (const float*)logits, n, temperature, cand);
    hipLaunchKernelGGL(sample_topk_stage2, dim3(1), dim3(1024), 0, st, (const unsigned long long*)cand, chunks * SK_KEEP, top_k < n ? top_k : n,
                       top_p, r, d_out_token, h_mirror);
    return last_launch_status();
}

// the repeat penalty alone (greedy decoding with a penalty: penalty, then ntk_argmax)
int ntk_repeat_penalty(float* logits, int n, const int* d_recent, int n_recent, float repeat_penalty, void* stream) {
    if (!logits || (n_recent > 0 && !d_recent)) return NTK_E_NULL;
    if (n <= 0 || n_recent < 0) return NTK_E_SHAPE;
    if (repeat_penalty > 1.0f && n_recent > 0)
        hipLaunchKernelGGL(sample_penalty_kernel, dim3(1), dim3(64), 0, resolve_stream(stream), logits, n, d_recent, n_recent, repeat_penalty);
    return last_launch_status();
}

}  // extern "C"
