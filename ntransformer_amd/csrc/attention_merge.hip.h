// attention_merge.hip.h -- the split-KV decode attention as ONE launch: the workgroup that finishes LAST among the `nsplit` workgroups of a head
// (walk form, attention.hip) or of a KV head (matrix-core form, attention_mfma.hip) merges their partial softmax states itself, instead of a second
// launch (attention_split_combine_kernel) behind a kernel boundary.  Same operations in the same order as that kernel -- the weights exp(m_s - M),
// L and every output element summed in split order with one fmaf per split, o / L at the end -- so the two forms give identical bits
// (tests/test_hip_kernels.py::test_attention_decode_split_merged_equals_the_two_launch_form).
//
// Hand-off (the write-through form of cdna_hip_programming.md's in-launch split-K recipe): the partial states are stored with relaxed agent-scope atomic
// stores (`sc1`: written through to memory, 4 bytes each -- the kernels' natural store width), every wave waits for its stores (`s_waitcnt vmcnt(0)`),
// the workgroup meets, thread 0 adds 1 to the head's counter (relaxed, agent scope); the workgroup that reads nsplit - 1 is the last one: it puts the
// counter back to 0 (the next launch is behind a kernel boundary) and reads the states with relaxed agent-scope atomic loads (`sc1`).  No fence: the first
// form of this file -- a release fence in every thread, an acquire fence in the last workgroup, plain loads -- was correct and 8-9 us per layer SLOWER
// than the two launches (profiles/NEGATIVE_RESULTS.md 8).  The counters (one u32 per head, in front of the partial states:
// ntk_attention_split_scratch_bytes / ntk_attention_split_scratch_init) are zero before the first launch and after every launch.  nsplit <= 64.
#pragma once
#include "common.hip.h"

namespace ntk {

constexpr int ATT_MERGE_MAX_SPLITS = 64;
// bytes of counters in front of the partial states (n_heads u32, rounded up to 256 bytes)
__host__ __device__ inline size_t att_merge_header_bytes(int n_heads) { return ((size_t)n_heads * 4 + 255) / 256 * 256; }

// a partial-state word: written through (MERGE) or an ordinary store (the merge launch is behind a kernel boundary)
template <bool MERGE> __device__ __forceinline__ void att_part_store(float* p, float v) {
    if constexpr (MERGE) __hip_atomic_store(reinterpret_cast<unsigned*>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
__device__ __forceinline__ float att_part_load(const float* p) {
    return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// called by ALL threads of the workgroup, behind its partial-state stores; true in every thread of the last workgroup to arrive.
// flag: an LDS word of the kernel's one LDS object that nothing reads any more
__device__ __forceinline__ bool att_merge_arrive(unsigned* counter, int nsplit, int tid, volatile int* flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's partial-state stores have been written through
    __syncthreads();
    if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = old == (unsigned)nsplit - 1u;
        if (last) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *flag = last;
    }
    __syncthreads();
    return *flag != 0;
}

// One WAVE (all 64 lanes active) merges the nsplit <= 64 states ph[split][hd + 2] = (acc[hd], m, l) of one head for the output elements
// d = d0 + 64 e, e < ND (elements >= hd are skipped); out = the head's row of the output.
template <int ND>
__device__ __forceinline__ void att_merge_head_wave(float* __restrict__ out, const float* __restrict__ ph, int hd, int nsplit, int lane, int d0) {
    constexpr int B = 32;
    const int st = hd + 2;
    float m = -INFINITY, l = 0.0f;
    {
        const float* ps = ph + (size_t)min(lane, nsplit - 1) * st + hd;
        const float mv = att_part_load(ps), lv = att_part_load(ps + 1);
        if (lane < nsplit) { m = mv; l = lv; }
    }
    int dl[ND];
#pragma unroll
    for (int e = 0; e < ND; ++e) dl[e] = min(d0 + 64 * e, hd - 1);
    float v[ND][B];
#pragma unroll
    for (int e = 0; e < ND; ++e)
#pragma unroll
        for (int u = 0; u < B; ++u) v[e][u] = att_part_load(ph + (size_t)min(u, nsplit - 1) * st + dl[e]);
    const float M = wave_max(m);
    const float w = (m == -INFINITY) ? 0.0f : expf(m - M);   // (a split that saw no position: weight 0; lanes past nsplit: m = -inf, l = 0)
    float L = 0.0f, o[ND];
#pragma unroll
    for (int e = 0; e < ND; ++e) o[e] = 0.0f;
    const int n32 = (nsplit + B - 1) / B * B;
    for (int s0 = 0; s0 < n32; s0 += B) {   // (uniform: one or two trips)
        if (s0 > 0) {
#pragma unroll
            for (int e = 0; e < ND; ++e)
#pragma unroll
                for (int u = 0; u < B; ++u) v[e][u] = att_part_load(ph + (size_t)min(s0 + u, nsplit - 1) * st + dl[e]);
        }
#pragma unroll
        for (int u = 0; u < B; ++u) {
            // (s0 is 0 or 32: the lane index of a readlane must be a constant or an SGPR)
            const float ws = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w), s0 + u));
            const float ls = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(l), s0 + u));
            L = fmaf(ws, ls, L);
#pragma unroll
            for (int e = 0; e < ND; ++e) o[e] = fmaf(ws, v[e][u], o[e]);   // (past the last split: weight 0 x a finite duplicate)
        }
    }
#pragma unroll
    for (int e = 0; e < ND; ++e)
        if (d0 + 64 * e < hd) out[d0 + 64 * e] = o[e] / L;
}

}  // namespace ntk
