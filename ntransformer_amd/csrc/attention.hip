// attention.hip -- RoPE, KV store and GQA KV-cache attention for gfx950 (wave64).
//
// Replaces: launch_rope (reference src/cuda/rotary.cu:16-140), launch_copy_to_kv_cache, launch_attention_decode,
// launch_attention_prefill (reference src/cuda/attention.cu:108-425).  Same data contract: q/k/v and the
// output are F32, the caches are IEEE half [max_seq][n_kv_heads][head_dim] per layer, math is F32, softmax
// subtracts the row maximum, GQA maps head h to kv head h / (n_heads / n_kv_heads).
//
// What is different from the reference kernels: K and V rows are read 16 bytes per lane (8 halves), head_dim/8
// lanes share one cache row so a wave covers 64/(head_dim/8) positions per instruction; the score dot products
// reduce with wave shuffles instead of a serial per-thread loop; the P.V product is split over positions
// across all waves (the reference walks the positions serially with one thread per output dim).  The
// engine's decode path additionally fuses RoPE + KV store into the attention launch and takes the position
// from device memory so the launch can be replayed from a hipGraph.
#include "common.hip.h"
#include "attention_merge.hip.h"
#include <cfloat>
#include <cstdlib>

namespace ntk {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void unpack8(const u32x4 r, float (&f)[8]) {
    f[0] = h2f((uint16_t)(r.x & 0xFFFF)); f[1] = h2f((uint16_t)(r.x >> 16));
    f[2] = h2f((uint16_t)(r.y & 0xFFFF)); f[3] = h2f((uint16_t)(r.y >> 16));
    f[4] = h2f((uint16_t)(r.z & 0xFFFF)); f[5] = h2f((uint16_t)(r.z >> 16));
    f[6] = h2f((uint16_t)(r.w & 0xFFFF)); f[7] = h2f((uint16_t)(r.w >> 16));
}

// acc[0..7] += (the eight halves of r) * p, one v_fma_mix_f32 each: the instruction converts its F16 operand on the fly (exactly: the same
// bits as v_cvt_f32_f16 + v_fma_f32).  hipcc folds the conversion of the K rows into it by itself but turns the P.V update into v_cvt +
// v_pk_fma_f32 -- 1.5 issue slots per product (a packed FMA costs two, tools/probes/mfma_valu_probe.hip) against 1 here.  One asm block
// per position, behind an s_nop: p comes straight out of v_exp_f32, and a VALU instruction that reads a transcendental's result needs a
// wait state the compiler supplies for its own instructions but cannot supply inside opaque asm (without it the first product of every
// position read the PREVIOUS p in the lanes the quarter-rate v_exp had not written yet: found by the parity tests).
#ifdef NTK_ATTN_NO_ASM
__device__ __forceinline__ void pv_update(float (&acc)[8], const u32x4 r, const float p) {
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        acc[2 * i] = fmaf(p, h2f((uint16_t)(w[i] & 0xFFFFu)), acc[2 * i]);
        acc[2 * i + 1] = fmaf(p, h2f((uint16_t)(w[i] >> 16)), acc[2 * i + 1]);
    }
}
#else
__device__ __forceinline__ void pv_update(float (&acc)[8], const u32x4 r, const float p) {
    asm("s_nop 1\n\t"
        "v_fma_mix_f32 %0, %8, %12, %0 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %1, %8, %12, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %2, %9, %12, %2 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %3, %9, %12, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %4, %10, %12, %4 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %5, %10, %12, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %6, %11, %12, %6 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %7, %11, %12, %7 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7])
        : "v"(r.x), "v"(r.y), "v"(r.z), "v"(r.w), "v"(p));
}
#endif

// rotation of one (x0, x1) pair, reference rotary.cu:46-60
__device__ __forceinline__ void rope_pair(float& x0, float& x1, int pos, int pair_idx, int head_dim, float theta, float fscale) {
    // powf evaluated in double and rounded once: reproduces a correctly-rounded powf (what IEEE libm gives the
    // oracle); a 1-ulp slip in the frequency is a 4e-4 rad phase error at position 4095
    const float freq = 1.0f / (float)pow((double)theta, (double)((2.0f * pair_idx) / head_dim));
    const float angle = pos * freq * fscale;
    const float c = cosf(angle), s = sinf(angle);
    const float a = x0, b = x1;
    rope_rotate(a, b, c, s, x0, x1);
}

// ---------------------------------------------------------------------------------------------
// Core: softmax(q . K^T * scale) . V for ONE (head, query) pair, executed by one workgroup.
//   qs      : LDS, post-RoPE query [hd]
//   n_cache : keys/values 0..n_cache-1 are read from the cache
//   extra   : optional one more (key, value) pair held in LDS as floats (the token being decoded)
//   sc      : LDS scores [n_cache + 1]; part: LDS [nwaves][hd]; red: LDS [16]
// LPR = lanes per cache row (head_dim / 8); 0 selects the generic any-head_dim path.
// ---------------------------------------------------------------------------------------------
template <int LPR>
__device__ void attend(float* out, const float* qs, const uint16_t* kc, const uint16_t* vc, int n_cache,
                       const float* k_extra, const float* v_extra, int kv_head, int n_kv_heads, int hd,
                       float scale, float* sc, float* part, float* red) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    const size_t stride = (size_t)n_kv_heads * hd;
    const uint16_t* kbase = kc + (size_t)kv_head * hd;
    const uint16_t* vbase = vc + (size_t)kv_head * hd;
    const int n_keys = n_cache + (k_extra ? 1 : 0);

    // ---- phase 1: scores ------------------------------------------------------------------------
    if constexpr (LPR > 0) {
        constexpr int PPW = 64 / LPR;               // positions per wave instruction
        const int sub = lane / LPR, part_i = lane % LPR;
        float qreg[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) qreg[j] = qs[8 * part_i + j];
        for (int p0 = wave * PPW; p0 < n_cache; p0 += nwaves * PPW) {
            const int pos = p0 + sub;
            float s = 0.0f;
            if (pos < n_cache) {
                const u32x4 raw = *reinterpret_cast<const u32x4*>(kbase + pos * stride + 8 * part_i);
                float kf[8];
                unpack8(raw, kf);
#pragma unroll
                for (int j = 0; j < 8; ++j) s = fmaf(qreg[j], kf[j], s);
            }
            s = group_sum<LPR>(s);
            if (part_i == 0 && pos < n_cache) sc[pos] = s * scale;
        }
    } else {
        for (int pos = tid; pos < n_cache; pos += blockDim.x) {
            const uint16_t* k = kbase + pos * stride;
            float s = 0.0f;
            for (int d = 0; d < hd; ++d) s = fmaf(qs[d], h2f(k[d]), s);
            sc[pos] = s * scale;
        }
    }
    if (k_extra && wave == 0) {
        float s = 0.0f;
        for (int d = lane; d < hd; d += 64) s = fmaf(qs[d], k_extra[d], s);
        s = wave_sum(s);
        if (lane == 0) sc[n_cache] = s * scale;
    }
    __syncthreads();

    // ---- phase 2: softmax over sc[0..n_keys) -----------------------------------------------------
    float m = -FLT_MAX;
    for (int pos = tid; pos < n_keys; pos += blockDim.x) m = fmaxf(m, sc[pos]);
    m = block_max(m, red);
    float l = 0.0f;
    for (int pos = tid; pos < n_keys; pos += blockDim.x) {
        const float e = expf(sc[pos] - m);
        sc[pos] = e;
        l += e;
    }
    l = block_sum(l, red);
    const float inv = (l > 0.0f) ? 1.0f / l : 0.0f;   // reference attention.cu:293 (prefill guard); decode never hits 0
    __syncthreads();

    // ---- phase 3: out[d] = inv * sum_pos e[pos] V[pos][d] ------------------------------------------
    if constexpr (LPR > 0) {
        constexpr int PPW = 64 / LPR;
        const int sub = lane / LPR, part_i = lane % LPR;
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.0f;
        for (int p0 = wave * PPW; p0 < n_cache; p0 += nwaves * PPW) {
            const int pos = p0 + sub;
            if (pos < n_cache) {
                const float pw = sc[pos];
                const u32x4 raw = *reinterpret_cast<const u32x4*>(vbase + pos * stride + 8 * part_i);
                float vf[8];
                unpack8(raw, vf);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = fmaf(pw, vf[j], acc[j]);
            }
        }
#pragma unroll
        for (int off = LPR; off < 64; off <<= 1) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += __shfl_xor(acc[j], off, 64);
        }
        if (sub == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) part[wave * hd + 8 * part_i + j] = acc[j];
        }
        __syncthreads();
        for (int d = tid; d < hd; d += blockDim.x) {
            float t = 0.0f;
            for (int w = 0; w < nwaves; ++w) t += part[w * hd + d];
            if (v_extra) t = fmaf(sc[n_cache], v_extra[d], t);
            out[d] = t * inv;
        }
    } else {
        for (int d = tid; d < hd; d += blockDim.x) {
            float t = 0.0f;
            for (int pos = 0; pos < n_cache; ++pos) t = fmaf(sc[pos], h2f(vbase[pos * stride + d]), t);
            if (v_extra) t = fmaf(sc[n_cache], v_extra[d], t);
            out[d] = t * inv;
        }
    }
}

template <int LPR>
__global__ __launch_bounds__(256) void attention_kernel(float* __restrict__ output, const float* __restrict__ Q,
                                                        const uint16_t* __restrict__ kc, const uint16_t* __restrict__ vc,
                                                        int n_keys_base, int causal, int n_heads, int n_kv_heads, int hd,
                                                        float scale) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int head = blockIdx.x, qi = blockIdx.y;
    const int n_keys = causal ? n_keys_base + qi + 1 : n_keys_base;   // prefill: keys 0..start_pos+qi
    float* qs = lds;                        // [hd]
    float* red = qs + hd;                   // [16]
    float* part = red + 16;                 // [nwaves][hd]
    float* sc = part + (blockDim.x >> 6) * hd;
    const size_t qoff = ((size_t)qi * n_heads + head) * hd;
    for (int d = threadIdx.x; d < hd; d += blockDim.x) qs[d] = Q[qoff + d];
    __syncthreads();
    attend<LPR>(output + qoff, qs, kc, vc, n_keys, nullptr, nullptr, head / (n_heads / n_kv_heads), n_kv_heads, hd,
                scale, sc, part, red);
}

template <int LPR>
__global__ __launch_bounds__(256) void attention_decode_fused_kernel(
    float* __restrict__ output, const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
    uint16_t* __restrict__ kc, uint16_t* __restrict__ vc, const int* __restrict__ d_pos, int n_heads, int n_kv_heads,
    int hd, int max_seq, float scale, float theta, float fscale) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int head = blockIdx.x;
    const int group = n_heads / n_kv_heads, kv_head = head / group;
    const int pos = *d_pos;
    float* qs = lds;                        // [hd] post-RoPE query
    float* kx = qs + hd;                    // [hd] post-RoPE key of this token, rounded through half
    float* vx = kx + hd;                    // [hd] value of this token, rounded through half
    float* red = vx + hd;
    float* part = red + 16;
    float* sc = part + (blockDim.x >> 6) * hd;
    const int half_dim = hd / 2;
    const size_t cache_row = ((size_t)pos * n_kv_heads + kv_head) * hd;
    const bool writer = (head % group == 0) && pos < max_seq;   // one workgroup per kv head stores the row
    for (int i = threadIdx.x; i < half_dim; i += blockDim.x) {
        float a = q[(size_t)head * hd + i], b = q[(size_t)head * hd + i + half_dim];
        rope_pair(a, b, pos, i, hd, theta, fscale);
        qs[i] = a; qs[i + half_dim] = b;
        float ka = k[(size_t)kv_head * hd + i], kb = k[(size_t)kv_head * hd + i + half_dim];
        rope_pair(ka, kb, pos, i, hd, theta, fscale);
        const uint16_t ha = f2h(ka), hb = f2h(kb);          // reference attention.cu:338 (__float2half, RNE)
        kx[i] = h2f(ha); kx[i + half_dim] = h2f(hb);
        if (writer) { kc[cache_row + i] = ha; kc[cache_row + i + half_dim] = hb; }
    }
    for (int i = threadIdx.x; i < hd; i += blockDim.x) {
        const uint16_t hv = f2h(v[(size_t)kv_head * hd + i]);
        vx[i] = h2f(hv);
        if (writer) vc[cache_row + i] = hv;
    }
    __syncthreads();
    attend<LPR>(output + (size_t)head * hd, qs, kc, vc, pos, kx, vx, kv_head, n_kv_heads, hd, scale, sc, part, red);
}

// ---------------------------------------------------------------------------------------------
// Decode attention, single pass (engine path).  The reference kernel (attention.cu:108-202) and `attend` above make
// three passes over LDS scores with ~8 workgroup barriers; at decode lengths the launch is pure latency, so here
// every (wave, 16-lane group) streams its own positions with an online softmax (running max m, sum l, 8 output
// dims per lane) and the 16 partial states merge once through LDS.  Same F32 math on the same half-rounded K/V; the
// summation order differs (|d out| ~1e-6).  The token being decoded comes from LDS (kx/vx), not from the cache row
// another workgroup is writing.
// The launch is a chain of memory round trips (position -> cache rows -> next rows ...), and the walk is arranged around it
// (round 2, "v3": 6.7 -> 5.1 us per layer at position 128, 11.8 -> 7.4 at 320, 97 -> 49 at 4095; with 8 splits 21.7 -> 15.9):
//   * nothing that can be requested without the position waits for it: q, k, v, the frequencies and the first four cache
//     rows of every position group (row indices clamped to the cache, validity applied later) are in flight before *d_pos
//     is consumed;
//   * four positions per group are in flight instead of one (the next four are requested as soon as the current four are
//     unpacked, under a uniform branch; rows past the position are fetched and ignored, so no load is predicated per lane);
//   * the token being decoded is taken by its group after the loop (its place in that group's order);
//   * RoPE of q, RoPE of k and the conversion of v run on different waves, sin and cos share one argument reduction, and the
//     new cache row is stored at the very end (a store in front of the walk sits in the same in-order counter as the row loads);
//   * the merge computes each group's weight once.
// SPLIT: workgroup (head, sp) of nsplit takes the positions sp * G + g + j * nsplit * G of group g and leaves its un-normalised
// state (acc[hd], m, l) in `output` = part[head][sp] for attention_split_combine_kernel; otherwise nsplit = 1 and `output` is the
// head's normalised result.
// ---------------------------------------------------------------------------------------------
template <int LPR, int D, bool SPLIT, bool MERGE = false>   // MERGE: the partial state is written through (attention_merge.hip.h)
__device__ __forceinline__ void attention_decode_walk(
    float* __restrict__ output, const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
    uint16_t* __restrict__ kc, uint16_t* __restrict__ vc, const int* __restrict__ d_pos, const float* __restrict__ inv_freq,
    const int n_heads, const int n_kv_heads, const int hd, const int max_seq, const float scale, const float theta, const float fscale,
    const int head, const int sp, const int nsplit) {
    constexpr int PPW = 64 / LPR, NW = 4, G = NW * PPW;   // D: positions in flight per group
    constexpr float EMPTY = -3.0e38f;   // running maximum of a group that has seen nothing (finite: exp(EMPTY - x) = 0, EMPTY - EMPTY = 0)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int group = n_heads / n_kv_heads, kv_head = head / group;
    const int stepG = nsplit * G, first = sp * G;   // this workgroup's positions: first + g + j * stepG
    float* qs = lds;              // [hd] post-RoPE query; after the walk: [G] merge weights
    float* kx = qs + hd;          // [hd] post-RoPE key of this token, rounded through half
    float* vx = kx + hd;          // [hd] value of this token, rounded through half
    float* ms = vx + hd;          // [G] running maxima
    float* ls = ms + G;           // [G] running sums
    float* accs = ls + G;         // [G][hd]
    const int half_dim = hd / 2;
    const size_t stride = (size_t)n_kv_heads * hd;
    const int sub = lane / LPR, part_i = lane % LPR, g = wave * PPW + sub;
    const uint16_t* kbase = kc + (size_t)kv_head * hd + 8 * part_i;
    const uint16_t* vbase = vc + (size_t)kv_head * hd + 8 * part_i;
    const int pmax = max_seq - 1;

    // this thread's share of the new token: wave 0 rotates q, wave 1 rotates and stores k, waves 2-3 store v (hd <= 256)
    const int ri = tid & 63, role = tid >> 6;
    float in_a = 0.0f, in_b = 0.0f, in_c = 0.0f, in_d = 0.0f, freq = 0.0f;
    const bool rot = role < 2 && ri < half_dim;
    if (rot) {   // pairs (ri, ri + hd/2) and, for hd = 256, (ri + 64, ri + 64 + hd/2)
        const float* src = role == 0 ? q + (size_t)head * hd : k + (size_t)kv_head * hd;
        in_a = src[ri]; in_b = src[ri + half_dim];
        freq = inv_freq ? inv_freq[ri] : 1.0f / (float)pow((double)theta, (double)((2.0f * ri) / hd));
        if (half_dim > 64) { in_c = src[ri + 64]; in_d = src[ri + 64 + half_dim]; }
    }
    float vin0 = 0.0f, vin1 = 0.0f;
    const int vi = tid - 128;
    if (role >= 2) {
        if (vi < hd) vin0 = v[(size_t)kv_head * hd + vi];
        if (vi + 128 < hd) vin1 = v[(size_t)kv_head * hd + vi + 128];
    }

    // (after the loads RoPE waits for: a CU serves its requests roughly in order)
    // Rows are addressed by 32-bit byte offsets inside the layer's cache (max_seq * row bytes < 4 GiB: host-checked), advanced by a constant
    // per batch and clamped to the last row: an add and a min per row where the 64-bit form spent a quarter-rate multiply and a 64-bit
    // multiply-add (36 of the walk's ~250 issue slots per batch of 4 positions).
    const unsigned row_bytes = (unsigned)stride * 2u, off_max = (unsigned)pmax * row_bytes, adv = (unsigned)(stepG * D) * row_bytes;
    const char* kb = reinterpret_cast<const char*>(kbase);
    const char* vb = reinterpret_cast<const char*>(vbase);
    unsigned roff[D];
    u32x4 kraw[D], vraw[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        roff[d] = min((unsigned)(first + g + stepG * d) * row_bytes, off_max);
        kraw[d] = *reinterpret_cast<const u32x4*>(kb + roff[d]);
        vraw[d] = *reinterpret_cast<const u32x4*>(vb + roff[d]);
    }
    const int pos = *d_pos;
    const size_t cache_row = (size_t)pos * stride + (size_t)kv_head * hd;
    const bool writer = (head % group == 0) && sp == 0 && pos < max_seq;
    // the new cache row is stored at the very END of the kernel (values kept in registers): a store in front of the walk would
    // sit in the same in-order counter as the row loads and make the first wait of the walk a wait for its acknowledgement
    uint16_t st_h[4] = {0, 0, 0, 0};
    if (rot) {
        // reference rotary.cu:46-60; inv_freq (engine) holds 1/powf(theta, 2i/hd) computed once on the host
        auto rotate = [&](const int i, const float a, const float b, const float f, uint16_t& ha, uint16_t& hb) {
            const float angle = pos * f * fscale;
            float c, sn;
            sincosf(angle, &sn, &c);   // one argument reduction; bit-identical to sinf / cosf on gfx950 (tools/micro/sincos_check.hip)
            float ra, rb;
            rope_rotate(a, b, c, sn, ra, rb);
            if (role == 0) { qs[i] = ra; qs[i + half_dim] = rb; }
            else {
                ha = f2h(ra); hb = f2h(rb);   // attention.cu:338 (__float2half, RNE)
                kx[i] = h2f(ha); kx[i + half_dim] = h2f(hb);
            }
        };
        rotate(ri, in_a, in_b, freq, st_h[0], st_h[1]);
        if (half_dim > 64) {
            const int i2 = ri + 64;
            const float f2 = inv_freq ? inv_freq[i2] : 1.0f / (float)pow((double)theta, (double)((2.0f * i2) / hd));
            rotate(i2, in_c, in_d, f2, st_h[2], st_h[3]);
        }
    }
    if (role >= 2) {
        if (vi < hd) { st_h[0] = f2h(vin0); vx[vi] = h2f(st_h[0]); }
        if (vi + 128 < hd) { st_h[1] = f2h(vin1); vx[vi + 128] = h2f(st_h[1]); }
    }
    __syncthreads();

    float qreg[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) qreg[j] = qs[8 * part_i + j];
    float m = EMPTY, l = 0.0f, acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.0f;
    // NP positions of the group's walk as ONE online-softmax update: their scores first, then one rescale of the running state by
    // exp(m - new max) and NP weights -- NP + 1 exponentials instead of 2 NP, and the hardware exponential (v_exp_f32 on x * log2 e,
    // ~1 ulp; the reference's CUDA build evaluates expf the same way under --use_fast_math, CMakeLists.txt:20) instead of libm's
    // ~15-instruction expf: the walk is bound by its VALU work (one wave per SIMD), 75 -> 45 instructions per position and head
    // (round 3; measured: 8 rows in flight or 8-wave workgroups instead changed nothing, profiles/r03_attention_kv_head_form.txt).
    // Invalid positions (past the token, or a clamped row) are exact no-ops.
    // MASKED = false: every one of the np positions exists (all batches but the last): no selects.  V stays packed (two halves per dword).
    auto batch = [&](const float (*kf)[8], const u32x4* vr, const bool* valid, const int np, const bool masked) {
        float sc[D];
        float mn = m;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (d >= np) { sc[d] = EMPTY; continue; }
            float t = 0.0f;
#pragma unroll
            for (int j = 0; j < 8; ++j) t = fmaf(qreg[j], kf[d][j], t);
            t = group_sum<LPR>(t) * scale;
            sc[d] = (!masked || valid[d]) ? t : EMPTY;
            mn = fmaxf(mn, sc[d]);
        }
        const float a = __expf(m - mn);
        l *= a;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] *= a;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (d >= np) continue;
            const float pw = (!masked || valid[d]) ? __expf(sc[d] - mn) : 0.0f;
            l += pw;
            pv_update(acc, vr[d], pw);
        }
        m = mn;
    };
    auto walk_batch = [&](const int base, const bool masked) {
        float kf[D][8];
        u32x4 vr[D];
        bool valid[D];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            valid[d] = base + g + stepG * d < pos;
            vr[d] = vraw[d];
            if (masked && !valid[d]) vr[d] = u32x4{0u, 0u, 0u, 0u};   // rows past the position hold anything (0 * NaN)
            unpack8(kraw[d], kf[d]);
        }
        __builtin_amdgcn_sched_barrier(0);   // the ring registers are free: the next batch may land in them
        if (base + stepG * D < pos) {   // uniform: another batch follows
#pragma unroll
            for (int d = 0; d < D; ++d) {
                roff[d] = min(roff[d] + adv, off_max);
                kraw[d] = *reinterpret_cast<const u32x4*>(kb + roff[d]);
                vraw[d] = *reinterpret_cast<const u32x4*>(vb + roff[d]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        batch(kf, vr, valid, D, masked);
    };
    int base = first;
    for (; base + stepG * (D - 1) + G <= pos; base += stepG * D) walk_batch(base, false);   // whole batches: every position of every group exists
    if (base < pos) walk_batch(base, true);                                                  // (uniform) the last one, masked
    {   // the token being decoded: from LDS (another workgroup is writing its cache row), by the group whose turn it is
        float kf[D][8];
        u32x4 vr[D];
        bool valid[D];
#pragma unroll
        for (int j = 0; j < 8; ++j) kf[0][j] = kx[8 * part_i + j];
        {   // vx holds the value rounded through half: its half bits again (exact)
            const float* vxp = vx + 8 * part_i;
            vr[0].x = (uint32_t)f2h(vxp[0]) | ((uint32_t)f2h(vxp[1]) << 16); vr[0].y = (uint32_t)f2h(vxp[2]) | ((uint32_t)f2h(vxp[3]) << 16);
            vr[0].z = (uint32_t)f2h(vxp[4]) | ((uint32_t)f2h(vxp[5]) << 16); vr[0].w = (uint32_t)f2h(vxp[6]) | ((uint32_t)f2h(vxp[7]) << 16);
        }
        valid[0] = first + g == pos % stepG;
        batch(kf, vr, valid, 1, true);
    }
    if (part_i == 0) { ms[g] = m; ls[g] = l; }
#pragma unroll
    for (int j = 0; j < 8; ++j) accs[g * hd + 8 * part_i + j] = acc[j];
    __syncthreads();
    float M = ms[0];
    for (int i = 1; i < G; ++i) M = fmaxf(M, ms[i]);
    if (tid < G) qs[tid] = expf(ms[tid] - M);   // 0 for groups that saw no position
    __syncthreads();
    for (int d = tid; d < hd; d += blockDim.x) {
        float L = 0.0f, o = 0.0f;
        for (int i = 0; i < G; ++i) {
            const float w = qs[i];
            L = fmaf(w, ls[i], L);
            o = fmaf(w, accs[i * hd + d], o);
        }
        if constexpr (SPLIT) {
            att_part_store<MERGE>(output + d, o);
            if (d == 0) { att_part_store<MERGE>(output + hd, M); att_part_store<MERGE>(output + hd + 1, L); }
        } else {
            output[(size_t)head * hd + d] = o / L;
        }
    }
    if (writer) {
        if (role == 1 && rot) {
            kc[cache_row + ri] = st_h[0]; kc[cache_row + ri + half_dim] = st_h[1];
            if (half_dim > 64) { kc[cache_row + ri + 64] = st_h[2]; kc[cache_row + ri + 64 + half_dim] = st_h[3]; }
        }
        if (role >= 2) {
            if (vi < hd) vc[cache_row + vi] = st_h[0];
            if (vi + 128 < hd) vc[cache_row + vi + 128] = st_h[1];
        }
    }
}


template <int LPR, int D>
__global__ __launch_bounds__(256) void attention_decode_fused_v3_kernel(
    float* __restrict__ output, const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
    uint16_t* __restrict__ kc, uint16_t* __restrict__ vc, const int* __restrict__ d_pos, const float* __restrict__ inv_freq,
    int n_heads, int n_kv_heads, int hd, int max_seq, float scale, float theta, float fscale) {
    attention_decode_walk<LPR, D, false>(output, q, k, v, kc, vc, d_pos, inv_freq, n_heads, n_kv_heads, hd, max_seq, scale, theta, fscale,
                                         (int)blockIdx.x, 0, 1);
}

// ---------------------------------------------------------------------------------------------
// Split-KV decode attention for long contexts.  One workgroup per head walks the cache serially, so the
// single-pass launch grows with the context (49 us per layer at position 4095, where the whole layer's KV is only 16.8 MB).
// Here `nsplit` workgroups share a head -- grid (n_heads, nsplit), positions interleaved so every split sees the same load --
// and a second launch (attention_split_combine_kernel) merges the nsplit partial states per head.  RoPE + KV store of the new
// token as in the single pass (store: split 0 of the first head of each KV group).  The engine picks single pass / 8 / 16
// splits by position (host side, one hipGraph per regime).
// XCD-aware head order: workgroups are dealt to the 8 XCDs round-robin by linear id and every XCD has its own L2.
// blockIdx.x = i -> kv head i % n_kv_heads: with 8 KV heads (every Llama-3 size) all query heads of a KV head, in all splits,
// run on ONE XCD, so its cache rows leave HBM once instead of once per query head (measured: 4x HBM traffic, 23.6 us per layer
// at position 4095 with the plain order).
// part layout: [n_heads][nsplit][hd + 2] = acc[hd], m, l
// ---------------------------------------------------------------------------------------------
// MERGE: the head's last workgroup to finish merges the nsplit states into output[head] itself (attention_merge.hip.h) -- no combine launch
template <int LPR, int D, bool MERGE>
__global__ __launch_bounds__(256) void attention_decode_split_kernel(
    float* __restrict__ part, const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
    uint16_t* __restrict__ kc, uint16_t* __restrict__ vc, const int* __restrict__ d_pos, const float* __restrict__ inv_freq,
    int n_heads, int n_kv_heads, int hd, int max_seq, float scale, float theta, float fscale, float* __restrict__ output,
    unsigned* __restrict__ counters) {
    const int group = n_heads / n_kv_heads;
    const int kv_head = blockIdx.x % n_kv_heads, head = kv_head * group + blockIdx.x / n_kv_heads;
    const int sp = blockIdx.y, nsplit = gridDim.y;
    attention_decode_walk<LPR, D, true, MERGE>(part + ((size_t)head * nsplit + sp) * (hd + 2), q, k, v, kc, vc, d_pos, inv_freq, n_heads,
                                               n_kv_heads, hd, max_seq, scale, theta, fscale, head, sp, nsplit);
    if constexpr (MERGE) {
        extern __shared__ __attribute__((aligned(16))) float lds[];   // (the walk's LDS: nothing reads it any more)
        if (!att_merge_arrive(counters + head, nsplit, (int)threadIdx.x, reinterpret_cast<volatile int*>(lds))) return;
        const int wave = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
        if (64 * wave < hd)   // (uniform per wave) hd 64: wave 0; 128: waves 0, 1; 256: all four
            att_merge_head_wave<1>(output + (size_t)head * hd, part + (size_t)head * nsplit * (hd + 2), hd, nsplit, lane, 64 * wave + lane);
    }
}

__global__ __launch_bounds__(128) void attention_split_combine_kernel(float* __restrict__ output, const float* __restrict__ part,
                                                                      int hd, int nsplit, int n_kv_heads) {
    // One workgroup per head, ONE memory round trip for everything: thread s requests split s's (m, l), then every thread requests its
    // output element of up to 32 splits at once; the weights exp(m_s - M) go through LDS while those loads are in flight; the sums run in
    // split order.  Every loop over the splits is unrolled in blocks of 32 over tables padded with exact no-ops (weight 0, l 0, m -inf): a
    // rolled loop pays an LDS round trip per split and sum -- the kernel trace of the 3.9K-context decode showed this launch at 5.1 us,
    // three quarters of it in three such loops over 32 splits (profiles/r04_rocprofv3_kernel_trace_8b_q8_0_ctx3900.txt, first pass).
    // (Round 3 walked the splits in two rolled loops of dependent GLOBAL loads.)  Workgroup b serves head
    // (b % n_kv_heads) * group + b / n_kv_heads: on the XCD (b % 8) whose L2 the partial states of that KV head were written through.
    constexpr int B = 32;
    __shared__ __attribute__((aligned(16))) float wsh[1024], lsh[1024];
    const int group = (int)gridDim.x / n_kv_heads;
    const int head = ((int)blockIdx.x % n_kv_heads) * group + (int)blockIdx.x / n_kv_heads, tid = threadIdx.x;
    const float* ph = part + (size_t)head * nsplit * (hd + 2);
    const int n32 = (nsplit + B - 1) / B * B;   // <= 1024 (host-checked)
    float mreg[8], lreg[8];                      // thread t: splits t, t + 128 ...
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        mreg[u] = -INFINITY; lreg[u] = 0.0f;
        if (128 * u < nsplit) {   // (uniform)
            const int s0 = tid + 128 * u;
            const float* ps = ph + (size_t)min(s0, nsplit - 1) * (hd + 2) + hd;
            const float mv = ps[0], lv = ps[1];
            if (s0 < nsplit) { mreg[u] = mv; lreg[u] = lv; }
        }
    }
    const int d0 = min(tid, hd - 1);
    float v0[B];
#pragma unroll
    for (int u = 0; u < B; ++u) v0[u] = ph[(size_t)min(u, nsplit - 1) * (hd + 2) + d0];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 8; ++u)
        if (tid + 128 * u < n32) { wsh[tid + 128 * u] = mreg[u]; lsh[tid + 128 * u] = lreg[u]; }
    __syncthreads();
    float M = -INFINITY;
    for (int s0 = 0; s0 < n32; s0 += B) {
#pragma unroll
        for (int u = 0; u < B; ++u) M = fmaxf(M, wsh[s0 + u]);
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 8; ++u)   // (a split that saw no position: m = -inf, or the walk's finite EMPTY: weight 0 either way)
        if (tid + 128 * u < n32) wsh[tid + 128 * u] = (mreg[u] == -INFINITY) ? 0.0f : expf(mreg[u] - M);
    __syncthreads();
    float L = 0.0f;
    for (int s0 = 0; s0 < n32; s0 += B) {
#pragma unroll
        for (int u = 0; u < B; ++u) L = fmaf(wsh[s0 + u], lsh[s0 + u], L);   // split order, like the output sums
    }
    for (int d = tid; d < hd; d += blockDim.x) {
        float o = 0.0f;
        for (int s0 = 0; s0 < n32; s0 += B) {
            float v[B];
            if (s0 == 0 && d == d0) {
#pragma unroll
                for (int u = 0; u < B; ++u) v[u] = v0[u];
            } else {
#pragma unroll
                for (int u = 0; u < B; ++u) v[u] = ph[(size_t)min(s0 + u, nsplit - 1) * (hd + 2) + d];
            }
#pragma unroll
            for (int u = 0; u < B; ++u) o = fmaf(wsh[s0 + u], v[u], o);   // (past the last split: weight 0 x a finite duplicate)
        }
        output[(size_t)head * hd + d] = o / L;
    }
}

// ---------------------------------------------------------------------------------------------
// Prompt attention, flash style (replaces attention_prefill_kernel, reference attention.cu:216-311,
// whose one-block-per-(head, query) form re-reads every K / V row once per query: 32 x 1024 blocks each walking up
// to 1024 keys for a 1024-token prompt).  One workgroup per (head, tile of 32 queries): the keys the tile can see go
// through LDS in tiles of 64 cache rows (K and V, raw halves, 32 KB), every wave owns 8 of the queries and keeps an online
// softmax state (running max, sum, 8 output dims per lane) per query in registers -- the decode kernel's inner loop with
// the cache rows coming from LDS and shared by the 32 queries.  K / V traffic per head falls from T^2 / 2 rows to
// T^2 / 64; no score row in LDS, so no limit on the context.  F32 math on the same half-rounded K / V, causal limit
// start_pos + query index; only the summation order differs from the reference kernel (|d out| ~1e-6).
// ---------------------------------------------------------------------------------------------
constexpr int FA_QT = 32;    // queries per workgroup
constexpr int FA_KT = 64;    // cache rows per LDS tile
constexpr int FA_QPW = 8;    // queries per wave (4 waves)

template <int LPR>
__global__ __launch_bounds__(256) void attention_prefill_tiled_kernel(float* __restrict__ output, const float* __restrict__ Q,
                                                                     const uint16_t* __restrict__ kc, const uint16_t* __restrict__ vc,
                                                                     int T, int start_pos, int n_heads, int n_kv_heads, float scale) {
    constexpr int HD = 8 * LPR, PPW = 64 / LPR;
    extern __shared__ __attribute__((aligned(16))) uint8_t fa_lds[];
    uint16_t* kt = reinterpret_cast<uint16_t*>(fa_lds);                 // [FA_KT][HD] halves
    uint16_t* vt = kt + FA_KT * HD;                                     // [FA_KT][HD]
    float* qt = reinterpret_cast<float*>(vt + FA_KT * HD);              // [FA_QT][HD] queries of the tile
    const int head = blockIdx.x, q0 = blockIdx.y * FA_QT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kv_head = head / (n_heads / n_kv_heads);
    const size_t stride = (size_t)n_kv_heads * HD;
    const int sub = lane / LPR, part_i = lane % LPR;
    const int nq = min(FA_QT, T - q0);                                  // queries in this tile
    for (int i = tid; i < nq * HD; i += 256) qt[i] = Q[((size_t)(q0 + i / HD) * n_heads + head) * HD + (i % HD)];
    float m[FA_QPW], l[FA_QPW], acc[FA_QPW][8];
#pragma unroll
    for (int u = 0; u < FA_QPW; ++u) {
        m[u] = -INFINITY; l[u] = 0.0f;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[u][j] = 0.0f;
    }
    const int n_keys = start_pos + q0 + nq;                             // keys 0 .. n_keys-1 are visible to the last query of the tile
    for (int k0 = 0; k0 < n_keys; k0 += FA_KT) {
        __syncthreads();                                                // the previous tile has been consumed (and qt is written)
        const int nk = min(FA_KT, n_keys - k0);
        for (int i = tid; i < nk * LPR; i += 256) {                     // 16-byte pieces, contiguous per cache row
            const int r = i / LPR, c = i % LPR;
            const size_t g = (size_t)(k0 + r) * stride + (size_t)kv_head * HD + 8 * c;
            *reinterpret_cast<u32x4*>(kt + r * HD + 8 * c) = *reinterpret_cast<const u32x4*>(kc + g);
            *reinterpret_cast<u32x4*>(vt + r * HD + 8 * c) = *reinterpret_cast<const u32x4*>(vc + g);
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < FA_QPW; ++u) {
            const int qi = wave * FA_QPW + u;                           // query of the tile (wave-uniform)
            if (qi >= nq) continue;
            const int last = start_pos + q0 + qi;                       // causal limit: keys <= last
            if (k0 > last) continue;
            float qreg[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) qreg[j] = qt[qi * HD + 8 * part_i + j];
            const int kend = min(nk, last - k0 + 1);
            for (int r0 = 0; r0 < kend; r0 += PPW) {
                const int r = r0 + sub, rr = min(r, kend - 1);          // past the causal limit: a valid row (finite values), weight 0
                float kf[8], vf[8];
                unpack8(*reinterpret_cast<const u32x4*>(kt + rr * HD + 8 * part_i), kf);
                unpack8(*reinterpret_cast<const u32x4*>(vt + rr * HD + 8 * part_i), vf);
                float sc = 0.0f;
#pragma unroll
                for (int j = 0; j < 8; ++j) sc = fmaf(qreg[j], kf[j], sc);
                sc = group_sum<LPR>(sc);
                sc = r < kend ? sc * scale : -INFINITY;
                const float mn = fmaxf(m[u], sc);
                const float a = (mn == -INFINITY) ? 1.0f : expf(m[u] - mn), pw = (mn == -INFINITY) ? 0.0f : expf(sc - mn);
                l[u] = fmaf(l[u], a, pw);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[u][j] = fmaf(acc[u][j], a, pw * vf[j]);
                m[u] = mn;
            }
        }
    }
    // merge the PPW position groups of the wave per query, normalise, store
#pragma unroll
    for (int u = 0; u < FA_QPW; ++u) {
        const int qi = wave * FA_QPW + u;
        if (qi >= nq) continue;
        float mm = m[u], ll = l[u];
#pragma unroll
        for (int off = LPR; off < 64; off <<= 1) {
            const float mo = __shfl_xor(mm, off, 64), lo = __shfl_xor(ll, off, 64);
            const float mn = fmaxf(mm, mo);
            const float wa = (mm == -INFINITY) ? 0.0f : expf(mm - mn), wb = (mo == -INFINITY) ? 0.0f : expf(mo - mn);
            ll = fmaf(ll, wa, lo * wb);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[u][j] = fmaf(acc[u][j], wa, __shfl_xor(acc[u][j], off, 64) * wb);
            mm = mn;
        }
        if (sub == 0) {
            const float inv = ll > 0.0f ? 1.0f / ll : 0.0f;             // reference attention.cu:293
            float* o = output + ((size_t)(q0 + qi) * n_heads + head) * HD + 8 * part_i;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = acc[u][j] * inv;
        }
    }
}

__global__ void rope_kernel(float* __restrict__ q, float* __restrict__ k, const int* __restrict__ positions, int seq_len,
                            int n_heads, int n_kv_heads, int head_dim, float theta, float fscale, int interleaved) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int half_dim = head_dim / 2;
    const int total_q = seq_len * n_heads * half_dim, total_k = seq_len * n_kv_heads * half_dim;
    if (idx >= total_q + total_k) return;
    const bool is_key = idx >= total_q;
    const int local = is_key ? idx - total_q : idx;
    const int n_h = is_key ? n_kv_heads : n_heads;
    const int pair = local % half_dim, head = (local / half_dim) % n_h, sp = local / (half_dim * n_h);
    float* data = (is_key ? k : q) + ((size_t)sp * n_h + head) * head_dim;
    const int i0 = interleaved ? 2 * pair : pair, i1 = interleaved ? 2 * pair + 1 : pair + half_dim;
    float a = data[i0], b = data[i1];
    rope_pair(a, b, positions[sp], pair, head_dim, theta, fscale);
    data[i0] = a;
    data[i1] = b;
}

// The same rotation for a prompt: one workgroup per token evaluates the token's head_dim / 2 (cos, sin) pairs ONCE -- the correctly
// rounded frequency, cosf, sinf of rope_pair(), bit for bit -- and applies them to all n_heads + n_kv_heads heads (rope_kernel
// re-evaluates pow / cosf / sinf for every head: 40 times per pair on the 8B shapes, 29 us per 1024-token layer against the ~8 us the
// 42 MB of traffic take).  head_dim <= 256.
__global__ __launch_bounds__(256) void rope_rows_kernel(float* __restrict__ q, float* __restrict__ k, const int* __restrict__ positions,
                                                        int n_heads, int n_kv_heads, int head_dim, float theta, float fscale, int interleaved) {
    __shared__ float cs[2][128];
    const int sp = blockIdx.x, half_dim = head_dim / 2;
    const int pos = positions[sp];
    for (int i = threadIdx.x; i < half_dim; i += blockDim.x) {
        const float freq = 1.0f / (float)pow((double)theta, (double)((2.0f * i) / head_dim));
        const float angle = pos * freq * fscale;
        cs[0][i] = cosf(angle);
        cs[1][i] = sinf(angle);
    }
    __syncthreads();
    const int total = (n_heads + n_kv_heads) * half_dim;
    for (int idx = threadIdx.x; idx < total; idx += blockDim.x) {
        const int pair = idx % half_dim, head = idx / half_dim;
        float* data = head < n_heads ? q + ((size_t)sp * n_heads + head) * head_dim : k + ((size_t)sp * n_kv_heads + (head - n_heads)) * head_dim;
        const int i0 = interleaved ? 2 * pair : pair, i1 = interleaved ? 2 * pair + 1 : pair + half_dim;
        const float a = data[i0], b = data[i1], c = cs[0][pair], sn = cs[1][pair];
        rope_rotate(a, b, c, sn, data[i0], data[i1]);
    }
}

// ... and the same launch also stores the token's K (rotated) and V rows into the half-precision caches (kv_store_kernel's conversions of the same
// values: identical cache rows); q is rotated in place, k and v are only read.  One launch instead of two per prompt layer.
__global__ __launch_bounds__(256) void rope_kv_store_rows_kernel(float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                                 const int* __restrict__ positions, int n_heads, int n_kv_heads, int head_dim, float theta,
                                                                 float fscale, int interleaved, uint16_t* __restrict__ kc, uint16_t* __restrict__ vc,
                                                                 int start_pos, int max_seq) {
    __shared__ float cs[2][128];
    const int sp = blockIdx.x, half_dim = head_dim / 2;
    const int pos = positions[sp];
    for (int i = threadIdx.x; i < half_dim; i += blockDim.x) {
        const float freq = 1.0f / (float)pow((double)theta, (double)((2.0f * i) / head_dim));
        const float angle = pos * freq * fscale;
        cs[0][i] = cosf(angle);
        cs[1][i] = sinf(angle);
    }
    __syncthreads();
    const int cp = start_pos + sp, per_pos = n_kv_heads * head_dim;
    const bool store = cp < max_seq;   // reference attention.cu:336
    const int total = (n_heads + n_kv_heads) * half_dim;
    for (int idx = threadIdx.x; idx < total; idx += blockDim.x) {
        const int pair = idx % half_dim, head = idx / half_dim;
        const int i0 = interleaved ? 2 * pair : pair, i1 = interleaved ? 2 * pair + 1 : pair + half_dim;
        const float c = cs[0][pair], sn = cs[1][pair];
        if (head < n_heads) {
            float* data = q + ((size_t)sp * n_heads + head) * head_dim;
            const float a = data[i0], b = data[i1];
            rope_rotate(a, b, c, sn, data[i0], data[i1]);
        } else {
            const int kh = head - n_heads;
            const float* data = k + ((size_t)sp * n_kv_heads + kh) * head_dim;
            float ra, rb;
            rope_rotate(data[i0], data[i1], c, sn, ra, rb);
            if (store) {
                uint16_t* row = kc + (size_t)cp * per_pos + (size_t)kh * head_dim;
                row[i0] = f2h(ra);
                row[i1] = f2h(rb);
            }
        }
    }
    if (store)
        for (int e = threadIdx.x; e < per_pos; e += blockDim.x) vc[(size_t)cp * per_pos + e] = f2h(v[(size_t)sp * per_pos + e]);
}

__global__ void kv_store_kernel(uint16_t* __restrict__ kc, uint16_t* __restrict__ vc, const float* __restrict__ k,
                                const float* __restrict__ v, int total, int per_pos, int start_pos, int max_seq) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int t = idx / per_pos, e = idx % per_pos;
    const int cp = start_pos + t;
    if (cp >= max_seq) return;   // reference attention.cu:336
    kc[(size_t)cp * per_pos + e] = f2h(k[idx]);
    vc[(size_t)cp * per_pos + e] = f2h(v[idx]);
}

static size_t attn_lds(int hd, int n_keys, int nvec) {
    return sizeof(float) * ((size_t)nvec * hd + 16 + 4 * (size_t)hd + (size_t)n_keys + 1);
}

int launch_attention_decode_kvhead_mfma(float* part, const float* q, const float* k, const float* v, uint16_t* kc, uint16_t* vc,
                                        const int* d_pos, const float* inv_freq, int nh, int nkv, int max_seq, float scale, float theta,
                                        float fscale, int nsplit, float* merged_output, unsigned* counters, hipStream_t st);   // attention_mfma.hip
int launch_attention_prefill_mfma(float* out, const float* Q, const uint16_t* kc, const uint16_t* vc, int T, int start_pos, int nh, int nkv,
                                  int hd, float scale, hipStream_t st);   // attention_mfma.hip

static int launch_attention(float* out, const float* Q, const void* kc, const void* vc, int T, int n_keys_base, int causal,
                            int nh, int nkv, int hd, float scale, hipStream_t st) {
    if (nh <= 0 || nkv <= 0 || hd <= 0 || nh % nkv != 0 || T < 0) return NTK_E_SHAPE;
    if (T == 0) return NTK_OK;
    const int max_keys = causal ? n_keys_base + T : n_keys_base;
    const bool aligned = (reinterpret_cast<uintptr_t>(kc) & 15) == 0 && (reinterpret_cast<uintptr_t>(vc) & 15) == 0;
    const uint16_t* k16 = static_cast<const uint16_t*>(kc);
    const uint16_t* v16 = static_cast<const uint16_t*>(vc);
    static const bool tiled_off = NTK_TUNE_ENV_INT("NTK_PREFILL_ATTENTION_1TO1", 0) != 0;      // (tuning builds only: the older prompt kernels)
    static const bool mfma_off = NTK_TUNE_ENV_INT("NTK_PREFILL_ATTENTION_NO_MFMA", 0) != 0;
    if (causal && T > 1 && aligned && hd == 128 && !tiled_off && !mfma_off) {   // prompt, head_dim 128: F16 matrix cores (attention_mfma.hip)
        const int rc = launch_attention_prefill_mfma(out, Q, k16, v16, T, n_keys_base, nh, nkv, hd, scale, st);
        if (rc != NTK_E_SHAPE && rc != NTK_E_ALIGN) return rc;
    }
    if (causal && T > 1 && aligned && (hd == 64 || hd == 128 || hd == 256) && !tiled_off) {   // prompt: flash-style tiles
        const size_t fl = (size_t)2 * FA_KT * hd * sizeof(uint16_t) + (size_t)FA_QT * hd * sizeof(float);
        const dim3 fgrid(nh, (T + FA_QT - 1) / FA_QT);
        if (hd == 128) hipLaunchKernelGGL(attention_prefill_tiled_kernel<16>, fgrid, dim3(256), fl, st, out, Q, k16, v16, T, n_keys_base, nh, nkv, scale);
        else if (hd == 64) hipLaunchKernelGGL(attention_prefill_tiled_kernel<8>, fgrid, dim3(256), fl, st, out, Q, k16, v16, T, n_keys_base, nh, nkv, scale);
        else {
            static bool once = hipFuncSetAttribute(reinterpret_cast<const void*>(attention_prefill_tiled_kernel<32>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)fl) == hipSuccess;
            if (!once) return NTK_E_LAUNCH;
            hipLaunchKernelGGL(attention_prefill_tiled_kernel<32>, fgrid, dim3(256), fl, st, out, Q, k16, v16, T, n_keys_base, nh, nkv, scale);
        }
        return last_launch_status();
    }
    const size_t lds = attn_lds(hd, max_keys, 1);
    if (lds > 160 * 1024) return NTK_E_SHAPE;
    dim3 grid(nh, T), block(256);
    // more than 64 KiB of dynamic LDS (contexts beyond ~15K keys) must be opted into per kernel
#define NTK_ATT(LPR_)                                                                                                   \
    do {                                                                                                                \
        if (lds > 64 * 1024)                                                                                            \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attention_kernel<LPR_>),                            \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                            \
        hipLaunchKernelGGL(attention_kernel<LPR_>, grid, block, lds, st, out, Q, k16, v16, n_keys_base, causal, nh, nkv, hd, scale); \
    } while (0)
    if (aligned && hd == 128) NTK_ATT(16);
    else if (aligned && hd == 64) NTK_ATT(8);
    else if (aligned && hd == 256) NTK_ATT(32);
    else NTK_ATT(0);
#undef NTK_ATT
    return last_launch_status();
}

}  // namespace ntk

extern "C" {

int ntk_rope(float* q, float* k, const int* positions, int /*batch_size*/, int seq_len, int n_heads, int n_kv_heads,
             int head_dim, float theta_base, float freq_scale, int interleaved, void* stream) {
    if (!q || !k || !positions) return NTK_E_NULL;
    if (seq_len < 0 || n_heads < 0 || n_kv_heads < 0 || head_dim <= 0 || (head_dim & 1)) return NTK_E_SHAPE;
    const int total = seq_len * (n_heads + n_kv_heads) * (head_dim / 2);
    if (total == 0) return NTK_OK;
    if (seq_len >= 4 && head_dim <= 256)   // prompts: the token's (cos, sin) pairs once for all heads
        hipLaunchKernelGGL(ntk::rope_rows_kernel, dim3(seq_len), dim3(256), 0, ntk::resolve_stream(stream), q, k, positions, n_heads, n_kv_heads,
                           head_dim, theta_base, freq_scale, interleaved);
    else
        hipLaunchKernelGGL(ntk::rope_kernel, dim3((total + 255) / 256), dim3(256), 0, ntk::resolve_stream(stream), q, k,
                           positions, seq_len, n_heads, n_kv_heads, head_dim, theta_base, freq_scale, interleaved);
    return ntk::last_launch_status();
}

int ntk_rope_kv_store(float* q, const float* k, const float* v, const int* positions, int seq_len, int n_heads, int n_kv_heads, int head_dim,
                      float theta_base, float freq_scale, int interleaved, void* k_cache, void* v_cache, int start_pos, int max_seq, void* stream) {
    if (!q || !k || !v || !positions || !k_cache || !v_cache) return NTK_E_NULL;
    if (seq_len < 0 || n_heads <= 0 || n_kv_heads <= 0 || head_dim <= 0 || (head_dim & 1) || head_dim > 256 || start_pos < 0 || max_seq <= 0) return NTK_E_SHAPE;
    if (seq_len == 0) return NTK_OK;
    hipLaunchKernelGGL(ntk::rope_kv_store_rows_kernel, dim3(seq_len), dim3(256), 0, ntk::resolve_stream(stream), q, k, v, positions, n_heads, n_kv_heads,
                       head_dim, theta_base, freq_scale, interleaved, static_cast<uint16_t*>(k_cache), static_cast<uint16_t*>(v_cache), start_pos, max_seq);
    return ntk::last_launch_status();
}

int ntk_copy_to_kv_cache(void* k_cache, void* v_cache, const float* k, const float* v, int seq_len, int n_kv_heads,
                         int head_dim, int start_pos, int max_seq, void* stream) {
    if (!k_cache || !v_cache || !k || !v) return NTK_E_NULL;
    if (seq_len < 0 || n_kv_heads <= 0 || head_dim <= 0 || start_pos < 0) return NTK_E_SHAPE;
    const int per = n_kv_heads * head_dim, total = seq_len * per;
    if (total == 0) return NTK_OK;
    hipLaunchKernelGGL(ntk::kv_store_kernel, dim3((total + 255) / 256), dim3(256), 0, ntk::resolve_stream(stream),
                       static_cast<uint16_t*>(k_cache), static_cast<uint16_t*>(v_cache), k, v, total, per, start_pos, max_seq);
    return ntk::last_launch_status();
}

int ntk_attention_decode(float* output, const float* q, const void* k_cache, const void* v_cache, int seq_len, int n_heads,
                         int n_kv_heads, int head_dim, int /*max_seq*/, float scale, void* stream) {
    if (!output || !q || !k_cache || !v_cache) return NTK_E_NULL;
    if (seq_len <= 0) return NTK_E_SHAPE;
    return ntk::launch_attention(output, q, k_cache, v_cache, 1, seq_len, 0, n_heads, n_kv_heads, head_dim, scale,
                                 ntk::resolve_stream(stream));
}

int ntk_attention_prefill(float* output, const float* Q, const void* k_cache, const void* v_cache, int seq_len, int start_pos,
                          int n_heads, int n_kv_heads, int head_dim, int /*max_seq*/, float scale, void* stream) {
    if (!output || !Q || !k_cache || !v_cache) return NTK_E_NULL;
    if (seq_len < 0 || start_pos < 0) return NTK_E_SHAPE;
    return ntk::launch_attention(output, Q, k_cache, v_cache, seq_len, start_pos, 1, n_heads, n_kv_heads, head_dim, scale,
                                 ntk::resolve_stream(stream));
}

int ntk_attention_decode_fused(float* output, const float* q, const float* k, const float* v, void* k_cache, void* v_cache,
                               const int* d_pos, const float* inv_freq, int n_heads, int n_kv_heads, int head_dim, int max_seq,
                               float scale, float theta_base, float freq_scale, void* stream) {
    if (!output || !q || !k || !v || !k_cache || !v_cache || !d_pos) return NTK_E_NULL;
    if (n_heads <= 0 || n_kv_heads <= 0 || n_heads % n_kv_heads != 0 || head_dim <= 0 || (head_dim & 1) || max_seq <= 0)
        return NTK_E_SHAPE;
    if ((size_t)max_seq * n_kv_heads * head_dim * 2 >= 0xF0000000ull) return NTK_E_SHAPE;   // (32-bit row offsets inside one layer's cache)
    hipStream_t st = ntk::resolve_stream(stream);
    const bool aligned = (reinterpret_cast<uintptr_t>(k_cache) & 15) == 0 && (reinterpret_cast<uintptr_t>(v_cache) & 15) == 0;
    uint16_t* k16 = static_cast<uint16_t*>(k_cache);
    uint16_t* v16 = static_cast<uint16_t*>(v_cache);
    if (aligned && (head_dim == 128 || head_dim == 64 || head_dim == 256)) {   // single-pass kernel
        const int G = 4 * (64 / (head_dim / 8));
        const size_t lds = sizeof(float) * ((size_t)3 * head_dim + 2 * G + (size_t)G * head_dim);
#define NTK_ATTV(...) hipLaunchKernelGGL((ntk::attention_decode_fused_v3_kernel<__VA_ARGS__>), dim3(n_heads), dim3(256), lds, st, output, q, k, v, \
                                           k16, v16, d_pos, inv_freq, n_heads, n_kv_heads, head_dim, max_seq, scale, theta_base, freq_scale)
        if (head_dim == 128) NTK_ATTV(16, 4);
        else if (head_dim == 64) NTK_ATTV(8, 4);
        else NTK_ATTV(32, 4);
#undef NTK_ATTV
        return ntk::last_launch_status();
    }
    const size_t lds = ntk::attn_lds(head_dim, max_seq, 3);
    if (lds > 160 * 1024) return NTK_E_SHAPE;
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ntk::attention_decode_fused_kernel<0>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(ntk::attention_decode_fused_kernel<0>, dim3(n_heads), dim3(256), lds, st, output, q, k, v, k16, v16, d_pos,
                       n_heads, n_kv_heads, head_dim, max_seq, scale, theta_base, freq_scale);
    return ntk::last_launch_status();
}

size_t ntk_attention_split_scratch_bytes(int n_heads, int head_dim, int nsplit) {
    return ntk::att_merge_header_bytes(n_heads) + (size_t)n_heads * (size_t)nsplit * (size_t)(head_dim + 2) * sizeof(float);
}

int ntk_attention_split_scratch_init(float* scratch, int n_heads, void* stream) {
    if (!scratch) return NTK_E_NULL;
    if (n_heads <= 0) return NTK_E_SHAPE;
    return hipMemsetAsync(scratch, 0, ntk::att_merge_header_bytes(n_heads), ntk::resolve_stream(stream)) == hipSuccess ? NTK_OK : NTK_E_LAUNCH;
}

static int attention_decode_split_impl(float* output, const float* q, const float* k, const float* v, void* k_cache, void* v_cache,
                                       const int* d_pos, const float* inv_freq, int n_heads, int n_kv_heads, int head_dim, int max_seq,
                                       float scale, float theta_base, float freq_scale, int nsplit, float* scratch_all, void* stream, bool merged) {
    if (!output || !q || !k || !v || !k_cache || !v_cache || !d_pos || !scratch_all) return NTK_E_NULL;
    if (n_heads <= 0 || n_kv_heads <= 0 || n_heads % n_kv_heads != 0 || max_seq <= 0 || nsplit < 1 || nsplit > 1024)
        return NTK_E_SHAPE;
    if (merged && nsplit > ntk::ATT_MERGE_MAX_SPLITS) return NTK_E_SHAPE;
    unsigned* counters = reinterpret_cast<unsigned*>(scratch_all);   // (zero outside the merged launches)
    float* scratch = reinterpret_cast<float*>(reinterpret_cast<uint8_t*>(scratch_all) + ntk::att_merge_header_bytes(n_heads));
    if (head_dim != 128 && head_dim != 64 && head_dim != 256) return NTK_E_SHAPE;   // 16-byte row pieces: the engine falls back to the single pass otherwise
    if ((size_t)max_seq * n_kv_heads * head_dim * 2 >= 0xF0000000ull) return NTK_E_SHAPE;   // (32-bit row offsets inside one layer's cache)
    if ((reinterpret_cast<uintptr_t>(k_cache) & 15) || (reinterpret_cast<uintptr_t>(v_cache) & 15)) return NTK_E_ALIGN;
    hipStream_t st = ntk::resolve_stream(stream);
    uint16_t* k16 = static_cast<uint16_t*>(k_cache);
    uint16_t* v16 = static_cast<uint16_t*>(v_cache);
    // head_dim 128, at most 16 query heads per KV head, 16 splits or more: the matrix-core form, one workgroup per (KV head, split), every
    // cache row read once (attention_mfma.hip).  With fewer splits a wave of that form walks four or more 32-row chunks per 4096 positions
    // and the per-query-head walk below is faster (measured, tools/attn_bench.py over 40 rotating layer caches, 8B geometry, us per layer
    // incl. the combine launch: 1023 positions 8.7 (walk, 8 splits) vs 10.2 (matrix cores, 8) -- 2047: 10.1 vs 11.1 (16) -- 4095: 13.5 vs
    // 12.75 (32); 70B geometry 4095: 16.5 vs 13.5; inside the engine the forms tie between 2300 and 3300 positions: profiles/r04_attention_kvhead_form.txt)
    static const int kvhead_form = NTK_TUNE_ENV_INT("NTK_ATTN_KVHEAD", 1);   // (tuning builds: 0 = the per-query-head walk, 2 = the matrix-core form at any split count)
    if (head_dim == 128 && n_heads / n_kv_heads <= 16 && ((kvhead_form == 1 && nsplit >= 16) || kvhead_form == 2)) {
        const int rc = ntk::launch_attention_decode_kvhead_mfma(scratch, q, k, v, k16, v16, d_pos, inv_freq, n_heads, n_kv_heads, max_seq,
                                                                scale, theta_base, freq_scale, nsplit, merged ? output : nullptr, counters, st);
        if (rc != NTK_OK || merged) return rc;
        hipLaunchKernelGGL(ntk::attention_split_combine_kernel, dim3(n_heads), dim3(128), 0, st, output, scratch, head_dim, nsplit, n_kv_heads);
        return ntk::last_launch_status();
    }
    const int G = 4 * (64 / (head_dim / 8));
    const size_t lds = sizeof(float) * ((size_t)3 * head_dim + 2 * G + (size_t)G * head_dim);
#define NTK_ATTSP(...) hipLaunchKernelGGL((ntk::attention_decode_split_kernel<__VA_ARGS__>), dim3(n_heads, nsplit), dim3(256), lds, st, scratch, q, k, \
                                           v, k16, v16, d_pos, inv_freq, n_heads, n_kv_heads, head_dim, max_seq, scale, theta_base, freq_scale, output, counters)
    static const int split_d = NTK_TUNE_ENV_INT("NTK_ATTN_SPLIT_D", 4);   // (tuning builds: rows in flight per position group)
    if (merged) {
        if (head_dim == 128) NTK_ATTSP(16, 4, true);
        else if (head_dim == 64) NTK_ATTSP(8, 4, true);
        else NTK_ATTSP(32, 4, true);
        return ntk::last_launch_status();
    }
    if (head_dim == 128 && split_d == 8) NTK_ATTSP(16, 8, false);
    else if (head_dim == 128) NTK_ATTSP(16, 4, false);
    else if (head_dim == 64) NTK_ATTSP(8, 4, false);
    else NTK_ATTSP(32, 4, false);
#undef NTK_ATTSP
    if (ntk::last_launch_status() != NTK_OK) return NTK_E_LAUNCH;
    hipLaunchKernelGGL(ntk::attention_split_combine_kernel, dim3(n_heads), dim3(128), 0, st, output, scratch, head_dim, nsplit, n_kv_heads);
    return ntk::last_launch_status();
}

int ntk_attention_decode_split(float* output, const float* q, const float* k, const float* v, void* k_cache, void* v_cache,
                               const int* d_pos, const float* inv_freq, int n_heads, int n_kv_heads, int head_dim, int max_seq,
                               float scale, float theta_base, float freq_scale, int nsplit, float* scratch, void* stream) {
    return attention_decode_split_impl(output, q, k, v, k_cache, v_cache, d_pos, inv_freq, n_heads, n_kv_heads, head_dim, max_seq, scale, theta_base,
                                       freq_scale, nsplit, scratch, stream, false);
}

int ntk_attention_decode_split_merged(float* output, const float* q, const float* k, const float* v, void* k_cache, void* v_cache,
                                      const int* d_pos, const float* inv_freq, int n_heads, int n_kv_heads, int head_dim, int max_seq,
                                      float scale, float theta_base, float freq_scale, int nsplit, float* scratch, void* stream) {
    return attention_decode_split_impl(output, q, k, v, k_cache, v_cache, d_pos, inv_freq, n_heads, n_kv_heads, head_dim, max_seq, scale, theta_base,
                                       freq_scale, nsplit, scratch, stream, true);
}

}  // extern "C"
