// attention_mfma.hip -- causal prompt attention on the F16 matrix cores (second half).
//
// Replaces, for head_dim 128 and prompts, the reference's attention_prefill_kernel (reference src/cuda/attention.cu:216-311: one
// block per (head, query) that walks the whole prefix) and supersedes attention.hip's tiled VALU kernel, which the profile of a
// 1024-token 8B prompt showed at 722 us per layer = 12 TFLOP/s, 20 % of the prompt pass.  Same definition: scores = scale * q . k
// over the F16 cache rows 0 .. start_pos + query, softmax with max subtraction, output = sum p v / sum p, F32 accumulation.
//
// Arithmetic.  K and V are F16 in the cache (exact MFMA operands).  The F32 operands -- q * scale and the probabilities p -- are
// split into an F16 value and the F16 rounding of the remainder (hi + lo: 22 mantissa bits; what is lost is below 2^-22 of the
// operand, far inside the 1e-3 logit budget), so every product runs on v_mfma_f32_16x16x32_f16 with F32 accumulation: two
// MFMAs per operand pair.
//
// Decomposition (flash-attention-2 shape, transposed so that a lane owns ONE query):
//   * workgroup = 4 waves = 64 consecutive queries of one head; a wave owns 16 queries; key tiles of 64 cache rows are staged
//     through LDS once per workgroup: K row-major (pitch 272 B: the 16 rows of an operand read hit 64 banks), V TRANSPOSED
//     ([head_dim][64 keys], pitch 136 B) because both MFMA operands want their K dimension contiguous per lane;
//   * S^T = K . Q^T: A = K tile (M = 16 keys), B = Q (N = 16 queries, kept in registers for the whole kernel), 4 key blocks x 4
//     head_dim chunks x (hi, lo) = 32 MFMAs per tile.  The accumulator of lane (i, g) holds query i, keys 4g .. 4g+3 of each
//     block: all of a lane's scores belong to ITS query, so the online softmax is lane-local arithmetic plus two cross-lane
//     steps (the four lanes i, i+16, i+32, i+48 of a query);
//   * O^T += V^T . P^T: A = V^T (M = 16 head_dim rows), B = P^T -- and the accumulator layout of S^T IS the B-operand layout of
//     P^T for the key order {4g..4g+3, 16+4g..16+4g+3} of a 32-key chunk, which V^T is simply read in: no transposition of P;
//     8 head_dim blocks x 2 key chunks x (hi, lo) = 32 MFMAs per tile;
//   * causal: a workgroup walks key tiles 0 .. its last query's position, a wave skips the tiles beyond its own last query, the
//     diagonal tile is masked per element; workgroups with the longest prefixes are dispatched first.
// Bound: MFMA (64 MFMAs = 1024 cycles per 16 queries x 64 keys and SIMD; the softmax adds ~200 VALU instructions per tile).
#include "common.hip.h"

namespace ntk {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

constexpr int AM_QT = 64;     // queries per workgroup
constexpr int AM_KT = 64;     // cache rows per tile
constexpr int AM_HD = 128;
constexpr int AM_KSTR = AM_HD + 8;   // halves per K row in LDS (272 B)
constexpr int AM_VSTR = AM_KT + 4;   // halves per V^T row in LDS (136 B)

// x = hi + lo with hi = (half)x, lo = (half)(x - hi)
__device__ __forceinline__ void am_split8(const float (&x)[8], f16x8& hi, f16x8& lo) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const _Float16 h = (_Float16)x[e];
        hi[e] = h;
        lo[e] = (_Float16)(x[e] - (float)h);
    }
}

__global__ __launch_bounds__(256, 2) void attention_prefill_mfma_kernel(float* __restrict__ output, const float* __restrict__ Q,
                                                                        const uint16_t* __restrict__ kc, const uint16_t* __restrict__ vc,
                                                                        int T, int start_pos, int n_heads, int n_kv_heads, float scale) {
    __shared__ __attribute__((aligned(16))) uint16_t kt[AM_KT * AM_KSTR];    // 17 KB
    __shared__ __attribute__((aligned(16))) uint16_t vt[AM_HD * AM_VSTR];    // 17 KB, transposed
    const int head = blockIdx.x;
    const int q0 = ((int)gridDim.y - 1 - (int)blockIdx.y) * AM_QT;           // longest prefixes first
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, g = lane >> 4;
    const int kv_head = head / (n_heads / n_kv_heads);
    const size_t stride = (size_t)n_kv_heads * AM_HD;                        // halves between cache rows
    const int nq = min(AM_QT, T - q0);
    const int wq0 = q0 + 16 * wave, qi = wq0 + i;                            // this lane's query
    const int my_limit = start_pos + qi;                                     // keys <= my_limit are visible to it
    const int wave_limit = start_pos + min(wq0 + 15, T - 1);                 // ... to the wave's last query
    const bool wave_live = wq0 < T;

    // Q operand: (q * scale) as hi + lo halves, head_dim chunk c (32 wide), this lane's slots 8g .. 8g+7 -- registers for good
    f16x8 qh[4], ql[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = 0.0f;
        if (qi < T) {
            const float* qp = Q + ((size_t)qi * n_heads + head) * AM_HD + 32 * c + 8 * g;
            const float4 a = *reinterpret_cast<const float4*>(qp), b = *reinterpret_cast<const float4*>(qp + 4);
            x[0] = a.x * scale; x[1] = a.y * scale; x[2] = a.z * scale; x[3] = a.w * scale;
            x[4] = b.x * scale; x[5] = b.y * scale; x[6] = b.z * scale; x[7] = b.w * scale;
        }
        am_split8(x, qh[c], ql[c]);
    }
    f32x4 o[8];   // O^T: head_dim rows 16 ht + 4g + e of this lane's query
#pragma unroll
    for (int ht = 0; ht < 8; ++ht) o[ht] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    float m_run = -INFINITY, l_run = 0.0f;

    const int n_keys = start_pos + q0 + nq;   // cache rows 0 .. n_keys-1 are visible to the tile's last query
    uint32_t k_ofs[4], v_ofs[4];               // this thread's pieces of a tile, in halves from the tile's first row (a tile spans < 2^32 halves)
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int p = tid + 256 * n;
        k_ofs[n] = (uint32_t)((p >> 4) * stride + (size_t)kv_head * AM_HD + 8 * (p & 15));
        v_ofs[n] = (uint32_t)((4 * (tid >> 4) + n) * stride + (size_t)kv_head * AM_HD + 8 * (tid & 15));
    }
    for (int k0 = 0; k0 < n_keys; k0 += AM_KT) {
        __syncthreads();                       // the previous tile has been consumed
        // ---- stage the tile: 16-byte pieces, 16 per cache row, rows past the end repeat the last one (masked below) ----
        // K row-major: piece p -> row p / 16, piece p % 16.  V transposed: a thread takes piece c of the FOUR rows 4 rg .. 4 rg + 3 and
        // writes, per head_dim value, the four rows' halves as one 8-byte store (V^T[8c + e][4 rg .. 4 rg + 3]) -- 8 stores per thread and
        // tile where one store per half was 32.  (Row addresses: per-thread offsets fixed before the loop + a uniform k0 * stride; only
        // the launch's last tile can run past the end and takes the clamped form.)
        u32x4 kk[4], vv[4];
        if (k0 + AM_KT <= n_keys) {
            const size_t tile_ofs = (size_t)k0 * stride;
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                kk[n] = *reinterpret_cast<const u32x4*>(kc + tile_ofs + k_ofs[n]);
                vv[n] = *reinterpret_cast<const u32x4*>(vc + tile_ofs + v_ofs[n]);
            }
        } else {
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const int p = tid + 256 * n;
                kk[n] = *reinterpret_cast<const u32x4*>(kc + (size_t)min(k0 + (p >> 4), n_keys - 1) * stride + (size_t)kv_head * AM_HD + 8 * (p & 15));
                vv[n] = *reinterpret_cast<const u32x4*>(vc + (size_t)min(k0 + 4 * (tid >> 4) + n, n_keys - 1) * stride + (size_t)kv_head * AM_HD + 8 * (tid & 15));
            }
        }
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int p = tid + 256 * n;
            *reinterpret_cast<u32x4*>(kt + (p >> 4) * AM_KSTR + 8 * (p & 15)) = kk[n];
        }
        {
            const int c = tid & 15, rg = tid >> 4;
#pragma unroll
            for (int e2 = 0; e2 < 4; ++e2) {   // dword e2 of the pieces holds head_dim values 8c + 2 e2 (low half) and 8c + 2 e2 + 1 (high half)
                const uint32_t w0 = vv[0][e2], w1 = vv[1][e2], w2 = vv[2][e2], w3 = vv[3][e2];
                const u32x2 lo = {__builtin_amdgcn_perm(w1, w0, 0x05040100u), __builtin_amdgcn_perm(w3, w2, 0x05040100u)};
                const u32x2 hi = {__builtin_amdgcn_perm(w1, w0, 0x07060302u), __builtin_amdgcn_perm(w3, w2, 0x07060302u)};
                *reinterpret_cast<u32x2*>(vt + (8 * c + 2 * e2) * AM_VSTR + 4 * rg) = lo;
                *reinterpret_cast<u32x2*>(vt + (8 * c + 2 * e2 + 1) * AM_VSTR + 4 * rg) = hi;
            }
        }
        __syncthreads();
        if (!wave_live || k0 > wave_limit) continue;   // wave-uniform: nothing of this tile is visible to the wave's queries

        // ---- S^T = K . Q^T: 4 key blocks of 16, lane (i, g): query i, keys k0 + 16 mt + 4g + e ------------------------------
        f32x4 s[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const f16x8 ka = *reinterpret_cast<const f16x8*>(kt + (16 * mt + i) * AM_KSTR + 32 * c + 8 * g);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ka, qh[c], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ka, ql[c], acc, 0, 0, 0);
            }
            s[mt] = acc;
        }
        // ---- online softmax, one query per lane (its 16 scores here + the three other lanes of the query) --------------------
        float m_tile = -INFINITY;
        if (k0 + AM_KT - 1 <= start_pos + wq0 && wq0 + 15 < T) {   // wave-uniform: the whole tile is visible to every query of the wave
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int e = 0; e < 4; ++e) m_tile = fmaxf(m_tile, s[mt][e]);
        } else {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int key = k0 + 16 * mt + 4 * g + e;
                    s[mt][e] = key <= my_limit ? s[mt][e] : -INFINITY;
                    m_tile = fmaxf(m_tile, s[mt][e]);
                }
        }
        m_tile = fmaxf(m_tile, __shfl_xor(m_tile, 16, 64));
        m_tile = fmaxf(m_tile, __shfl_xor(m_tile, 32, 64));
        const float m_new = fmaxf(m_run, m_tile);
        // (rows of queries past T, or a tile wholly beyond this query's limit: everything masked, m_new may still be -inf)
        const float m_use = m_new == -INFINITY ? 0.0f : m_new;
        // the hardware exponential (v_exp_f32 on x log2 e, ~1 ulp), like the decode walk of attention.hip: 17 per tile and lane, and libm's
        // expf is ~15 instructions each where the kernel is bound by its VALU work (the reference's CUDA build evaluates expf the same
        // way under --use_fast_math, CMakeLists.txt:20)
        const float alpha = __expf(m_run - m_use);     // exp(-inf) = 0 on the first visible tile
        float l_tile = 0.0f;
        f16x8 ph[2], pl[2];                            // P^T operands of the two 32-key chunks
#pragma unroll
        for (int kc2 = 0; kc2 < 2; ++kc2) {
            float pv[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                pv[e] = __expf(s[2 * kc2][e] - m_use);         // keys 32 kc + 4g + e
                pv[4 + e] = __expf(s[2 * kc2 + 1][e] - m_use); // keys 32 kc + 16 + 4g + e
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) l_tile += pv[e];
            am_split8(pv, ph[kc2], pl[kc2]);
        }
        l_tile += __shfl_xor(l_tile, 16, 64);
        l_tile += __shfl_xor(l_tile, 32, 64);
        l_run = l_run * alpha + l_tile;
        m_run = m_new;
        // ---- O^T = alpha O^T + V^T . P^T ------------------------------------------------------------------------------------------
        if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {   // (x * 1.0f is x: skipping the rescale where no lane's maximum moved is exact)
#pragma unroll
            for (int ht = 0; ht < 8; ++ht) o[ht] = o[ht] * alpha;
        }
#pragma unroll
        for (int ht = 0; ht < 8; ++ht) {
            f32x4 acc = o[ht];
#pragma unroll
            for (int kc2 = 0; kc2 < 2; ++kc2) {
                const uint16_t* vrow = vt + (16 * ht + i) * AM_VSTR + 32 * kc2 + 4 * g;
                const u32x2 v0 = *reinterpret_cast<const u32x2*>(vrow), v1 = *reinterpret_cast<const u32x2*>(vrow + 16);
                const f16x8 va = __builtin_bit_cast(f16x8, u32x4{v0.x, v0.y, v1.x, v1.y});
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(va, ph[kc2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(va, pl[kc2], acc, 0, 0, 0);
            }
            o[ht] = acc;
        }
    }
    if (qi < T) {
        const float inv = l_run > 0.0f ? 1.0f / l_run : 0.0f;   // reference attention.cu:293 guards sum > 0
        float* op = output + ((size_t)qi * n_heads + head) * AM_HD + 4 * g;
#pragma unroll
        for (int ht = 0; ht < 8; ++ht) *reinterpret_cast<f32x4*>(op + 16 * ht) = o[ht] * inv;
    }
}

// head_dim 128, 16-byte aligned caches and Q / output; T >= 1.  Returns NTK_E_SHAPE for anything else (caller: the tiled kernel).
int launch_attention_prefill_mfma(float* out, const float* Q, const uint16_t* kc, const uint16_t* vc, int T, int start_pos, int nh, int nkv,
                                  int hd, float scale, hipStream_t st) {
    if (hd != AM_HD || T < 1 || nh % nkv != 0) return NTK_E_SHAPE;
    if ((reinterpret_cast<uintptr_t>(Q) & 15) || (reinterpret_cast<uintptr_t>(out) & 15) || (reinterpret_cast<uintptr_t>(kc) & 15) ||
        (reinterpret_cast<uintptr_t>(vc) & 15))
        return NTK_E_ALIGN;
    const dim3 grid(nh, (T + AM_QT - 1) / AM_QT);
    hipLaunchKernelGGL(attention_prefill_mfma_kernel, grid, dim3(256), 0, st, out, Q, kc, vc, T, start_pos, nh, nkv, scale);
    return last_launch_status();
}

}  // namespace ntk
