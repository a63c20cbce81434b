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
#include "attention_merge.hip.h"

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

// =================================================================================================================================
// Long-context DECODE attention on the matrix cores: one workgroup per (KV head, split), the query heads that share the KV head are
// the 16-wide N dimension of the MFMAs (4 of 16 columns live for 8B, 8 for 70B -- the matrix cores are idle in decode anyway).
//
// Replaces, beyond a few hundred positions, the per-query-head walk of attention.hip (attention_decode_split_kernel; reference
// attention.cu:108-214 attention_decode_generic_kernel + rotary.cu:16-62 + attention.cu:316-342): that walk reads every cache row
// once per query head of the group (4x / 8x through L2) and spends ~45 VALU instructions per (position, head) at one wave per SIMD --
// 11.5 us per layer at 4095 positions against 2.6 us of cache bytes.  Here a cache row is read ONCE, straight into the A operand of
//   S^T[key][head] = K . (q scale)^T          (v_mfma_f32_16x16x32_f16, q split into an F16 value + the F16 rounding of the rest)
// and V goes through a wave-private transposed LDS image into
//   O^T[dim][head] += V^T . P^T               (P split the same way; the accumulator layout of S^T IS the B operand layout of P^T)
// exactly as in the prompt kernel above, with a lane owning ONE query head: the online softmax is lane-local plus two cross-lane steps.
//
// Work: the positions 0 .. pos (pos = the token being decoded) in chunks of 32 cache rows; chunk c belongs to wave (c mod W) of the
// W = 4 nsplit waves of a KV head, so a launch sized for its regime gives every wave one or two chunks, all requested in the first
// microsecond of the kernel (row addresses do not depend on the position: rows past it are loaded and masked by selects, never by
// arithmetic -- they may hold anything).  The token being decoded: RoPE of the group's queries by all workgroups; RoPE of k and the
// half rounding of v by the workgroup whose wave owns chunk pos / 32, which takes that row from LDS and stores it to the cache at the end.
// Output: un-normalised partial states part[head][split] = (acc[128], m, l) for attention_split_combine_kernel (attention.hip).
// =================================================================================================================================
constexpr int AD_CK = 32;                       // cache rows per chunk
constexpr int AD_VP = 40;                       // halves per V^T row in LDS (80 B: the 8-byte operand reads of 32 lanes cover 64 banks)
constexpr int AD_WAVE_LDS = AM_HD * AD_VP * 2;  // 10240 B per wave: V^T of its chunk; afterwards its partial output [head][128] floats

// MERGE: the KV head's last workgroup to finish merges the nsplit states of its query heads into `output` itself (attention_merge.hip.h)
template <bool MERGE>
__global__ __launch_bounds__(256) void attention_decode_kvhead_mfma_kernel(
    float* __restrict__ part, const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
    uint16_t* __restrict__ kc, uint16_t* __restrict__ vc, const int* __restrict__ d_pos, const float* __restrict__ inv_freq,
    const int n_heads, const int n_kv_heads, const int max_seq, const float scale, const float theta, const float fscale,
    float* __restrict__ output, unsigned* __restrict__ counters) {
    // one LDS object: [queries 16 x 128 f32][new k row, new v row as halves][4 wave regions][m, l of 4 waves x 16 heads]
    __shared__ __attribute__((aligned(16))) uint8_t smem[16 * AM_HD * 4 + 2 * AM_HD * 2 + 4 * AD_WAVE_LDS + 2 * 64 * 4];
    float* qs = reinterpret_cast<float*>(smem);
    uint16_t* knew = reinterpret_cast<uint16_t*>(smem + 16 * AM_HD * 4);
    uint16_t* vnew = knew + AM_HD;
    uint8_t* wl0 = smem + 16 * AM_HD * 4 + 2 * AM_HD * 2;
    float* ms = reinterpret_cast<float*>(wl0 + 4 * AD_WAVE_LDS);
    float* ls = ms + 64;

    const int kv_head = blockIdx.x, sp = blockIdx.y, nsplit = gridDim.y;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, g = lane >> 4;
    const int group = n_heads / n_kv_heads;               // query heads per KV head (<= 16: host-checked)
    const int W = 4 * nsplit, gw = 4 * sp + wave;         // waves of this KV head, this wave's index among them
    const unsigned row_bytes = (unsigned)n_kv_heads * AM_HD * 2u;
    const char* kb = reinterpret_cast<const char*>(kc) + (size_t)kv_head * AM_HD * 2;
    const char* vb = reinterpret_cast<const char*>(vc) + (size_t)kv_head * AM_HD * 2;
    const unsigned last_row = (unsigned)(max_seq - 1);

    // ---- the first chunk's rows, requested before anything else (32-bit byte offsets: max_seq * row_bytes < 4 GiB, host-checked) ----
    // K: block mt (16 keys), head_dim chunk c (32 wide): lane (i, g) holds dims 32c + 8g .. +7 of key 16 mt + i = the A operand itself.
    // V: combination n of (piece pc = 16 bytes of a row, row group rg = 4 rows): lane p = lane + 64 n -> pc = p & 15, rg = p >> 4.
    u32x4 kraw[2][4], vraw[2][4];
    auto request = [&](const int chunk) {
        const unsigned r0 = (unsigned)chunk * AD_CK;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const unsigned off = min(r0 + 16u * mt + (unsigned)i, last_row) * row_bytes + 16u * g;
#pragma unroll
            for (int c = 0; c < 4; ++c) kraw[mt][c] = *reinterpret_cast<const u32x4*>(kb + off + 64u * c);
        }
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int p = lane + 64 * n;
            const unsigned rbase = r0 + 4u * (unsigned)(p >> 4);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                vraw[n][r] = *reinterpret_cast<const u32x4*>(vb + min(rbase + r, last_row) * row_bytes + 16u * (p & 15));
        }
    };
    // The token's own inputs are requested FIRST (a CU returns its loads in request order: behind 16 KB of cache rows per wave they
    // would come back with HBM latency, and RoPE -- which the cache rows do not wait for -- would start two microseconds late).
    // Thread t: query pairs (ri, ri + 64) of heads t / 64 and t / 64 + 4 ...; threads 0-63 also the key pair, threads 128-255 a value.
    const int ri = tid & 63;
    float qa[4], qb[4];                                                  // group <= 16: at most 4 query pairs per thread
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int n = (tid >> 6) + 4 * u;
        qa[u] = qb[u] = 0.0f;
        if (n < group) {
            const float* src = q + ((size_t)kv_head * group + n) * AM_HD;
            qa[u] = src[ri]; qb[u] = src[ri + 64];
        }
    }
    float ka_in = 0.0f, kb_in = 0.0f, v_in = 0.0f;
    if (tid < 64) { const float* src = k + (size_t)kv_head * AM_HD; ka_in = src[tid]; kb_in = src[tid + 64]; }
    else if (tid >= 128) v_in = v[(size_t)kv_head * AM_HD + tid - 128];
    const float freq = inv_freq ? inv_freq[ri] : 1.0f / (float)pow((double)theta, (double)((2.0f * ri) / AM_HD));
    const int pos = *d_pos;
    __builtin_amdgcn_sched_barrier(0);   // (the compiler may not hoist the cache rows above the token's loads)
    request(gw);
    __builtin_amdgcn_sched_barrier(0);

    // ---- the token being decoded: RoPE (reference rotary.cu:46-60, the arithmetic of attention.hip's walk) ----
    float rc, rs;
    {
        const float angle = pos * freq * fscale;
        sincosf(angle, &rs, &rc);   // one argument reduction for the thread's pairs (same frequency index ri)
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int n = (tid >> 6) + 4 * u;
        if (n < group) {
            rope_rotate(qa[u], qb[u], rc, rs, qs[n * AM_HD + ri], qs[n * AM_HD + ri + 64]);
        }
    }
    const int own_chunk = pos / AD_CK;                                 // the chunk that contains the token being decoded
    const bool owner_wg = ((own_chunk % W) >> 2) == sp;                 // (uniform) ... belongs to a wave of this workgroup
    const bool writer = owner_wg && pos < max_seq;
    uint16_t st_h[2] = {0, 0};
    if (owner_wg) {
        if (tid < 64) {                                                 // key pair (tid, tid + 64): attention.cu:338 (__float2half, RNE)
            float ra, rb;
            rope_rotate(ka_in, kb_in, rc, rs, ra, rb);
            st_h[0] = f2h(ra); st_h[1] = f2h(rb);
            knew[tid] = st_h[0]; knew[tid + 64] = st_h[1];
        } else if (tid >= 128) {                                        // value element tid - 128
            st_h[0] = f2h(v_in);
            vnew[tid - 128] = st_h[0];
        }
    }
    __syncthreads();

    // Q operand: (q * scale) as hi + lo halves; lane (i, g): head i, dims 32c + 8g .. +7.  Heads past the group: zero columns.
    f16x8 qh[4], ql[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = 0.0f;
        if (i < group) {
            const float4 a = *reinterpret_cast<const float4*>(qs + i * AM_HD + 32 * c + 8 * g);
            const float4 b = *reinterpret_cast<const float4*>(qs + i * AM_HD + 32 * c + 8 * g + 4);
            x[0] = a.x * scale; x[1] = a.y * scale; x[2] = a.z * scale; x[3] = a.w * scale;
            x[4] = b.x * scale; x[5] = b.y * scale; x[6] = b.z * scale; x[7] = b.w * scale;
        }
        am_split8(x, qh[c], ql[c]);
    }

    f32x4 o[8];   // O^T: dims 16 ht + 4g + e of head i
#pragma unroll
    for (int ht = 0; ht < 8; ++ht) o[ht] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    float m_run = -INFINITY, l_run = 0.0f;
    uint16_t* vt = reinterpret_cast<uint16_t*>(wl0 + wave * AD_WAVE_LDS);   // this wave's V^T image [128 dims][AD_VP]
    const int nchunks = own_chunk + 1;

    for (int chunk = gw; chunk < nchunks; chunk += W) {
        const int r0 = chunk * AD_CK;
        const bool last = chunk == own_chunk;   // (uniform) holds the token being decoded, and rows past it
        u32x4 ka[2][4], vv[2][4];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int c = 0; c < 4; ++c) ka[mt][c] = kraw[mt][c];
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) vv[n][r] = vraw[n][r];
        if (last) {   // the new row from LDS; rows past it zeroed (V) -- their scores are masked below
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int key = r0 + 16 * mt + i;
                if (key == pos) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) ka[mt][c] = *reinterpret_cast<const u32x4*>(knew + 32 * c + 8 * g);
                }
            }
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int p = lane + 64 * n;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = r0 + 4 * (p >> 4) + r;
                    if (key == pos) vv[n][r] = *reinterpret_cast<const u32x4*>(vnew + 8 * (p & 15));
                    if (key > pos) vv[n][r] = u32x4{0u, 0u, 0u, 0u};
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (chunk + W < nchunks) request(chunk + W);   // (uniform) the ring registers are free: the next chunk may land in them
        __builtin_amdgcn_sched_barrier(0);

        // ---- V^T image: per head_dim value the four rows' halves as one 8-byte store; key group (4 rows) rg of dim d sits at
        //      column group rg ^ ((d >> 3) & 7) (the 16 pieces of a row would otherwise hit two banks) ----
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int p = lane + 64 * n, pc = p & 15, rg = p >> 4;
            const int col = 4 * (rg ^ (pc & 7));
#pragma unroll
            for (int e2 = 0; e2 < 4; ++e2) {   // dword e2 of a piece: dims 8 pc + 2 e2 (low half) and 8 pc + 2 e2 + 1 (high half)
                const uint32_t w0 = vv[n][0][e2], w1 = vv[n][1][e2], w2 = vv[n][2][e2], w3 = vv[n][3][e2];
                const u32x2 lo = {__builtin_amdgcn_perm(w1, w0, 0x05040100u), __builtin_amdgcn_perm(w3, w2, 0x05040100u)};
                const u32x2 hi = {__builtin_amdgcn_perm(w1, w0, 0x07060302u), __builtin_amdgcn_perm(w3, w2, 0x07060302u)};
                *reinterpret_cast<u32x2*>(vt + (8 * pc + 2 * e2) * AD_VP + col) = lo;
                *reinterpret_cast<u32x2*>(vt + (8 * pc + 2 * e2 + 1) * AD_VP + col) = hi;
            }
        }

        // ---- S^T = K . Q^T: lane (i, g): head i, keys r0 + 16 mt + 4g + e ----
        f32x4 s[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const f16x8 a = __builtin_bit_cast(f16x8, ka[mt][c]);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, qh[c], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, ql[c], acc, 0, 0, 0);
            }
            s[mt] = acc;
        }
        float m_tile = -INFINITY;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (last) s[mt][e] = (r0 + 16 * mt + 4 * g + e <= pos) ? s[mt][e] : -INFINITY;   // a select: the row may hold anything
                m_tile = fmaxf(m_tile, s[mt][e]);
            }
        m_tile = fmaxf(m_tile, __shfl_xor(m_tile, 16, 64));
        m_tile = fmaxf(m_tile, __shfl_xor(m_tile, 32, 64));
        const float m_new = fmaxf(m_run, m_tile);          // finite: every chunk holds at least one position <= pos
        const float alpha = __expf(m_run - m_new);         // exp(-inf) = 0 on the wave's first chunk
        float pv[8], l_tile = 0.0f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            pv[e] = __expf(s[0][e] - m_new);               // keys r0 + 4g + e
            pv[4 + e] = __expf(s[1][e] - m_new);           // keys r0 + 16 + 4g + e
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) l_tile += pv[e];
        f16x8 ph, pl;
        am_split8(pv, ph, pl);
        l_tile += __shfl_xor(l_tile, 16, 64);
        l_tile += __shfl_xor(l_tile, 32, 64);
        l_run = l_run * alpha + l_tile;
        m_run = m_new;

        // ---- O^T = alpha O^T + V^T . P^T ----
#pragma unroll
        for (int ht = 0; ht < 8; ++ht) {
            const int d = 16 * ht + i, sw = (d >> 3) & 7;
            const uint16_t* vrow = vt + d * AD_VP;
            const u32x2 v0 = *reinterpret_cast<const u32x2*>(vrow + 4 * (g ^ sw));          // keys 4g .. 4g+3
            const u32x2 v1 = *reinterpret_cast<const u32x2*>(vrow + 4 * ((4 + g) ^ sw));    // keys 16 + 4g .. 16 + 4g+3
            const f16x8 va = __builtin_bit_cast(f16x8, u32x4{v0.x, v0.y, v1.x, v1.y});
            f32x4 acc = o[ht] * alpha;
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(va, ph, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(va, pl, acc, 0, 0, 0);
            o[ht] = acc;
        }
    }

    // ---- merge the four waves (un-normalised states), write part[head][sp] = (acc[128], m, l) ----
    float* mine = reinterpret_cast<float*>(wl0 + wave * AD_WAVE_LDS);   // [head][128] (the wave's own LDS reads are behind it: in order)
    if (i < group) {
#pragma unroll
        for (int ht = 0; ht < 8; ++ht) *reinterpret_cast<f32x4*>(mine + i * AM_HD + 16 * ht + 4 * g) = o[ht];
        if (g == 0) { ms[wave * 16 + i] = m_run; ls[wave * 16 + i] = l_run; }
    }
    __syncthreads();
    for (int idx = tid; idx < group * AM_HD; idx += 256) {
        const int n = idx >> 7, d = idx & (AM_HD - 1);
        float M = -INFINITY;
#pragma unroll
        for (int w = 0; w < 4; ++w) M = fmaxf(M, ms[w * 16 + n]);
        float L = 0.0f, acc = 0.0f;
        if (M > -INFINITY) {
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const float mw = ms[w * 16 + n];
                const float wgt = mw > -INFINITY ? expf(mw - M) : 0.0f;
                L = fmaf(wgt, ls[w * 16 + n], L);
                acc = fmaf(wgt, reinterpret_cast<const float*>(wl0 + w * AD_WAVE_LDS)[n * AM_HD + d], acc);
            }
        }
        float* out = part + (((size_t)kv_head * group + n) * nsplit + sp) * (AM_HD + 2);
        att_part_store<MERGE>(out + d, acc);
        if (d == 0) { att_part_store<MERGE>(out + AM_HD, M); att_part_store<MERGE>(out + AM_HD + 1, L); }
    }
    if (writer) {   // the new cache row, at the very end (attention.hip: a store in front of the walk delays the first row)
        const size_t cache_row = (size_t)pos * n_kv_heads * AM_HD + (size_t)kv_head * AM_HD;
        if (tid < 64) { kc[cache_row + tid] = st_h[0]; kc[cache_row + tid + 64] = st_h[1]; }
        else if (tid >= 128) vc[cache_row + tid - 128] = st_h[0];
    }
    if constexpr (MERGE) {
        if (!att_merge_arrive(counters + kv_head, nsplit, tid, reinterpret_cast<volatile int*>(qs))) return;   // (qs: the queries, read long ago)
        for (int n = wave; n < group; n += 4) {   // a wave per query head of the group, both halves of its 128 elements
            const int head = kv_head * group + n;
            att_merge_head_wave<2>(output + (size_t)head * AM_HD, part + (size_t)head * nsplit * (AM_HD + 2), AM_HD, nsplit, lane, lane);
        }
    }
}

// head_dim 128, <= 16 query heads per KV head, 16-byte aligned caches (caller-checked): partial states for attention_split_combine_kernel
int launch_attention_decode_kvhead_mfma(float* part, const float* q, const float* k, const float* v, uint16_t* kc, uint16_t* vc,
                                        const int* d_pos, const float* inv_freq, int nh, int nkv, int max_seq, float scale, float theta,
                                        float fscale, int nsplit, float* merged_output, unsigned* counters, hipStream_t st) {
    if (nh % nkv != 0 || nh / nkv > 16 || nsplit < 1) return NTK_E_SHAPE;
    if (merged_output)
        hipLaunchKernelGGL(attention_decode_kvhead_mfma_kernel<true>, dim3(nkv, nsplit), dim3(256), 0, st, part, q, k, v, kc, vc, d_pos, inv_freq,
                           nh, nkv, max_seq, scale, theta, fscale, merged_output, counters);
    else
        hipLaunchKernelGGL(attention_decode_kvhead_mfma_kernel<false>, dim3(nkv, nsplit), dim3(256), 0, st, part, q, k, v, kc, vc, d_pos, inv_freq,
                           nh, nkv, max_seq, scale, theta, fscale, nullptr, nullptr);
    return last_launch_status();
}

}  // namespace ntk
