// gemm_bf16.hip -- batched prompt projections on the BF16 matrix cores, 64 tokens per pass over the weights.
//
// Replaces the reference's prefill, which runs launch_gemv once per prompt token and matrix
// (reference src/model/attention.cpp:144-162,200-210, src/model/ffn.cpp:96-133), and supersedes gemm_prefill.hip's F32-MFMA
// form (16 tokens per pass, 29-65 TFLOP/s of the 157 TFLOP/s F32 matrix rate) for the formats of the target models.
//
// Arithmetic -- exact products, F32 accumulation, like the GEMV:
//   * a GGUF weight is (integer) x (scale): the INTEGER part (Q8_0: -128..127, Q4_K: 0..15, Q6_K: -32..31) is exact in BF16;
//   * an F32 activation is split into three BF16 pieces x = x1 + x2 + x3 (8 + 8 + 8 mantissa bits, exact), so
//     sum_k q_k x_k = sum over pieces of BF16 x BF16 products, each exact in the F32 accumulator of v_mfma_f32_16x16x32_bf16;
//   * the per-block scale (FP16 d, 6-bit K-quant sub-scales, Q6_K's int8 sub-scales per 16 columns) multiplies the F32 block
//     sum afterwards, and the K-quant minimum enters as -dmin*m * sum_k x_k -- the same factorisation as reference
//     gemm.cu:129-141, 190-244, 421-459.  No activation or weight is rounded; only the summation order differs from the GEMV.
// Three BF16 MFMAs (K = 32) replace eight F32 MFMAs (K = 4) per 32 columns and 16 tokens: ~5x less matrix-pipe time, and 64
// tokens share every dequantised weight instead of 16.
//
// Decomposition (gfx950, wave64):
//   * pre-pass (split_x_kernel): X[T,in] F32 -> three BF16 planes in MFMA operand order, 1 KiB per (32-column step, plane,
//     16-token block), plus the per-(step, token) sums of x the K-quant minimum term needs; written once per GEMM, read from
//     L2 by every workgroup;
//   * main kernel: workgroup = 4 waves, wave = 16*RT output rows x 64 tokens (RT*4 accumulator tiles of 16x16); B operands of
//     two 32-column steps (24 KB) are staged through a double-buffered LDS ring by the whole workgroup and read by its 4 waves;
//     A operands come straight from the raw GGUF rows (a lane reads the 8 weights of one row that its MFMA slot needs, 2-byte
//     aligned loads, one step ahead of the MFMAs);
//   * MFMA operand slots: lane (i = lane % 16, g = lane / 16) holds columns {4g..4g+3} and {16+4g..16+4g+3} of the 32-column
//     step for row / token i -- the same permutation on A and B, so the dot product is unchanged, and the two halves are the
//     two 16-column sub-scale groups of Q6_K (which uses two K = 16 MFMAs per step).
// Bound: MFMA (BF16 dense 2.5 PFLOP/s, three products per weight-token pair).
#include "common.hip.h"
#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace ntk {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int GB_TOK = 64;           // tokens per pass
constexpr int GB_CH = 2;             // 32-column steps per LDS chunk
constexpr int GB_PIECE = 1024;       // bytes of one (step, plane, token block) operand record
constexpr int GB_STEP_BYTES = 3 * 4 * GB_PIECE;           // 12 KB of B operands per step
constexpr int GB_CHUNK_BYTES = GB_CH * GB_STEP_BYTES;      // 24 KB

__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {   // upper halves of two floats (exact for our small integers)
    return __builtin_amdgcn_perm(__float_as_uint(hi), __float_as_uint(lo), 0x07060302u);
}
typedef uint32_t __attribute__((aligned(2))) u32_a2;
typedef uint16_t __attribute__((aligned(2))) u16_a2;
__device__ __forceinline__ uint32_t ld32(const uint8_t* p) { return *reinterpret_cast<const u32_a2*>(p); }   // 2-byte aligned global dword
__device__ __forceinline__ uint16_t ld16(const uint8_t* p) { return *reinterpret_cast<const u16_a2*>(p); }

// ---- pre-pass: X -> BF16 planes in operand order + per-step sums ----------------------------------------------------------
// grid = in / 32 steps, block = 256 = 4 token blocks x 64 lanes
__global__ __launch_bounds__(256) void split_x_kernel(const float* __restrict__ X, int T, int in, u32x4* __restrict__ xb, float* __restrict__ xsum) {
    const int step = blockIdx.x, tb = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int j = lane & 15, g = lane >> 4, t = tb * 16 + j;
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = 0.0f;
    if (t < T) {
        const float* row = X + (size_t)t * in + step * 32 + 4 * g;
        const float4 a = *reinterpret_cast<const float4*>(row), b = *reinterpret_cast<const float4*>(row + 16);
        x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
    }
    float p1[8], p2[8], p3[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {   // truncation split: every difference is exact, the third piece holds the last <= 8 bits
        p1[e] = __uint_as_float(__float_as_uint(x[e]) & 0xFFFF0000u);
        const float r1 = x[e] - p1[e];
        p2[e] = __uint_as_float(__float_as_uint(r1) & 0xFFFF0000u);
        p3[e] = r1 - p2[e];
    }
    const size_t base = ((size_t)step * 3 * 4 + tb) * 64 + lane;   // plane p at + p * 4 * 64
    xb[base] = u32x4{pack_bf16(p1[0], p1[1]), pack_bf16(p1[2], p1[3]), pack_bf16(p1[4], p1[5]), pack_bf16(p1[6], p1[7])};
    xb[base + 256] = u32x4{pack_bf16(p2[0], p2[1]), pack_bf16(p2[2], p2[3]), pack_bf16(p2[4], p2[5]), pack_bf16(p2[6], p2[7])};
    xb[base + 512] = u32x4{pack_bf16(p3[0], p3[1]), pack_bf16(p3[2], p3[3]), pack_bf16(p3[4], p3[5]), pack_bf16(p3[6], p3[7])};
    // sum of the step's 32 activations of token t (fixed order: the lane's 8 in sequence, then the 4 column groups)
    float s = ((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7]));
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    if (g == 0) xsum[(size_t)step * GB_TOK + t] = s;
}

// ---- per-format A operand: the 8 integer weights of (row, 32-column step, slot group g) as BF16 + the block's scales ------
// load(): the raw dwords of the slot (issued GB_AD steps ahead of their use: the weights come from HBM); decode(): BF16 + scales
struct AOp {
    u32x4 a;            // 8 BF16 integers
    float s0, s1;       // scale of the slot's low / high 4 columns (equal unless the format scales per 16 columns)
    float mn;           // K-quant minimum term factor (dmin * m), 0 otherwise
};
template <int DT> struct DeqI;

template <> struct DeqI<NTK_DT_Q8_0> {   // types.h:104-108: half d, int8 qs[32]
    static constexpr int BW = 32, BB = 34;
    static constexpr bool SPLIT16 = false, HAS_MIN = false;
    struct Raw { uint32_t lo, hi, d; };
    __device__ static Raw load(const uint8_t* row, int step, int g) {
        const uint8_t* p = row + 34 * step;
        return Raw{ld32(p + 2 + 4 * g), ld32(p + 18 + 4 * g), (uint32_t)ld16(p)};
    }
    __device__ static AOp decode(const Raw& r, int /*step*/) {
        AOp o;
        o.a = u32x4{pack_bf16(sb2f(r.lo, 0), sb2f(r.lo, 1)), pack_bf16(sb2f(r.lo, 2), sb2f(r.lo, 3)),
                    pack_bf16(sb2f(r.hi, 0), sb2f(r.hi, 1)), pack_bf16(sb2f(r.hi, 2), sb2f(r.hi, 3))};
        o.s0 = o.s1 = h2f((uint16_t)r.d);
        o.mn = 0.0f;
        return o;
    }
};

template <> struct DeqI<NTK_DT_Q4_K> {   // types.h:112-117: half d, dmin; 12 packed 6-bit (scale, min); 128 bytes of nibbles
    static constexpr int BW = 256, BB = 144;
    static constexpr bool SPLIT16 = false, HAS_MIN = true;
    struct Raw { uint32_t h0, s0, s1, s2, lo, hi; };
    __device__ static Raw load(const uint8_t* row, int step, int g) {
        const uint8_t* p = row + 144 * (step >> 3);
        const uint8_t* q = p + 16 + 32 * ((step & 7) >> 1);
        return Raw{ld32(p), ld32(p + 4), ld32(p + 8), ld32(p + 12), ld32(q + 4 * g), ld32(q + 16 + 4 * g)};
    }
    __device__ static AOp decode(const Raw& r, int step) {
        const int j = step & 7;
        float sc, mn;
        kq_scale_min(r.s0, r.s1, r.s2, j, sc, mn);                            // gemm.cu:206-222
        const float d = h2f((uint16_t)(r.h0 & 0xFFFFu)), dmin = h2f((uint16_t)(r.h0 >> 16));
        const int sh = 4 * (j & 1);                                           // even sub-block: low nibbles, odd: high
        const uint32_t lo = (r.lo >> sh) & 0x0F0F0F0Fu, hi = (r.hi >> sh) & 0x0F0F0F0Fu;
        AOp o;
        o.a = u32x4{pack_bf16(ub2f(lo, 0), ub2f(lo, 1)), pack_bf16(ub2f(lo, 2), ub2f(lo, 3)),
                    pack_bf16(ub2f(hi, 0), ub2f(hi, 1)), pack_bf16(ub2f(hi, 2), ub2f(hi, 3))};
        o.s0 = o.s1 = d * sc;
        o.mn = dmin * mn;
        return o;
    }
};

template <> struct DeqI<NTK_DT_Q6_K> {   // types.h:132-137: ql[128], qh[64], int8 scales[16], half d
    static constexpr int BW = 256, BB = 210;
    static constexpr bool SPLIT16 = true, HAS_MIN = false;
    struct Raw { uint32_t l0, l1, h0, h1, scd; };
    __device__ static Raw load(const uint8_t* row, int step, int g) {
        const uint8_t* p = row + 210 * (step >> 3);
        const int j = step & 7, hf = j >> 2, t = j & 3;
        const uint8_t* ql = p + 64 * hf + 32 * (t & 1);
        const uint8_t* qh = p + 128 + 32 * hf;
        return Raw{ld32(ql + 4 * g), ld32(ql + 16 + 4 * g), ld32(qh + 4 * g), ld32(qh + 16 + 4 * g),
                   (uint32_t)ld16(p + 192 + 8 * hf + 2 * t) | ((uint32_t)ld16(p + 208) << 16)};   // two int8 sub-scales | half d
    }
    __device__ static AOp decode(const Raw& r, int step) {
        const int t = step & 3;
        const int sl = 4 * (t >> 1), sh = 2 * t;
        const uint32_t lo = ((r.l0 >> sl) & 0x0F0F0F0Fu) | (((r.h0 >> sh) & 0x03030303u) << 4);   // gemm.cu:421-459
        const uint32_t hi = ((r.l1 >> sl) & 0x0F0F0F0Fu) | (((r.h1 >> sh) & 0x03030303u) << 4);
        AOp o;   // q - 32: exact small integers
        o.a = u32x4{pack_bf16(ub2f(lo, 0) - 32.0f, ub2f(lo, 1) - 32.0f), pack_bf16(ub2f(lo, 2) - 32.0f, ub2f(lo, 3) - 32.0f),
                    pack_bf16(ub2f(hi, 0) - 32.0f, ub2f(hi, 1) - 32.0f), pack_bf16(ub2f(hi, 2) - 32.0f, ub2f(hi, 3) - 32.0f)};
        const float d = h2f((uint16_t)(r.scd >> 16));
        o.s0 = d * (float)(int)(int8_t)(r.scd & 0xFF);
        o.s1 = d * (float)(int)(int8_t)((r.scd >> 8) & 0xFF);
        o.mn = 0.0f;
        return o;
    }
};

constexpr int GB_AD = 4;   // weight slots are requested this many steps ahead of the MFMAs that consume them (2 LDS chunks)

struct GemmBParams {
    const uint8_t* W;
    const uint8_t* xb;      // operand planes [steps][3][4][64] x 16 B
    const float* xsum;      // [steps][64]
    float* Y;               // [T][out]
    const float* resid;     // optional [T][out], may alias Y
    int T, out, in, steps;
    unsigned row_bytes;
    int nsplit, steps_per_split;   // blockIdx.y = K split; nsplit > 1: partial sums go to `part` [split][T][out], summed by reduce_splits
    float* part;
};

// block = 256 threads = 4 waves; wave w of workgroup b: rows (4 b + w) * 16 RT ...
template <int DT, int RT>
__global__ __launch_bounds__(256, 2) void gemm_quant_bf16_kernel(const GemmBParams p) {
    using D = DeqI<DT>;
    extern __shared__ __attribute__((aligned(16))) uint8_t gb_lds[];
    uint8_t* bbuf = gb_lds;                                             // [2][GB_CHUNK_BYTES]
    float* xs = reinterpret_cast<float*>(gb_lds + 2 * GB_CHUNK_BYTES);  // [2][GB_CH][64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, g = lane >> 4;
    const int row0 = ((int)blockIdx.x * 4 + wave) * 16 * RT;
    const int ntb = (p.T + 15) >> 4;                                    // token blocks in use (wave-uniform)
    const uint8_t* arow[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) arow[rt] = p.W + (size_t)min(row0 + rt * 16 + i, p.out - 1) * p.row_bytes;
    f32x4 acc[RT][4];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int tb = 0; tb < 4; ++tb) acc[rt][tb] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    // this workgroup's K range (split-K: blockIdx.y), in steps of 32 columns; chunks of GB_CH steps counted from its start
    const int step_lo = (int)blockIdx.y * p.steps_per_split, step_hi = min(p.steps, step_lo + p.steps_per_split);
    const int nsteps = step_hi - step_lo;
    const int nchunks = (nsteps + GB_CH - 1) / GB_CH;
    // workgroup-cooperative B staging: a chunk is 24 KB contiguous in xb = 6 x 16 B per thread.  Two register sets: the chunk
    // after next is requested while the next one waits in registers for its LDS buffer (the planes come from L2, ~1 us away:
    // one chunk of MFMA work does not cover that)
    u32x4 bq[2][6];
    float xsq[2] = {0.0f, 0.0f};
    auto fetch_chunk = [&](int c, auto set_tag) {
        constexpr int SET = decltype(set_tag)::value;
        const u32x4* src = reinterpret_cast<const u32x4*>(p.xb + ((size_t)step_lo / GB_CH + c) * GB_CHUNK_BYTES);
        const int valid = (c < nchunks ? min(GB_CH, nsteps - c * GB_CH) : 0) * (GB_STEP_BYTES / 16);   // 16-byte pieces that exist
#pragma unroll
        for (int k = 0; k < 6; ++k) bq[SET][k] = (tid + 256 * k < valid) ? src[tid + 256 * k] : u32x4{0, 0, 0, 0};
        if (D::HAS_MIN && tid < GB_CH * 64) xsq[SET] = (tid + 0 < valid / (GB_STEP_BYTES / 16) * 64) ? p.xsum[((size_t)step_lo + (size_t)c * GB_CH) * 64 + tid] : 0.0f;
    };
    auto store_chunk = [&](int buf, auto set_tag) {
        constexpr int SET = decltype(set_tag)::value;
        u32x4* dst = reinterpret_cast<u32x4*>(bbuf + (size_t)buf * GB_CHUNK_BYTES);
#pragma unroll
        for (int k = 0; k < 6; ++k) dst[tid + 256 * k] = bq[SET][k];
        if (D::HAS_MIN && tid < GB_CH * 64) xs[buf * GB_CH * 64 + tid] = xsq[SET];
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    fetch_chunk(0, S0{});
    store_chunk(0, S0{});
    fetch_chunk(1, S1{});
    __syncthreads();

    typename D::Raw raw[GB_AD][RT];
#pragma unroll
    for (int u = 0; u < GB_AD; ++u)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) raw[u][rt] = D::load(arow[rt], min(step_lo + u, step_hi - 1), g);

    // steps in groups of GB_AD (= two LDS chunks): slot u of the register ring is consumed at step 4 grp + u and refilled with
    // the weights of step 4 (grp + 1) + u -- four steps (~one HBM round trip of MFMA work) ahead
    for (int grp = 0; grp * GB_AD < nsteps; ++grp) {
#pragma unroll
        for (int u = 0; u < GB_AD; ++u) {
            const int rel = grp * GB_AD + u;                           // step relative to the split's start
            if (rel >= nsteps) break;
            const int step = step_lo + rel;
            const int c = rel / GB_CH, sc = rel % GB_CH, buf = c & 1;   // (GB_AD = 2 GB_CH: chunk parity == u / GB_CH, compile time)
            if (sc == 0) {                                             // the chunk after next: requested now, into the set chunk c came from
                if (u / GB_CH == 0) fetch_chunk(c + 2, S0{}); else fetch_chunk(c + 2, S1{});
            }
            AOp a[RT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                a[rt] = D::decode(raw[u][rt], step);
                raw[u][rt] = D::load(arow[rt], min(step + GB_AD, step_hi - 1), g);
            }
            const u32x4* bs = reinterpret_cast<const u32x4*>(bbuf + (size_t)buf * GB_CHUNK_BYTES + (size_t)sc * GB_STEP_BYTES);
            // scales of the 4 accumulator rows this lane holds (rows 4g + e of the tile), from the lanes that decoded those rows
            float s0[RT][4], s1[RT][4], mn[RT][4];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    s0[rt][e] = __shfl(a[rt].s0, 4 * g + e, 64);
                    if (D::SPLIT16) s1[rt][e] = __shfl(a[rt].s1, 4 * g + e, 64);
                    if (D::HAS_MIN) mn[rt][e] = __shfl(a[rt].mn, 4 * g + e, 64);
                }
#pragma unroll
            for (int tb = 0; tb < 4; ++tb) {
                if (tb >= ntb) break;
                const u32x4 b0 = bs[(0 * 4 + tb) * 64 + lane], b1 = bs[(1 * 4 + tb) * 64 + lane], b2 = bs[(2 * 4 + tb) * 64 + lane];
                const float xsum_t = D::HAS_MIN ? xs[(buf * GB_CH + sc) * 64 + tb * 16 + i] : 0.0f;   // token = this lane's column
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    if constexpr (D::SPLIT16) {   // two 16-column groups with their own scale: K = 16 MFMAs on the operand halves
                        const s16x4 al = __builtin_bit_cast(s16x4, (uint64_t)a[rt].a.x | ((uint64_t)a[rt].a.y << 32));
                        const s16x4 ah = __builtin_bit_cast(s16x4, (uint64_t)a[rt].a.z | ((uint64_t)a[rt].a.w << 32));
                        f32x4 cl = {0.0f, 0.0f, 0.0f, 0.0f}, ch = {0.0f, 0.0f, 0.0f, 0.0f};
                        const u32x4 bb[3] = {b0, b1, b2};
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl) {
                            const s16x4 bl = __builtin_bit_cast(s16x4, (uint64_t)bb[pl].x | ((uint64_t)bb[pl].y << 32));
                            const s16x4 bh = __builtin_bit_cast(s16x4, (uint64_t)bb[pl].z | ((uint64_t)bb[pl].w << 32));
                            cl = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(al, bl, cl, 0, 0, 0);
                            ch = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, bh, ch, 0, 0, 0);
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[rt][tb][e] = fmaf(s1[rt][e], ch[e], fmaf(s0[rt][e], cl[e], acc[rt][tb][e]));
                    } else {
                        const bf16x8 av = __builtin_bit_cast(bf16x8, a[rt].a);
                        f32x4 cc = {0.0f, 0.0f, 0.0f, 0.0f};
                        cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, __builtin_bit_cast(bf16x8, b0), cc, 0, 0, 0);
                        cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, __builtin_bit_cast(bf16x8, b1), cc, 0, 0, 0);
                        cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, __builtin_bit_cast(bf16x8, b2), cc, 0, 0, 0);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float v = fmaf(s0[rt][e], cc[e], acc[rt][tb][e]);
                            if (D::HAS_MIN) v = fmaf(-mn[rt][e], xsum_t, v);   // - dmin * m * sum x   (gemm.cu:232-244)
                            acc[rt][tb][e] = v;
                        }
                    }
                }
            }
            if (sc == GB_CH - 1 || rel + 1 == nsteps) {
                // buffer buf^1 was last read during chunk c-1, which every wave left through the barrier below: free to refill
                // with chunk c+1, which has been waiting in the other register set since the start of chunk c-1
                if (c + 1 < nchunks) { if (u / GB_CH == 0) store_chunk(buf ^ 1, S1{}); else store_chunk(buf ^ 1, S0{}); }
                __syncthreads();
            }
        }
    }
    // ---- epilogue: accumulator element e of lane (i = token column, g) is row 4g + e of the tile -----------------------
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int r = row0 + rt * 16 + 4 * g;
        if (r >= p.out) continue;
#pragma unroll
        for (int tb = 0; tb < 4; ++tb) {
            const int t = tb * 16 + i;
            if (t >= p.T) continue;
            float4 v = {acc[rt][tb][0], acc[rt][tb][1], acc[rt][tb][2], acc[rt][tb][3]};
            if (p.nsplit > 1) {   // K split: this workgroup's partial sums, combined (fixed order) by reduce_splits_kernel
                *reinterpret_cast<float4*>(p.part + ((size_t)blockIdx.y * GB_TOK + t) * p.out + r) = v;
                continue;
            }
            float* y = p.Y + (size_t)t * p.out + r;
            if (p.resid) {
                const float4 rs = *reinterpret_cast<const float4*>(p.resid + (size_t)t * p.out + r);
                v.x += rs.x; v.y += rs.y; v.z += rs.z; v.w += rs.w;
            }
            *reinterpret_cast<float4*>(y) = v;
        }
    }
}

// Y[t][r] = sum over splits (in order) of part[s][t][r] (+ resid): one float4 per thread
__global__ __launch_bounds__(256) void reduce_splits_kernel(float* __restrict__ Y, const float* __restrict__ part, const float* __restrict__ resid,
                                                            int T, int out, int nsplit) {
    const size_t idx = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (idx >= (size_t)T * out) return;
    const size_t t = idx / out, r = idx % out;
    float4 v = *reinterpret_cast<const float4*>(part + t * out + r);
    for (int s = 1; s < nsplit; ++s) {
        const float4 a = *reinterpret_cast<const float4*>(part + ((size_t)s * GB_TOK + t) * out + r);
        v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
    }
    if (resid) {
        const float4 a = *reinterpret_cast<const float4*>(resid + idx);
        v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
    }
    *reinterpret_cast<float4*>(Y + idx) = v;
}

static size_t ws_planes_bytes(int in) { return (size_t)(in / 32) * GB_STEP_BYTES + (size_t)(in / 32) * GB_TOK * sizeof(float) + 256; }
constexpr int GB_MAX_SPLIT = 8;

template <int DT>
static int launch_gemm_bf16(float* Y, const void* W, const float* X, int T, int out, int in, const float* resid, void* ws, int reuse_x,
                            hipStream_t st) {
    using D = DeqI<DT>;
    if (in % D::BW != 0 || in % 32 != 0 || out % 16 != 0) return NTK_E_SHAPE;
    if ((reinterpret_cast<uintptr_t>(W) & 1) || (reinterpret_cast<uintptr_t>(X) & 15) || (reinterpret_cast<uintptr_t>(Y) & 15) ||
        (resid && (reinterpret_cast<uintptr_t>(resid) & 15)))
        return NTK_E_ALIGN;
    GemmBParams p{};
    p.W = static_cast<const uint8_t*>(W);
    p.T = T; p.out = out; p.in = in; p.steps = in / 32;
    p.row_bytes = (unsigned)((size_t)in / D::BW * D::BB);
    uint8_t* wsb = static_cast<uint8_t*>(ws);
    p.xb = wsb;
    p.xsum = reinterpret_cast<const float*>(wsb + (size_t)p.steps * GB_STEP_BYTES);
    p.part = reinterpret_cast<float*>(wsb + (ws_planes_bytes(in) + 255) / 256 * 256);
    p.Y = Y; p.resid = resid;
    if (!reuse_x)
        hipLaunchKernelGGL(split_x_kernel, dim3(p.steps), dim3(256), 0, st, X, T, in, reinterpret_cast<u32x4*>(wsb), const_cast<float*>(p.xsum));
    const size_t lds = (size_t)2 * GB_CHUNK_BYTES + (size_t)2 * GB_CH * 64 * sizeof(float);
    // 32 rows per wave when that still leaves >= 256 workgroups (the B operands are then read from LDS half as often per MFMA)
    const bool rt2 = out >= 256 * 128 / 2;
    const int row_wgs = rt2 ? (out + 127) / 128 : (out + 63) / 64;
    // K split: small matrices leave most CUs idle and a workgroup walks its K range serially (latency-bound): split K until the
    // grid has ~2 workgroups per CU, in whole groups of GB_AD steps
    int nsplit = 1;
    while (nsplit < GB_MAX_SPLIT && row_wgs * nsplit < 512 && p.steps / (nsplit * 2) >= 4 * GB_AD) nsplit *= 2;
    p.nsplit = nsplit;
    p.steps_per_split = ((p.steps + nsplit - 1) / nsplit + GB_AD - 1) / GB_AD * GB_AD;
    const dim3 grid(row_wgs, nsplit);
    if (rt2) hipLaunchKernelGGL((gemm_quant_bf16_kernel<DT, 2>), grid, dim3(256), lds, st, p);
    else hipLaunchKernelGGL((gemm_quant_bf16_kernel<DT, 1>), grid, dim3(256), lds, st, p);
    if (nsplit > 1) {
        const size_t n4 = ((size_t)T * out + 3) / 4;
        hipLaunchKernelGGL(reduce_splits_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, Y, (const float*)p.part, resid, T, out, nsplit);
    }
    return last_launch_status();
}

}  // namespace ntk

extern "C" {

size_t ntk_gemm_quant_workspace_bytes(int in_features, int out_features) {
    if (in_features <= 0 || out_features < 0) return 0;
    return (ntk::ws_planes_bytes((in_features + 31) / 32 * 32) + 255) / 256 * 256 +
           (size_t)ntk::GB_MAX_SPLIT * ntk::GB_TOK * (size_t)out_features * sizeof(float) + 256;
}

int ntk_gemm_quant_ws(float* Y, const void* W, const float* X, int n_tokens, int out_features, int in_features, int weight_dtype,
                      const float* resid, void* workspace, size_t workspace_bytes, int reuse_x, void* stream) {
    if (!Y || !W || !X || !workspace) return NTK_E_NULL;
    if (n_tokens < 0 || out_features < 0 || in_features <= 0) return NTK_E_SHAPE;
    if (workspace_bytes < ntk_gemm_quant_workspace_bytes(in_features, out_features) || (reinterpret_cast<uintptr_t>(workspace) & 15)) return NTK_E_SHAPE;
    if (weight_dtype != NTK_DT_Q8_0 && weight_dtype != NTK_DT_Q4_K && weight_dtype != NTK_DT_Q6_K) return NTK_E_DTYPE;
    if (n_tokens == 0 || out_features == 0) return NTK_OK;
    if (n_tokens > ntk::GB_TOK) reuse_x = 0;   // the planes hold one 64-token chunk at a time
    hipStream_t st = ntk::resolve_stream(stream);
    for (int t0 = 0; t0 < n_tokens; t0 += ntk::GB_TOK) {   // 64 tokens per pass over W
        const int T = std::min(ntk::GB_TOK, n_tokens - t0);
        float* y = Y + (size_t)t0 * out_features;
        const float* x = X + (size_t)t0 * in_features;
        const float* rs = resid ? resid + (size_t)t0 * out_features : nullptr;
        int rc;
        switch (weight_dtype) {
            case NTK_DT_Q8_0: rc = ntk::launch_gemm_bf16<NTK_DT_Q8_0>(y, W, x, T, out_features, in_features, rs, workspace, reuse_x, st); break;
            case NTK_DT_Q4_K: rc = ntk::launch_gemm_bf16<NTK_DT_Q4_K>(y, W, x, T, out_features, in_features, rs, workspace, reuse_x, st); break;
            default: rc = ntk::launch_gemm_bf16<NTK_DT_Q6_K>(y, W, x, T, out_features, in_features, rs, workspace, reuse_x, st); break;
        }
        if (rc != NTK_OK) return rc;
    }
    return NTK_OK;
}

}  // extern "C"
