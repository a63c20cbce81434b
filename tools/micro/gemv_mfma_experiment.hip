// gemv_mfma.hip -- batch-1 GEMV for the K-quant formats with the dot products on the BF16 matrix cores.
//
// Same operator as gemv.hip's gemv_quant_kernel (reference src/cuda/gemm.cu:158-255 Q4_K, :387-470 Q6_K): y = W . x with the
// weights in their GGUF block encoding, F32 activations, F32 accumulation.  Why a second kernel: the 4/6-bit decoders of
// gemv.hip are VALU-bound (84 % of the SIMDs' issue cycles: one v_cvt_f32_ubyte + half a v_pk_fma_f32 per weight, plus the
// nibble masks), so the K-quant launches reach 5.0-5.3 TB/s where Q8_0 reaches 6.45.  Here:
//   * a weight never becomes a float.  The 4-bit code n is turned into the BF16 number 16 + n by BIT operations
//     (0x4180 | n << 3: exponent of 16, n in the mantissa; Q6_K: 64 + q = 0x4280 | q << 1) -- two bit ops per dword of four
//     codes and one v_perm_b32 per pair -- and the offset leaves through the block's sum of x, which the K-quant minimum
//     term needs anyway:  d*sc * sum (16+n) x  -  (16 d*sc + dmin*m) * sum x;
//   * the products run on v_mfma_f32_16x16x32_bf16 with the activation split into three exact BF16 pieces (x = x1+x2+x3,
//     truncation split: BF16 x BF16 products are exact in the F32 accumulator), three chained MFMAs per 32-column sub-block
//     and 16 rows.  All 16 "token" rows of the A operand carry the same x, so every lane of a weight row's lane group ends
//     up with the sub-block's sum -- no zero padding, no masks;
//   * the per-(row, sub-block) scale work is spread over the four lanes (i, g = 0..3) that hold a weight row: lane g owns
//     sub-blocks g, 4 + g of the super-block, decodes only their 6-bit (scale, min) pairs and folds only their sums; the
//     four partial row sums meet once per row tile.
// Decomposition: workgroup = 8 waves sharing the activation planes in LDS (192 B per 32-column step: 3 pieces x 4 column
// groups x 8 BF16, + the steps' sums of x); a wave owns tiles of 32 rows (two MFMA row tiles) and a K range -- `ks` waves
// share a tile when the matrix has too few rows to occupy the chip, their partial sums meet in LDS in wave order
// (deterministic).  Weight bytes travel as in gemm_bf16.hip: 16-byte pieces of whole super-blocks, 9 (Q4_K) / 14 (Q6_K)
// consecutive lanes per row, two units in flight per wave in registers, a per-wave LDS image with a conflict-free pitch.
// Bound: HBM (algorithmic bytes = rows x row_bytes, as gemv.hip).
#include "common.hip.h"
#include <algorithm>
#include <cstdlib>

namespace ntk {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t __attribute__((aligned(2))) gm_u32_a2;
typedef uint16_t __attribute__((aligned(2))) gm_u16_a2;
__device__ __forceinline__ uint32_t gm_lds32(const uint8_t* p) { return *reinterpret_cast<const gm_u32_a2*>(p); }
__device__ __forceinline__ uint32_t gm_lds16(const uint8_t* p) { return *reinterpret_cast<const gm_u16_a2*>(p); }

constexpr int GM_WAVES = 8;     // waves per workgroup
constexpr int GM_RT = 2;        // MFMA row tiles per wave: 32 rows share every x operand read
constexpr int GM_ROWS = 16 * GM_RT;
constexpr int GM_NR = 2;        // super-blocks in flight per wave (register ring)
constexpr int GM_XSTEP = 192;   // bytes of activation operands per 32-column step: [piece 3][column group 4][8 BF16]

struct GmParams {
    const uint8_t* W;
    float* y;
    const float* x;
    const float* norm_w;   // RMSNorm prologue when non-null
    const float* resid;
    float eps;
    int out, in, steps;    // steps = in / 32
    unsigned row_bytes, w_last;
    int ks;                // waves sharing one 32-row tile (K split): 1, 2, 4 or 8
    int ntiles;            // ceil(out / 32)
};

__device__ __forceinline__ uint32_t gm_pack_bf16(float lo, float hi) {
    return __builtin_amdgcn_perm(__float_as_uint(hi), __float_as_uint(lo), 0x07060302u);
}

// ---- per-format pieces --------------------------------------------------------------------------------------------------
template <int DT> struct GmFmt;

template <> struct GmFmt<NTK_DT_Q4_K> {   // types.h:112-117: half d, dmin; 12 packed 6-bit (scale, min); 128 bytes of nibbles
    static constexpr int BW = 256, BB = 144, NCH = 9, STRIDE = 144, NSUM = 1;
    struct Hdr { u32x4 h; };
    __device__ static Hdr header(const uint8_t* row) { return Hdr{*reinterpret_cast<const u32x4*>(row)}; }
    // the 8 BF16 numbers 16 + n of (row, sub-block j, column group g): columns {4g..4g+3, 16+4g..16+4g+3} of the sub-block
    __device__ static u32x4 slot(const uint8_t* rowg, int j) {
        const uint32_t a = gm_lds32(rowg + 16 + 32 * (j >> 1)), b = gm_lds32(rowg + 32 + 32 * (j >> 1));
        uint32_t ta, tb;   // byte k = 0x80 | n_k << 3
        if (j & 1) { ta = ((a >> 1) & 0x78787878u) | 0x80808080u; tb = ((b >> 1) & 0x78787878u) | 0x80808080u; }   // high nibbles
        else { ta = ((a << 3) & 0x78787878u) | 0x80808080u; tb = ((b << 3) & 0x78787878u) | 0x80808080u; }
        const uint32_t k41 = 0x41414141u;
        return u32x4{__builtin_amdgcn_perm(k41, ta, 0x04010400u), __builtin_amdgcn_perm(k41, ta, 0x04030402u),
                     __builtin_amdgcn_perm(k41, tb, 0x04010400u), __builtin_amdgcn_perm(k41, tb, 0x04030402u)};
    }
    // lane g's sub-block j = 4 h + g: A = d * sc, B = 16 * d * sc + dmin * m   (gemm.cu:206-222, 232-244)
    __device__ static void scales(const Hdr& hd, int h, int g, float& A, float& B) {
        const uint32_t s0 = hd.h.y, s1 = hd.h.z, s2 = hd.h.w, sh = 8u * (uint32_t)g;
        uint32_t sc, mn;
        if (h == 0) {
            sc = (s0 >> sh) & 63u;
            mn = (s1 >> sh) & 63u;
        } else {
            const uint32_t hi = s2 >> sh;
            sc = (hi & 0xFu) | (((s0 >> sh) >> 2) & 0x30u);
            mn = ((hi >> 4) & 0xFu) | (((s1 >> sh) >> 2) & 0x30u);
        }
        const float d = h2f((uint16_t)(hd.h.x & 0xFFFFu)), dmin = h2f((uint16_t)(hd.h.x >> 16));
        A = d * (float)sc;
        B = fmaf(16.0f, A, dmin * (float)mn);
    }
};

// ---- the kernel ------------------------------------------------------------------------------------------------------------
template <int DT, bool NORM, int ABL = 0>
__global__ __launch_bounds__(64 * GM_WAVES, 1) void gemv_mfma_kernel(const GmParams p) {
    using F = GmFmt<DT>;
    constexpr int NCH = F::NCH, STRIDE = F::STRIDE;
    constexpr int PIECES = GM_ROWS * NCH, NLD = (PIECES + 63) / 64;
    extern __shared__ __attribute__((aligned(16))) uint8_t gm_lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, g = lane >> 4;
    // LDS: activation operands | step sums | reduction scratch | K-split partials | the waves' weight images
    uint8_t* xop = gm_lds;
    float* xsum = reinterpret_cast<float*>(gm_lds + (size_t)p.steps * GM_XSTEP);
    float* red = xsum + (size_t)p.steps * F::NSUM;
    float* kpart = red + 16;
    uint8_t* stage = reinterpret_cast<uint8_t*>(kpart + GM_WAVES * GM_ROWS) + (size_t)wave * (GM_ROWS * STRIDE);

    // ---- prologue: x (RMSNorm'ed) -> three exact BF16 pieces in MFMA operand order + the steps' sums of x --------------
    {
        const int ngroups = p.steps * 4;   // (step, column group): 8 activations each
        float ssq = 0.0f;
        if (NORM) {
            for (int q = tid; q < ngroups; q += 64 * GM_WAVES) {
                const float* xp = p.x + 32 * (q >> 2) + 4 * (q & 3);
                const float4 a = *reinterpret_cast<const float4*>(xp), b = *reinterpret_cast<const float4*>(xp + 16);
                ssq += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w + b.x * b.x + b.y * b.y + b.z * b.z + b.w * b.w;
            }
            ssq = block_sum(ssq, red);
        }
        const float rms_inv = NORM ? 1.0f / sqrtf(ssq / (float)p.in + p.eps) : 1.0f;   // rsqrtf(mean + eps), rmsnorm.cu:60-61
        for (int q0 = 0; q0 < ngroups; q0 += 64 * GM_WAVES) {
            const int q = q0 + tid;
            float x[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = 0.0f;
            if (q < ngroups) {
                const int off = 32 * (q >> 2) + 4 * (q & 3);
                const float4 a = *reinterpret_cast<const float4*>(p.x + off), b = *reinterpret_cast<const float4*>(p.x + off + 16);
                x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
                if (NORM) {   // x * rms_inv * w, the reference's association (rmsnorm.cu:68)
                    const float4 wa = *reinterpret_cast<const float4*>(p.norm_w + off), wb = *reinterpret_cast<const float4*>(p.norm_w + off + 16);
                    x[0] = x[0] * rms_inv * wa.x; x[1] = x[1] * rms_inv * wa.y; x[2] = x[2] * rms_inv * wa.z; x[3] = x[3] * rms_inv * wa.w;
                    x[4] = x[4] * rms_inv * wb.x; x[5] = x[5] * rms_inv * wb.y; x[6] = x[6] * rms_inv * wb.z; x[7] = x[7] * rms_inv * wb.w;
                }
            }
            float p1[8], p2[8], p3[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {   // truncation split: every difference is exact
                p1[e] = __uint_as_float(__float_as_uint(x[e]) & 0xFFFF0000u);
                const float r1 = x[e] - p1[e];
                p2[e] = __uint_as_float(__float_as_uint(r1) & 0xFFFF0000u);
                p3[e] = r1 - p2[e];
            }
            // sums of the group's 8 activations, then of the step's 4 groups (adjacent lanes): fixed order
            float s_lo = (x[0] + x[1]) + (x[2] + x[3]), s_hi = (x[4] + x[5]) + (x[6] + x[7]);
            if (F::NSUM == 1) {
                float s = s_lo + s_hi;
                s = group_sum<4>(s);
                if (q < ngroups && (q & 3) == 0) xsum[q >> 2] = s;
            } else {
                s_lo = group_sum<4>(s_lo);
                s_hi = group_sum<4>(s_hi);
                if (q < ngroups && (q & 3) == 0) { xsum[2 * (q >> 2)] = s_lo; xsum[2 * (q >> 2) + 1] = s_hi; }
            }
            if (q < ngroups) {
                u32x4* dst = reinterpret_cast<u32x4*>(xop + (size_t)(q >> 2) * GM_XSTEP + (size_t)(q & 3) * 16);
                dst[0] = u32x4{gm_pack_bf16(p1[0], p1[1]), gm_pack_bf16(p1[2], p1[3]), gm_pack_bf16(p1[4], p1[5]), gm_pack_bf16(p1[6], p1[7])};
                dst[4] = u32x4{gm_pack_bf16(p2[0], p2[1]), gm_pack_bf16(p2[2], p2[3]), gm_pack_bf16(p2[4], p2[5]), gm_pack_bf16(p2[6], p2[7])};
                dst[8] = u32x4{gm_pack_bf16(p3[0], p3[1]), gm_pack_bf16(p3[2], p3[3]), gm_pack_bf16(p3[4], p3[5]), gm_pack_bf16(p3[6], p3[7])};
            }
        }
    }
    __syncthreads();

    // ---- tiles ------------------------------------------------------------------------------------------------------------
    const int ks = p.ks, tiles_per_wg = GM_WAVES / ks;
    const int kslice = wave % ks, slot_in_wg = wave / ks;
    const int units_per_slice = p.steps / 8 / ks;       // super-blocks of this wave's K range (a multiple of GM_NR)
    const int unit0 = kslice * units_per_slice;
    const int rounds = (p.ntiles + (int)gridDim.x * tiles_per_wg - 1) / ((int)gridDim.x * tiles_per_wg);
    const uint8_t* xg = xop + g * 16;                   // this lane's column group inside a step's operand record
    const uint8_t* img[GM_RT];
#pragma unroll
    for (int rt = 0; rt < GM_RT; ++rt) img[rt] = stage + (rt * 16 + i) * STRIDE + 4 * g;

    for (int round = 0; round < rounds; ++round) {
        const int tile = (round * (int)gridDim.x + (int)blockIdx.x) * tiles_per_wg + slot_in_wg;
        const bool live = tile < p.ntiles;              // wave-uniform; dead waves still join the barriers below
        float acc[GM_RT];
#pragma unroll
        for (int rt = 0; rt < GM_RT; ++rt) acc[rt] = 0.0f;
        if (live) {
            const int row0 = tile * GM_ROWS;
            uint32_t w_off[NLD], s_off[NLD];
#pragma unroll
            for (int n = 0; n < NLD; ++n) {
                const int q = min(64 * n + lane, PIECES - 1), r = q / NCH, c = q - r * NCH;
                w_off[n] = (uint32_t)min(row0 + r, p.out - 1) * p.row_bytes + 16u * c;
                s_off[n] = (uint32_t)(r * STRIDE + 16 * c);
            }
            u32x4 ring[GM_NR][NLD];
            auto load_unit = [&](int k, int urel) {   // past the end: the last unit again, never consumed
                const uint32_t uoff = (uint32_t)(unit0 + min(urel, units_per_slice - 1)) * F::BB;
#pragma unroll
                for (int n = 0; n < NLD; ++n) ring[k][n] = *reinterpret_cast<const u32x4*>(p.W + min(w_off[n] + uoff, p.w_last));
            };
#pragma unroll
            for (int k = 0; k < GM_NR; ++k) load_unit(k, k);
            for (int trip = 0; trip * GM_NR < units_per_slice; ++trip) {
#pragma unroll
                for (int k = 0; k < GM_NR; ++k) {
                    const int urel = trip * GM_NR + k;
#pragma unroll
                    for (int n = 0; n < NLD; ++n) *reinterpret_cast<u32x4*>(stage + s_off[n]) = ring[k][n];
                    load_unit(k, urel + GM_NR);
                    __builtin_amdgcn_sched_barrier(0);
                    typename F::Hdr hdr[GM_RT];
#pragma unroll
                    for (int rt = 0; rt < GM_RT; ++rt) hdr[rt] = F::header(img[rt] - 4 * g);
                    const int step0 = (unit0 + urel) * 8;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {   // four sub-blocks at a time: lane group g folds sub-block 4 h + g
                        float dsel[GM_RT];
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) {
                            const int j = 4 * h + jj;
                            const uint8_t* xs = xg + (size_t)(step0 + j) * GM_XSTEP;
                            const uint8_t* xs_ = (ABL & 2) ? xg : xs;
                            const bf16x8 x1 = *reinterpret_cast<const bf16x8*>(xs_), x2 = *reinterpret_cast<const bf16x8*>(xs_ + 64),
                                         x3 = *reinterpret_cast<const bf16x8*>(xs_ + 128);
#pragma unroll
                            for (int rt = 0; rt < GM_RT; ++rt) {
                                u32x4 wraw = (ABL & 4) ? u32x4{gm_lds32(img[rt] + 16 + 32 * (j >> 1)), gm_lds32(img[rt] + 32 + 32 * (j >> 1)), 0x41804180u, 0x41804180u} : F::slot(img[rt], j);
                                const bf16x8 wv = __builtin_bit_cast(bf16x8, wraw);
                                f32x4 d = {0.0f, 0.0f, 0.0f, 0.0f};
                                if (ABL & 1) { d[0] = __uint_as_float(wraw.x) + __uint_as_float(wraw.z) * x1[0]; } else {
                                d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x1, wv, d, 0, 0, 0);
                                d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x2, wv, d, 0, 0, 0);
                                d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x3, wv, d, 0, 0, 0); }
                                dsel[rt] = (jj == 0 || g == jj) ? d[0] : dsel[rt];   // every lane of the row holds the sum; lane group jj keeps it
                            }
                        }
                        const float sx = xsum[step0 + 4 * h + g];
#pragma unroll
                        for (int rt = 0; rt < GM_RT; ++rt) {
                            float A, B;
                            F::scales(hdr[rt], h, g, A, B);
                            acc[rt] = fmaf(A, dsel[rt], acc[rt]);
                            acc[rt] = fmaf(-B, sx, acc[rt]);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // the four lane groups of a row hold the sums of their sub-blocks: meet in every lane (fixed order)
#pragma unroll
            for (int rt = 0; rt < GM_RT; ++rt) {
                const float a1 = acc[rt] + __shfl_xor(acc[rt], 16, 64);
                acc[rt] = a1 + __shfl_xor(a1, 32, 64);
            }
        }
        // ---- K split: partial row sums meet in LDS, wave order; the slice-0 wave of a tile finishes the rows ---------------
        if (ks > 1) {
            __syncthreads();   // (previous round's readers are done)
            if (live && g == 0) {
#pragma unroll
                for (int rt = 0; rt < GM_RT; ++rt) kpart[wave * GM_ROWS + rt * 16 + i] = acc[rt];
            }
            __syncthreads();
        }
        if (live && kslice == 0 && g == 0) {
#pragma unroll
            for (int rt = 0; rt < GM_RT; ++rt) {
                const int r = tile * GM_ROWS + rt * 16 + i;
                if (r >= p.out) continue;
                float v = acc[rt];
                for (int s = 1; s < ks; ++s) v += kpart[(wave + s) * GM_ROWS + rt * 16 + i];
                if (p.resid) v += p.resid[r];
                p.y[r] = v;
            }
        }
    }
}

static size_t gm_lds_bytes(int steps, int nsum, int stride) {
    return (size_t)steps * GM_XSTEP + (size_t)steps * nsum * 4 + 16 * 4 + (size_t)GM_WAVES * GM_ROWS * 4 + (size_t)GM_WAVES * GM_ROWS * stride;
}

template <int DT>
static int launch_gemv_mfma(float* y, const void* W, const float* x, int out, int in, const float* norm_w, float eps, const float* resid,
                            hipStream_t st) {
    using F = GmFmt<DT>;
    if (in <= 0 || in % F::BW != 0 || out <= 0) return NTK_E_SHAPE;
    const size_t row_bytes = (size_t)in / F::BW * F::BB;
    if ((size_t)out * row_bytes > 0xFFFFFF00ull) return NTK_E_SHAPE;
    if ((reinterpret_cast<uintptr_t>(W) & 15) || (reinterpret_cast<uintptr_t>(x) & 15) || (norm_w && (reinterpret_cast<uintptr_t>(norm_w) & 15)))
        return NTK_E_ALIGN;
    GmParams p{};
    p.W = static_cast<const uint8_t*>(W);
    p.y = y; p.x = x; p.norm_w = norm_w; p.resid = resid; p.eps = eps;
    p.out = out; p.in = in; p.steps = in / 32;
    p.row_bytes = (unsigned)row_bytes;
    p.w_last = (unsigned)((size_t)out * row_bytes - 16);
    p.ntiles = (out + GM_ROWS - 1) / GM_ROWS;
    const size_t lds = gm_lds_bytes(p.steps, F::NSUM, F::STRIDE);
    if (lds > 160 * 1024) return NTK_E_SHAPE;   // in_features beyond ~16K: the activation operands do not fit (caller: gemv.hip's kernel)
    // K split: as many waves per tile as it takes to put ~2048 waves to work, in whole ring turns of super-blocks
    const int units = in / 256;
    static const int force_ks = [] { const char* e = getenv("NTK_GEMV_MFMA_KS"); return e ? atoi(e) : 0; }();
    int ks = 1;
    while (ks < GM_WAVES && (long)p.ntiles * ks < 2048 && units % (ks * 2 * GM_NR) == 0) ks *= 2;
    if (force_ks == 1 || force_ks == 2 || force_ks == 4 || force_ks == 8) ks = force_ks;
    if (units % (ks * GM_NR) != 0) return NTK_E_SHAPE;
    p.ks = ks;
    const int tiles_per_wg = GM_WAVES / ks;
    static const int max_wg = [] { const char* e = getenv("NTK_GEMV_MFMA_WG"); return e ? std::max(1, atoi(e)) : 256; }();
    const int grid = std::max(1, std::min(max_wg, (p.ntiles + tiles_per_wg - 1) / tiles_per_wg));
    using KernelFn = void (*)(const GmParams);
    static const int abl = [] { const char* e = getenv("NTK_GEMV_MFMA_ABL"); return e ? atoi(e) : 0; }();
    KernelFn fn = norm_w ? (KernelFn)gemv_mfma_kernel<DT, true> : (KernelFn)gemv_mfma_kernel<DT, false>;
    if (abl == 1) fn = (KernelFn)gemv_mfma_kernel<DT, false, 1>;
    if (abl == 2) fn = (KernelFn)gemv_mfma_kernel<DT, false, 2>;
    if (abl == 4) fn = (KernelFn)gemv_mfma_kernel<DT, false, 4>;
    if (abl == 6) fn = (KernelFn)gemv_mfma_kernel<DT, false, 6>;
    if (abl == 7) fn = (KernelFn)gemv_mfma_kernel<DT, false, 7>;
    if (lds > 64 * 1024) {
        static bool once = hipFuncSetAttribute((const void*)gemv_mfma_kernel<DT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess &&
                           hipFuncSetAttribute((const void*)gemv_mfma_kernel<DT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
        if (!once) return NTK_E_SHAPE;
    }
    hipLaunchKernelGGL(fn, dim3(grid), dim3(64 * GM_WAVES), lds, st, p);
    return last_launch_status();
}

}  // namespace ntk

extern "C" {

// experimental entry point (the engine reaches this kernel through ntk_gemv / ntk_gemv_fused)
int ntk_gemv_mfma(float* y, const void* W, const float* x, int out_features, int in_features, int weight_dtype, const float* norm_w, float eps,
                  const float* resid, void* stream) {
    if (!y || !W || !x) return NTK_E_NULL;
    hipStream_t st = ntk::resolve_stream(stream);
    switch (weight_dtype) {
        case NTK_DT_Q4_K: return ntk::launch_gemv_mfma<NTK_DT_Q4_K>(y, W, x, out_features, in_features, norm_w, eps, resid, st);
        default: return NTK_E_DTYPE;
    }
}

}  // extern "C"
