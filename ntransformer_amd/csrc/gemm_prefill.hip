// gemm_prefill.hip -- batched prompt projections on the matrix cores: Y[T,out] = X[T,in] . W[out,in]^T,
// W in raw GGUF blocks, T <= 16 tokens per pass.
//
// Replaces: the reference's prefill, which runs launch_gemv once per prompt token
// (reference src/model/attention.cpp:144-162,200-210 and src/model/ffn.cpp:96-133): T passes over every weight
// matrix.  Here each matrix is streamed from HBM ONCE per 16 tokens.
//
// Arithmetic: weights are dequantised to F32 exactly as the GEMV kernels define them (reference gemm.cu:60-75,
// 129-141, 190-244, 297-354, 421-459 / SURVEY.md Appendix A) and multiplied with the F32 activations on
// v_mfma_f32_16x16x4_f32 -- F32 in, F32 accumulate: no activation rounding, so the prompt's KV cache and logits stay
// within the same 1e-3 budget as the per-token path (only the summation order differs).
//
// Decomposition (gfx950, wave64) -- activation-stationary:
//   * one workgroup of 16 waves per CU; wave w owns the 256-column slice w of a 4096-column K chunk and keeps ITS
//     activations (16 tokens x 256 columns, 64 VGPRs: the B operands, B[k][j]: lane = j + 16k) in registers while
//     the workgroup walks its tiles of 16 output rows (tile = blockIdx.x, + gridDim.x, ...);
//   * per tile wave w fetches the bytes of ROW w of the chunk (contiguous in HBM: 1 KiB per wave instruction, prefetched one
//     tile ahead in VGPRs) into a shared LDS image; after a barrier it computes on SLICE w of all 16 rows: lane
//     (r = lane%16, g = lane/16) dequantises the 8 weights of row r, columns 32s + 8g..8g+7 of every 32-column
//     sub-block s and feeds them as the A operand (A[i][k]: lane = i + 16k).  No global loads inside the MFMA loop;
//   * the 16 partial 16x16 accumulators meet in LDS (fixed order: deterministic), one barrier per tile; rows longer
//     than 4096 columns take further K chunks that accumulate into Y (same thread owns an element in every chunk).
// Bound: MFMA (f32 16x16x4 = 64 FLOP/clk/SIMD): 2 weights/clk/SIMD = 4.9e12 weights/s, the HBM rate of Q8_0 at
// 5.2 TB/s -- i.e. balanced for Q8_0, compute-bound for the 4-6 bit formats.  T tokens cost one pass for T <= 16.
// Measured (round 1): 45-65 TFLOP/s of the 157 TFLOP/s bound.  Cycle trace of a tile round (tools/gemm_trace.py): stage 500,
// barrier 300, MFMA phase 6000 for the first wave but ~12500 until the last of the 16 waves reaches the next barrier (8192
// of matrix-pipe work per SIMD: the dequantisation's VALU issue competes with the MFMA issue), partials + reduce 1000.
// A 32-token form on v_mfma_f32_32x32x2 (twice the matrix work per dequantised value) measured the same time per token.
#include "common.hip.h"
#include <algorithm>
#include <cstdlib>

namespace ntk {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int GM_ROWS = 16;      // output rows per tile (MFMA M)
constexpr int GM_TOK = 16;       // tokens per pass (MFMA N)
constexpr int GM_SLICE = 256;    // columns per wave step
constexpr int GM_WAVES = 16;      // waves per workgroup = 256-column slices of K handled side by side (one workgroup per CU;
                                  // 8 waves x 2 workgroups measured 10 % slower: more K chunks, the same cost per round)
constexpr int GM_RPW = GM_ROWS / GM_WAVES;   // tile rows fetched by one wave

// Phase trace (tuning builds only: make HIPFLAGS+=-DNTK_GEMM_TRACE): thread 0 of workgroup 0 records the cycle counter at
// the phase boundaries of its tiles; read back with ntk_debug_gemm_trace(), printed by tools/gemm_trace.py.
#ifdef NTK_GEMM_TRACE
__device__ unsigned long long g_gemm_trace[256];
#define GM_STAMP() do { if (trace_on && tslot < 256) g_gemm_trace[tslot++] = __builtin_readcyclecounter(); } while (0)
#else
#define GM_STAMP() do { } while (0)
#endif

// tuning ablations, compile-time only (make HIPFLAGS+=-DNTK_GEMM_ABLATE=n, tools/gpu_gemm_ablate.sh): 1 = skip the
// dequantisation, 2 = VALU FMAs instead of MFMA, 4 = no weight loads after the first tile.  0 in the product: the
// sub-block loop stays one basic block.
#ifndef NTK_GEMM_ABLATE
#define NTK_GEMM_ABLATE 0
#endif
constexpr int kGemmAblate = NTK_GEMM_ABLATE;

template <int DT> struct GFmt;
// BW/BB: weights / bytes per block.  PIECES: 16-byte pieces that cover one row's slice bytes at any 2-byte alignment.
template <> struct GFmt<NTK_DT_Q8_0> { static constexpr int BW = 32, BB = 34, SB = 272; };
template <> struct GFmt<NTK_DT_Q4_0> { static constexpr int BW = 32, BB = 18, SB = 144; };
template <> struct GFmt<NTK_DT_Q4_K> { static constexpr int BW = 256, BB = 144, SB = 144; };
template <> struct GFmt<NTK_DT_Q5_K> { static constexpr int BW = 256, BB = 176, SB = 176; };
template <> struct GFmt<NTK_DT_Q6_K> { static constexpr int BW = 256, BB = 210, SB = 210; };

struct GemmParams {
    const uint8_t* W;     // 16-byte-aligned-down base
    int delta;            // true W = W + delta (0..15, even)
    const float* X;       // [T][in]
    float* Y;             // [T][out]
    const float* resid;   // optional [T][out], may alias Y
    int T, out, in;
    unsigned row_bytes;
    int nslices;          // ceil(in / 256)
};

// The 8 weights of (row image `st`, 32-column sub-block s of the slice, column group g) as F32.
//   st: LDS image of the row's slice bytes, byte k of the slice at st[k] (st itself carries the row's alignment shift)
template <int DT> struct Deq;

__device__ __forceinline__ void bytes8_to_f32(uint32_t lo, uint32_t hi, float (&f)[8]) {
    f[0] = ub2f(lo, 0); f[1] = ub2f(lo, 1); f[2] = ub2f(lo, 2); f[3] = ub2f(lo, 3);
    f[4] = ub2f(hi, 0); f[5] = ub2f(hi, 1); f[6] = ub2f(hi, 2); f[7] = ub2f(hi, 3);
}

template <> struct Deq<NTK_DT_Q8_0> {   // gemm.cu:129-141: w = d * q
    __device__ static void run(const uint8_t* st, int shift, int s, int g, float (&a)[8]) {
        const int ob = shift + 34 * s;
        const float d = h2f(lds_u16_at(st, ob));
        uint32_t q[2];
        lds_read_dwords<2>(q, st, ob + 2 + 8 * g);
#pragma unroll
        for (int j = 0; j < 4; ++j) { a[j] = d * sb2f(q[0], j); a[4 + j] = d * sb2f(q[1], j); }
    }
};

template <> struct Deq<NTK_DT_Q4_0> {   // gemm.cu:60-75: columns 0..15 low nibbles, 16..31 high nibbles; w = d (n - 8)
    __device__ static void run(const uint8_t* st, int shift, int s, int g, float (&a)[8]) {
        const int ob = shift + 18 * s;
        const float d = h2f(lds_u16_at(st, ob));
        uint32_t q[2];
        lds_read_dwords<2>(q, st, ob + 2 + 8 * (g & 1));
        const int sh = 4 * (g >> 1);
        float n[8];
        bytes8_to_f32((q[0] >> sh) & 0x0F0F0F0Fu, (q[1] >> sh) & 0x0F0F0F0Fu, n);
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = d * (n[j] - 8.0f);
    }
};

// K-quant (scale, min) of sub-block s (types.h:112-117, gemm.cu:206-222); hd = the block's first 16 bytes
__device__ __forceinline__ void kq_pair(const uint32_t (&hd)[4], int s, float& dsc, float& dmn) {
    const float d = h2f((uint16_t)(hd[0] & 0xFFFFu)), dmin = h2f((uint16_t)(hd[0] >> 16));
    const uint32_t s0 = hd[1], s1 = hd[2], s2 = hd[3];
    uint32_t sc, mn;
    if (s < 4) {
        sc = (s0 >> (8 * s)) & 63u;
        mn = (s1 >> (8 * s)) & 63u;
    } else {
        const int b = 8 * (s - 4);
        sc = ((s2 >> b) & 0xFu) | (((s0 >> (b + 6)) & 3u) << 4);
        mn = ((s2 >> (b + 4)) & 0xFu) | (((s1 >> (b + 6)) & 3u) << 4);
    }
    dsc = d * (float)sc;
    dmn = dmin * (float)mn;
}

template <> struct Deq<NTK_DT_Q4_K> {   // gemm.cu:190-244: w = d sc n - dmin m
    __device__ static void run(const uint8_t* st, int shift, int s, int g, float (&a)[8]) {
        uint32_t hd[4], q[2];
        lds_read_dwords<4>(hd, st, shift);
        float dsc, dmn;
        kq_pair(hd, s, dsc, dmn);
        lds_read_dwords<2>(q, st, shift + 16 + 32 * (s >> 1) + 8 * g);
        const int sh = 4 * (s & 1);
        float n[8];
        bytes8_to_f32((q[0] >> sh) & 0x0F0F0F0Fu, (q[1] >> sh) & 0x0F0F0F0Fu, n);
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = fmaf(dsc, n[j], -dmn);
    }
};

template <> struct Deq<NTK_DT_Q5_K> {   // gemm.cu:297-354: fifth bit = bit s of qh[l]
    __device__ static void run(const uint8_t* st, int shift, int s, int g, float (&a)[8]) {
        uint32_t hd[4], q[2], h[2];
        lds_read_dwords<4>(hd, st, shift);
        float dsc, dmn;
        kq_pair(hd, s, dsc, dmn);
        lds_read_dwords<2>(h, st, shift + 16 + 8 * g);
        lds_read_dwords<2>(q, st, shift + 48 + 32 * (s >> 1) + 8 * g);
        const int sh = 4 * (s & 1);
        float n[8];
        bytes8_to_f32(((q[0] >> sh) & 0x0F0F0F0Fu) | (((h[0] >> s) & 0x01010101u) << 4),
                      ((q[1] >> sh) & 0x0F0F0F0Fu) | (((h[1] >> s) & 0x01010101u) << 4), n);
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = fmaf(dsc, n[j], -dmn);
    }
};

template <> struct Deq<NTK_DT_Q6_K> {   // gemm.cu:421-459: sub-block s = 4 half + type; w = d sc[8 half + 2 type + l/16] (q - 32)
    __device__ static void run(const uint8_t* st, int shift, int s, int g, float (&a)[8]) {
        const int hf = s >> 2, ty = s & 3;
        uint32_t q[2], h[2];
        lds_read_dwords<2>(q, st, shift + 64 * hf + 32 * (ty & 1) + 8 * g);
        lds_read_dwords<2>(h, st, shift + 128 + 32 * hf + 8 * g);
        const uint32_t scw = (uint32_t)lds_u16_at(st, shift + 192 + 8 * hf + 2 * ty);   // the sub-block's two int8 sub-scales
        const float d = h2f(lds_u16_at(st, shift + 208));
        const float dsc = d * sb2f(scw, 0) , dsc1 = d * sb2f(scw, 1);
        const float sc = (g >> 1) ? dsc1 : dsc;                                          // l / 16 = g / 2
        const int sh = 4 * (ty >> 1);
        float n[8];
        bytes8_to_f32(((q[0] >> sh) & 0x0F0F0F0Fu) | (((h[0] >> (2 * ty)) & 0x03030303u) << 4),
                      ((q[1] >> sh) & 0x0F0F0F0Fu) | (((h[1] >> (2 * ty)) & 0x03030303u) << 4), n);
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = sc * (n[j] - 32.0f);
    }
};

// FULL: in % 256 == 0, every slice has its eight sub-blocks (no per-sub-block branches: the compiler can then move the
// next sub-block's LDS reads above the current MFMAs)
template <int DT, bool FULL>
__global__ __launch_bounds__(64 * GM_WAVES) void gemm_quant_mfma_kernel(const GemmParams p) {
    using F = GFmt<DT>;
    constexpr int CB = GM_WAVES * F::SB;                          // bytes of one row inside a K chunk (GM_WAVES slices)
    constexpr int NLR = (CB + 15 + 1023) / 1024;                  // 1 KiB wave loads that cover them at any alignment
    constexpr int RPITCH = NLR * 1024 + 16;                       // LDS bytes per staged row (+16: rows start on different banks)
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint8_t* tile_img = smem;                                     // [16 rows][RPITCH]: row w + GM_WAVES q is written by wave w
    float* red = reinterpret_cast<float*>(smem + (size_t)GM_ROWS * RPITCH);   // [2][GM_WAVES][256]
    const int r = lane & 15, g = lane >> 4;
    const int ntiles = (p.out + GM_ROWS - 1) / GM_ROWS;
    const int tok = lane & 15;                                    // B / D column of this lane
    const float* xrow = p.X + (size_t)min(tok, p.T - 1) * p.in;
    const int my_tiles = blockIdx.x < ntiles ? (ntiles - 1 - blockIdx.x) / (int)gridDim.x + 1 : 0;
    const int nchunks = (p.nslices + GM_WAVES - 1) / GM_WAVES;
    int parity = 0;
#ifdef NTK_GEMM_TRACE
    const bool trace_on = blockIdx.x == 0 && tid == 0;
    int tslot = 0;
#endif
    GM_STAMP();                                                    // kernel entry

    for (int chunk = 0; chunk < nchunks; ++chunk) {
        // ---- this wave's 256-column slice of the chunk; its activations stay in registers for every tile ----------
        const int slice = chunk * GM_WAVES + wave;
        const bool active = slice < p.nslices;                    // wave-uniform
        const int col0 = slice * GM_SLICE;
        const int nsub = active ? min(GM_SLICE, p.in - col0) / 32 : 0;   // 32-column sub-blocks (8; fewer in a ragged tail)
        float xr[8][8];   // B operand: xr[s][j] = X[tok][col0 + 32 s + 8 g + j]
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const bool have = s < nsub && tok < p.T;
            const float* xs = xrow + (have ? col0 + 32 * s + 8 * g : 0);
            const float4 v0 = *reinterpret_cast<const float4*>(xs), v1 = *reinterpret_cast<const float4*>(xs + 4);
            xr[s][0] = have ? v0.x : 0.0f; xr[s][1] = have ? v0.y : 0.0f; xr[s][2] = have ? v0.z : 0.0f; xr[s][3] = have ? v0.w : 0.0f;
            xr[s][4] = have ? v1.x : 0.0f; xr[s][5] = have ? v1.y : 0.0f; xr[s][6] = have ? v1.z : 0.0f; xr[s][7] = have ? v1.w : 0.0f;
        }
        // ---- weights: wave w fetches ROW w of the tile -- the chunk's bytes of one row are contiguous in HBM, 1 KiB per
        //      wave instruction -- and after a barrier computes on SLICE w of all 16 rows: the transposition runs through LDS.
        //      All offsets are 32-bit (the host checks the matrix is < 4 GiB).
        const unsigned cbyte = (unsigned)chunk * CB;                                   // chunk start inside a row
        const unsigned clen = min((unsigned)CB, p.row_bytes - cbyte);                  // bytes of this chunk in a row (the last chunk may be short)
        u32x4 pf[GM_RPW][NLR];
        auto issue = [&](int tile) {
#pragma unroll
            for (int q = 0; q < GM_RPW; ++q) {
                const unsigned grow = (unsigned)min(tile * GM_ROWS + wave + GM_WAVES * q, p.out - 1);   // ragged last tile: re-read the last row
                const unsigned rel = (unsigned)p.delta + grow * p.row_bytes + cbyte;
                const unsigned last = ((rel & 15u) + clen - 1u) & ~15u;
                const uint8_t* a = p.W + (rel & ~15u);
#pragma unroll
                for (int j = 0; j < NLR; ++j)
                    pf[q][j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(a + min(16u * (unsigned)(lane + 64 * j), last)));
            }
        };
        if (my_tiles > 0) issue(blockIdx.x);
        GM_STAMP();                                                // chunk prologue done (x in registers, first rows requested)

        for (int tl = 0; tl < my_tiles; ++tl) {
            const int tile = blockIdx.x + tl * gridDim.x, row0 = tile * GM_ROWS;
#pragma unroll
            for (int q = 0; q < GM_RPW; ++q)
#pragma unroll
                for (int j = 0; j < NLR; ++j)
                    *reinterpret_cast<u32x4*>(tile_img + (wave + GM_WAVES * q) * RPITCH + 16 * (lane + 64 * j)) = pf[q][j];
            GM_STAMP();                                            // rows landed + staged
            __syncthreads();                                       // B1: the tile image is complete
            GM_STAMP();                                            // past B1
            // what the reduction adds to (residual / earlier K chunks) is requested BEFORE the prefetch: memory returns in
            // order, so waiting for it later must not mean waiting for the next tile's rows
            float base = 0.0f;
            const int oi = tid >> 4, oj = tid & 15;
            const bool owner = tid < 256 && row0 + oi < p.out && oj < p.T;
            const size_t o = (size_t)oj * p.out + row0 + oi;
            if (owner && (chunk > 0 || p.resid != nullptr)) base = chunk == 0 ? p.resid[o] : p.Y[o];
            // the next tile's rows fly during this tile's MFMAs (unconditional -- the last tile re-reads itself -- so the
            // number of loads behind `base` is static and hipcc can wait with vmcnt(NLR) instead of vmcnt(0))
            if (!(kGemmAblate & 4)) issue(tl + 1 < my_tiles ? tile + (int)gridDim.x : tile);
            f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};   // two chains: an MFMA waits for its accumulator
            if (active) {
                const unsigned grow = (unsigned)min(row0 + r, p.out - 1);
                const int shift = (int)(((unsigned)p.delta + grow * p.row_bytes + cbyte) & 15u) + wave * F::SB;
                const uint8_t* st = tile_img + r * RPITCH;
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    if (FULL || s < nsub) {
                        float a[8];
                        if (kGemmAblate & 1) {
#pragma unroll
                            for (int j = 0; j < 8; ++j) a[j] = 1.0f;
                        } else {
                            Deq<DT>::run(st, shift, s, g, a);
                        }
                        if (kGemmAblate & 2) {
#pragma unroll
                            for (int j = 0; j < 8; ++j) acc0[j & 3] += a[j] * xr[s][j];
                        } else {
#pragma unroll
                            for (int j = 0; j < 8; ++j) {
                                if (s & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], xr[s][j], acc1, 0, 0, 0);
                                else acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], xr[s][j], acc0, 0, 0, 0);
                            }
                        }
                    }
                }
            }
            // ---- cross-wave reduction: D[i][j], i = 4 (lane / 16) + reg, j = lane % 16.  `red` is double-buffered, so the
            //      barrier below is also the one that frees the tile image for the next tile's rows.
            GM_STAMP();                                            // MFMAs issued
            float* rd = red + parity * (GM_WAVES * 256);
            parity ^= 1;
#pragma unroll
            for (int e = 0; e < 4; ++e) rd[wave * 256 + (4 * g + e) * 16 + tok] = acc0[e] + acc1[e];
            GM_STAMP();                                            // partials written
            __syncthreads();                                       // B2
            GM_STAMP();                                            // past B2
            if (owner) {   // chunk 0 writes (adding the residual), later K chunks accumulate: the same thread owns the element
                float t = 0.0f;
#pragma unroll
                for (int w = 0; w < GM_WAVES; ++w) t += rd[w * 256 + tid];
                p.Y[o] = base + t;
            }
            GM_STAMP();                                            // reduced + stored
        }
        __syncthreads();   // the last tile's readers are done before the next chunk's first rows are staged
    }
}

template <int DT>
static int launch_gemm(float* Y, const void* W, const float* X, int T, int out, int in, const float* resid, hipStream_t st) {
    using F = GFmt<DT>;
    if (in <= 0 || in % F::BW != 0 || in % 32 != 0) return NTK_E_SHAPE;
    const uintptr_t w = reinterpret_cast<uintptr_t>(W);
    if (w & 1) return NTK_E_ALIGN;
    if ((reinterpret_cast<uintptr_t>(X) & 15) != 0 || (in % 4) != 0) return NTK_E_ALIGN;
    GemmParams p{};
    p.W = reinterpret_cast<const uint8_t*>(w & ~(uintptr_t)15);
    p.delta = (int)(w & 15);
    p.X = X; p.Y = Y; p.resid = resid;
    p.T = T; p.out = out; p.in = in;
    const size_t row_bytes = (size_t)in / F::BW * F::BB;
    if ((size_t)out * row_bytes > 0xFFFFFFF0ull) return NTK_E_SHAPE;
    p.row_bytes = (unsigned)row_bytes;
    p.nslices = (in + GM_SLICE - 1) / GM_SLICE;
    constexpr int CB = GM_WAVES * F::SB;
    constexpr int NLR = (CB + 15 + 1023) / 1024;
    constexpr int RPITCH = NLR * 1024 + 16;
    const size_t lds = (size_t)GM_ROWS * RPITCH + (size_t)2 * GM_WAVES * 256 * sizeof(float);
    static bool once = [] {
        return hipFuncSetAttribute((const void*)gemm_quant_mfma_kernel<DT, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   160 * 1024) == hipSuccess &&
               hipFuncSetAttribute((const void*)gemm_quant_mfma_kernel<DT, false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   160 * 1024) == hipSuccess;
    }();
    if (!once || lds > 160 * 1024) return NTK_E_SHAPE;
    const int ntiles = (out + GM_ROWS - 1) / GM_ROWS;
    const int grid = std::min(ntiles, 256 * 16 / GM_WAVES);   // 16 waves per CU; a workgroup walks its tiles (prefetching the next)
    if (in % GM_SLICE == 0) hipLaunchKernelGGL((gemm_quant_mfma_kernel<DT, true>), dim3(grid), dim3(64 * GM_WAVES), lds, st, p);
    else hipLaunchKernelGGL((gemm_quant_mfma_kernel<DT, false>), dim3(grid), dim3(64 * GM_WAVES), lds, st, p);
    return last_launch_status();
}

}  // namespace ntk

#ifdef NTK_GEMM_TRACE
extern "C" int ntk_debug_gemm_trace(unsigned long long* out256) {
    return hipMemcpyFromSymbol(out256, HIP_SYMBOL(ntk::g_gemm_trace), 256 * sizeof(unsigned long long)) == hipSuccess ? 0 : -3;
}
#endif

extern "C" int ntk_gemm_quant(float* Y, const void* W, const float* X, int n_tokens, int out_features, int in_features,
                              int weight_dtype, const float* resid, void* stream) {
    if (!Y || !W || !X) return NTK_E_NULL;
    if (n_tokens < 0 || out_features < 0 || in_features <= 0) return NTK_E_SHAPE;
    if (n_tokens == 0 || out_features == 0) return NTK_OK;
    hipStream_t st = ntk::resolve_stream(stream);
    for (int t0 = 0; t0 < n_tokens; t0 += ntk::GM_TOK) {   // 16 tokens per pass over W
        const int T = std::min(ntk::GM_TOK, n_tokens - t0);
        float* y = Y + (size_t)t0 * out_features;
        const float* x = X + (size_t)t0 * in_features;
        const float* rs = resid ? resid + (size_t)t0 * out_features : nullptr;
        int rc;
        switch (weight_dtype) {
            case NTK_DT_Q8_0: rc = ntk::launch_gemm<NTK_DT_Q8_0>(y, W, x, T, out_features, in_features, rs, st); break;
            case NTK_DT_Q4_0: rc = ntk::launch_gemm<NTK_DT_Q4_0>(y, W, x, T, out_features, in_features, rs, st); break;
            case NTK_DT_Q4_K: rc = ntk::launch_gemm<NTK_DT_Q4_K>(y, W, x, T, out_features, in_features, rs, st); break;
            case NTK_DT_Q5_K: rc = ntk::launch_gemm<NTK_DT_Q5_K>(y, W, x, T, out_features, in_features, rs, st); break;
            case NTK_DT_Q6_K: rc = ntk::launch_gemm<NTK_DT_Q6_K>(y, W, x, T, out_features, in_features, rs, st); break;
            default: return NTK_E_DTYPE;
        }
        if (rc != NTK_OK) return rc;
    }
    return NTK_OK;
}
