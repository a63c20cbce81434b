// gemv_core.hip.h -- per-format decode of the dequant-fused GEMV (shared by gemv.hip and decode_persistent.hip).
//
// Replaces the arithmetic of the reference's gemv_{q4_0,q8_0,q4_k,q5_k,q6_k}_kernel (reference src/cuda/gemm.cu:32-470):
// same per-block math (integer quant x F32 activation, F32 accumulate, FP16 scale per block / 6-bit sub-scale per
// sub-block), on the lane <-> column decomposition described in gemv.hip.
#pragma once
#include "common.hip.h"
#include <algorithm>

namespace ntk {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// 16-byte global load the compiler does not see in its s_waitcnt bookkeeping (base: uniform pointer, off: bytes)
__device__ __forceinline__ u32x4 asm_load16(const void* base, unsigned off) {
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(off), "s"(base) : "memory");
    return v;
}

constexpr int RB = 4;            // rows (items) per batch between cross-wave combines (8 measured the same)
constexpr int XPITCH = 68;       // floats per 64-column lane row in the prologue LDS image (conflict-free b128)
constexpr int MAX_SEG = 3;
// tuning ablations, compile-time only (make HIPFLAGS+=-DNTK_GEMV_ABLATE=n; profiles/r01_gemv_ablation.txt):
// 1 = skip the x prologue, 2 = skip the decode, 4 = skip LDS staging.  0 in the product.
#ifndef NTK_GEMV_ABLATE
#define NTK_GEMV_ABLATE 0
#endif
constexpr int kAblate = NTK_GEMV_ABLATE;

// Launch timeline (tuning builds only: make HIPFLAGS+=-DNTK_GEMV_TRACE): thread 0 of every workgroup records the constant
// 100 MHz clock at the phase boundaries of its launch; read back with ntk_debug_gemv_trace(), printed by tools/gemv_trace.py.
#ifdef NTK_GEMV_TRACE
constexpr int GT_SLOTS = 64, GT_WG = 512, GT_EV = 14;
__device__ unsigned long long g_gemv_trace[GT_SLOTS][GT_WG][GT_EV];
#define GV_STAMP(ev) do { asm volatile("" ::: "memory"); gv_t[ev] = __builtin_amdgcn_s_memrealtime(); asm volatile("" ::: "memory"); } while (0)
#else
#define GV_STAMP(ev) do {} while (0)
#endif

template <int DT> struct Fmt;
// BW/BB: weights / bytes per GGUF block; NL: 1 KiB chunks per <=4096-column slice (+15 alignment bytes);
// MINW: waves per SIMD the register allocator must leave room for (4 -> <=128 VGPRs, 3 -> <=168: the 5/6-bit
// decoders keep more packed dwords live and would otherwise spill the prefetch registers)
// NBUF: row slices per wave of an LDS-DMA ring (EXPERIMENTS=1 builds: the persistent token kernel)
template <> struct Fmt<NTK_DT_Q8_0> { static constexpr int BW = 32, BB = 34, NL = 5, MINW = 4, NBUF = 1; };
template <> struct Fmt<NTK_DT_Q4_0> { static constexpr int BW = 32, BB = 18, NL = 3, MINW = 4, NBUF = 3; };
template <> struct Fmt<NTK_DT_Q4_K> { static constexpr int BW = 256, BB = 144, NL = 3, MINW = 4, NBUF = 3; };
template <> struct Fmt<NTK_DT_Q5_K> { static constexpr int BW = 256, BB = 176, NL = 3, MINW = 4, NBUF = 2; };
template <> struct Fmt<NTK_DT_Q6_K> { static constexpr int BW = 256, BB = 210, NL = 4, MINW = 4, NBUF = 2; };
// bytes of one ring slot: a <= 4096-column slice + its alignment shift, rounded to 16, + 16 (the row's residual value)
template <int DT> constexpr int DMA_SLOT = ((4096 / Fmt<DT>::BW * Fmt<DT>::BB + 30) / 16) * 16 + 16;
// (Round 3 also built the row stream of gemv.hip on such a ring -- two or three row slices in flight per wave instead of one -- and measured
// NOTHING: Q4_K gate|up 16.7 -> 16.3 us, LM head 52.8 -> 52.7, 70B gate|up 48.7 -> 49.1 (profiles/r03_gemv_lds_dma_ring_experiment.txt):
// the long K-quant launches were bound by VALU issue, not by bytes in flight.  That form was removed in round 4, when the matrix-core
// GEMV of gemv_rp.hip took the K-quant launches of the engine.)

// ---- LDS-DMA: up to 1 KiB per wave instruction straight from HBM into the wave's ring slot (no VGPR bounce, no ds_write) ----
// lds_dst: wave-uniform LDS byte address the wave's lane 0 writes (lane l writes lds_dst + 16 l / 4 l); gsrc: this lane's source.
// M0 is compiler-reserved: saved, set and restored inside the one statement that uses it (cdna_hip_programming.md 5.7).  Inactive
// lanes (EXEC) move nothing.  The compiler does not count these loads: the waits are explicit (wait_vm).
__device__ __forceinline__ void dma16(uint32_t lds_dst, const uint8_t* gsrc) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void dma4(uint32_t lds_dst, const float* gsrc) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// wait until at most n of the wave's vector-memory operations are outstanding (they complete in order): n = the operations issued
// AFTER the last DMA instruction of the row that must have landed.  An operation issued and not counted only makes the wait
// stricter than needed, never weaker; counting one that was not issued would be the bug.
__device__ __forceinline__ void wait_vm(unsigned n) {   // wave-uniform
    if (n >= 14) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
    else if (n == 13) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
    else if (n == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (n == 11) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
    else if (n == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if (n == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    else if (n == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (n == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else if (n == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (n == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if (n == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (n == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if (n == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if (n == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// formats that have a 16-byte-aligned fast decoder (others instantiate only the general one)
template <int DT> constexpr bool A16_OK = (DT == NTK_DT_Q4_K || DT == NTK_DT_Q5_K);

struct GemvSeg {
    const uint8_t* W;   // 16-byte-aligned-down base of the segment
    float* y;
    int rows;
    int delta;          // true W = W + delta (0..15)
};

struct GemvParams {
    GemvSeg seg[MAX_SEG];
    int nseg;
    int total_rows;     // silu_pair: rows of ONE matrix
    const float* x;
    int in;
    int ns;             // column slices per row (waves cooperating on one row)
    int rw;             // row groups per workgroup
    int slice_cols;
    int nbatch;         // uniform batch count per row group
    int x_vec;          // x (and norm_w) 16-byte aligned -> float4 loads
    const float* norm_w;
    float eps;
    const float* resid;
    int silu_pair;
    unsigned row_bytes;
#ifdef NTK_GEMV_TRACE
    int trace_slot;     // tuning builds: which record of g_gemv_trace this launch fills
#endif
};

// ------------------------------------------------------------------------------------------------
// Per-format decode of one lane's 64 columns from the staged byte image.
//   st    : wave-private LDS image of the slice's bytes, st[shift + k] = byte k of the slice
//   ncols : how many of the lane's 64 columns exist (0, 32 or 64)
//   xr    : the lane's activations, sx16/sx32 their run sums
// ------------------------------------------------------------------------------------------------
// A16: every LDS offset the decoder touches is 16-byte aligned (K-quant rows whose bytes start 16-byte aligned)
template <int DT, bool A16> struct Dot;

typedef float f32x2 __attribute__((ext_vector_type(2)));
// two FMAs per instruction (v_pk_fma_f32): the dot products are VALU-issue bound for the 4/5/6-bit formats
__device__ __forceinline__ f32x2 pkfma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
// hide a value from the optimiser so that (x & 0x0F0F0F0F) stays ONE v_and and each byte converts with
// v_cvt_f32_ubyteN (otherwise hipcc re-derives every nibble with its own v_bfe_u32 + v_cvt_f32_ubyte0)
__device__ __forceinline__ uint32_t opaque(uint32_t v) { asm("" : "+v"(v)); return v; }
__device__ __forceinline__ f32x2 ub01(uint32_t w) { return f32x2{ub2f(w, 0), ub2f(w, 1)}; }
__device__ __forceinline__ f32x2 ub23(uint32_t w) { return f32x2{ub2f(w, 2), ub2f(w, 3)}; }
__device__ __forceinline__ float hsum(f32x2 a, f32x2 b) { return (a.x + a.y) + (b.x + b.y); }

// x2[i] = (x[2i], x[2i+1]) of the lane's 64 columns

template <bool A16> struct Dot<NTK_DT_Q8_0, A16> {   // reference gemm.cu:129-141: sum += d * sum_j q_j x_j
    __device__ static float run(const uint8_t* st, int shift, int lane, int ncols, const f32x2 (&x2)[32],
                                const float (&)[4], const float (&)[2]) {
        float acc = 0.0f;
        const int o = shift + 68 * lane;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (ncols > 32 * h) {
                const int ob = o + 34 * h;
                const float d = h2f(lds_u16_at(st, ob));
                uint32_t q[8];
                lds_read_dwords<8>(q, st, ob + 2);
                f32x2 a0 = {0.0f, 0.0f}, a1 = {0.0f, 0.0f};
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    a0 = pkfma(f32x2{sb2f(q[i], 0), sb2f(q[i], 1)}, x2[16 * h + 2 * i], a0);
                    a1 = pkfma(f32x2{sb2f(q[i], 2), sb2f(q[i], 3)}, x2[16 * h + 2 * i + 1], a1);
                }
                acc = fmaf(d, hsum(a0, a1), acc);
            }
        }
        return acc;
    }
};

template <bool A16> struct Dot<NTK_DT_Q4_0, A16> {   // reference gemm.cu:60-75: w_j = d (lo-8), w_{j+16} = d (hi-8)
    __device__ static float run(const uint8_t* st, int shift, int lane, int ncols, const f32x2 (&x2)[32],
                                const float (&)[4], const float (&sx32)[2]) {
        float acc = 0.0f;
        const int o = shift + 36 * lane;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (ncols > 32 * h) {
                const int ob = o + 18 * h;
                const float d = h2f(lds_u16_at(st, ob));
                uint32_t q[4];
                lds_read_dwords<4>(q, st, ob + 2);
                f32x2 a0 = {0.0f, 0.0f}, a1 = {0.0f, 0.0f};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint32_t lo = opaque(q[i] & 0x0F0F0F0Fu), hi = opaque(q[i] & 0xF0F0F0F0u);   // hi bytes = 16 n, x pre-scaled
                    a0 = pkfma(ub01(lo), x2[16 * h + 2 * i], a0);
                    a1 = pkfma(ub23(lo), x2[16 * h + 2 * i + 1], a1);
                    a0 = pkfma(ub01(hi), x2[16 * h + 8 + 2 * i], a0);
                    a1 = pkfma(ub23(hi), x2[16 * h + 8 + 2 * i + 1], a1);
                }
                acc = fmaf(d, fmaf(-8.0f, sx32[h], hsum(a0, a1)), acc);   // sum (n-8) x = sum n x - 8 sum x
            }
        }
        return acc;
    }
};

// K-quant header: d, dmin and the 6-bit (scale, min) pairs of sub-blocks 2c, 2c+1 (types.h:112-117, gemm.cu:206-222).
// Branch-free in the lane-constant c.  The 12 packed bytes are the dwords s0 / s1 / s2; all eight scales (mins) are
// first laid out as the bytes of two dwords each -- sub-blocks 0..3: s0 & 0x3F (s1 & 0x3F); sub-blocks 4..7:
// (s2 & 0x0F) | ((s0 >> 6) << 4)   ((s2 >> 4) | ((s1 >> 6) << 4)) -- then one select and one shift pick the pair.
template <bool A16>
__device__ __forceinline__ void kq_header(const uint8_t* st, int ob, int c, float& d1, float& m1, float& d2, float& m2) {
    uint32_t hd[4];
    if constexpr (A16) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(st + ob);
        hd[0] = v.x; hd[1] = v.y; hd[2] = v.z; hd[3] = v.w;
    } else {
        lds_read_dwords<4>(hd, st, ob);
    }
    const float d = h2f((uint16_t)(hd[0] & 0xFFFFu)), dmin = h2f((uint16_t)(hd[0] >> 16));
    const uint32_t s0 = hd[1], s1 = hd[2], s2 = hd[3];
    const uint32_t sc_lo = s0 & 0x3F3F3F3Fu, mn_lo = s1 & 0x3F3F3F3Fu;
    const uint32_t sc_hi = (s2 & 0x0F0F0F0Fu) | ((s0 >> 2) & 0x30303030u);
    const uint32_t mn_hi = ((s2 >> 4) & 0x0F0F0F0Fu) | ((s1 >> 2) & 0x30303030u);
    const bool upper = c >= 2;
    const int sh = 16 * (c & 1);
    const uint32_t sc = (upper ? sc_hi : sc_lo) >> sh, mn = (upper ? mn_hi : mn_lo) >> sh;
    d1 = d * ub2f(sc, 0); m1 = dmin * ub2f(mn, 0);
    d2 = d * ub2f(sc, 1); m2 = dmin * ub2f(mn, 1);
}

// N dwords at a 16-byte aligned LDS offset (A16) or at any even offset
template <int N, bool A16>
__device__ __forceinline__ void lds_read_q(uint32_t (&dst)[N], const uint8_t* st, int off) {
    if constexpr (A16) {
        static_assert(N % 4 == 0, "b128 reads");
#pragma unroll
        for (int i = 0; i < N / 4; ++i) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(st + off + 16 * i);
            dst[4 * i] = v.x; dst[4 * i + 1] = v.y; dst[4 * i + 2] = v.z; dst[4 * i + 3] = v.w;
        }
    } else {
        lds_read_dwords<N>(dst, st, off);
    }
}

template <bool A16> struct Dot<NTK_DT_Q4_K, A16> {   // reference gemm.cu:190-244
    __device__ static float run(const uint8_t* st, int shift, int lane, int ncols, const f32x2 (&x2)[32],
                                const float (&)[4], const float (&sx32)[2]) {
        if (ncols <= 0) return 0.0f;
        const int c = lane & 3, ob = shift + 144 * (lane >> 2);
        float d1, m1, d2, m2;
        kq_header<A16>(st, ob, c, d1, m1, d2, m2);
        uint32_t q[8];
        lds_read_q<8, A16>(q, st, ob + 16 + 32 * c);
        f32x2 l0 = {0.0f, 0.0f}, l1 = {0.0f, 0.0f}, h0 = {0.0f, 0.0f}, h1 = {0.0f, 0.0f};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint32_t lo = opaque(q[i] & 0x0F0F0F0Fu), hi = opaque(q[i] & 0xF0F0F0F0u);   // hi bytes = 16 n, x pre-scaled
            l0 = pkfma(ub01(lo), x2[2 * i], l0);
            l1 = pkfma(ub23(lo), x2[2 * i + 1], l1);
            h0 = pkfma(ub01(hi), x2[16 + 2 * i], h0);
            h1 = pkfma(ub23(hi), x2[16 + 2 * i + 1], h1);
        }
        float t = d1 * hsum(l0, l1);
        t = fmaf(-m1, sx32[0], t);
        t = fmaf(d2, hsum(h0, h1), t);
        t = fmaf(-m2, sx32[1], t);
        return t;
    }
};

// ---- integer activations (round 2): the lane's 64 activations as three byte planes per 32-column sub-block ------------------
// x_k ~ X_k * 2^(e-22) with X_k = rint(x_k * 2^(22-e)), e = exponent of the sub-block's largest |x| (so |X_k| < 2^22: the error per
// term, half a unit = at most 2^-22 of that maximum, is at the level of the F32 rounding of the sub-block sum it replaces).  X_k in two's complement
// = b0 + 256 b1 + 65536 s2 with b0, b1 unsigned bytes and s2 the signed top byte: plane p of four consecutive columns is one dword,
// and sum_k n_k X_k = udot4(n, b0) + 256 udot4(n, b1) + 65536 sdot4(n, s2) -- v_dot4_u32_u8 / v_dot4_i32_i8 issue at the rate of
// the byte converts (profiles/r02_valu_issue_rates.txt), so 8 weights cost 3 mask operations + 6 dot instructions instead of 2 masks
// + 8 converts + 4 packed FMAs.  The planes are built once per workgroup in the prologue's LDS image (gemv.hip).
struct XInt {
    uint32_t d[3][16];   // [plane][dword j]: columns 4j .. 4j+3 of the lane's 64 (j < 8: first 32-column group, j >= 8: second)
    float inv[2];        // 2^(e-22) of the two groups
    float sx[4];         // sums of x of the four 16-column runs (K-quant minimum term / Q6_K's -32 offset)
};
template <int DT> struct DotI;

template <> struct DotI<NTK_DT_Q4_K> {   // reference gemm.cu:190-244 (rows 16-byte aligned: the A16 form)
    __device__ static float run(const uint8_t* st, int lane, int ncols, const XInt& xi) {
        if (ncols <= 0) return 0.0f;
        const int c = lane & 3, ob = 144 * (lane >> 2);
        float d1, m1, d2, m2;
        kq_header<true>(st, ob, c, d1, m1, d2, m2);
        uint32_t q[8];
        lds_read_q<8, true>(q, st, ob + 16 + 32 * c);
        uint32_t a0 = 0, a1 = 0, b0 = 0, b1 = 0;
        int a2 = 0, b2 = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint32_t lo = q[i] & 0x0F0F0F0Fu, hi = (q[i] >> 4) & 0x0F0F0F0Fu;   // sub-block 2c / 2c + 1, columns 4i .. 4i+3
            a0 = __builtin_amdgcn_udot4(lo, xi.d[0][i], a0, false);
            a1 = __builtin_amdgcn_udot4(lo, xi.d[1][i], a1, false);
            a2 = __builtin_amdgcn_sdot4((int)lo, (int)xi.d[2][i], a2, false);
            b0 = __builtin_amdgcn_udot4(hi, xi.d[0][8 + i], b0, false);
            b1 = __builtin_amdgcn_udot4(hi, xi.d[1][8 + i], b1, false);
            b2 = __builtin_amdgcn_sdot4((int)hi, (int)xi.d[2][8 + i], b2, false);
        }
        // the planes recombine in INTEGER arithmetic: |sum_k n_k X_k| <= 15 * 32 * 2^22 < 2^31, so a0 + 256 a1 + 65536 a2 is the exact
        // sub-block sum (two v_lshl_add_u32), rounded to F32 once (before: three converts and two FMAs)
        const float fa = (float)(int)(a0 + (a1 << 8) + ((uint32_t)a2 << 16)) * xi.inv[0];
        const float fb = (float)(int)(b0 + (b1 << 8) + ((uint32_t)b2 << 16)) * xi.inv[1];
        float t = d1 * fa;
        t = fmaf(-m1, xi.sx[0] + xi.sx[1], t);
        t = fmaf(d2, fb, t);
        t = fmaf(-m2, xi.sx[2] + xi.sx[3], t);
        return t;
    }
};

template <> struct DotI<NTK_DT_Q6_K> {   // reference gemm.cu:421-459; lane = (block, half hf, t) as Dot<Q6_K>, any even alignment
    __device__ static float run(const uint8_t* st, int shift, int lane, int ncols, const XInt& xi) {
        if (ncols <= 0) return 0.0f;
        const int t = lane & 1, hf = (lane >> 1) & 1, ob = shift + 210 * (lane >> 2);
        uint32_t scd[2];
        lds_read_dwords<2>(scd, st, ob + 192 + 8 * hf);                    // the half's 8 int8 sub-scales
        const uint32_t sc_lo = scd[0] >> (16 * t), sc_hi = scd[1] >> (16 * t);
        const float d = h2f(lds_u16_at(st, ob + 208));
        float S[4];   // [type * 2 + is]
#pragma unroll
        for (int is = 0; is < 2; ++is) {   // the two 16-column runs of each type
            uint32_t A[4], H[4];
            lds_read_dwords<4>(A, st, ob + 64 * hf + 32 * t + 16 * is);
            lds_read_dwords<4>(H, st, ob + 128 + 32 * hf + 16 * is);
            uint32_t a0 = 0, a1 = 0, b0 = 0, b1 = 0;
            int a2 = 0, b2 = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t hs = H[i] >> (2 * t);
                const uint32_t qa = (A[i] & 0x0F0F0F0Fu) | ((hs & 0x03030303u) << 4);      // q in 0..63 (the -32 leaves through sx)
                const uint32_t qb = ((A[i] >> 4) & 0x0F0F0F0Fu) | (hs & 0x30303030u);
                a0 = __builtin_amdgcn_udot4(qa, xi.d[0][4 * is + i], a0, false);
                a1 = __builtin_amdgcn_udot4(qa, xi.d[1][4 * is + i], a1, false);
                a2 = __builtin_amdgcn_sdot4((int)qa, (int)xi.d[2][4 * is + i], a2, false);
                b0 = __builtin_amdgcn_udot4(qb, xi.d[0][8 + 4 * is + i], b0, false);
                b1 = __builtin_amdgcn_udot4(qb, xi.d[1][8 + 4 * is + i], b1, false);
                b2 = __builtin_amdgcn_sdot4((int)qb, (int)xi.d[2][8 + 4 * is + i], b2, false);
            }
            // (q <= 63 over 16 columns can reach 2^32: the top plane stays a float term; the two unsigned planes combine as integers)
            S[is] = fmaf(65536.0f, (float)a2, (float)(a0 + (a1 << 8))) * xi.inv[0];
            S[2 + is] = fmaf(65536.0f, (float)b2, (float)(b0 + (b1 << 8))) * xi.inv[1];
        }
        float bs = sb2f(sc_lo, 0) * fmaf(-32.0f, xi.sx[0], S[0]);
        bs = fmaf(sb2f(sc_lo, 1), fmaf(-32.0f, xi.sx[1], S[1]), bs);
        bs = fmaf(sb2f(sc_hi, 0), fmaf(-32.0f, xi.sx[2], S[2]), bs);
        bs = fmaf(sb2f(sc_hi, 1), fmaf(-32.0f, xi.sx[3], S[3]), bs);
        return d * bs;
    }
};

template <bool A16> struct Dot<NTK_DT_Q5_K, A16> {   // reference gemm.cu:297-354
    __device__ static float run(const uint8_t* st, int shift, int lane, int ncols, const f32x2 (&x2)[32],
                                const float (&)[4], const float (&sx32)[2]) {
        if (ncols <= 0) return 0.0f;
        const int c = lane & 3, ob = shift + 176 * (lane >> 2);
        float d1, m1, d2, m2;
        kq_header<A16>(st, ob, c, d1, m1, d2, m2);
        f32x2 l0 = {0.0f, 0.0f}, l1 = {0.0f, 0.0f}, h0 = {0.0f, 0.0f}, h1 = {0.0f, 0.0f};
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {   // two 16-byte halves: keeps the live dword count low
            uint32_t qh[4], ql[4];
            lds_read_q<4, A16>(qh, st, ob + 16 + 16 * hh);
            lds_read_q<4, A16>(ql, st, ob + 48 + 32 * c + 16 * hh);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                // rotate right by 2c - 4: bit 2c of every qh byte lands on bit 4, bit 2c + 1 on bit 5 (what wraps in from the
                // neighbouring byte falls outside the masks)
                const uint32_t r = __builtin_amdgcn_alignbit(qh[i], qh[i], (uint32_t)((2 * c + 28) & 31));
                const uint32_t lo = opaque((ql[i] & 0x0F0F0F0Fu) | (r & 0x10101010u));
                const uint32_t hi = opaque(((ql[i] >> 4) & 0x0F0F0F0Fu) | ((r >> 1) & 0x10101010u));
                l0 = pkfma(ub01(lo), x2[8 * hh + 2 * i], l0);
                l1 = pkfma(ub23(lo), x2[8 * hh + 2 * i + 1], l1);
                h0 = pkfma(ub01(hi), x2[16 + 8 * hh + 2 * i], h0);
                h1 = pkfma(ub23(hi), x2[16 + 8 * hh + 2 * i + 1], h1);
            }
        }
        float t = d1 * hsum(l0, l1);
        t = fmaf(-m1, sx32[0], t);
        t = fmaf(d2, hsum(h0, h1), t);
        t = fmaf(-m2, sx32[1], t);
        return t;
    }
};

// Q6_K (reference gemm.cu:421-459).  lane = (block, half hf, t).  Within a half the 128 weights are
//   q1: ql[l] & 15 | (qh[l] & 3) << 4 -> column l          q3: ql[l] >> 4 | ((qh[l] >> 4) & 3) << 4 -> column 64 + l
//   q2: ql[32+l] & 15 | ((qh[l] >> 2) & 3) << 4 -> 32 + l   q4: ql[32+l] >> 4 | ((qh[l] >> 6) & 3) << 4 -> 96 + l
// so lane t takes the 32 bytes ql[32t .. 32t+31] whole: their LOW nibbles are columns 32t + l (q1 / q2), their HIGH
// nibbles columns 64 + 32t + l (q3 / q4), with the 2-bit tops at bits 2t and 4 + 2t of qh[l].  The lane's activations
// are loaded to match (kernel: read_own_row): x2[0..15] = columns 128hf + 32t + [0,32), x2[16..31] = the same + 64.
// Seven VALU ops turn one ql dword + one qh dword into 8 weights; sub-scale index = l/16 + {2t, 4 + 2t}.
template <bool A16> struct Dot<NTK_DT_Q6_K, A16> {
    __device__ static float run(const uint8_t* st, int shift, int lane, int ncols, const f32x2 (&x2)[32],
                                const float (&sx16)[4], const float (&)[2]) {
        if (ncols <= 0) return 0.0f;
        const int t = lane & 1, hf = (lane >> 1) & 1, ob = shift + 210 * (lane >> 2);
        uint32_t scd[2];
        lds_read_dwords<2>(scd, st, ob + 192 + 8 * hf);                    // the half's 8 int8 sub-scales
        const uint32_t sc_lo = scd[0] >> (16 * t), sc_hi = scd[1] >> (16 * t);   // bytes 0,1: is = 0,1
        const float d = h2f(lds_u16_at(st, ob + 208));
        float S[4];   // [type * 2 + is]
#pragma unroll
        for (int is = 0; is < 2; ++is) {   // the two 16-column runs of each type (sub-scale index is)
            uint32_t A[4], H[4];
            lds_read_dwords<4>(A, st, ob + 64 * hf + 32 * t + 16 * is);
            lds_read_dwords<4>(H, st, ob + 128 + 32 * hf + 16 * is);
            f32x2 a0 = {0.0f, 0.0f}, a1 = {0.0f, 0.0f}, b0 = {0.0f, 0.0f}, b1 = {0.0f, 0.0f};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t hs = H[i] >> (2 * t);
                const uint32_t qa = opaque((A[i] & 0x0F0F0F0Fu) | ((hs & 0x03030303u) << 4));
                const uint32_t qb = opaque(((A[i] >> 4) & 0x0F0F0F0Fu) | (hs & 0x30303030u));
                a0 = pkfma(ub01(qa), x2[8 * is + 2 * i], a0);
                a1 = pkfma(ub23(qa), x2[8 * is + 2 * i + 1], a1);
                b0 = pkfma(ub01(qb), x2[16 + 8 * is + 2 * i], b0);
                b1 = pkfma(ub23(qb), x2[16 + 8 * is + 2 * i + 1], b1);
            }
            S[is] = hsum(a0, a1);
            S[2 + is] = hsum(b0, b1);
        }
        // sum (q-32) x = sum q x - 32 sum x, per 16-column run, times the run's int8 sub-scale
        float bs = sb2f(sc_lo, 0) * fmaf(-32.0f, sx16[0], S[0]);
        bs = fmaf(sb2f(sc_lo, 1), fmaf(-32.0f, sx16[1], S[1]), bs);
        bs = fmaf(sb2f(sc_hi, 0), fmaf(-32.0f, sx16[2], S[2]), bs);
        bs = fmaf(sb2f(sc_hi, 1), fmaf(-32.0f, sx16[3], S[3]), bs);
        return d * bs;
    }
};


}  // namespace ntk
