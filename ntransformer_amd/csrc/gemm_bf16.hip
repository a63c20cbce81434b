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
constexpr int GB_MAX_SPLIT = 8;     // K splits x token chunks of one launch never exceed this (size of the partial-sum area)
constexpr int GB_MAX_CHUNKS = 16;   // 64-token chunks per launch (32 measured no better: 15.1k vs 16.0k tok/s at 2048 tokens)
constexpr int GB_PIECE = 1024;       // bytes of one (step, plane, token block) operand record
constexpr int GB_STEP_BYTES = 3 * 4 * GB_PIECE;           // 12 KB of B operands per step

__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {   // upper halves of two floats (exact for our small integers)
    return __builtin_amdgcn_perm(__float_as_uint(hi), __float_as_uint(lo), 0x07060302u);
}
typedef uint32_t __attribute__((aligned(2))) u32_a2;
typedef uint16_t __attribute__((aligned(2))) u16_a2;
__device__ __forceinline__ uint32_t ld32(const uint8_t* p) { return *reinterpret_cast<const u32_a2*>(p); }   // 2-byte aligned global dword
__device__ __forceinline__ uint16_t ld16(const uint8_t* p) { return *reinterpret_cast<const u16_a2*>(p); }

// ---- pre-pass: X -> BF16 planes in operand order + per-step sums ----------------------------------------------------------
// grid = (in / 32 steps + 1 (a record of zeros), 64-token chunks), block = 256 = 4 token blocks x 64 lanes
__global__ __launch_bounds__(256) void split_x_kernel(const float* __restrict__ X, int T, int in, u32x4* __restrict__ xb, float* __restrict__ xsum,
                                                      size_t chunk_bytes) {
    const int step = blockIdx.x, tb = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int j = lane & 15, g = lane >> 4, t = tb * 16 + j;
    // blockIdx.y = 64-token chunk: its tokens, its planes
    X += (size_t)blockIdx.y * GB_TOK * in;
    T = min(GB_TOK, T - (int)blockIdx.y * GB_TOK);
    xb = reinterpret_cast<u32x4*>(reinterpret_cast<uint8_t*>(xb) + blockIdx.y * chunk_bytes);
    xsum = reinterpret_cast<float*>(reinterpret_cast<uint8_t*>(xsum) + blockIdx.y * chunk_bytes);
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = 0.0f;
    if (t < T && step * 32 < in) {   // block in/32 writes the all-zero record that K ranges rounded up to whole trips read
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

// ---- per-format weight operand ---------------------------------------------------------------------------------------------
// The raw GGUF rows travel HBM -> registers -> a per-wave LDS image in UNITS of whole blocks (Q8_0: 4 blocks = 136 B, K-quants:
// one 256-column super-block), as 16-byte pieces of the 16-byte-aligned window that covers the unit: 16 bytes per lane and
// NCH consecutive lanes per row, i.e. >= 144 contiguous bytes per row and request (a lane-per-slot gather of 4-byte pieces,
// 16 bytes per row and request, tops out near 2.4 TB/s in the address coalescer).  Each 32-column step then reads its slot
// (row i, columns {4g..4g+3, 16+4g..16+4g+3}) out of the image with 2-byte-aligned LDS dword reads.  STRIDE (bytes between row
// images) = 4 x an odd number of dwords... chosen so that the 16 rows x 4 column groups of one read hit 64 different banks.
struct AOp {
    u32x4 a;            // 8 BF16 integers
    float s0, s1;       // scale of the slot's low / high 4 columns (equal unless the format scales per 16 columns)
    float mn;           // K-quant minimum term factor (dmin * m), 0 otherwise
};
__device__ __forceinline__ uint32_t lds32(const uint8_t* p) { return *reinterpret_cast<const u32_a2*>(p); }   // 2-byte aligned LDS dword
__device__ __forceinline__ uint32_t lds16(const uint8_t* p) { return *reinterpret_cast<const u16_a2*>(p); }
template <int DT> struct DeqI;

template <> struct DeqI<NTK_DT_Q8_0> {   // types.h:104-108: half d, int8 qs[32]
    static constexpr int BW = 32, BB = 34;
    static constexpr int SPU = 4, UB = 136, NCH = 10, STRIDE = 176;   // window: shift (0 or 8) + 136 <= 160
    static constexpr bool SPLIT16 = false, HAS_MIN = false;
    struct Hdr {};
    __device__ static Hdr header(const uint8_t*) { return Hdr{}; }
    // row = the row's image + the unit's shift, rowg = row + 4 g; j = step within the unit
    __device__ static AOp step(const uint8_t* row, const uint8_t* rowg, const Hdr&, int j) {
        const uint32_t lo = lds32(rowg + 34 * j + 2), hi = lds32(rowg + 34 * j + 18);
        AOp o;
        o.a = u32x4{pack_bf16(sb2f(lo, 0), sb2f(lo, 1)), pack_bf16(sb2f(lo, 2), sb2f(lo, 3)),
                    pack_bf16(sb2f(hi, 0), sb2f(hi, 1)), pack_bf16(sb2f(hi, 2), sb2f(hi, 3))};
        o.s0 = o.s1 = h2f((uint16_t)lds16(row + 34 * j));
        o.mn = 0.0f;
        return o;
    }
};

template <> struct DeqI<NTK_DT_Q4_K> {   // types.h:112-117: half d, dmin; 12 packed 6-bit (scale, min); 128 bytes of nibbles
    static constexpr int BW = 256, BB = 144;
    static constexpr int SPU = 8, UB = 144, NCH = 9, STRIDE = 144;     // rows are 16-byte aligned: no shift
    static constexpr bool SPLIT16 = false, HAS_MIN = true;
    struct Hdr { u32x4 h; };   // d | dmin, 12 scale bytes
    __device__ static Hdr header(const uint8_t* row) { return Hdr{*reinterpret_cast<const u32x4*>(row)}; }
    __device__ static AOp step(const uint8_t* /*row*/, const uint8_t* rowg, const Hdr& hd, int j) {
        float sc, mn;
        kq_scale_min(hd.h.y, hd.h.z, hd.h.w, j, sc, mn);                       // gemm.cu:206-222
        const float d = h2f((uint16_t)(hd.h.x & 0xFFFFu)), dmin = h2f((uint16_t)(hd.h.x >> 16));
        const int sh = 4 * (j & 1);                                           // even sub-block: low nibbles, odd: high
        const uint32_t lo = (lds32(rowg + 16 + 32 * (j >> 1)) >> sh) & 0x0F0F0F0Fu;
        const uint32_t hi = (lds32(rowg + 32 + 32 * (j >> 1)) >> sh) & 0x0F0F0F0Fu;
        AOp o;
        o.a = u32x4{pack_bf16(ub2f(lo, 0), ub2f(lo, 1)), pack_bf16(ub2f(lo, 2), ub2f(lo, 3)),
                    pack_bf16(ub2f(hi, 0), ub2f(hi, 1)), pack_bf16(ub2f(hi, 2), ub2f(hi, 3))};
        o.s0 = o.s1 = d * sc;
        o.mn = dmin * mn;
        return o;
    }
};

template <> struct DeqI<NTK_DT_Q5_K> {   // types.h:122-128: half d, dmin; 12 packed 6-bit (scale, min); qh[32]; ql[128]
    static constexpr int BW = 256, BB = 176;
    static constexpr int SPU = 8, UB = 176, NCH = 11, STRIDE = 176;    // rows are 16-byte aligned: no shift
    static constexpr bool SPLIT16 = false, HAS_MIN = true;
    struct Hdr { u32x4 h; };   // d | dmin, 12 scale bytes
    __device__ static Hdr header(const uint8_t* row) { return Hdr{*reinterpret_cast<const u32x4*>(row)}; }
    __device__ static AOp step(const uint8_t* /*row*/, const uint8_t* rowg, const Hdr& hd, int j) {
        float sc, mn;
        kq_scale_min(hd.h.y, hd.h.z, hd.h.w, j, sc, mn);                       // gemm.cu:206-222 (same packing as Q4_K)
        const float d = h2f((uint16_t)(hd.h.x & 0xFFFFu)), dmin = h2f((uint16_t)(hd.h.x >> 16));
        const int sh = 4 * (j & 1);                                           // even sub-block: low nibbles, odd: high
        // fifth bit of column l of sub-block j: bit j of qh[l]   (gemm.cu:297-354: u1 = 1 << 2c, u2 = 2 << 2c)
        const uint32_t lo = ((lds32(rowg + 48 + 32 * (j >> 1)) >> sh) & 0x0F0F0F0Fu) | (((lds32(rowg + 16) >> j) & 0x01010101u) << 4);
        const uint32_t hi = ((lds32(rowg + 64 + 32 * (j >> 1)) >> sh) & 0x0F0F0F0Fu) | (((lds32(rowg + 32) >> j) & 0x01010101u) << 4);
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
    static constexpr int SPU = 8, UB = 210, NCH = 14, STRIDE = 240;    // window: shift (even, <= 14) + 210 <= 224
    static constexpr bool SPLIT16 = true, HAS_MIN = false;
    struct Hdr { float d; };
    __device__ static Hdr header(const uint8_t* row) { return Hdr{h2f((uint16_t)lds16(row + 208))}; }
    __device__ static AOp step(const uint8_t* row, const uint8_t* rowg, const Hdr& hd, int j) {
        const int hf = j >> 2, t = j & 3;
        const uint8_t* ql = rowg + 64 * hf + 32 * (t & 1);
        const uint8_t* qh = rowg + 128 + 32 * hf;
        const int sl = 4 * (t >> 1), sh = 2 * t;
        const uint32_t lo = ((lds32(ql) >> sl) & 0x0F0F0F0Fu) | (((lds32(qh) >> sh) & 0x03030303u) << 4);         // gemm.cu:421-459
        const uint32_t hi = ((lds32(ql + 16) >> sl) & 0x0F0F0F0Fu) | (((lds32(qh + 16) >> sh) & 0x03030303u) << 4);
        AOp o;   // q - 32: exact small integers
        o.a = u32x4{pack_bf16(ub2f(lo, 0) - 32.0f, ub2f(lo, 1) - 32.0f), pack_bf16(ub2f(lo, 2) - 32.0f, ub2f(lo, 3) - 32.0f),
                    pack_bf16(ub2f(hi, 0) - 32.0f, ub2f(hi, 1) - 32.0f), pack_bf16(ub2f(hi, 2) - 32.0f, ub2f(hi, 3) - 32.0f)};
        const uint32_t sc = lds16(row + 192 + 8 * hf + 2 * t);   // the two int8 sub-scales of the step's 16-column halves
        o.s0 = hd.d * (float)(int)(int8_t)(sc & 0xFF);
        o.s1 = hd.d * (float)(int)(int8_t)(sc >> 8);
        o.mn = 0.0f;
        return o;
    }
};

constexpr int GB_NR = 2;      // weight units in flight per wave (register ring): 8 (Q8_0) / 16 (K-quants) steps ahead of the MFMAs
constexpr int GB_SLOTS = 4;   // LDS ring of B-operand step records (12 KB each), filled by LDS-DMA GB_SLOTS - 1 steps ahead
constexpr int GB_XS_OFF = GB_SLOTS * GB_STEP_BYTES;   // then [GB_SLOTS][64] floats: the steps' per-token sums of x (K-quant minimum)
constexpr int GB_STAGE_OFF = GB_XS_OFF + GB_SLOTS * GB_TOK * 4;   // then the 4 waves' weight images
template <int DT, int RT> constexpr int gb_lds_bytes() { return GB_STAGE_OFF + 4 * 16 * RT * DeqI<DT>::STRIDE; }

constexpr int GB_MAX_SEG = 3;   // matrices sharing X in one launch (Q | K | V, gate | up)
struct GemmBSeg {
    const uint8_t* W;
    float* Y;               // [T][out]
    float* part;            // K split: [split][T][out] partial sums of this matrix
    int out, tile0;         // rows; first row tile of this matrix in the launch's tile numbering
    unsigned w_last;        // out * row_bytes - 16: the last 16-byte piece of the matrix (requests past the end re-read it)
};
struct GemmBParams {
    GemmBSeg seg[GB_MAX_SEG];
    int nseg;
    const uint8_t* xb;      // operand planes [steps][3][4][64] x 16 B
    const float* xsum;      // [steps][64]
    const float* resid;     // optional [T][out] (single matrix only), may alias Y
    int T, in, steps;       // T = tokens of the launch (<= 16 chunks of 64)
    unsigned row_bytes;
    int nsplit, steps_per_split;   // blockIdx.y = K split; nsplit > 1: partial sums go to seg.part, summed by reduce_splits
    int chunks, row_wgs;    // 64-token chunks of this launch (blockIdx.x enumerates (row tile, chunk), see below); row tiles of all matrices
    size_t chunk_bytes;     // between the planes (and sums) of consecutive chunks
};

// 16 B per lane, global -> LDS without passing through registers (gfx950 LDS-DMA, b128 form): lane l's 16 bytes land at
// M0 + imm + 16 l, read from gsrc + imm.  The compiler does not count these requests: the waits on them are explicit (vmcnt).
__device__ __forceinline__ void gb_dma16(uint32_t lds_dst, const uint8_t* gsrc) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void gb_dma16x3(uint32_t lds_dst, const uint8_t* gsrc) {   // 3 KB per wave: +0, +1024, +2048 on both sides
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                 "global_load_lds_dwordx4 %1, off offset:1024\n\tglobal_load_lds_dwordx4 %1, off offset:2048\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// block = 256 threads = 4 waves; wave w of workgroup b owns weight rows (4 b + w) * 16 RT ... + 16 RT and all 64 tokens.
// MFMA roles: the activation planes are the A operand (M = 16 tokens), the weights the B operand (N = 16 weight rows), so the
// accumulator of lane (i, g) holds weight row i for tokens 4 g + e -- the row's scale is the one this lane decoded (no
// cross-lane traffic), and the step's sum of x comes as one float4 from LDS.
// One loop trip = GB_NR units of SPU steps, straight-line (no exits inside: the s_waitcnt counts are exact).  Per step s:
//   wait until the DMA of step s has landed | barrier | at a unit's first step: the unit's raw rows go from their ring registers
//   to the wave's LDS image and the ring slot is re-requested GB_NR units ahead | read + decode the step's weight slot | DMA step
//   s + GB_SLOTS - 1 into the slot step s - 1 just vacated | 12 RT MFMAs.
template <int DT, int RT>
__global__ __launch_bounds__(256, 2) void gemm_quant_bf16_kernel(const GemmBParams p) {
    using D = DeqI<DT>;
    constexpr int SPU = D::SPU, NCH = D::NCH, STRIDE = D::STRIDE;
    constexpr int ROWS = 16 * RT, PIECES = ROWS * NCH, NLD = (PIECES + 63) / 64;   // 16-byte pieces of a unit; requests per lane
    static_assert((GB_NR * SPU) % GB_SLOTS == 0, "a trip must cover whole turns of the B ring");
    extern __shared__ __attribute__((aligned(16))) uint8_t gb_lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, g = lane >> 4;
    // blockIdx.x -> (row tile, token chunk).  Workgroups go to the 8 XCDs round-robin; the `chunks` workgroups that share a row
    // tile's weights get ids 8 apart (same XCD, same L2, dispatched back to back): id = 8 * (chunks * (tile / 8) + chunk) + tile % 8
    const int bid = (int)blockIdx.x;
    const int xcd = bid & 7, within = bid >> 3;
    const int chunk = within % p.chunks, tile = (within / p.chunks) * 8 + xcd;
    if (tile >= p.row_wgs) return;   // row tiles are padded to a multiple of 8
    // which matrix of the launch this row tile belongs to (workgroup-uniform)
    int sidx = 0;
    if (p.nseg > 1 && tile >= p.seg[1].tile0) sidx = 1;
    if (p.nseg > 2 && tile >= p.seg[2].tile0) sidx = 2;
    const uint8_t* const segW = p.seg[sidx].W;
    float* const segY = p.seg[sidx].Y;
    float* const segPart = p.seg[sidx].part;
    const int seg_out = p.seg[sidx].out;
    const unsigned seg_w_last = p.seg[sidx].w_last;
    const int Tc = min(GB_TOK, p.T - chunk * GB_TOK);
    const int row0 = ((tile - p.seg[sidx].tile0) * 4 + wave) * ROWS;
    f32x4 acc[RT][4];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int tb = 0; tb < 4; ++tb) acc[rt][tb] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    // this workgroup's K range (split-K: blockIdx.y), in steps of 32 columns; whole trips of GB_NR * SPU steps
    const int step_lo = (int)blockIdx.y * p.steps_per_split, step_hi = min(p.steps, step_lo + p.steps_per_split);
    const int nsteps = step_hi - step_lo;
    const int unit_lo = step_lo / SPU, nunits = (nsteps + SPU - 1) / SPU;

    // weight pieces of this lane: piece q = 64 n + lane of the wave's unit -> row q / NCH, 16-byte piece q % NCH of the 16-byte
    // aligned window that covers the row's unit (rows need not be 16-byte aligned: every row has its own window start and shift)
    uint8_t* stage = gb_lds + GB_STAGE_OFF + (size_t)wave * (ROWS * STRIDE);
    uint32_t w_row[NLD], w_c16[NLD], s_off[NLD];
#pragma unroll
    for (int n = 0; n < NLD; ++n) {
        const int q = min(64 * n + lane, PIECES - 1), r = q / NCH, c = q - r * NCH;
        w_row[n] = (uint32_t)min(row0 + r, seg_out - 1) * p.row_bytes;
        w_c16[n] = 16u * c;
        s_off[n] = (uint32_t)(r * STRIDE + 16 * c);
    }
    u32x4 ring[GB_NR][NLD];
    auto load_unit = [&](int k, int urel) {   // unit `urel` of this split (past the end: the last one again, multiplied by zeros)
        const uint32_t uoff = (uint32_t)(unit_lo + min(urel, nunits - 1)) * D::UB;
#pragma unroll
        for (int n = 0; n < NLD; ++n)
            ring[k][n] = *reinterpret_cast<const u32x4*>(segW + min(((w_row[n] + uoff) & ~15u) + w_c16[n], seg_w_last));
    };
    auto stage_unit = [&](int k) {
#pragma unroll
        for (int n = 0; n < NLD; ++n) *reinterpret_cast<u32x4*>(stage + s_off[n]) = ring[k][n];
    };

    const uint32_t lds0 = (uint32_t)(uintptr_t)gb_lds;   // generic -> LDS address: the low 32 bits
    const uint8_t* xb_thread = p.xb + (size_t)chunk * p.chunk_bytes + (size_t)wave * 3072 + (size_t)lane * 16;   // wave w copies bytes [3072 w, 3072 w + 3072) of a step record
    constexpr int ND = D::HAS_MIN ? 4 : 3;               // DMA requests per step and wave
    auto dma_step = [&](int rel, int slot) {             // step record `rel` (past the end: the record of zeros)
        const int s = rel < nsteps ? step_lo + rel : p.steps;
        const uint8_t* src = xb_thread + (size_t)s * GB_STEP_BYTES;
        gb_dma16x3(__builtin_amdgcn_readfirstlane(lds0 + (uint32_t)slot * GB_STEP_BYTES + (uint32_t)wave * 3072u), src);
        if (D::HAS_MIN && lane < 4)   // the step's 64 sums: 16 floats per wave
            gb_dma16(__builtin_amdgcn_readfirstlane(lds0 + GB_XS_OFF + (uint32_t)slot * 256u + (uint32_t)wave * 64u),
                     reinterpret_cast<const uint8_t*>(p.xsum) + (size_t)chunk * p.chunk_bytes + ((size_t)s * GB_TOK + wave * 16 + lane * 4) * sizeof(float));
    };
    const uint8_t* img[RT];      // this lane's row images (row rt*16 + i of the wave's tile)
    uint32_t my_row[RT];         // and the rows' byte offsets in W: the unit's bytes start `(my_row + unit offset) & 15` into the image
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        img[rt] = stage + (rt * 16 + i) * STRIDE;
        my_row[rt] = (uint32_t)min(row0 + rt * 16 + i, seg_out - 1) * p.row_bytes;
    }
    typename D::Hdr hdr[RT];
    const uint8_t* cur[RT];      // img + the staged unit's shift
    auto enter_unit = [&](int unit) {   // the image now holds `unit`
        const uint32_t uoff = (uint32_t)(unit_lo + min(unit, nunits - 1)) * D::UB;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            cur[rt] = img[rt] + ((my_row[rt] + uoff) & 15u);
            hdr[rt] = D::header(cur[rt]);
        }
    };
    // Prologue, in the steady state's request order (the waits below count requests): the ring; DMA 0, 1; unit 0 staged and its
    // ring slot re-requested; DMA 2; step 0 decoded.
#pragma unroll
    for (int k = 0; k < GB_NR; ++k) load_unit(k, k);
    dma_step(0, 0);
    dma_step(1, 1);
    stage_unit(0);
    load_unit(0, GB_NR);
    dma_step(2, 2);
    static_assert(GB_SLOTS == 4, "the prologue is written out for a B ring of 4");
    enter_unit(0);
    AOp a[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) a[rt] = D::step(cur[rt], cur[rt] + 4 * g, hdr[rt], 0);

    for (int trip = 0; trip * (GB_NR * SPU) < nsteps; ++trip) {
#pragma unroll
        for (int k = 0; k < GB_NR; ++k) {
#pragma unroll
            for (int j = 0; j < SPU; ++j) {
                const int rel = (trip * GB_NR + k) * SPU + j;        // step relative to the split's start
                const int slot = (k * SPU + j) % GB_SLOTS;           // its B ring slot (compile time)
                // requests younger than the DMA of this step: the DMAs of the two steps since, and the next unit's weight requests if
                // they went out in one of those two steps (they go out in a unit's LAST step, ahead of that step's DMA)
                if (j == 0 || j == 1) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * ND + NLD) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * ND) : "memory");
                if (j == SPU - 1) {   // the weights of this step were decoded a step ago: the image is free for the next unit
                    stage_unit((k + 1) % GB_NR);
                    load_unit((k + 1) % GB_NR, trip * GB_NR + k + 1 + GB_NR);
                    enter_unit(trip * GB_NR + k + 1);
                }
                dma_step(rel + GB_SLOTS - 1, (slot + GB_SLOTS - 1) % GB_SLOTS);
                // one scheduling region from here to the end of the step: the next step's weight slot is read and decoded in the
                // shadow of this step's MFMAs
                AOp an[RT];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) an[rt] = D::step(cur[rt], cur[rt] + 4 * g, hdr[rt], (j + 1) % SPU);

                const u32x4* bs = reinterpret_cast<const u32x4*>(gb_lds + (size_t)slot * GB_STEP_BYTES);
                const f32x4* xs = reinterpret_cast<const f32x4*>(gb_lds + GB_XS_OFF + (size_t)slot * 256);
                // all B operands of the step up front (48 registers), then 4 RT independent accumulation chains, plane-major: a
                // chain's next MFMA is 4 RT - 1 issues away, so none waits for its predecessor
                u32x4 b[3][4];
                f32x4 xsum_t[4];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                    for (int tb = 0; tb < 4; ++tb) b[pl][tb] = bs[(pl * 4 + tb) * 64 + lane];
#pragma unroll
                for (int tb = 0; tb < 4; ++tb) xsum_t[tb] = D::HAS_MIN ? xs[tb * 4 + g] : f32x4{0.0f, 0.0f, 0.0f, 0.0f};   // tokens tb*16 + 4g + e
                if constexpr (D::SPLIT16) {   // two 16-column groups with their own scale: K = 16 MFMAs on the operand halves
                    f32x4 cl[RT][4], ch[RT][4];
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                        for (int tb = 0; tb < 4; ++tb) {
                            const s16x4 xl = __builtin_bit_cast(s16x4, (uint64_t)b[pl][tb].x | ((uint64_t)b[pl][tb].y << 32));
                            const s16x4 xh = __builtin_bit_cast(s16x4, (uint64_t)b[pl][tb].z | ((uint64_t)b[pl][tb].w << 32));
#pragma unroll
                            for (int rt = 0; rt < RT; ++rt) {
                                const s16x4 wl = __builtin_bit_cast(s16x4, (uint64_t)a[rt].a.x | ((uint64_t)a[rt].a.y << 32));
                                const s16x4 wh = __builtin_bit_cast(s16x4, (uint64_t)a[rt].a.z | ((uint64_t)a[rt].a.w << 32));
                                const f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
                                cl[rt][tb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(xl, wl, pl ? cl[rt][tb] : z, 0, 0, 0);
                                ch[rt][tb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(xh, wh, pl ? ch[rt][tb] : z, 0, 0, 0);
                            }
                        }
#pragma unroll
                    for (int tb = 0; tb < 4; ++tb)
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                acc[rt][tb][e] = fmaf(a[rt].s1, ch[rt][tb][e], fmaf(a[rt].s0, cl[rt][tb][e], acc[rt][tb][e]));
                } else {
                    f32x4 cc[RT][4];
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                        for (int tb = 0; tb < 4; ++tb)
#pragma unroll
                            for (int rt = 0; rt < RT; ++rt) {
                                const f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
                                cc[rt][tb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b[pl][tb]), __builtin_bit_cast(bf16x8, a[rt].a),
                                                                                     pl ? cc[rt][tb] : z, 0, 0, 0);
                            }
#pragma unroll
                    for (int tb = 0; tb < 4; ++tb)
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                float v = fmaf(a[rt].s0, cc[rt][tb][e], acc[rt][tb][e]);
                                if (D::HAS_MIN) v = fmaf(-a[rt].mn, xsum_t[tb][e], v);   // - dmin * m * sum x   (gemm.cu:232-244)
                                acc[rt][tb][e] = v;
                            }
                }
                // issue order of the region: every LDS read first, then the MFMAs with the next step's decode (and this step's
                // scale-FMAs once their chains close) in their shadow
                if constexpr (!D::SPLIT16) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 64, 0);
#pragma unroll
                    for (int n = 0; n < 12 * RT; ++n) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                    }
                }
                // the accumulators and the next operands are complete HERE: without this anchor the instruction selector parks every
                // step's scale-FMAs at the end of the trip (they have no memory dependence) and the products of 16 steps sit in
                // registers until then
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
                    for (int tb = 0; tb < 4; ++tb) asm volatile("" : "+v"(acc[rt][tb]));
                    a[rt] = an[rt];
                    asm volatile("" : "+v"(a[rt].a), "+v"(a[rt].s0), "+v"(a[rt].s1), "+v"(a[rt].mn));
                }
                __builtin_amdgcn_sched_barrier(0);   // no motion of memory requests across steps (the waits count them in order)
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no DMA may still be in flight towards LDS when the workgroup retires
    // ---- epilogue: accumulator element e of lane (i = weight row of the tile, g) is token tb*16 + 4g + e of the chunk ------
    const size_t tok0 = (size_t)chunk * GB_TOK;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int r = row0 + rt * 16 + i;
        if (r >= seg_out) continue;
        if (p.nsplit > 1) {   // K split: this workgroup's partial sums, combined (fixed order) by reduce_splits_kernel
#pragma unroll
            for (int tb = 0; tb < 4; ++tb)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int t = tb * 16 + 4 * g + e;
                    if (t < Tc) segPart[((size_t)blockIdx.y * p.T + tok0 + t) * seg_out + r] = acc[rt][tb][e];
                }
            continue;
        }
        float rs[4][4];
#pragma unroll
        for (int tb = 0; tb < 4; ++tb)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int t = tb * 16 + 4 * g + e;
                rs[tb][e] = (p.resid && t < Tc) ? p.resid[(tok0 + t) * seg_out + r] : 0.0f;
            }
#pragma unroll
        for (int tb = 0; tb < 4; ++tb)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int t = tb * 16 + 4 * g + e;
                if (t < Tc) segY[(tok0 + t) * seg_out + r] = acc[rt][tb][e] + rs[tb][e];
            }
    }
}

// Y[t][r] = sum over splits (in order) of part[s][t][r] (+ resid): one float4 per thread, blockIdx.y = matrix of the launch
struct ReduceArgs {
    float* Y[GB_MAX_SEG];
    const float* part[GB_MAX_SEG];
    int out[GB_MAX_SEG];
    const float* resid;
    int nseg, T, nsplit;
};
__global__ __launch_bounds__(256) void reduce_splits_kernel(const ReduceArgs a) {
    const int sg = blockIdx.y;
    float* Y = a.Y[0];
    const float* part = a.part[0];
    int out = a.out[0];
    if (sg == 1) { Y = a.Y[1]; part = a.part[1]; out = a.out[1]; }
    if (sg == 2) { Y = a.Y[2]; part = a.part[2]; out = a.out[2]; }
    const size_t idx = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (idx >= (size_t)a.T * out) return;
    float4 v = *reinterpret_cast<const float4*>(part + idx);
    for (int s = 1; s < a.nsplit; ++s) {
        const float4 q = *reinterpret_cast<const float4*>(part + (size_t)s * a.T * out + idx);
        v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
    }
    if (a.resid) {
        const float4 q = *reinterpret_cast<const float4*>(a.resid + idx);
        v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
    }
    *reinterpret_cast<float4*>(Y + idx) = v;
}

// one chunk's planes + sums (+ the record of zeros), rounded to 256 B
static size_t ws_chunk_bytes(int in) { return ((size_t)(in / 32 + 1) * (GB_STEP_BYTES + GB_TOK * sizeof(float)) + 255) / 256 * 256; }

struct HostSeg { float* Y; const void* W; int out; };

// T <= GB_MAX_CHUNKS * 64 = 1024 tokens in one launch; nseg matrices [out_s][in] of one format sharing X
template <int DT>
static int launch_gemm_bf16(const HostSeg* segs, int nseg, const float* X, int T, int in, const float* resid, void* ws, int reuse_x,
                            hipStream_t st) {
    using D = DeqI<DT>;
    constexpr int TRIP = GB_NR * D::SPU;   // steps per loop trip: K ranges are whole trips
    if (nseg < 1 || nseg > GB_MAX_SEG || (resid && nseg != 1) || in % D::BW != 0) return NTK_E_SHAPE;
    const size_t row_bytes = (size_t)in / D::BW * D::BB;
    long out_total = 0;
    for (int i = 0; i < nseg; ++i) {
        if (segs[i].out <= 0 || segs[i].out % 16 != 0 || (size_t)segs[i].out * row_bytes > 0xFFFFFF00ull) return NTK_E_SHAPE;   // 32-bit piece offsets
        if ((reinterpret_cast<uintptr_t>(segs[i].W) & 15) || (reinterpret_cast<uintptr_t>(segs[i].Y) & 15)) return NTK_E_ALIGN;
        out_total += segs[i].out;
    }
    if ((reinterpret_cast<uintptr_t>(X) & 15) || (resid && (reinterpret_cast<uintptr_t>(resid) & 15))) return NTK_E_ALIGN;
    GemmBParams p{};
    p.nseg = nseg;
    p.T = T; p.in = in; p.steps = in / 32;
    p.chunks = (T + GB_TOK - 1) / GB_TOK;
    p.chunk_bytes = ws_chunk_bytes(in);
    p.row_bytes = (unsigned)row_bytes;
    uint8_t* wsb = static_cast<uint8_t*>(ws);
    p.xb = wsb;
    p.xsum = reinterpret_cast<const float*>(wsb + (size_t)(p.steps + 1) * GB_STEP_BYTES);
    p.resid = resid;
    if (!reuse_x)
        hipLaunchKernelGGL(split_x_kernel, dim3(p.steps + 1, p.chunks), dim3(256), 0, st, X, T, in, reinterpret_cast<u32x4*>(wsb), const_cast<float*>(p.xsum),
                           p.chunk_bytes);
    // Rows per wave (RT x 16): a workgroup streams ALL B operands of its K range from L2 whatever its height, so taller tiles
    // cut that traffic and the LDS reads per MFMA; K is then split (in whole trips) while fewer than one workgroup per CU exists.
    static const int force_rt = [] { const char* e = getenv("NTK_GEMM_RT"); return e ? atoi(e) : 0; }();
    static const int want_wgs = [] { const char* e = getenv("NTK_GEMM_WGS"); return e ? atoi(e) : 256; }();
    int rt = out_total >= 2048 ? 2 : 1;
    if (force_rt == 1 || force_rt == 2) rt = force_rt;
    int tiles = 0;
    float* part = reinterpret_cast<float*>(wsb + (size_t)GB_MAX_CHUNKS * p.chunk_bytes);
    for (int i = 0; i < nseg; ++i) {
        p.seg[i].W = static_cast<const uint8_t*>(segs[i].W);
        p.seg[i].Y = segs[i].Y;
        p.seg[i].out = segs[i].out;
        p.seg[i].w_last = (unsigned)((size_t)segs[i].out * row_bytes - 16);
        p.seg[i].tile0 = tiles;
        tiles += (segs[i].out + 64 * rt - 1) / (64 * rt);
    }
    p.row_wgs = tiles;
    const int trips = (p.steps + TRIP - 1) / TRIP;
    int nsplit = 1;
    while (nsplit * 2 * p.chunks <= GB_MAX_SPLIT && p.row_wgs * p.chunks * nsplit < want_wgs && trips / (nsplit * 2) >= 1) nsplit *= 2;
    const int tps = (trips + nsplit - 1) / nsplit;   // trips per split
    nsplit = (trips + tps - 1) / tps;                // no empty split
    p.nsplit = nsplit;
    p.steps_per_split = tps * TRIP;
    for (int i = 0; i < nseg; ++i) {                 // partial-sum areas, one after the other: nsplit x T x out_i floats each
        p.seg[i].part = part;
        part += (size_t)nsplit * T * segs[i].out;
    }
    const dim3 grid((unsigned)((p.row_wgs + 7) / 8 * 8 * p.chunks), nsplit);
    const size_t lds2 = gb_lds_bytes<DT, 2>(), lds1 = gb_lds_bytes<DT, 1>();
    if (rt == 2) hipLaunchKernelGGL((gemm_quant_bf16_kernel<DT, 2>), grid, dim3(256), lds2, st, p);
    else hipLaunchKernelGGL((gemm_quant_bf16_kernel<DT, 1>), grid, dim3(256), lds1, st, p);
    if (nsplit > 1) {
        ReduceArgs ra{};
        ra.nseg = nseg; ra.T = T; ra.nsplit = nsplit; ra.resid = resid;
        size_t biggest = 0;
        for (int i = 0; i < nseg; ++i) {
            ra.Y[i] = segs[i].Y; ra.part[i] = p.seg[i].part; ra.out[i] = segs[i].out;
            biggest = std::max(biggest, ((size_t)T * segs[i].out + 3) / 4);
        }
        hipLaunchKernelGGL(reduce_splits_kernel, dim3((unsigned)((biggest + 255) / 256), nseg), dim3(256), 0, st, ra);
    }
    return last_launch_status();
}

}  // namespace ntk

extern "C" {

size_t ntk_gemm_quant_workspace_bytes(int in_features, int out_features) {
    if (in_features <= 0 || out_features < 0) return 0;
    return (size_t)ntk::GB_MAX_CHUNKS * ntk::ws_chunk_bytes((in_features + 31) / 32 * 32) +
           (size_t)ntk::GB_MAX_SPLIT * ntk::GB_TOK * (size_t)out_features * sizeof(float) + 256;
}

static int gemm_ws_dispatch(const ntk::HostSeg* segs, int nseg, const float* X, int n_tokens, int in_features, int weight_dtype, const float* resid,
                            void* workspace, int reuse_x, hipStream_t st) {
    constexpr int PASS = ntk::GB_MAX_CHUNKS * ntk::GB_TOK;
    if (n_tokens > PASS) reuse_x = 0;   // the planes hold one pass (1024 tokens) at a time
    for (int t0 = 0; t0 < n_tokens; t0 += PASS) {   // up to 16 x 64 tokens per launch: the chunks share the weights in L2
        const int T = std::min(PASS, n_tokens - t0);
        ntk::HostSeg sg[ntk::GB_MAX_SEG];
        for (int i = 0; i < nseg; ++i) sg[i] = ntk::HostSeg{segs[i].Y + (size_t)t0 * segs[i].out, segs[i].W, segs[i].out};
        const float* x = X + (size_t)t0 * in_features;
        const float* rs = resid ? resid + (size_t)t0 * segs[0].out : nullptr;
        int rc;
        switch (weight_dtype) {
            case NTK_DT_Q8_0: rc = ntk::launch_gemm_bf16<NTK_DT_Q8_0>(sg, nseg, x, T, in_features, rs, workspace, reuse_x, st); break;
            case NTK_DT_Q4_K: rc = ntk::launch_gemm_bf16<NTK_DT_Q4_K>(sg, nseg, x, T, in_features, rs, workspace, reuse_x, st); break;
            case NTK_DT_Q5_K: rc = ntk::launch_gemm_bf16<NTK_DT_Q5_K>(sg, nseg, x, T, in_features, rs, workspace, reuse_x, st); break;
            default: rc = ntk::launch_gemm_bf16<NTK_DT_Q6_K>(sg, nseg, x, T, in_features, rs, workspace, reuse_x, st); break;
        }
        if (rc != NTK_OK) return rc;
    }
    return NTK_OK;
}

int ntk_gemm_quant_ws(float* Y, const void* W, const float* X, int n_tokens, int out_features, int in_features, int weight_dtype,
                      const float* resid, void* workspace, size_t workspace_bytes, int reuse_x, void* stream) {
    if (!Y || !W || !X || !workspace) return NTK_E_NULL;
    if (n_tokens < 0 || out_features < 0 || in_features <= 0) return NTK_E_SHAPE;
    if (workspace_bytes < ntk_gemm_quant_workspace_bytes(in_features, out_features) || (reinterpret_cast<uintptr_t>(workspace) & 15)) return NTK_E_SHAPE;
    if (weight_dtype != NTK_DT_Q8_0 && weight_dtype != NTK_DT_Q4_K && weight_dtype != NTK_DT_Q5_K && weight_dtype != NTK_DT_Q6_K) return NTK_E_DTYPE;
    if (n_tokens == 0 || out_features == 0) return NTK_OK;
    const ntk::HostSeg sg{Y, W, out_features};
    return gemm_ws_dispatch(&sg, 1, X, n_tokens, in_features, weight_dtype, resid, workspace, reuse_x, ntk::resolve_stream(stream));
}

// several matrices of one format sharing X (Q | K | V, gate | up) in ONE launch: segs[i] = {Y_i [n_tokens][rows_i], W_i, rows_i}
// (ntk_gemv_seg: y, W, rows, dtype -- the dtypes must agree).  workspace: ntk_gemm_quant_workspace_bytes(in, sum of rows).
int ntk_gemm_quant_ws_multi(const ntk_gemv_seg* segs, int nseg, const float* X, int n_tokens, int in_features, void* workspace,
                            size_t workspace_bytes, int reuse_x, void* stream) {
    if (!segs || !X || !workspace) return NTK_E_NULL;
    if (nseg < 1 || nseg > ntk::GB_MAX_SEG || n_tokens < 0 || in_features <= 0) return NTK_E_SHAPE;
    ntk::HostSeg sg[ntk::GB_MAX_SEG];
    long total = 0;
    for (int i = 0; i < nseg; ++i) {
        if (!segs[i].W || !segs[i].y) return NTK_E_NULL;
        if (segs[i].rows <= 0 || segs[i].dtype != segs[0].dtype) return NTK_E_SHAPE;
        sg[i] = ntk::HostSeg{segs[i].y, segs[i].W, segs[i].rows};
        total += segs[i].rows;
    }
    const int dt = segs[0].dtype;
    if (dt != NTK_DT_Q8_0 && dt != NTK_DT_Q4_K && dt != NTK_DT_Q5_K && dt != NTK_DT_Q6_K) return NTK_E_DTYPE;
    if (workspace_bytes < ntk_gemm_quant_workspace_bytes(in_features, (int)total) || (reinterpret_cast<uintptr_t>(workspace) & 15)) return NTK_E_SHAPE;
    if (n_tokens == 0) return NTK_OK;
    return gemm_ws_dispatch(sg, nseg, X, n_tokens, in_features, dt, nullptr, workspace, reuse_x, ntk::resolve_stream(stream));
}

}  // extern "C"
