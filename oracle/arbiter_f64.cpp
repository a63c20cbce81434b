// oracle/arbiter_f64.cpp
// =====================================================================================
// TEST INFRASTRUCTURE ONLY (same rule as ref_launchers_cpu.cpp: nothing under ntransformer_amd/ may
// include, link or call this file; tests/ use it as a checker, never as the thing measured or shipped).
//
// A FLOAT64 ARBITER of the resident decode path -- the mathematical function the reference's F32 CUDA
// kernels approximate (reference src/model/transformer.cpp:604-669 over src/cuda/{gemm,rmsnorm,rotary,
// attention,elementwise}.cu), evaluated with every accumulation in double.  It is NOT a restatement of
// the reference's summation order (that is ref_launchers_cpu.cpp); it is the third party that says which
// of two F32 implementations is closer to the truth, and how far any F32 implementation can be.
//
// What stays exactly as the reference DEFINES it (data, not arithmetic):
//   * the weights: GGUF block decoding, reference src/core/types.h:96-137 and gemm.cu:52-75 (Q4_0),
//     :129-141 (Q8_0), :190-244 (Q4_K), :297-354 (Q5_K), :421-459 (Q6_K) -- here each weight's real value
//     (d * q, d * sc * n - dmin * m, d * sc * (q - 32)) formed in double;
//   * the rounding of K and V to IEEE half on their way into the cache (attention.cu:338-339,
//     __float2half: round to nearest even) -- the one discontinuity on the path; here the double value is
//     rounded ONCE to half (no double rounding through float), and arb_kv_store reports how far each
//     value sat from its rounding boundary, so that a test can tell a legitimate flip of an F32
//     implementation (pre-rounding value within F32 error of the boundary) from a wrong value;
//   * the RoPE angle: freq = 1.0f / powf(theta, 2p/hd) and angle = pos * freq * scale are F32 values in the
//     reference (rotary.cu:46-49); the arbiter takes those F32 numbers as the definition of the angle and
//     evaluates cos / sin of them in double;
//   * the F32 constants scale = 1/sqrtf(hd) (attention.cpp:21) and eps.
// Everything else (dot products, sum of squares, softmax, P.V, SiLU, residual adds) is double.
//
// PARITY PIN: tests/test_arbiter_cpu.py checks the arbiter against the F32 restatement (which carries the
// reference's own known-answer vectors) on every operator and on the golden models: agreement to F32
// rounding level, and exact agreement of the block decoders.
// =====================================================================================
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

enum { DT_F32 = 0, DT_F16 = 1, DT_Q8_0 = 2, DT_Q4_0 = 3, DT_Q4_K = 4, DT_Q6_K = 5, DT_Q5_K = 6 };   // reference types.h:24-35

inline double h2d(uint16_t h) {   // exact
    const int sign = h >> 15, exp = (h >> 10) & 0x1F, man = h & 0x3FF;
    double v;
    if (exp == 0) v = std::ldexp((double)man, -24);
    else if (exp == 31) v = man ? NAN : INFINITY;
    else v = std::ldexp((double)(man | 0x400), exp - 25);
    return sign ? -v : v;
}
inline uint16_t rd16(const uint8_t* p) { uint16_t v; std::memcpy(&v, p, 2); return v; }

// double -> IEEE half, round to nearest even, ONE rounding.  *mid_dist (optional) = |x - nearest rounding boundary| (absolute):
// an implementation whose pre-rounding value differs from x by less than that cannot round differently.
uint16_t d2h(double x, double* mid_dist) {
    const uint16_t sign = std::signbit(x) ? 0x8000u : 0u;
    double ax = std::fabs(x);
    if (mid_dist) *mid_dist = INFINITY;
    if (std::isnan(ax)) return (uint16_t)(sign | 0x7E00u);
    if (ax >= 65520.0) { if (mid_dist) *mid_dist = ax - 65520.0; return (uint16_t)(sign | 0x7C00u); }
    if (ax == 0.0) { if (mid_dist) *mid_dist = std::ldexp(1.0, -25); return sign; }
    int e;
    (void)std::frexp(ax, &e);           // ax = m * 2^e, 0.5 <= m < 1  ->  ax in [2^(e-1), 2^e)
    int E = e - 1;
    if (E < -14) E = -14;               // subnormal halves share the ulp 2^-24
    const double ulp = std::ldexp(1.0, E - 10);
    const double t = ax / ulp;          // exact (power of two)
    double n = std::nearbyint(t);       // default rounding mode: to nearest even
    if (mid_dist) { const double fl = std::floor(t); *mid_dist = std::fabs((t - fl) - 0.5) * ulp; }
    uint32_t bits;
    if (E == -14 && n < 1024.0) bits = (uint32_t)n;                       // subnormal (or zero)
    else {
        if (n >= 2048.0) { n = 1024.0; ++E; }
        bits = ((uint32_t)(E + 15) << 10) | ((uint32_t)n - 1024u);
    }
    return (uint16_t)(sign | bits);
}

inline void kq_scale_min(const uint8_t* sc, int j, int& s, int& m) {   // gemm.cu:206-222
    if (j < 4) { s = sc[j] & 0x3F; m = sc[j + 4] & 0x3F; }
    else { s = (sc[j + 4] & 0x0F) | ((sc[j - 4] >> 6) << 4); m = (sc[j + 4] >> 4) | ((sc[j] >> 6) << 4); }
}

size_t row_bytes(int dt, int in) {
    switch (dt) {
        case DT_F32: return (size_t)in * 4;
        case DT_F16: return (size_t)in * 2;
        case DT_Q8_0: return (size_t)(in / 32) * 34;
        case DT_Q4_0: return (size_t)(in / 32) * 18;
        case DT_Q4_K: return (size_t)(in / 256) * 144;
        case DT_Q5_K: return (size_t)(in / 256) * 176;
        case DT_Q6_K: return (size_t)(in / 256) * 210;
        default: return 0;
    }
}

// the real values of one row's weights
void dequant_row(double* w, const uint8_t* row, int in, int dt) {
    switch (dt) {
        case DT_F32: { const float* r = (const float*)row; for (int i = 0; i < in; ++i) w[i] = r[i]; break; }
        case DT_F16: { const uint16_t* r = (const uint16_t*)row; for (int i = 0; i < in; ++i) w[i] = h2d(r[i]); break; }
        case DT_Q8_0:
            for (int b = 0; b < in / 32; ++b) {
                const uint8_t* blk = row + (size_t)b * 34;
                const double d = h2d(rd16(blk));
                const int8_t* q = (const int8_t*)(blk + 2);
                for (int j = 0; j < 32; ++j) w[b * 32 + j] = d * q[j];
            }
            break;
        case DT_Q4_0:
            for (int b = 0; b < in / 32; ++b) {
                const uint8_t* blk = row + (size_t)b * 18;
                const double d = h2d(rd16(blk));
                const uint8_t* q = blk + 2;
                for (int j = 0; j < 16; ++j) {
                    w[b * 32 + j] = d * ((q[j] & 0x0F) - 8);
                    w[b * 32 + j + 16] = d * ((q[j] >> 4) - 8);
                }
            }
            break;
        case DT_Q4_K:
            for (int b = 0; b < in / 256; ++b) {
                const uint8_t* blk = row + (size_t)b * 144;
                const double d = h2d(rd16(blk)), dmin = h2d(rd16(blk + 2));
                const uint8_t* q = blk + 16;
                for (int c = 0; c < 4; ++c) {
                    int sl, ml, sh, mh;
                    kq_scale_min(blk + 4, 2 * c, sl, ml);
                    kq_scale_min(blk + 4, 2 * c + 1, sh, mh);
                    const double d1 = d * sl, m1 = dmin * ml, d2 = d * sh, m2 = dmin * mh;
                    double* y = w + b * 256 + c * 64;
                    for (int l = 0; l < 32; ++l) {
                        y[l] = d1 * (q[l] & 0x0F) - m1;
                        y[l + 32] = d2 * (q[l] >> 4) - m2;
                    }
                    q += 32;
                }
            }
            break;
        case DT_Q5_K:
            for (int b = 0; b < in / 256; ++b) {
                const uint8_t* blk = row + (size_t)b * 176;
                const double d = h2d(rd16(blk)), dmin = h2d(rd16(blk + 2));
                const uint8_t* qh = blk + 16;
                const uint8_t* ql = blk + 48;
                int u1 = 1, u2 = 2;
                for (int c = 0; c < 4; ++c) {
                    int sl, ml, sh, mh;
                    kq_scale_min(blk + 4, 2 * c, sl, ml);
                    kq_scale_min(blk + 4, 2 * c + 1, sh, mh);
                    const double d1 = d * sl, m1 = dmin * ml, d2 = d * sh, m2 = dmin * mh;
                    double* y = w + b * 256 + c * 64;
                    for (int l = 0; l < 32; ++l) {
                        y[l] = d1 * ((ql[l] & 0x0F) + ((qh[l] & u1) ? 16 : 0)) - m1;
                        y[l + 32] = d2 * ((ql[l] >> 4) + ((qh[l] & u2) ? 16 : 0)) - m2;
                    }
                    ql += 32; u1 <<= 2; u2 <<= 2;
                }
            }
            break;
        case DT_Q6_K:
            for (int b = 0; b < in / 256; ++b) {
                const uint8_t* blk = row + (size_t)b * 210;
                const double d = h2d(rd16(blk + 208));
                const uint8_t* ql = blk; const uint8_t* qh = blk + 128; const int8_t* sc = (const int8_t*)(blk + 192);
                double* y = w + b * 256;
                for (int hf = 0; hf < 2; ++hf) {
                    for (int l = 0; l < 32; ++l) {
                        const int is = l / 16;
                        const int q1 = (int)((ql[l] & 0xF) | (((qh[l] >> 0) & 3) << 4)) - 32;
                        const int q2 = (int)((ql[l + 32] & 0xF) | (((qh[l] >> 2) & 3) << 4)) - 32;
                        const int q3 = (int)((ql[l] >> 4) | (((qh[l] >> 4) & 3) << 4)) - 32;
                        const int q4 = (int)((ql[l + 32] >> 4) | (((qh[l] >> 6) & 3) << 4)) - 32;
                        y[l] = d * sc[is + 0] * q1;
                        y[l + 32] = d * sc[is + 2] * q2;
                        y[l + 64] = d * sc[is + 4] * q3;
                        y[l + 96] = d * sc[is + 6] * q4;
                    }
                    y += 128; ql += 64; qh += 32; sc += 8;
                }
            }
            break;
        default: for (int i = 0; i < in; ++i) w[i] = 0.0;
    }
}

}  // namespace

extern "C" {

int arb_abi_version(void) { return 1; }

void arb_set_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n > 0 ? n : 1);
#else
    (void)n;
#endif
}

// the real weights of `rows` rows (a test hook for the decoders)
void arb_dequant(double* out, const void* W, int rows, int in, int dtype) {
    const size_t rb = row_bytes(dtype, in);
    for (int r = 0; r < rows; ++r) dequant_row(out + (size_t)r * in, (const uint8_t*)W + (size_t)r * rb, in, dtype);
}

// Y[t][r] = sum_i W[r][i] * X[t][i]   (launch_gemv for every token; W decoded once per row)
int arb_gemm(double* Y, const void* W, const double* X, int T, int out_f, int in_f, int dtype) {
    const size_t rb = row_bytes(dtype, in_f);
    if (!rb) return -1;
    const uint8_t* base = (const uint8_t*)W;
#pragma omp parallel
    {
        std::vector<double> w((size_t)in_f);
#pragma omp for schedule(static)
        for (int r = 0; r < out_f; ++r) {
            dequant_row(w.data(), base + (size_t)r * rb, in_f, dtype);
            for (int t = 0; t < T; ++t) {
                const double* x = X + (size_t)t * in_f;
                double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
                int i = 0;
                for (; i + 4 <= in_f; i += 4) {
                    a0 += w[i] * x[i]; a1 += w[i + 1] * x[i + 1]; a2 += w[i + 2] * x[i + 2]; a3 += w[i + 3] * x[i + 3];
                }
                for (; i < in_f; ++i) a0 += w[i] * x[i];
                Y[(size_t)t * out_f + r] = (a0 + a1) + (a2 + a3);
            }
        }
    }
    return 0;
}

// rmsnorm.cu:16-70: y = x * (1 / sqrt(mean(x^2) + eps)) * w, eps and w F32 data
void arb_rmsnorm(double* y, const double* x, const float* w, int batch, int hidden, float eps) {
    for (int b = 0; b < batch; ++b) {
        const double* xr = x + (size_t)b * hidden;
        double* yr = y + (size_t)b * hidden;
        double ss = 0.0;
        for (int i = 0; i < hidden; ++i) ss += xr[i] * xr[i];
        const double inv = 1.0 / std::sqrt(ss / hidden + (double)eps);
        for (int i = 0; i < hidden; ++i) yr[i] = xr[i] * inv * (double)w[i];
    }
}

// rotary.cu:16-62 (non-interleaved pairs (i, i + hd/2)).  The angle is the reference's F32 number.
void arb_rope(double* q, double* k, const int* positions, int T, int nh, int nkv, int hd, float theta, float fscale) {
    const int half = hd / 2;
    for (int is_key = 0; is_key < 2; ++is_key) {
        double* data = is_key ? k : q;
        const int n_h = is_key ? nkv : nh;
        for (int t = 0; t < T; ++t)
            for (int h = 0; h < n_h; ++h) {
                double* v = data + ((size_t)t * n_h + h) * hd;
                for (int p = 0; p < half; ++p) {
                    const float freq = 1.0f / powf(theta, (2.0f * p) / hd);
                    const float angle = positions[t] * freq * fscale;
                    const double c = std::cos((double)angle), s = std::sin((double)angle);
                    const double x0 = v[p], x1 = v[p + half];
                    v[p] = x0 * c - x1 * s;
                    v[p + half] = x1 * c + x0 * s;
                }
            }
    }
}

// attention.cu:316-342: cache[start + t][e] = half(k[t][e]); mid_k / mid_v (may be NULL): distance of every value to its rounding
// boundary, same shape as k / v
void arb_kv_store(uint16_t* kc, uint16_t* vc, const double* k, const double* v, int T, int per, int start, int max_seq,
                  double* mid_k, double* mid_v) {
    for (int t = 0; t < T; ++t) {
        const int cp = start + t;
        if (cp >= max_seq) continue;
        for (int e = 0; e < per; ++e) {
            kc[(size_t)cp * per + e] = d2h(k[(size_t)t * per + e], mid_k ? mid_k + (size_t)t * per + e : nullptr);
            vc[(size_t)cp * per + e] = d2h(v[(size_t)t * per + e], mid_v ? mid_v + (size_t)t * per + e : nullptr);
        }
    }
}

// attention.cu:108-202 / :216-311: causal GQA attention of T queries at positions start.. over the half cache
void arb_attention(double* out, const double* Q, const uint16_t* kc, const uint16_t* vc, int T, int start, int nh, int nkv, int hd,
                   float scale) {
    const int group = nh / nkv;
    const size_t stride = (size_t)nkv * hd;
#pragma omp parallel for collapse(2) schedule(dynamic)
    for (int t = 0; t < T; ++t)
        for (int h = 0; h < nh; ++h) {
            const int n_keys = start + t + 1, kvh = h / group;
            const double* q = Q + ((size_t)t * nh + h) * hd;
            double* o = out + ((size_t)t * nh + h) * hd;
            std::vector<double> s((size_t)n_keys);
            double mx = -INFINITY;
            for (int p = 0; p < n_keys; ++p) {
                const uint16_t* kr = kc + (size_t)p * stride + (size_t)kvh * hd;
                double a = 0.0;
                for (int d = 0; d < hd; ++d) a += q[d] * h2d(kr[d]);
                s[p] = a * (double)scale;
                mx = std::fmax(mx, s[p]);
            }
            double tot = 0.0;
            for (int p = 0; p < n_keys; ++p) { s[p] = std::exp(s[p] - mx); tot += s[p]; }
            for (int d = 0; d < hd; ++d) o[d] = 0.0;
            for (int p = 0; p < n_keys; ++p) {
                const uint16_t* vr = vc + (size_t)p * stride + (size_t)kvh * hd;
                const double w = s[p] / tot;
                for (int d = 0; d < hd; ++d) o[d] += w * h2d(vr[d]);
            }
        }
}

// gemm.cu:719-724
void arb_silu_mul(double* out, const double* g, const double* u, long n) {
    for (long i = 0; i < n; ++i) out[i] = g[i] / (1.0 + std::exp(-g[i])) * u[i];
}

double arb_h2d(uint16_t h) { return h2d(h); }
uint16_t arb_d2h(double x) { return d2h(x, nullptr); }

}  // extern "C"
