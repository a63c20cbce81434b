// integration/nt_cuda_launchers.cpp -- REFERENCE-SIDE BINDING (what a maintainer of xaskasdf/ntransformer adds).
//
// Defines the 17 `nt::cuda::launch_*` functions declared in the reference's src/cuda/kernels.h:14-71 with their
// exact signatures, each forwarding to the C ABI of libntransformer_hip.so (include/ntk.h).  Compiled INSTEAD of
// src/cuda/{gemm,attention,rmsnorm,rotary,elementwise,softmax}.cu; every caller in src/model/*.cpp and
// tests/test_gemm.cpp stays untouched.  Error behaviour is the reference's: launches are fire-and-forget, an
// unsupported GEMV dtype only prints (src/cuda/gemm.cu:801-803, :866-868).
#include "cuda/kernels.h"   // the reference's own header (include path: <reference>/src)
#include "ntk.h"
#include <cstdio>

namespace nt {
namespace cuda {

static void report(const char* what, int st) {
    if (st != NTK_OK) fprintf(stderr, "%s: %s\n", what, ntk_status_string(st));
}

void launch_rmsnorm(float* o, const float* in, const float* w, int batch, int hidden, float eps, void* s) {
    report("launch_rmsnorm", ntk_rmsnorm(o, in, w, batch, hidden, eps, s));
}
void launch_rmsnorm_f16(void* o, const float* in, const float* w, int batch, int hidden, float eps, void* s) {
    report("launch_rmsnorm_f16", ntk_rmsnorm_f16(o, in, w, batch, hidden, eps, s));
}
void launch_rope(float* q, float* k, const int* pos, int batch, int seq_len, int nh, int nkv, int hd, float theta, float fscale,
                 bool interleaved, void* s) {
    report("launch_rope", ntk_rope(q, k, pos, batch, seq_len, nh, nkv, hd, theta, fscale, interleaved ? 1 : 0, s));
}
void launch_softmax(float* o, const float* in, int rows, int cols, void* s) { report("launch_softmax", ntk_softmax(o, in, rows, cols, s)); }
void launch_masked_softmax(float* o, const float* in, const bool* mask, int rows, int cols, void* s) {
    report("launch_masked_softmax", ntk_masked_softmax(o, in, reinterpret_cast<const uint8_t*>(mask), rows, cols, s));
}
void launch_gemv(float* y, const void* W, const float* x, int out_f, int in_f, DType dt, void* s) {
    const int st = ntk_gemv(y, W, x, out_f, in_f, (int)dt, s);
    if (st == NTK_E_DTYPE) fprintf(stderr, "Unsupported dtype for GEMV: %s\n", dtype_name(dt));
    else report("launch_gemv", st);
}
void launch_gemv_add(float* y, const void* W, const float* x, int out_f, int in_f, DType dt, void* s) {
    const int st = ntk_gemv_add(y, W, x, out_f, in_f, (int)dt, s);
    if (st == NTK_E_DTYPE) fprintf(stderr, "launch_gemv_add: only F16 supported (got %s)\n", dtype_name(dt));
    else report("launch_gemv_add", st);
}
void launch_gemm_f32(float* C, const float* A, const float* B, int M, int N, int K, void* s) { report("launch_gemm_f32", ntk_gemm_f32(C, A, B, M, N, K, s)); }
void launch_silu_mul(float* o, const float* g, const float* u, int n, void* s) { report("launch_silu_mul", ntk_silu_mul(o, g, u, n, s)); }
void launch_add_bias(float* y, const float* b, int n, void* s) { report("launch_add_bias", ntk_add_bias(y, b, n, s)); }
void launch_attention_decode(float* o, const float* q, const void* kc, const void* vc, int seq_len, int nh, int nkv, int hd, int max_seq,
                             float scale, void* s) {
    report("launch_attention_decode", ntk_attention_decode(o, q, kc, vc, seq_len, nh, nkv, hd, max_seq, scale, s));
}
void launch_attention_prefill(float* o, const float* Q, const void* kc, const void* vc, int seq_len, int start_pos, int nh, int nkv, int hd,
                              int max_seq, float scale, void* s) {
    report("launch_attention_prefill", ntk_attention_prefill(o, Q, kc, vc, seq_len, start_pos, nh, nkv, hd, max_seq, scale, s));
}
void launch_copy_to_kv_cache(void* kc, void* vc, const float* k, const float* v, int seq_len, int nkv, int hd, int start_pos, int max_seq, void* s) {
    report("launch_copy_to_kv_cache", ntk_copy_to_kv_cache(kc, vc, k, v, seq_len, nkv, hd, start_pos, max_seq, s));
}
void launch_add(float* o, const float* a, const float* b, int n, void* s) { report("launch_add", ntk_add(o, a, b, n, s)); }
void launch_add_inplace(float* a, const float* b, int n, void* s) { report("launch_add_inplace", ntk_add_inplace(a, b, n, s)); }
void launch_copy(float* d, const float* src, int n, void* s) { report("launch_copy", ntk_copy(d, src, n, s)); }
void launch_cosine_similarity(float* r, const float* a, const float* b, int n, void* s) { report("launch_cosine_similarity", ntk_cosine_similarity(r, a, b, n, s)); }

}  // namespace cuda
}  // namespace nt
