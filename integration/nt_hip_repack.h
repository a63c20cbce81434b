// integration/nt_hip_repack.h -- REFERENCE-SIDE BINDING, optional part: the matrix-core decode GEMV for resident K-quant matrices.
//
// The decode GEMV of libntransformer_hip.so for Q4_K / Q5_K / Q6_K on the int8 matrix cores (csrc/gemv_rp.hip, ntk_gemv_rp) reads a repacked
// copy of the tensor that its caller owns (include/ntk.h: ntk_rp_bytes / ntk_rp_pack).  For the reference that caller is this binding:
//
//     // src/model/attention.cpp, Attention::set_weights (:80-95), after the assignments -- and the same five lines in FFN::init (ffn.cpp:7-29):
//     #include "nt_hip_repack.h"
//     for (const Tensor* w : {&wq_, &wk_, &wv_, &wo_})
//         nt::cuda::hip_register_resident_weight(w->data(), (int)w->shape()[0], (int)w->shape()[1], w->dtype(), stream);
//
// After that nt::cuda::launch_gemv(y, W, x, out, in, dtype, stream) on a registered pointer runs on the repacked copy; everything else about the
// call is unchanged (same arguments, same results within the GEMV tolerance: tests/test_gemv_rp.py).  Only for weights that stay where they are
// (the resident path, transformer.cpp:604-669): the streaming / tiered modes rewrite their buffers in place and must not register them.
// Unmodified binaries: NT_HIP_AUTO_REPACK=1 registers a K-quant pointer at its first launch_gemv (same caveat); NT_HIP_REPACK_STATS=1 prints at
// exit how many launches ran on which kernel.
#pragma once
#include "core/types.h"   // nt::DType (the reference's own header)

namespace nt {
namespace cuda {

// Packs the matrix (once, on the device, stream-ordered) and routes later launch_gemv calls on W to the matrix-core GEMV.  Returns false and
// leaves the pointer on the raw-GGUF path when the format / shape has no repacked form (anything but Q4_K / Q5_K / Q6_K with in % 256 == 0) or
// device memory is short.  Costs rp_bytes ~ 1.00-1.03 x the tensor's GGUF bytes of HBM per registered matrix.
bool hip_register_resident_weight(const void* W, int out_features, int in_features, DType dtype, void* stream);
// Frees every repacked copy (synchronises the device first).
void hip_release_resident_weights();

}  // namespace cuda
}  // namespace nt
