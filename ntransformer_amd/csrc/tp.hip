// tp.hip -- the exchange step of tensor-parallel decoding (no reference counterpart: the reference is
// single-GPU).  One process per GPU, every rank holds a 1/W slice of each projection (rows of Wq/Wk/Wv/gate/up, columns of
// Wo/down: engine/model.cpp), so the outputs of Wo and of down are PARTIAL sums of the full vector: twice per layer the W
// partial vectors (H floats at decode, T x H for a prompt) must be added up on every rank.
//
// That all-reduce is hand-rolled over peer memory (xGMI loads) rather than an RCCL call: the message is 16-32 KB, i.e. pure
// latency, and a one-shot "everyone reads everyone" exchange is one kernel on the rank's own stream (hipGraph-capturable, no
// second communicator stream, no host involvement):
//   * every rank owns a communication buffer [flags | slot 0 | slot 1] that its peers map (hipIpc handles across processes);
//   * the producing GEMV / GEMM writes the rank's partial vector straight into the slot of the current call (call k uses slot
//     k & 1).  Visibility to the PEERS does not rest on the kernel boundary alone: the communication buffer is allocated
//     FINE-GRAINED (ntk_tp_comm_alloc: hipExtMallocWithFlags(hipDeviceMallocFinegrained)), i.e. the producer's stores are not
//     held dirty in any of the eight XCD L2s but written through to the memory side, where the peers' system-scope loads
//     (sc0 sc1, past every cache) find them; the kernel boundary between the producer and this kernel then only has to order the
//     flag store behind those stores.  (A release fence executed by the publishing workgroup would write back the L2 of ITS XCD
//     only -- the producer's stores came from all eight -- so it is not what this design relies on.)  Unmeasured across xGMI: the
//     boxes of this project have one GPU; NTK_TP_COARSE=1 falls back to an ordinary allocation for A/B on real hardware;
//   * tp_allreduce_add_kernel: publish flag[slot] = epoch * 1024 + k + 1 (write-through, system scope), wait until every
//     peer's flag[slot] has reached that value (bounded spin: a lost peer sets an error word instead of hanging the GPU), then
//     hidden += sum over ranks IN RANK ORDER of their slots, peers read past the caches (sc0 sc1).  Every rank adds the same
//     numbers in the same order: hidden stays bit-identical on all ranks, so sampling needs no further exchange.
//   * slot reuse: a rank writes slot k & 1 for call k + 2 only after it finished call k + 1, which it could only finish once
//     every peer had published call k + 1, i.e. had finished reading call k.  Two slots suffice.
//   * epoch: a device-side counter advanced once per forward / token (ntk_tp_advance_epoch), so a captured token graph can be
//     replayed: call indices are constants of the graph, the epoch is data.
#include "common.hip.h"
#include <algorithm>
#include <cstdlib>

namespace ntk {

constexpr int TP_MAX_WORLD = 8;
constexpr size_t TP_HDR = 256;   // bytes: flag of slot 0 @0, of slot 1 @64, error word @128, epoch @192

struct TpArgs {
    uint8_t* base[TP_MAX_WORLD];   // every rank's communication buffer (own one included), as mapped on THIS device
    float* hidden;
    size_t max_floats;             // capacity of one slot
    int rank, world, n;
    unsigned call_index;
};

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned tp_load_sys(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ f32x4 tp_load16_sys(const float* p) {   // 16 bytes past every cache (the peer rewrites the slot call after call)
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

__global__ __launch_bounds__(256) void tp_allreduce_add_kernel(const TpArgs a) {
    const unsigned slot = a.call_index & 1u;
    uint8_t* mine = a.base[a.rank];
    const unsigned expected = *reinterpret_cast<const unsigned*>(mine + 192) * 1024u + a.call_index + 1u;
    if (threadIdx.x == 0) {
        // this rank's partial vector was written by the previous kernel on this stream: it is in memory.  Publish, then wait.
        if (blockIdx.x == 0) __hip_atomic_store(reinterpret_cast<unsigned*>(mine + 64 * slot), expected, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        int ok = 1;
        for (int r = 0; r < a.world; ++r) {
            if (r == a.rank) continue;
            const unsigned* f = reinterpret_cast<const unsigned*>(a.base[r] + 64 * slot);
            int budget = 1 << 22;   // ~1 s: a peer that never arrives must not hang the device
            while ((int)(tp_load_sys(f) - expected) < 0 && --budget) __builtin_amdgcn_s_sleep(8);
            if (!budget) ok = 0;
        }
        if (!ok) __hip_atomic_store(reinterpret_cast<unsigned*>(mine + 128), expected, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __syncthreads();   // the workgroup proceeds once thread 0 has seen every peer (or given up: the error word says so)
    const size_t data_off = TP_HDR + (size_t)slot * a.max_floats * sizeof(float);
    for (size_t idx = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; idx < (size_t)a.n; idx += (size_t)gridDim.x * 256 * 4) {
        f32x4 v = *reinterpret_cast<const f32x4*>(a.hidden + idx);
        for (int r = 0; r < a.world; ++r) {   // rank order on every rank: bit-identical sums
            const float* src = reinterpret_cast<const float*>(a.base[r] + data_off) + idx;
            v += (r == a.rank) ? *reinterpret_cast<const f32x4*>(src) : tp_load16_sys(src);
        }
        *reinterpret_cast<f32x4*>(a.hidden + idx) = v;
    }
}

__global__ void tp_advance_epoch_kernel(unsigned* epoch) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *epoch += 1u;
}

}  // namespace ntk

extern "C" {

using namespace ntk;

size_t ntk_tp_comm_bytes(size_t max_floats) { return TP_HDR + 2 * ((max_floats + 3) / 4 * 4) * sizeof(float) + 256; }

// a communication buffer: fine-grained device memory (see the header comment), ordinary device memory if the runtime refuses or
// NTK_TP_COARSE=1; freed with nt_hip_free
void* ntk_tp_comm_alloc(size_t bytes) {
    void* p = nullptr;
    if (NTK_TUNE_ENV_INT("NTK_TP_COARSE", 0) == 0) {   // (tuning builds: ordinary device memory, for an A/B on real multi-GPU hardware)
        if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained) == hipSuccess && p) return p;
        (void)hipGetLastError();
        p = nullptr;
    }
    if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}

// zero the flags, the error word and the epoch of a freshly allocated communication buffer (before the peers map it)
int ntk_tp_comm_reset(void* comm, void* stream) {
    if (!comm) return NTK_E_NULL;
    NTK_HIP_TRY(hipMemsetAsync(comm, 0, TP_HDR, resolve_stream(stream)));
    return NTK_OK;
}

// where the producer of call `call_index` writes this rank's partial vector
float* ntk_tp_slot(void* comm, size_t max_floats, unsigned call_index) {
    if (!comm) return nullptr;
    return reinterpret_cast<float*>(static_cast<uint8_t*>(comm) + TP_HDR + (size_t)(call_index & 1u) * ((max_floats + 3) / 4 * 4) * sizeof(float));
}

// hidden[0..n) += sum over ranks of their slot of this call.  peers: the world's communication buffers as mapped here
// (peers[rank] = this rank's own).  n % 4 == 0, n <= max_floats, hidden 16-byte aligned, call_index < 1023.
int ntk_tp_allreduce_add(float* hidden, void* const* peers, int rank, int world, size_t max_floats, unsigned call_index, int n, void* stream) {
    if (!hidden || !peers) return NTK_E_NULL;
    if (world < 1 || world > TP_MAX_WORLD || rank < 0 || rank >= world || n < 0 || (n & 3) || (size_t)n > max_floats || call_index >= 1023u)
        return NTK_E_SHAPE;
    if (reinterpret_cast<uintptr_t>(hidden) & 15) return NTK_E_ALIGN;
    if (n == 0) return NTK_OK;
    TpArgs a{};
    for (int r = 0; r < world; ++r) {
        if (!peers[r]) return NTK_E_NULL;
        a.base[r] = static_cast<uint8_t*>(peers[r]);
    }
    a.hidden = hidden; a.max_floats = (max_floats + 3) / 4 * 4; a.rank = rank; a.world = world; a.n = n; a.call_index = call_index;
    const int grid = std::max(1, std::min(64, (n / 4 + 255) / 256));
    hipLaunchKernelGGL(tp_allreduce_add_kernel, dim3(grid), dim3(256), 0, resolve_stream(stream), a);
    return last_launch_status();
}

int ntk_tp_advance_epoch(void* comm, void* stream) {
    if (!comm) return NTK_E_NULL;
    hipLaunchKernelGGL(tp_advance_epoch_kernel, dim3(1), dim3(64), 0, resolve_stream(stream), reinterpret_cast<unsigned*>(static_cast<uint8_t*>(comm) + 192));
    return last_launch_status();
}

// after a synchronise: 0, or the (epoch * 1024 + call + 1) of the last call that gave up waiting for a peer
unsigned ntk_tp_error(void* comm) {
    unsigned v = 0;
    if (!comm || hipMemcpy(&v, static_cast<uint8_t*>(comm) + 128, 4, hipMemcpyDeviceToHost) != hipSuccess) return ~0u;
    return v;
}

// hipIpc plumbing for the one-process-per-GPU deployment: 64-byte handles travel over whatever the launcher has (a file,
// torch.distributed's gloo store, MPI)
int ntk_ipc_export(void* devptr, void* handle64) {
    if (!devptr || !handle64) return NTK_E_NULL;
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "handle size");
    NTK_HIP_TRY(hipIpcGetMemHandle(static_cast<hipIpcMemHandle_t*>(handle64), devptr));
    return NTK_OK;
}
int ntk_ipc_open(const void* handle64, void** devptr) {
    if (!handle64 || !devptr) return NTK_E_NULL;
    hipIpcMemHandle_t h;
    __builtin_memcpy(&h, handle64, 64);
    NTK_HIP_TRY(hipIpcOpenMemHandle(devptr, h, hipIpcMemLazyEnablePeerAccess));
    return NTK_OK;
}
int ntk_ipc_close(void* devptr) {
    if (!devptr) return NTK_E_NULL;
    NTK_HIP_TRY(hipIpcCloseMemHandle(devptr));
    return NTK_OK;
}

}  // extern "C"
