// oracle/shim/cuda_runtime.h -- TEST INFRASTRUCTURE ONLY (never shipped, never on the product path).
//
// Host stand-in for the CUDA runtime so that the reference's *host* translation units
// (/root/reference/src/{core,model,inference,memory,utils}/*, main.cpp) compile and link with
// plain g++ into a CPU-only binary (oracle/_ref/*).  "Device" memory is host memory, streams and
// events are inert handles, every copy is a memcpy.  Written from the list of runtime symbols the
// reference uses (grep over /root/reference/src); it is not a CUDA
// compatibility layer for the product -- the product (ntransformer_amd/csrc) talks to HIP directly.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <chrono>

typedef int cudaError_t;
enum : int { cudaSuccess = 0, cudaErrorMemoryAllocation = 2, cudaErrorInvalidValue = 1 };

struct nt_shim_stream_ { int id; };
struct nt_shim_event_ { double t_ms; };
typedef nt_shim_stream_* cudaStream_t;
typedef nt_shim_event_* cudaEvent_t;

enum cudaMemcpyKind {
    cudaMemcpyHostToHost = 0,
    cudaMemcpyHostToDevice = 1,
    cudaMemcpyDeviceToHost = 2,
    cudaMemcpyDeviceToDevice = 3,
    cudaMemcpyDefault = 4
};

enum : unsigned {
    cudaStreamNonBlocking = 1u,
    cudaEventDisableTiming = 2u,
    cudaHostRegisterReadOnly = 8u
};

enum cudaFuncAttribute { cudaFuncAttributeMaxDynamicSharedMemorySize = 8 };

struct cudaDeviceProp {
    char name[256];
    size_t totalGlobalMem;
    int multiProcessorCount;
    int major, minor;
    int maxThreadsPerBlock;
    int warpSize;
    size_t sharedMemPerBlock;
};

inline const char* cudaGetErrorString(cudaError_t e) {
    return e == cudaSuccess ? "no error" : (e == cudaErrorMemoryAllocation ? "out of memory" : "error");
}

// ---- memory: device == host -------------------------------------------------------------
inline cudaError_t cudaMalloc(void** p, size_t n) {
    *p = std::malloc(n ? n : 1);
    return *p ? cudaSuccess : cudaErrorMemoryAllocation;
}
template <typename T> inline cudaError_t cudaMalloc(T** p, size_t n) {
    return cudaMalloc(reinterpret_cast<void**>(p), n);
}
inline cudaError_t cudaFree(void* p) { std::free(p); return cudaSuccess; }
inline cudaError_t cudaMallocHost(void** p, size_t n) { return cudaMalloc(p, n); }
template <typename T> inline cudaError_t cudaMallocHost(T** p, size_t n) {
    return cudaMalloc(reinterpret_cast<void**>(p), n);
}
inline cudaError_t cudaFreeHost(void* p) { std::free(p); return cudaSuccess; }
inline cudaError_t cudaHostRegister(void*, size_t, unsigned) { return cudaSuccess; }
inline cudaError_t cudaHostUnregister(void*) { return cudaSuccess; }
inline cudaError_t cudaMemcpy(void* d, const void* s, size_t n, cudaMemcpyKind) {
    if (n) std::memmove(d, s, n);
    return cudaSuccess;
}
inline cudaError_t cudaMemcpyAsync(void* d, const void* s, size_t n, cudaMemcpyKind k, cudaStream_t = nullptr) {
    return cudaMemcpy(d, s, n, k);
}
inline cudaError_t cudaMemset(void* p, int v, size_t n) { if (n) std::memset(p, v, n); return cudaSuccess; }
inline cudaError_t cudaMemGetInfo(size_t* free_b, size_t* total_b) {
    // Pretend to be a large accelerator so the reference never picks a streaming tier.
    *total_b = (size_t)288 << 30;
    *free_b = (size_t)256 << 30;
    return cudaSuccess;
}

// ---- device -------------------------------------------------------------------------------
inline cudaError_t cudaGetDeviceCount(int* n) { *n = 1; return cudaSuccess; }
inline cudaError_t cudaSetDevice(int) { return cudaSuccess; }
inline cudaError_t cudaDeviceSynchronize() { return cudaSuccess; }
inline cudaError_t cudaGetDeviceProperties(cudaDeviceProp* p, int) {
    std::memset(p, 0, sizeof(*p));
    std::strncpy(p->name, "host-cpu (oracle shim, no GPU)", sizeof(p->name) - 1);
    p->totalGlobalMem = (size_t)288 << 30;
    p->multiProcessorCount = 1;
    p->major = 0; p->minor = 0;
    p->maxThreadsPerBlock = 1024;
    p->warpSize = 32;
    p->sharedMemPerBlock = 64 * 1024;
    return cudaSuccess;
}
template <typename F> inline cudaError_t cudaFuncSetAttribute(F, cudaFuncAttribute, int) { return cudaSuccess; }

// ---- streams / events: inert ----------------------------------------------------------------
inline cudaError_t cudaStreamCreateWithFlags(cudaStream_t* s, unsigned) {
    *s = new nt_shim_stream_{0};
    return cudaSuccess;
}
inline cudaError_t cudaStreamDestroy(cudaStream_t s) { delete s; return cudaSuccess; }
inline cudaError_t cudaStreamSynchronize(cudaStream_t) { return cudaSuccess; }
inline cudaError_t cudaStreamWaitEvent(cudaStream_t, cudaEvent_t, unsigned) { return cudaSuccess; }
inline cudaError_t cudaEventCreate(cudaEvent_t* e) { *e = new nt_shim_event_{0.0}; return cudaSuccess; }
inline cudaError_t cudaEventCreateWithFlags(cudaEvent_t* e, unsigned) { return cudaEventCreate(e); }
inline cudaError_t cudaEventDestroy(cudaEvent_t e) { delete e; return cudaSuccess; }
inline cudaError_t cudaEventRecord(cudaEvent_t e, cudaStream_t = nullptr) {
    using clk = std::chrono::steady_clock;
    e->t_ms = std::chrono::duration<double, std::milli>(clk::now().time_since_epoch()).count();
    return cudaSuccess;
}
inline cudaError_t cudaEventSynchronize(cudaEvent_t) { return cudaSuccess; }
inline cudaError_t cudaEventElapsedTime(float* ms, cudaEvent_t a, cudaEvent_t b) {
    *ms = (float)(b->t_ms - a->t_ms);
    return cudaSuccess;
}
