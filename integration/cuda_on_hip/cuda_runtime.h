// integration/cuda_on_hip/cuda_runtime.h -- REFERENCE-SIDE BINDING (not part of libntransformer_hip.so).
//
// The reference's *host* sources include <cuda_runtime.h> for a handful of runtime calls that sit outside
// its CUDADevice / nt_cuda_* abstraction:
//     src/model/transformer.cpp:28,387   cudaFreeHost / cudaMallocHost          (pinned scalar for the layer-skip path)
//     src/model/transformer.cpp:879-881  cudaMemcpyAsync + cudaStreamSynchronize (same path)
//     src/memory/streamer.cu             cudaMalloc / cudaMallocHost / cudaHostRegister / events / cudaMemGetInfo
//                                        (the SLEP streamer: linked because Transformer owns a LayerStreamer member,
//                                         never entered on the resident path)
//     src/core/tensor.cpp, allocator.cpp cudaMalloc / cudaFree / cudaGetErrorString
// A maintainer building the reference for MI355X puts this directory first on the include path, compiles
// integration/nt_cuda_launchers.cpp + integration/hip_device.cpp instead of src/cuda/*.cu + src/core/device.cu, and links
// libntransformer_hip.so + libamdhip64: no reference source file changes.  Each name below is the HIP runtime's
// own entry point for the same operation; nothing is emulated.  oracle/Makefile builds `_ref/ref_logits_hip` and
// `_ref/ntransformer_ref_hip` (the reference's Transformer / CLI, unmodified, decoding on the MI355X kernels) this way.
#pragma once
#ifndef __HIP_PLATFORM_AMD__
#define __HIP_PLATFORM_AMD__ 1
#endif
#include <hip/hip_runtime_api.h>

typedef hipError_t  cudaError_t;
typedef hipStream_t cudaStream_t;
typedef hipEvent_t  cudaEvent_t;
typedef hipMemcpyKind cudaMemcpyKind;
typedef hipDeviceProp_t cudaDeviceProp;

#define cudaSuccess                 hipSuccess
#define cudaErrorMemoryAllocation   hipErrorOutOfMemory
#define cudaMemcpyHostToHost        hipMemcpyHostToHost
#define cudaMemcpyHostToDevice      hipMemcpyHostToDevice
#define cudaMemcpyDeviceToHost      hipMemcpyDeviceToHost
#define cudaMemcpyDeviceToDevice    hipMemcpyDeviceToDevice
#define cudaMemcpyDefault           hipMemcpyDefault
#define cudaStreamNonBlocking       hipStreamNonBlocking
#define cudaEventDisableTiming      hipEventDisableTiming
#define cudaHostRegisterReadOnly    hipHostRegisterReadOnly

#define cudaGetErrorString          hipGetErrorString
#define cudaMalloc                  hipMalloc
#define cudaFree                    hipFree
#define cudaMallocHost              hipHostMalloc
#define cudaFreeHost                hipHostFree
#define cudaHostRegister            hipHostRegister
#define cudaHostUnregister          hipHostUnregister
#define cudaMemcpy                  hipMemcpy
#define cudaMemcpyAsync             hipMemcpyAsync
#define cudaMemset                  hipMemset
#define cudaMemGetInfo              hipMemGetInfo
#define cudaGetDeviceCount          hipGetDeviceCount
#define cudaSetDevice               hipSetDevice
#define cudaGetDeviceProperties     hipGetDeviceProperties
#define cudaDeviceSynchronize       hipDeviceSynchronize
#define cudaStreamCreateWithFlags   hipStreamCreateWithFlags
#define cudaStreamDestroy           hipStreamDestroy
#define cudaStreamSynchronize       hipStreamSynchronize
#define cudaStreamWaitEvent         hipStreamWaitEvent
#define cudaEventCreate             hipEventCreate
#define cudaEventCreateWithFlags    hipEventCreateWithFlags
#define cudaEventDestroy            hipEventDestroy
#define cudaEventRecord             hipEventRecord
#define cudaEventSynchronize        hipEventSynchronize
#define cudaEventElapsedTime        hipEventElapsedTime
