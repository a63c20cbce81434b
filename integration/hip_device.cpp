// integration/hip_device.cpp -- REFERENCE-SIDE BINDING: nt::CUDADevice (declared in the reference's
// src/core/device.h:36-72) implemented on the runtime surface of libntransformer_hip.so.  Compiled INSTEAD of
// src/core/device.cu; the extern "C" nt_cuda_* functions of device.h:79-88 are exported by the library itself.
#include "core/device.h"   // the reference's own header
#include "ntk.h"
#include <cstdio>
#include <cstring>

namespace nt {

CUDADevice& CUDADevice::instance() {
    static CUDADevice dev;
    return dev;
}
CUDADevice::~CUDADevice() {}

bool CUDADevice::init(int device_id) {
    if (initialized_) return true;
    if (ntk_device_count() <= device_id) { fprintf(stderr, "No HIP devices found\n"); return false; }
    if (ntk_device_init(device_id) != NTK_OK) return false;
    info_.device_id = device_id;
    ntk_device_name(info_.name, sizeof(info_.name));
    size_t fr = 0, tot = 0;
    ntk_device_mem_info(&fr, &tot);
    info_.total_vram = tot;
    info_.free_vram = fr;
    info_.warp_size = 64;
    for (int i = 0; i < STREAM_COUNT; ++i) streams_[i] = ntk_stream(i);
    initialized_ = true;
    return true;
}
void CUDADevice::synchronize() { ntk_device_synchronize(); }
void CUDADevice::synchronize_stream(StreamType st) { ntk_stream_synchronize(streams_[st]); }
void* CUDADevice::create_event() { return ntk_event_create(); }
void CUDADevice::destroy_event(void* e) { ntk_event_destroy(e); }
void CUDADevice::record_event(void* e, StreamType st) { ntk_event_record(e, streams_[st]); }
void CUDADevice::wait_event(StreamType st, void* e) { ntk_stream_wait_event(streams_[st], e); }   // stream side, like device.cu:96-101
float CUDADevice::elapsed_ms(void* a, void* b) {
    float ms = 0;
    ntk_event_synchronize(b);
    ntk_event_elapsed_ms(a, b, &ms);
    return ms;
}
size_t CUDADevice::free_vram() const { size_t f = 0, t = 0; ntk_device_mem_info(&f, &t); return f; }
size_t CUDADevice::total_vram() const { return info_.total_vram; }
void CUDADevice::print_info() const { fprintf(stderr, "=== GPU Device ===\nName: %s\nVRAM: %.1f GB\n", info_.name, info_.total_vram / 1073741824.0); }
void CUDADevice::memcpy_h2d_async(void* d, const void* s, size_t n, StreamType st) { ntk_memcpy_h2d_async(d, s, n, streams_[st]); }
void CUDADevice::memcpy_d2h_async(void* d, const void* s, size_t n, StreamType st) { ntk_memcpy_d2h_async(d, s, n, streams_[st]); }

}  // namespace nt
