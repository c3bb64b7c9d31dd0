// ott_audit.h — device-affinity audit build of libotters_hip (make audit: -DOTT_DEVICE_AUDIT -> libotters_hip_audit.so).
// Test infrastructure: never part of the product library (ott_internal.h includes this file only under OTT_DEVICE_AUDIT).
//
// What it is for.  The in-process multi-GPU store (ott_multi.hip) runs a shard per device, each from its own host thread, and
// the pool's boxes have ONE GPU: every device list the tests can use repeats ordinal 0, so a call made for shard 3 while
// shard 1's device is current — a missed use_device() on a shard thread, in the background plane builder, in drain() — does
// the right thing by accident there and corrupts memory (or fails with an invalid handle) on eight real GPUs.  This build
// makes that mistake visible on one GPU: use_device() records the store's LOGICAL device id (ott_store::logical; with option
// "multi_fake_distinct" every shard has its own) in a thread-local, every stream / event / device allocation remembers the
// logical id it was created under, and every HIP call the library makes is checked:
//   * allocation, launch (hipLaunchKernelGGL is redefined here), async copy / memset, event record, stream wait / synchronize / query: the stream (and the
//     event recorded on it) must belong to the thread's current logical device, which must have been selected by use_device;
//   * device buffers written by memset / D2D copies / named by OTT_AUDIT_PTR must belong to the current logical device (peer
//     copies name both sides and are checked against the PHYSICAL ordinals they pass);
//   * hipEventElapsedTime needs two events of one device; a raw hipSetDevice() does not compile.
// A violation prints one line to stderr, counts (ott_audit_violations()) and aborts unless OTT_AUDIT_ABORT=0.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>

struct ott_store;

namespace ott {
namespace audit {
hipError_t set_device(int device, int logical);
hipStream_t on(hipStream_t s, const char* file, int line);  // a launch on `s`
void ptr(const void* p, const ott_store* s, const char* file, int line);
hipError_t Malloc(void** p, size_t n, const char* file, int line);
hipError_t Free(void* p, const char* file, int line);
hipError_t StreamCreateWithFlags(hipStream_t* s, unsigned flags, const char* file, int line);
hipError_t StreamDestroy(hipStream_t s, const char* file, int line);
hipError_t EventCreate(hipEvent_t* e, const char* file, int line);
hipError_t EventCreateWithFlags(hipEvent_t* e, unsigned flags, const char* file, int line);
hipError_t EventDestroy(hipEvent_t e, const char* file, int line);
hipError_t EventRecord(hipEvent_t e, hipStream_t s, const char* file, int line);
hipError_t EventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b, const char* file, int line);
hipError_t StreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags, const char* file, int line);
hipError_t StreamSynchronize(hipStream_t s, const char* file, int line);
hipError_t StreamQuery(hipStream_t s, const char* file, int line);
hipError_t EventSynchronize(hipEvent_t e, const char* file, int line);
hipError_t MemsetD32Async(hipDeviceptr_t dst, int v, size_t count, hipStream_t s, const char* file, int line);
hipError_t MemcpyAsync(void* dst, const void* src, size_t n, hipMemcpyKind kind, hipStream_t s, const char* file, int line);
hipError_t Memcpy2DAsync(void* dst, size_t dpitch, const void* src, size_t spitch, size_t w, size_t h, hipMemcpyKind kind, hipStream_t s,
                         const char* file, int line);
hipError_t MemsetAsync(void* dst, int v, size_t n, hipStream_t s, const char* file, int line);
hipError_t Memcpy(void* dst, const void* src, size_t n, hipMemcpyKind kind, const char* file, int line);
hipError_t Memcpy2D(void* dst, size_t dpitch, const void* src, size_t spitch, size_t w, size_t h, hipMemcpyKind kind, const char* file, int line);
hipError_t MemcpyPeerAsync(void* dst, int ddev, const void* src, int sdev, size_t n, hipStream_t s, const char* file, int line);
hipError_t MemcpyPeer(void* dst, int ddev, const void* src, int sdev, size_t n, const char* file, int line);
hipError_t MemGetInfo(size_t* free_b, size_t* total_b, const char* file, int line);
}  // namespace audit
inline hipError_t use_device_raw(int device, int logical) { return audit::set_device(device, logical); }
}  // namespace ott

#ifndef OTT_AUDIT_IMPL  // (ott_audit.hip itself calls the real entry points)
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernelName, numBlocks, numThreads, memPerBlock, streamId, ...)                                        \
    do {                                                                                                                          \
        (kernelName)<<<(numBlocks), (numThreads), (memPerBlock), ::ott::audit::on((streamId), __FILE__, __LINE__)>>>(__VA_ARGS__); \
    } while (0)
#define OTT_AUDIT_PTR(p, store) (::ott::audit::ptr((p), (store), __FILE__, __LINE__))
#define hipSetDevice(...) OTT_DEVICE_AUDIT_raw_hipSetDevice_use_ott_use_device_instead(__VA_ARGS__)
#define hipMalloc(p, n) ::ott::audit::Malloc((void**)(p), (n), __FILE__, __LINE__)
#define hipFree(p) ::ott::audit::Free((void*)(p), __FILE__, __LINE__)
#define hipStreamCreateWithFlags(s, f) ::ott::audit::StreamCreateWithFlags((s), (f), __FILE__, __LINE__)
#define hipStreamCreate(s) OTT_DEVICE_AUDIT_use_hipStreamCreateWithFlags(s)
#define hipStreamDestroy(s) ::ott::audit::StreamDestroy((s), __FILE__, __LINE__)
#define hipEventCreate(e) ::ott::audit::EventCreate((e), __FILE__, __LINE__)
#define hipEventCreateWithFlags(e, f) ::ott::audit::EventCreateWithFlags((e), (f), __FILE__, __LINE__)
#define hipEventDestroy(e) ::ott::audit::EventDestroy((e), __FILE__, __LINE__)
#define hipEventRecord(e, s) ::ott::audit::EventRecord((e), (s), __FILE__, __LINE__)
#define hipEventElapsedTime(ms, a, b) ::ott::audit::EventElapsedTime((ms), (a), (b), __FILE__, __LINE__)
#define hipStreamWaitEvent(s, e, f) ::ott::audit::StreamWaitEvent((s), (e), (f), __FILE__, __LINE__)
#define hipStreamSynchronize(s) ::ott::audit::StreamSynchronize((s), __FILE__, __LINE__)
#define hipStreamQuery(s) ::ott::audit::StreamQuery((s), __FILE__, __LINE__)
#define hipEventSynchronize(e) ::ott::audit::EventSynchronize((e), __FILE__, __LINE__)
#define hipMemsetD32Async(d, v, n, st) ::ott::audit::MemsetD32Async((d), (v), (n), (st), __FILE__, __LINE__)
#define hipMemcpyAsync(d, s, n, k, st) ::ott::audit::MemcpyAsync((d), (s), (n), (k), (st), __FILE__, __LINE__)
#define hipMemcpy2DAsync(d, dp, s, sp, w, h, k, st) ::ott::audit::Memcpy2DAsync((d), (dp), (s), (sp), (w), (h), (k), (st), __FILE__, __LINE__)
#define hipMemsetAsync(d, v, n, st) ::ott::audit::MemsetAsync((d), (v), (n), (st), __FILE__, __LINE__)
#define hipMemcpy(d, s, n, k) ::ott::audit::Memcpy((d), (s), (n), (k), __FILE__, __LINE__)
#define hipMemcpy2D(d, dp, s, sp, w, h, k) ::ott::audit::Memcpy2D((d), (dp), (s), (sp), (w), (h), (k), __FILE__, __LINE__)
#define hipMemcpyPeerAsync(d, dd, s, sd, n, st) ::ott::audit::MemcpyPeerAsync((d), (dd), (s), (sd), (n), (st), __FILE__, __LINE__)
#define hipMemcpyPeer(d, dd, s, sd, n) ::ott::audit::MemcpyPeer((d), (dd), (s), (sd), (n), __FILE__, __LINE__)
#define hipMemGetInfo(f, t) ::ott::audit::MemGetInfo((f), (t), __FILE__, __LINE__)
#endif
