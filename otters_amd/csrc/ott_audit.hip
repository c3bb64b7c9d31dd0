// ott_audit.hip — the checks behind the device-affinity audit build (see ott_audit.h).  Compiled into libotters_hip_audit.so
// only (make audit); in the product library this translation unit is empty.
#ifdef OTT_DEVICE_AUDIT
#define OTT_AUDIT_IMPL
#include "ott_audit.h"

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <unordered_map>

#include "ott_internal.h"

namespace ott {
namespace audit {
namespace {

constexpr int NONE = INT32_MIN;
thread_local int t_logical = NONE;   // logical device of the last use_device() on this thread
thread_local int t_physical = -1;

struct Alloc {
    size_t size;
    int logical, physical;
};
std::mutex g_mu;
std::unordered_map<void*, int> g_streams, g_events;  // handle -> logical device it was created under
std::map<uintptr_t, Alloc> g_allocs;                 // base -> allocation
std::atomic<int> g_violations{0};
std::atomic<int> g_abort{-1};  // -1 = not read yet

void violation(const char* file, int line, const std::string& what) {
    g_violations.fetch_add(1);
    fprintf(stderr, "OTT_DEVICE_AUDIT violation at %s:%d: %s (thread's logical device: %s)\n", file, line, what.c_str(),
            t_logical == NONE ? "none selected" : std::to_string(t_logical).c_str());
    int ab = g_abort.load();
    if (ab < 0) {
        const char* e = getenv("OTT_AUDIT_ABORT");
        ab = (e && e[0] == '0') ? 0 : 1;
        g_abort.store(ab);
    }
    if (ab) abort();
}

bool need_cur(const char* file, int line, const char* what) {
    if (t_logical != NONE) return true;
    violation(file, line, std::string(what) + " before any use_device() on this thread");
    return false;
}

void chk_handle(const std::unordered_map<void*, int>& reg, void* h, const char* kind, const char* what, const char* file, int line, bool same_dev = true) {
    if (!need_cur(file, line, what)) return;
    int lg = NONE;
    {
        std::lock_guard<std::mutex> g(g_mu);
        auto it = reg.find(h);
        if (it != reg.end()) lg = it->second;
    }
    if (lg == NONE) {
        violation(file, line, std::string(what) + ": " + kind + " was not created by the library");
        return;
    }
    if (same_dev && lg != t_logical)
        violation(file, line, std::string(what) + ": " + kind + " belongs to logical device " + std::to_string(lg));
}
void chk_stream(hipStream_t s, const char* what, const char* file, int line) { chk_handle(g_streams, (void*)s, "the stream", what, file, line); }
void chk_event(hipEvent_t e, const char* what, const char* file, int line, bool same_dev = true) { chk_handle(g_events, (void*)e, "the event", what, file, line, same_dev); }

bool find_alloc(const void* p, Alloc& out) {
    std::lock_guard<std::mutex> g(g_mu);
    auto it = g_allocs.upper_bound((uintptr_t)p);
    if (it == g_allocs.begin()) return false;
    --it;
    if ((uintptr_t)p >= it->first + it->second.size) return false;
    out = it->second;
    return true;
}
// a device buffer the call touches must live on the thread's current logical device (host / pinned pointers are not registered)
void chk_buf(const void* p, const char* what, const char* file, int line) {
    Alloc a;
    if (!p || !find_alloc(p, a)) return;
    if (a.logical != t_logical)
        violation(file, line, std::string(what) + ": the device buffer belongs to logical device " + std::to_string(a.logical));
}
void chk_buf_phys(const void* p, int dev, const char* what, const char* file, int line) {
    Alloc a;
    if (!p || !find_alloc(p, a)) return;
    if (a.physical != dev)
        violation(file, line, std::string(what) + ": the buffer lives on device " + std::to_string(a.physical) + ", the call names device " + std::to_string(dev));
}

}  // namespace

hipError_t set_device(int device, int logical) {
    t_logical = logical;
    t_physical = device;
    return hipSetDevice(device);
}

hipStream_t on(hipStream_t s, const char* file, int line) {
    chk_stream(s, "kernel launch", file, line);
    return s;
}

void ptr(const void* p, const ott_store* s, const char* file, int line) {
    Alloc a;
    if (!p || !find_alloc(p, a)) return;
    if (a.logical != s->logical)
        violation(file, line, "a buffer of logical device " + std::to_string(a.logical) + " is handed to work of logical device " + std::to_string(s->logical));
}

hipError_t Malloc(void** p, size_t n, const char* file, int line) {
    need_cur(file, line, "hipMalloc");
    const hipError_t e = hipMalloc(p, n);
    if (e == hipSuccess && *p) {
        std::lock_guard<std::mutex> g(g_mu);
        g_allocs[(uintptr_t)*p] = Alloc{n, t_logical, t_physical};
    }
    return e;
}
hipError_t Free(void* p, const char* file, int line) {
    if (p) {
        need_cur(file, line, "hipFree");
        Alloc a;
        if (find_alloc(p, a) && a.logical != t_logical)
            violation(file, line, "hipFree: the buffer belongs to logical device " + std::to_string(a.logical));
        std::lock_guard<std::mutex> g(g_mu);
        g_allocs.erase((uintptr_t)p);
    }
    return hipFree(p);
}
hipError_t StreamCreateWithFlags(hipStream_t* s, unsigned flags, const char* file, int line) {
    need_cur(file, line, "hipStreamCreateWithFlags");
    const hipError_t e = hipStreamCreateWithFlags(s, flags);
    if (e == hipSuccess) {
        std::lock_guard<std::mutex> g(g_mu);
        g_streams[(void*)*s] = t_logical;
    }
    return e;
}
hipError_t StreamDestroy(hipStream_t s, const char* file, int line) {
    chk_stream(s, "hipStreamDestroy", file, line);
    {
        std::lock_guard<std::mutex> g(g_mu);
        g_streams.erase((void*)s);
    }
    return hipStreamDestroy(s);
}
hipError_t EventCreate(hipEvent_t* e, const char* file, int line) {
    need_cur(file, line, "hipEventCreate");
    const hipError_t rc = hipEventCreate(e);
    if (rc == hipSuccess) {
        std::lock_guard<std::mutex> g(g_mu);
        g_events[(void*)*e] = t_logical;
    }
    return rc;
}
hipError_t EventCreateWithFlags(hipEvent_t* e, unsigned flags, const char* file, int line) {
    need_cur(file, line, "hipEventCreateWithFlags");
    const hipError_t rc = hipEventCreateWithFlags(e, flags);
    if (rc == hipSuccess) {
        std::lock_guard<std::mutex> g(g_mu);
        g_events[(void*)*e] = t_logical;
    }
    return rc;
}
hipError_t EventDestroy(hipEvent_t e, const char* file, int line) {
    chk_event(e, "hipEventDestroy", file, line);
    {
        std::lock_guard<std::mutex> g(g_mu);
        g_events.erase((void*)e);
    }
    return hipEventDestroy(e);
}
hipError_t EventRecord(hipEvent_t e, hipStream_t s, const char* file, int line) {
    chk_stream(s, "hipEventRecord", file, line);
    chk_event(e, "hipEventRecord", file, line);
    return hipEventRecord(e, s);
}
hipError_t EventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b, const char* file, int line) {
    chk_event(a, "hipEventElapsedTime", file, line);
    chk_event(b, "hipEventElapsedTime", file, line);
    return hipEventElapsedTime(ms, a, b);
}
hipError_t StreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags, const char* file, int line) {
    chk_stream(s, "hipStreamWaitEvent", file, line);
    chk_event(e, "hipStreamWaitEvent", file, line, /*same_dev=*/false);  // waiting for ANOTHER device's event is the point
    return hipStreamWaitEvent(s, e, flags);
}
hipError_t StreamSynchronize(hipStream_t s, const char* file, int line) {
    chk_stream(s, "hipStreamSynchronize", file, line);
    return hipStreamSynchronize(s);
}
hipError_t StreamQuery(hipStream_t s, const char* file, int line) {
    chk_stream(s, "hipStreamQuery", file, line);
    return hipStreamQuery(s);
}
hipError_t EventSynchronize(hipEvent_t e, const char* file, int line) {
    chk_event(e, "hipEventSynchronize", file, line);
    return hipEventSynchronize(e);
}
hipError_t MemsetD32Async(hipDeviceptr_t dst, int v, size_t count, hipStream_t s, const char* file, int line) {
    chk_stream(s, "hipMemsetD32Async", file, line);
    chk_buf((const void*)dst, "hipMemsetD32Async", file, line);
    return hipMemsetD32Async(dst, v, count, s);
}
hipError_t MemcpyAsync(void* dst, const void* src, size_t n, hipMemcpyKind kind, hipStream_t s, const char* file, int line) {
    chk_stream(s, "hipMemcpyAsync", file, line);
    chk_buf(dst, "hipMemcpyAsync (destination)", file, line);
    chk_buf(src, "hipMemcpyAsync (source)", file, line);
    return hipMemcpyAsync(dst, src, n, kind, s);
}
hipError_t Memcpy2DAsync(void* dst, size_t dpitch, const void* src, size_t spitch, size_t w, size_t h, hipMemcpyKind kind, hipStream_t s,
                         const char* file, int line) {
    chk_stream(s, "hipMemcpy2DAsync", file, line);
    chk_buf(dst, "hipMemcpy2DAsync (destination)", file, line);
    chk_buf(src, "hipMemcpy2DAsync (source)", file, line);
    return hipMemcpy2DAsync(dst, dpitch, src, spitch, w, h, kind, s);
}
hipError_t MemsetAsync(void* dst, int v, size_t n, hipStream_t s, const char* file, int line) {
    chk_stream(s, "hipMemsetAsync", file, line);
    chk_buf(dst, "hipMemsetAsync", file, line);
    return hipMemsetAsync(dst, v, n, s);
}
hipError_t Memcpy(void* dst, const void* src, size_t n, hipMemcpyKind kind, const char* file, int line) {
    need_cur(file, line, "hipMemcpy");
    chk_buf(dst, "hipMemcpy (destination)", file, line);
    chk_buf(src, "hipMemcpy (source)", file, line);
    return hipMemcpy(dst, src, n, kind);
}
hipError_t Memcpy2D(void* dst, size_t dpitch, const void* src, size_t spitch, size_t w, size_t h, hipMemcpyKind kind, const char* file, int line) {
    need_cur(file, line, "hipMemcpy2D");
    chk_buf(dst, "hipMemcpy2D (destination)", file, line);
    chk_buf(src, "hipMemcpy2D (source)", file, line);
    return hipMemcpy2D(dst, dpitch, src, spitch, w, h, kind);
}
hipError_t MemcpyPeerAsync(void* dst, int ddev, const void* src, int sdev, size_t n, hipStream_t s, const char* file, int line) {
    chk_stream(s, "hipMemcpyPeerAsync", file, line);
    chk_buf_phys(dst, ddev, "hipMemcpyPeerAsync (destination)", file, line);
    chk_buf_phys(src, sdev, "hipMemcpyPeerAsync (source)", file, line);
    return hipMemcpyPeerAsync(dst, ddev, src, sdev, n, s);
}
hipError_t MemcpyPeer(void* dst, int ddev, const void* src, int sdev, size_t n, const char* file, int line) {
    need_cur(file, line, "hipMemcpyPeer");
    chk_buf_phys(dst, ddev, "hipMemcpyPeer (destination)", file, line);
    chk_buf_phys(src, sdev, "hipMemcpyPeer (source)", file, line);
    return hipMemcpyPeer(dst, ddev, src, sdev, n);
}
hipError_t MemGetInfo(size_t* free_b, size_t* total_b, const char* file, int line) {
    need_cur(file, line, "hipMemGetInfo");
    return hipMemGetInfo(free_b, total_b);
}

}  // namespace audit
}  // namespace ott

extern "C" {

// violations seen so far in this process (0 is what a test run must end with)
int ott_audit_violations(void) { return ott::audit::g_violations.load(); }

// The audit checks itself: three deliberate mistakes on `device` — a stream of one logical device synchronised while another
// is current, an event recorded on another logical device's stream, a launch-side stream check from a thread that never
// selected a device — must each be noticed.  Returns how many were (3), without aborting and without counting them.
int ott_audit_selftest(int device) {
    using namespace ott::audit;
    const int abort_was = g_abort.exchange(0);
    const int before = g_violations.load();
    const int logical_was = t_logical, physical_was = t_physical;
    hipStream_t sa = nullptr;
    hipEvent_t eb = nullptr;
    (void)set_device(device, 7001);
    (void)StreamCreateWithFlags(&sa, hipStreamNonBlocking, __FILE__, __LINE__);
    (void)set_device(device, 7002);
    (void)EventCreate(&eb, __FILE__, __LINE__);
    (void)StreamSynchronize(sa, __FILE__, __LINE__);  // mistake 1: shard 7001's stream while 7002 is current
    (void)set_device(device, 7001);
    (void)EventRecord(eb, sa, __FILE__, __LINE__);    // mistake 2: 7002's event on 7001's stream
    t_logical = NONE;
    (void)on(sa, __FILE__, __LINE__);                  // mistake 3: a launch from a thread that selected no device
    (void)set_device(device, 7001);
    (void)StreamSynchronize(sa, __FILE__, __LINE__);
    (void)StreamDestroy(sa, __FILE__, __LINE__);
    (void)set_device(device, 7002);
    (void)EventDestroy(eb, __FILE__, __LINE__);
    const int seen = g_violations.load() - before;
    g_violations.store(before);
    g_abort.store(abort_was);
    t_logical = logical_was;
    t_physical = physical_was;
    return seen;
}

}  // extern "C"
#endif  // OTT_DEVICE_AUDIT
