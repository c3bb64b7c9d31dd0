// fake_rccl.cpp — a stand-in for librccl with N ranks on ONE GPU.  TEST INFRASTRUCTURE (tests/ only; the product never links it).
//
// Why: the in-process multi-GPU store (otters_amd/csrc/ott_multi.hip) exchanges its shards' candidate blocks with
//   ncclCommInitAll(G devices) ... ncclGroupStart(); G x ncclAllGather(send_g, recv_g, block, comm_g, stream_g); ncclGroupEnd();
// and RCCL refuses two ranks on one device — on the pool's one-GPU boxes that branch could only ever run with G = 1.  This
// library exports the ten nccl* entry points libotters_hip binds (ott_comm.hip: rccl()) with the semantics that branch relies
// on, implemented as stream-ordered copies on the one device, so the grouped branch — buffer sizing (recv = block x G on every
// rank), stream ordering, drain(), ott_store_transport() == "rccl" — runs with G = 2 / 4 / 8.  It is loaded through
// OTT_RCCL_LIBRARY=<path> (read once per process).  What it cannot stand in for is the transport itself (xGMI / sockets).
//
// Semantics kept from NCCL: calls made between ncclGroupStart and ncclGroupEnd are only issued at the outermost ncclGroupEnd;
// an all-gather over a communicator clique needs one call per rank with the same count, otherwise ncclInvalidUsage (real NCCL
// would hang: an error is the testable form of that); rank r's receive buffer gets the ranks' blocks in rank order; rank r's
// part is ordered behind everything queued earlier on EVERY rank's stream (an event per rank) and runs on rank r's stream.
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <string.h>

#include <mutex>
#include <vector>

extern "C" {

typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4, ncclInvalidUsage = 5 } ncclResult_t;
typedef struct { char internal[128]; } ncclUniqueId;

}  // extern "C"

namespace {

struct Clique {
    int n = 0;
    int live = 0;
    std::vector<hipEvent_t> ev;  // one per rank: "everything queued on rank r's stream so far"
};
struct Comm {
    Clique* clique;
    int rank, dev;
};
struct Op {
    const void* send;
    void* recv;
    size_t bytes;
    Comm* comm;
    hipStream_t stream;
};

thread_local int t_depth = 0;
thread_local std::vector<Op> t_ops;
std::mutex g_mu;
uint64_t g_gathers = 0;  // grouped all-gathers issued (fake_rccl_gathers)

size_t dtype_size(int dt) {
    switch (dt) {
        case 0: case 1: return 1;   // int8 / uint8
        case 2: case 3: case 7: return 4;  // int32 / uint32 / float32
        case 4: case 5: case 8: return 8;  // int64 / uint64 / float64
        case 6: case 9: return 2;   // float16 / bfloat16
        default: return 0;
    }
}

ncclResult_t issue(std::vector<Op>& ops) {
    // group the pending calls by clique; every clique must be complete
    while (!ops.empty()) {
        Clique* cl = ops[0].comm->clique;
        std::vector<Op> mine((size_t)cl->n, Op{nullptr, nullptr, 0, nullptr, nullptr});
        int seen = 0;
        for (size_t i = 0; i < ops.size();) {
            if (ops[i].comm->clique != cl) {
                i++;
                continue;
            }
            const int r = ops[i].comm->rank;
            if (mine[(size_t)r].comm) return ncclInvalidUsage;  // two calls for one rank in one group
            mine[(size_t)r] = ops[i];
            seen++;
            ops.erase(ops.begin() + (long)i);
        }
        if (seen != cl->n) return ncclInvalidUsage;  // a rank is missing: real NCCL would wait for it for ever
        for (int r = 1; r < cl->n; r++)
            if (mine[(size_t)r].bytes != mine[0].bytes) return ncclInvalidArgument;
        std::lock_guard<std::mutex> g(g_mu);
        for (int r = 0; r < cl->n; r++) {
            if (hipSetDevice(mine[(size_t)r].comm->dev) != hipSuccess) return ncclUnhandledCudaError;
            if (hipEventRecord(cl->ev[(size_t)r], mine[(size_t)r].stream) != hipSuccess) return ncclUnhandledCudaError;
        }
        for (int r = 0; r < cl->n; r++) {
            const Op& me = mine[(size_t)r];
            if (hipSetDevice(me.comm->dev) != hipSuccess) return ncclUnhandledCudaError;
            for (int j = 0; j < cl->n; j++)
                if (j != r && hipStreamWaitEvent(me.stream, cl->ev[(size_t)j], 0) != hipSuccess) return ncclUnhandledCudaError;
            for (int j = 0; j < cl->n; j++)
                if (hipMemcpyAsync((char*)me.recv + (size_t)j * me.bytes, mine[(size_t)j].send, me.bytes, hipMemcpyDeviceToDevice, me.stream) != hipSuccess)
                    return ncclUnhandledCudaError;
        }
        g_gathers++;
    }
    return ncclSuccess;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetVersion(int* version) {
    if (!version) return ncclInvalidArgument;
    *version = 9900001;  // no RCCL release: "fake" to whoever prints it
    return ncclSuccess;
}
const char* ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "no error";
        case ncclUnhandledCudaError: return "unhandled hip error (fake_rccl)";
        case ncclInvalidArgument: return "invalid argument (fake_rccl: the ranks' counts differ)";
        case ncclInvalidUsage: return "invalid usage (fake_rccl: a group does not hold exactly one call per rank of the communicator clique)";
        default: return "error (fake_rccl)";
    }
}
ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    if (!id) return ncclInvalidArgument;
    memset(id, 0, sizeof(*id));
    memcpy(id->internal, "fake_rccl", 10);
    return ncclSuccess;
}
ncclResult_t ncclCommInitAll(void** comms, int ndev, const int* devlist) {
    if (!comms || ndev < 1) return ncclInvalidArgument;
    Clique* cl = new Clique();
    cl->n = cl->live = ndev;
    cl->ev.resize((size_t)ndev);
    for (int r = 0; r < ndev; r++) {
        const int dev = devlist ? devlist[r] : r;
        if (hipSetDevice(dev) != hipSuccess || hipEventCreateWithFlags(&cl->ev[(size_t)r], hipEventDisableTiming) != hipSuccess) return ncclUnhandledCudaError;
        comms[r] = new Comm{cl, r, dev};
    }
    return ncclSuccess;
}
// one process = one rank: only a world of one can be formed without a transport
ncclResult_t ncclCommInitRank(void** comm, int nranks, ncclUniqueId, int rank) {
    if (!comm || nranks != 1 || rank != 0) return ncclInvalidUsage;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return ncclUnhandledCudaError;
    return ncclCommInitAll(comm, 1, &dev);
}
ncclResult_t ncclCommDestroy(void* c) {
    Comm* cm = (Comm*)c;
    if (!cm) return ncclInvalidArgument;
    std::lock_guard<std::mutex> g(g_mu);
    Clique* cl = cm->clique;
    (void)hipSetDevice(cm->dev);
    (void)hipEventDestroy(cl->ev[(size_t)cm->rank]);
    if (--cl->live == 0) delete cl;
    delete cm;
    return ncclSuccess;
}
ncclResult_t ncclCommCount(const void* c, int* count) {
    if (!c || !count) return ncclInvalidArgument;
    *count = ((const Comm*)c)->clique->n;
    return ncclSuccess;
}
ncclResult_t ncclGroupStart() {
    t_depth++;
    return ncclSuccess;
}
ncclResult_t ncclGroupEnd() {
    if (t_depth <= 0) return ncclInvalidUsage;
    if (--t_depth > 0) return ncclSuccess;
    std::vector<Op> ops;
    ops.swap(t_ops);
    return issue(ops);
}
ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, int dtype, void* comm, hipStream_t stream) {
    const size_t es = dtype_size(dtype);
    if (!comm || !es || (count && (!send || !recv))) return ncclInvalidArgument;
    t_ops.push_back(Op{send, recv, count * es, (Comm*)comm, stream});
    if (t_depth > 0) return ncclSuccess;
    std::vector<Op> ops;
    ops.swap(t_ops);
    return issue(ops);  // outside a group: complete only for a clique of one
}

// how many all-gathers (one per clique and group) have been issued in this process: the tests assert the branch really ran
uint64_t fake_rccl_gathers(void) {
    std::lock_guard<std::mutex> g(g_mu);
    return g_gathers;
}

}  // extern "C"
