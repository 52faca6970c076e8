// Data-parallel gradient exchange behind the C ABI: RCCL (librccl.so, loaded at run time) all-reduce of slices of the
// flat fp32 gradient buffer over xGMI, plus the few stream / event primitives the host needs to overlap the collective
// with the rest of the backward tape (an event recorded INSIDE the backward hipGraph marks a gradient bucket final; the
// communication stream waits for it and launches the bucket's all-reduce while the compute stream keeps going).
//
// The reference has no communication layer at all ("Collective call sites: none"); this is the one new
// component BASELINE.json asks for, inserted between loss.backward() and optimizer.step() (train_model.py:121-122).
//
// RCCL is bound with dlopen/dlsym instead of a link-time dependency: a PyTorch-ROCm process already carries its own copy
// of librccl.so, and binding by handle keeps this library's calls on exactly the copy the host selected (uz_comm_load).
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>
#include "uz_common.h"

namespace {

// the slice of rccl.h this file needs (ABI-stable: NCCL 2.x)
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
enum { ncclSuccess = 0 };
enum { ncclSum = 0, ncclAvg = 4 };
enum { ncclFloat32 = 7 };

struct Rccl {
    void* handle = nullptr;
    int (*GetUniqueId)(ncclUniqueId*) = nullptr;
    int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*CommCount)(ncclComm_t, int*) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    int (*GetVersion)(int*) = nullptr;
} g;

struct Comm {
    ncclComm_t comm;
    int rank, nranks;
};

int load(const char* path) {
    if (g.handle) return 0;
    const char* cands[] = {path, "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* c : cands) {
        if (!c || !*c) continue;
        g.handle = dlopen(c, RTLD_NOW | RTLD_LOCAL);
        if (g.handle) break;
    }
    if (!g.handle) return uz::fail("comm: cannot load librccl.so (%s)", dlerror());
#define UZ_SYM(field, name)                                                           \
    do {                                                                              \
        *reinterpret_cast<void**>(&g.field) = dlsym(g.handle, name);                  \
        if (!g.field) { g.handle = nullptr; return uz::fail("comm: librccl.so lacks %s", name); } \
    } while (0)
    UZ_SYM(GetUniqueId, "ncclGetUniqueId");
    UZ_SYM(CommInitRank, "ncclCommInitRank");
    UZ_SYM(CommDestroy, "ncclCommDestroy");
    UZ_SYM(CommCount, "ncclCommCount");
    UZ_SYM(AllReduce, "ncclAllReduce");
    UZ_SYM(Broadcast, "ncclBroadcast");
    UZ_SYM(GroupStart, "ncclGroupStart");
    UZ_SYM(GroupEnd, "ncclGroupEnd");
    UZ_SYM(GetErrorString, "ncclGetErrorString");
    UZ_SYM(GetVersion, "ncclGetVersion");
#undef UZ_SYM
    return 0;
}

int nccl_fail(const char* what, int rc) { return uz::fail("comm: %s failed: %s", what, g.GetErrorString ? g.GetErrorString(rc) : "?"); }

}  // namespace

extern "C" int uz_comm_load(const char* librccl_path) { return load(librccl_path); }

extern "C" int uz_comm_version(void) {
    if (load(nullptr)) return -1;
    int v = 0;
    return g.GetVersion(&v) == ncclSuccess ? v : -1;
}

extern "C" int uz_comm_unique_id(void* out_128B) {
    if (int rc = load(nullptr)) return rc;
    ncclUniqueId id;
    if (int rc = g.GetUniqueId(&id)) return nccl_fail("ncclGetUniqueId", rc);
    memcpy(out_128B, id.internal, sizeof(id.internal));
    return 0;
}

extern "C" int uz_comm_init(int rank, int nranks, const void* unique_id_128B, void** comm_out) {
    if (int rc = load(nullptr)) return rc;
    UZ_REQUIRE(comm_out && unique_id_128B && nranks >= 1 && rank >= 0 && rank < nranks, "comm_init: bad arguments");
    ncclUniqueId id;
    memcpy(id.internal, unique_id_128B, sizeof(id.internal));
    Comm* c = new Comm{nullptr, rank, nranks};
    if (int rc = g.CommInitRank(&c->comm, nranks, id, rank)) { delete c; return nccl_fail("ncclCommInitRank", rc); }
    *comm_out = c;
    return 0;
}

extern "C" void uz_comm_destroy(void* comm) {
    Comm* c = static_cast<Comm*>(comm);
    if (!c) return;
    if (g.CommDestroy) g.CommDestroy(c->comm);
    delete c;
}

// the communicator's size as RCCL reports it (ncclCommCount), not the number the host asked for
extern "C" int uz_comm_size(void* comm) {
    Comm* c = static_cast<Comm*>(comm);
    if (!c) return -1;
    int n = -1;
    if (!g.CommCount || g.CommCount(c->comm, &n) != ncclSuccess) return -1;
    return n;
}

// In-place mean over the ranks of flat[0..count): ONE ncclAllReduce(avg, f32) enqueued on `stream` (asynchronous).
extern "C" int uz_allreduce_mean_f32(void* comm, float* flat, size_t count, void* stream) {
    Comm* c = static_cast<Comm*>(comm);
    UZ_REQUIRE(c && flat, "allreduce_mean: null communicator or buffer");
    if (count == 0) return 0;
    if (int rc = g.AllReduce(flat, flat, count, ncclFloat32, ncclAvg, c->comm, uz::S(stream))) return nccl_fail("ncclAllReduce", rc);
    return 0;
}

// Several slices in ONE RCCL group launch (contiguous buckets handed over together): offs_counts = {off0, n0, off1, n1, ...}
extern "C" int uz_allreduce_mean_f32_multi(void* comm, float* flat, const int64_t* offs_counts, int n_slices, void* stream) {
    Comm* c = static_cast<Comm*>(comm);
    UZ_REQUIRE(c && flat && (offs_counts || n_slices == 0), "allreduce_mean_multi: null argument");
    if (int rc = g.GroupStart()) return nccl_fail("ncclGroupStart", rc);
    for (int i = 0; i < n_slices; ++i) {
        float* p = flat + offs_counts[2 * i];
        if (offs_counts[2 * i + 1] <= 0) continue;
        if (int rc = g.AllReduce(p, p, (size_t)offs_counts[2 * i + 1], ncclFloat32, ncclAvg, c->comm, uz::S(stream))) {
            g.GroupEnd();
            return nccl_fail("ncclAllReduce", rc);
        }
    }
    if (int rc = g.GroupEnd()) return nccl_fail("ncclGroupEnd", rc);
    return 0;
}

// Parameter broadcast from `root` at start-up (replicas must start identical).
extern "C" int uz_broadcast_f32(void* comm, float* flat, size_t count, int root, void* stream) {
    Comm* c = static_cast<Comm*>(comm);
    UZ_REQUIRE(c && flat, "broadcast: null communicator or buffer");
    if (count == 0) return 0;
    if (int rc = g.Broadcast(flat, flat, count, ncclFloat32, root, c->comm, uz::S(stream))) return nccl_fail("ncclBroadcast", rc);
    return 0;
}

// ---------------------------------------------------------------- streams / events for the overlap
extern "C" int uz_stream_create(void** stream_out, int high_priority) {
    UZ_REQUIRE(stream_out, "stream_create: null");
    hipStream_t s;
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    const hipError_t e = hipStreamCreateWithPriority(&s, hipStreamNonBlocking, high_priority ? hi : lo);
    if (e != hipSuccess) return uz::fail("stream_create: %s", hipGetErrorString(e));
    *stream_out = s;
    return 0;
}
extern "C" void uz_stream_destroy(void* stream) { if (stream) (void)hipStreamDestroy(uz::S(stream)); }
extern "C" int uz_stream_synchronize(void* stream) {
    const hipError_t e = hipStreamSynchronize(uz::S(stream));
    return e == hipSuccess ? 0 : uz::fail("stream_synchronize: %s", hipGetErrorString(e));
}
extern "C" int uz_event_create(void** event_out, int timing) {
    UZ_REQUIRE(event_out, "event_create: null");
    hipEvent_t ev;
    const hipError_t e = hipEventCreateWithFlags(&ev, timing ? hipEventDefault : hipEventDisableTiming);
    if (e != hipSuccess) return uz::fail("event_create: %s", hipGetErrorString(e));
    *event_out = ev;
    return 0;
}
extern "C" void uz_event_destroy(void* event) { if (event) (void)hipEventDestroy(static_cast<hipEvent_t>(event)); }
// Works both on a live stream and on a stream under capture.  Under capture the record becomes an event-record NODE of the
// graph being built (added explicitly behind the stream's current capture dependencies, which it then replaces), so every
// later replay of the graph re-records the event at that point of the DAG and streams outside the graph can wait for it.
extern "C" int uz_event_record(void* event, void* stream) {
    hipStream_t st = uz::S(stream);
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(st, &cs);
    if (cs != hipStreamCaptureStatusActive) {
        const hipError_t e = hipEventRecord(static_cast<hipEvent_t>(event), st);
        return e == hipSuccess ? 0 : uz::fail("event_record: %s", hipGetErrorString(e));
    }
    unsigned long long id = 0;
    hipGraph_t graph = nullptr;
    const hipGraphNode_t* deps = nullptr;
    size_t nd = 0;
    hipError_t e = hipStreamGetCaptureInfo_v2(st, &cs, &id, &graph, &deps, &nd);
    if (e != hipSuccess || !graph) return uz::fail("event_record: hipStreamGetCaptureInfo_v2: %s", hipGetErrorString(e));
    hipGraphNode_t node = nullptr;
    e = hipGraphAddEventRecordNode(&node, graph, deps, nd, static_cast<hipEvent_t>(event));
    if (e != hipSuccess) return uz::fail("event_record: hipGraphAddEventRecordNode: %s", hipGetErrorString(e));
    e = hipStreamUpdateCaptureDependencies(st, &node, 1, hipStreamSetCaptureDependencies);
    return e == hipSuccess ? 0 : uz::fail("event_record: hipStreamUpdateCaptureDependencies: %s", hipGetErrorString(e));
}
extern "C" int uz_stream_wait_event(void* stream, void* event) {
    const hipError_t e = hipStreamWaitEvent(uz::S(stream), static_cast<hipEvent_t>(event), 0);
    return e == hipSuccess ? 0 : uz::fail("stream_wait_event: %s", hipGetErrorString(e));
}
extern "C" int uz_event_elapsed_ms(void* start, void* stop, float* ms_out) {
    const hipError_t e = hipEventElapsedTime(ms_out, static_cast<hipEvent_t>(start), static_cast<hipEvent_t>(stop));
    return e == hipSuccess ? 0 : uz::fail("event_elapsed: %s", hipGetErrorString(e));
}
