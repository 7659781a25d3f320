// lsq_comm.hip -- the ONE collective of the batch-sharded backward, issued by the library itself (include/lsq_hip.h,
// "rank communicator").  Host code only.
//
// The reference has no distributed code; north_star shards the batch over the GPUs of a node
// with "a single RCCL all-reduce over xGMI for the scale/shift gradient scalars".  A rank's step on BASELINE config 4 is
// ~90 us of GPU time, and torch.distributed's all_reduce costs ~60 us of HOST time per call (Work object, event pool,
// watchdog bookkeeping, Python: profiles/r04_module_sync_cost.txt) -- one enqueue away from host-bound.  Here the same RCCL
// call is made directly: ncclAllReduce on a communicator of the library's own, either on the caller's stream or (begin /
// end) on a side stream ordered with two events, so a 16-24-byte reduction overlaps the next step's kernels and costs the
// host a handful of HIP calls.
//
// RCCL is NOT a link dependency: the library is resolved at the first lsq_hip_comm_* call with dlopen -- the copy PyTorch has
// already loaded when there is one (RTLD_NOLOAD on its name: one RCCL per process), the system's otherwise -- so liblsq_hip.so
// loads and every other entry point works on a box without RCCL.
#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>

#include <hip/hip_runtime.h>

#include "../../include/lsq_hip.h"

namespace lsq {
char* error_buffer();                 // lsq_capi.hip: the thread-local message behind lsq_hip_last_error()
constexpr size_t kErrorBytes = 512;
}  // namespace lsq

namespace {

// the subset of rccl.h this file needs (layout-compatible: an opaque handle, a 128-byte id, int enums)
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[LSQ_COMM_ID_BYTES]; } ncclUniqueId;
enum { ncclSuccess = 0 };
enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3 };
enum { ncclFloat32 = 7, ncclFloat64 = 8 };

struct Rccl {
    void* handle = nullptr;
    int (*GetUniqueId)(ncclUniqueId*) = nullptr;
    int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    int (*GetVersion)(int*) = nullptr;
    char error[256] = "";
};

int fail(int code, const char* fmt, ...) {      // the message lsq_hip_last_error() returns (the calling thread's buffer)
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(lsq::error_buffer(), lsq::kErrorBytes, fmt, ap);
    va_end(ap);
    return code;
}

const Rccl* rccl() {
    static Rccl lib;
    static std::once_flag once;
    std::call_once(once, [] {
        // the copy this process already runs (PyTorch's: libtorch_hip.so needs "librccl.so") before a second one
        const char* loaded[] = {"librccl.so", "librccl.so.1"};
        for (const char* n : loaded) {
            if (lib.handle) break;
            lib.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        }
        const char* fresh[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : fresh) {
            if (lib.handle) break;
            lib.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        }
        if (!lib.handle) {
            snprintf(lib.error, sizeof(lib.error), "RCCL not found (dlopen librccl.so: %s)", dlerror());
            return;
        }
        auto sym = [&](const char* name) -> void* {
            void* p = dlsym(lib.handle, name);
            if (!p && !lib.error[0]) snprintf(lib.error, sizeof(lib.error), "RCCL lacks %s", name);
            return p;
        };
        lib.GetUniqueId = reinterpret_cast<decltype(lib.GetUniqueId)>(sym("ncclGetUniqueId"));
        lib.CommInitRank = reinterpret_cast<decltype(lib.CommInitRank)>(sym("ncclCommInitRank"));
        lib.CommDestroy = reinterpret_cast<decltype(lib.CommDestroy)>(sym("ncclCommDestroy"));
        lib.AllReduce = reinterpret_cast<decltype(lib.AllReduce)>(sym("ncclAllReduce"));
        lib.GetErrorString = reinterpret_cast<decltype(lib.GetErrorString)>(sym("ncclGetErrorString"));
        lib.GetVersion = reinterpret_cast<decltype(lib.GetVersion)>(sym("ncclGetVersion"));
    });
    return lib.error[0] ? nullptr : &lib;
}

int rccl_status(const Rccl* r, int rc, const char* what) {
    if (rc == ncclSuccess) return LSQ_OK;
    return fail(LSQ_ECOMM, "%s: RCCL error %d (%s)", what, rc, r->GetErrorString ? r->GetErrorString(rc) : "?");
}

int hip_status(hipError_t e, const char* what) {
    if (e == hipSuccess) return LSQ_OK;
    return fail(static_cast<int>(e), "%s: %s (%s)", what, hipGetErrorName(e), hipGetErrorString(e));
}

constexpr int kTickets = 8;      // begin / end pairs that may be outstanding at once

}  // namespace

constexpr int kCandidates = 6;   // streams the side stream is picked from (lsq_hip_comm_tune)

struct lsq_comm {
    ncclComm_t comm;
    int rank, nranks, device;
    hipStream_t side;                      // the stream the overlapped reductions run on
    hipStream_t cand[kCandidates];         // ... chosen among these by lsq_hip_comm_tune (cand[0] until then)
    int n_cand;
    int first_was_parked;                  // cand[0] was once some communicator's side stream (taken from the parked list): never destroyed
    std::atomic<int> picked;               // lsq_hip_comm_info's side-stream choice: 0 not tuned, 2 none ran apart, 2 + k = cand[k - 1]
    int event_system_fence;                // the events below were created with (1) / without (0) the system-scope fence
    hipEvent_t ready[kTickets];            // recorded on the caller's stream: the buffer's producer has been enqueued
    hipEvent_t done[kTickets];             // recorded on `side` behind the reduction
    hipEvent_t joined[kTickets];           // lsq_hip_comm_join: recorded on `side` at the time of the join
    std::atomic<uint32_t> next_join;
    std::atomic<uint32_t> next;
};

namespace {

// Side streams outlive their communicator (lsq_hip_comm_destroy says why); the next communicator of the process takes one
// over instead of creating another, so what stays behind is bounded by the communicators alive at once, not by how many were
// ever created.
std::mutex g_parked_mutex;
hipStream_t g_parked[16];
int g_parked_device[16];
int g_n_parked = 0;

hipStream_t take_parked_stream(int device) {
    std::lock_guard<std::mutex> lock(g_parked_mutex);
    for (int i = 0; i < g_n_parked; ++i)
        if (g_parked_device[i] == device) {
            hipStream_t s = g_parked[i];
            g_parked[i] = g_parked[g_n_parked - 1];
            g_parked_device[i] = g_parked_device[g_n_parked - 1];
            --g_n_parked;
            return s;
        }
    return nullptr;
}

void park_stream(hipStream_t s, int device) {
    std::lock_guard<std::mutex> lock(g_parked_mutex);
    if (g_n_parked < 16) { g_parked[g_n_parked] = s; g_parked_device[g_n_parked] = device; ++g_n_parked; }
}

// The two events of a begin order work of ONE device (the reduction reads what a kernel of the caller's stream wrote, the
// caller's stream reads what the reduction wrote): the agent-scope release every kernel ends with is enough, the system-scope
// fence an event records by default -- a writeback + invalidate of the L2s, paid again by the work behind it, 2-8 us per
// step of a BASELINE-config-4 shard (profiles/r05_comm_cost.txt) -- is not needed.  Whether a given RCCL transport agrees
// is for the host layer to CHECK before it relies on it (torchlsq.distributed.native_comm does: a run of reductions with
// changing values through begin / side stream / join, and lsq_hip_comm_configure(event_system_fence = 1) if it fails).
int create_events(lsq_comm* c, int system_fence) {
    const unsigned flags = system_fence ? hipEventDisableTiming : (hipEventDisableTiming | hipEventDisableSystemFence);
    hipError_t e = hipSuccess;
    for (int i = 0; i < kTickets; ++i) c->ready[i] = c->done[i] = c->joined[i] = nullptr;
    for (int i = 0; i < kTickets && e == hipSuccess; ++i) {
        e = hipEventCreateWithFlags(&c->ready[i], flags);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->done[i], flags);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->joined[i], flags);
    }
    c->event_system_fence = system_fence ? 1 : 0;
    return e == hipSuccess ? LSQ_OK : hip_status(e, "rank communicator: hipEventCreateWithFlags");
}

void destroy_events(lsq_comm* c) {
    for (int i = 0; i < kTickets; ++i) {
        if (c->ready[i]) (void)hipEventDestroy(c->ready[i]);
        if (c->done[i]) (void)hipEventDestroy(c->done[i]);
        if (c->joined[i]) (void)hipEventDestroy(c->joined[i]);
        c->ready[i] = c->done[i] = c->joined[i] = nullptr;
    }
}

int check_options(const lsq_comm_options* o, int* system_fence) {
    *system_fence = 0;
    if (!o) return LSQ_OK;
    if (o->size < static_cast<int32_t>(sizeof(int32_t) * 2)) return fail(LSQ_EINVAL, "lsq_comm_options: size %d", o->size);
    if (o->event_system_fence != 0 && o->event_system_fence != 1)
        return fail(LSQ_EINVAL, "lsq_comm_options: event_system_fence must be 0 or 1");
    *system_fence = o->event_system_fence;
    return LSQ_OK;
}

}  // namespace

extern "C" {

int lsq_hip_comm_unique_id(void* id) {
    if (!id) return fail(LSQ_EINVAL, "comm_unique_id: NULL buffer");
    const Rccl* r = rccl();
    if (!r) return fail(LSQ_ECOMM, "comm_unique_id: RCCL is not available on this system");
    ncclUniqueId u;
    if (int rc = rccl_status(r, r->GetUniqueId(&u), "ncclGetUniqueId")) return rc;
    std::memcpy(id, u.internal, LSQ_COMM_ID_BYTES);
    return LSQ_OK;
}

int lsq_hip_comm_create(const void* id, int32_t rank, int32_t nranks, const lsq_comm_options* options, lsq_comm** out) {
    if (!id || !out) return fail(LSQ_EINVAL, "comm_create: NULL argument");
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(LSQ_EINVAL, "comm_create: rank %d of %d", rank, nranks);
    int system_fence = 0;
    if (int rc = check_options(options, &system_fence)) return rc;
    const Rccl* r = rccl();
    if (!r) return fail(LSQ_ECOMM, "comm_create: RCCL is not available on this system");
    lsq_comm* c = new lsq_comm();
    c->rank = rank; c->nranks = nranks; c->next.store(0);
    if (int rc = hip_status(hipGetDevice(&c->device), "hipGetDevice")) { delete c; return rc; }
    ncclUniqueId u;
    std::memcpy(u.internal, id, LSQ_COMM_ID_BYTES);
    if (int rc = rccl_status(r, r->CommInitRank(&c->comm, nranks, u, rank), "ncclCommInitRank")) { delete c; return rc; }
    // The side stream is an ordinary non-blocking stream -- but WHICH one matters (lsq_hip_comm_tune): a few candidates now
    // (the first: one a destroyed communicator left behind, when there is one), the choice when the caller asks for it.
    hipError_t e = hipSuccess;
    c->n_cand = 0;
    c->picked.store(0);
    for (int i = 0; i < kCandidates && e == hipSuccess; ++i) {
        c->cand[i] = i == 0 ? take_parked_stream(c->device) : nullptr;
        if (i == 0) c->first_was_parked = c->cand[0] != nullptr;
        if (!c->cand[i]) e = hipStreamCreateWithFlags(&c->cand[i], hipStreamNonBlocking);
        if (e == hipSuccess) c->n_cand = i + 1;
    }
    c->side = c->n_cand > 0 ? c->cand[0] : nullptr;
    int rc = e == hipSuccess ? create_events(c, system_fence) : hip_status(e, "comm_create: side stream");
    c->next_join.store(0);
    if (rc != LSQ_OK) {      // give back what was made so far: events, candidate streams (the first may be a parked one), the RCCL communicator
        destroy_events(c);
        for (int i = 0; i < c->n_cand; ++i) {
            if (i == 0 && c->first_was_parked) park_stream(c->cand[0], c->device);
            else (void)hipStreamDestroy(c->cand[i]);
        }
        r->CommDestroy(c->comm);
        delete c;
        return rc;
    }
    *out = c;
    return LSQ_OK;
}

int lsq_hip_comm_configure(lsq_comm* c, const lsq_comm_options* options) {
    if (!c || !options) return fail(LSQ_EINVAL, "comm_configure: NULL argument");
    int system_fence = 0;
    if (int rc = check_options(options, &system_fence)) return rc;
    if (system_fence == c->event_system_fence) return LSQ_OK;
    // the events in flight belong to reductions already begun: let them finish, then swap the whole set
    if (int rc = hip_status(hipStreamSynchronize(c->side), "comm_configure: hipStreamSynchronize")) return rc;
    destroy_events(c);
    return create_events(c, system_fence);
}

// A HARDWARE queue of its own for the side stream.  HIP multiplexes its streams onto GPU_MAX_HW_QUEUES (4) hardware queues
// per priority level, PyTorch's pool of 32 streams has filled them long before a communicator is created, and a cross-stream
// wait parked in the SAME hardware queue as the compute stream stalls that stream's next kernel behind it: a BASELINE-config-4
// shard step went 89 -> 106 us that way (profiles/r05_comm_cost.txt), exactly like with torch.distributed's own stream, and
// 94 us with GPU_MAX_HW_QUEUES=8, where the streams happened to land apart.  A stream at another PRIORITY gets another queue
// pool, but the streaming kernels next to a high- or low-priority queue ran 1.4-2.6 x slower (same file).  There is no API
// that tells which queue a stream is on, so it is MEASURED against the stream the caller computes on: a long fill on that
// stream, a 4-byte fill on a candidate right behind it -- a candidate whose little fill finishes before the long one does
// is on another queue.  ~1 ms.  This is the ONE entry point of the library that allocates (up to 256 MB of scratch, freed
// before it returns) and synchronises (`stream` and the candidates): a setup call, made once, at the same point on every
// rank, never while `stream` is capturing and never between a begin and its end / join.
int lsq_hip_comm_tune(lsq_comm* c, void* stream) {
    if (!c) return fail(LSQ_EINVAL, "comm_tune: NULL communicator");
    hipStream_t caller = static_cast<hipStream_t>(stream);
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(caller, &st) != hipSuccess || st != hipStreamCaptureStatusNone) {
        (void)hipGetLastError();
        return fail(LSQ_EINVAL, "comm_tune: the stream is capturing (tune before the capture)");
    }
    c->picked.store(2);              // whatever happens below: a choice was attempted, `side` stays valid
    size_t free_b = 0, total_b = 0;
    if (hipError_t e = hipMemGetInfo(&free_b, &total_b); e != hipSuccess) return hip_status(e, "comm_tune: hipMemGetInfo");
    const size_t bytes = std::min<size_t>(size_t{256} << 20, free_b / 8);
    if (bytes < (size_t{32} << 20)) return LSQ_OK;     // not enough room for a fill that outlasts a launch: keep the first candidate
    char* scratch = nullptr;
    if (hipError_t e = hipMalloc(reinterpret_cast<void**>(&scratch), bytes + 256); e != hipSuccess) {
        (void)hipGetLastError();
        return LSQ_OK;                                   // same: nothing measured, nothing changed
    }
    if (hipError_t e = hipStreamSynchronize(c->side); e != hipSuccess) { (void)hipFree(scratch); return hip_status(e, "comm_tune: hipStreamSynchronize"); }
    hipEvent_t long_done = nullptr, small_done = nullptr;
    bool ok = hipEventCreate(&long_done) == hipSuccess && hipEventCreate(&small_done) == hipSuccess;
    int best = -1;
    for (int i = 0; ok && i < c->n_cand && best < 0; ++i) {
        int apart = 0;
        for (int rep = 0; ok && rep < 2; ++rep) {
            ok = hipStreamSynchronize(caller) == hipSuccess && hipStreamSynchronize(c->cand[i]) == hipSuccess &&
                 hipMemsetAsync(scratch + 256, rep, bytes, caller) == hipSuccess && hipEventRecord(long_done, caller) == hipSuccess &&
                 hipMemsetAsync(scratch, rep, 4, c->cand[i]) == hipSuccess && hipEventRecord(small_done, c->cand[i]) == hipSuccess &&
                 hipEventSynchronize(long_done) == hipSuccess && hipEventSynchronize(small_done) == hipSuccess;
            float ms = 0.0f;
            if (ok && hipEventElapsedTime(&ms, small_done, long_done) == hipSuccess && ms > 0.002f) ++apart;   // the small fill ended first
        }
        if (apart == 2) best = i;
    }
    (void)hipGetLastError();
    if (long_done) (void)hipEventDestroy(long_done);
    if (small_done) (void)hipEventDestroy(small_done);
    (void)hipFree(scratch);
    if (best > 0) c->side = c->cand[best];
    c->picked.store(2 + (best < 0 ? 0 : best + 1));
    return LSQ_OK;
}

int lsq_hip_comm_destroy(lsq_comm* c) {
    if (!c) return LSQ_OK;
    const Rccl* r = rccl();
    (void)hipStreamSynchronize(c->side);
    destroy_events(c);
    // The side stream itself is NOT destroyed: the host layer may have handed it to its allocator as a consumer of buffers
    // (torch: record_stream), which records an event on it when such a buffer is freed -- possibly long after this call
    // (seen: a segmentation fault at interpreter exit).  It is parked for the next communicator of this device instead.
    // (so is a first candidate that came from the parked list and lost the tuning: it WAS somebody's side stream)
    for (int i = 0; i < c->n_cand; ++i) {
        if (!c->cand[i] || c->cand[i] == c->side) continue;
        if (i == 0 && c->first_was_parked) park_stream(c->cand[0], c->device);
        else (void)hipStreamDestroy(c->cand[i]);
    }
    if (c->side) park_stream(c->side, c->device);
    int rc = r ? rccl_status(r, r->CommDestroy(c->comm), "ncclCommDestroy") : LSQ_OK;
    delete c;
    return rc;
}

int lsq_hip_comm_info(const lsq_comm* c, int32_t* out8) {
    if (!c || !out8) return fail(LSQ_EINVAL, "comm_info: NULL argument");
    int version = 0;
    const Rccl* r = rccl();
    if (r && r->GetVersion) (void)r->GetVersion(&version);
    out8[0] = c->rank; out8[1] = c->nranks; out8[2] = c->device; out8[3] = c->picked.load();
    out8[4] = c->event_system_fence; out8[5] = version; out8[6] = static_cast<int32_t>(c->next.load() & 0x7fffffffu); out8[7] = kTickets;
    return LSQ_OK;
}

void* lsq_hip_comm_side_stream(const lsq_comm* c) { return c ? static_cast<void*>(c->side) : nullptr; }

static int check_reduce(const lsq_comm* c, const void* send, void* recv, int64_t count, int dtype, int op, int* nccl_type,
                        int* nccl_op) {
    if (!c || !send || !recv) return fail(LSQ_EINVAL, "comm_all_reduce: NULL argument");
    if (count <= 0) return fail(LSQ_EINVAL, "comm_all_reduce: count must be positive");
    if (dtype != LSQ_F32 && dtype != LSQ_F64) return fail(LSQ_EINVAL, "comm_all_reduce: LSQ_F32 or LSQ_F64 elements");
    if (op < LSQ_COMM_SUM || op > LSQ_COMM_MAX) return fail(LSQ_EINVAL, "comm_all_reduce: unknown reduction %d", op);
    *nccl_type = dtype == LSQ_F64 ? ncclFloat64 : ncclFloat32;
    *nccl_op = op == LSQ_COMM_SUM ? ncclSum : (op == LSQ_COMM_MIN ? ncclMin : ncclMax);
    return LSQ_OK;
}

int lsq_hip_comm_all_reduce(lsq_comm* c, const void* send, void* recv, int64_t count, int dtype, int op, void* stream) {
    int t = 0, o = 0;
    if (int rc = check_reduce(c, send, recv, count, dtype, op, &t, &o)) return rc;
    const Rccl* r = rccl();
    return rccl_status(r, r->AllReduce(send, recv, static_cast<size_t>(count), t, o, c->comm, static_cast<hipStream_t>(stream)),
                       "ncclAllReduce");
}

int lsq_hip_comm_all_reduce_begin(lsq_comm* c, const void* send, void* recv, int64_t count, int dtype, int op, void* stream,
                                  int32_t* ticket) {
    int t = 0, o = 0;
    if (int rc = check_reduce(c, send, recv, count, dtype, op, &t, &o)) return rc;
    if (!ticket) return fail(LSQ_EINVAL, "comm_all_reduce_begin: NULL ticket");
    const Rccl* r = rccl();
    hipStream_t s = static_cast<hipStream_t>(stream);
    const uint32_t k = c->next.fetch_add(1) % kTickets;
    if (int rc = hip_status(hipEventRecord(c->ready[k], s), "comm_all_reduce_begin: hipEventRecord")) return rc;
    if (int rc = hip_status(hipStreamWaitEvent(c->side, c->ready[k], 0), "comm_all_reduce_begin: hipStreamWaitEvent")) return rc;
    if (int rc = rccl_status(r, r->AllReduce(send, recv, static_cast<size_t>(count), t, o, c->comm, c->side), "ncclAllReduce")) return rc;
    if (int rc = hip_status(hipEventRecord(c->done[k], c->side), "comm_all_reduce_begin: hipEventRecord")) return rc;
    *ticket = static_cast<int32_t>(k);
    return LSQ_OK;
}

int lsq_hip_comm_join(lsq_comm* c, void* stream) {
    if (!c) return fail(LSQ_EINVAL, "comm_join: NULL communicator");
    const uint32_t k = c->next_join.fetch_add(1) % kTickets;
    if (int rc = hip_status(hipEventRecord(c->joined[k], c->side), "comm_join: hipEventRecord")) return rc;
    return hip_status(hipStreamWaitEvent(static_cast<hipStream_t>(stream), c->joined[k], 0), "comm_join: hipStreamWaitEvent");
}

int lsq_hip_comm_all_reduce_end(lsq_comm* c, int32_t ticket, void* stream) {
    if (!c || ticket < 0 || ticket >= kTickets) return fail(LSQ_EINVAL, "comm_all_reduce_end: bad ticket %d", ticket);
    return hip_status(hipStreamWaitEvent(static_cast<hipStream_t>(stream), c->done[ticket], 0), "comm_all_reduce_end: hipStreamWaitEvent");
}

}  // extern "C"
