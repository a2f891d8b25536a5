// trk_api.hip -- host side of libtrk.so: the extern "C" entry points declared in
// include/trk.h.  HIP runtime only (no torch); RCCL is bound lazily with dlopen
// so that the library loads on machines without it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <chrono>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <zlib.h>
#include <cmath>
#include <deque>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/trk.h"
#include "../../include/trk_test.h"
#include "../../include/trk_vcf.h"
#include "trk_binom.h"
#include "trk_student.h"
#include "trk_internal.h"

// ---- RCCL (subset), resolved at run time --------------------------------------
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
enum { trkNcclUint8 = 1, trkNcclInt64 = 4 };
enum { trkNcclSum = 0 };
struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
static RcclApi g_rccl;
static std::string g_init_error;

static bool load_rccl(std::string& err) {
    if (g_rccl.handle) return true;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void* h = nullptr;
    for (const char* n : names) {
        h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) {
        err = std::string("cannot dlopen librccl: ") + (dlerror() ? dlerror() : "?");
        return false;
    }
    RcclApi a;
    a.handle = h;
    a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))dlsym(h, "ncclCommInitRank");
    a.CommDestroy = (decltype(a.CommDestroy))dlsym(h, "ncclCommDestroy");
    a.AllReduce = (decltype(a.AllReduce))dlsym(h, "ncclAllReduce");
    a.AllGather = (decltype(a.AllGather))dlsym(h, "ncclAllGather");
    a.GetErrorString = (decltype(a.GetErrorString))dlsym(h, "ncclGetErrorString");
    a.GroupStart = (decltype(a.GroupStart))dlsym(h, "ncclGroupStart");
    a.GroupEnd = (decltype(a.GroupEnd))dlsym(h, "ncclGroupEnd");
    if (!a.GetUniqueId || !a.CommInitRank || !a.AllReduce || !a.AllGather) {
        err = "librccl is missing required symbols";
        return false;
    }
    g_rccl = a;
    return true;
}

// ---- context ---------------------------------------------------------------------
struct ProfRec {
    int kernel;
    hipEvent_t start, stop;
};

// trk_thread_queue: the queue of the CALLING thread, when it has asked for one of its own (-1: the selected queue)
static thread_local int t_queue = -1;

struct InflateState;
static void inflate_state_free(InflateState* st);
struct DeflateState {          // trk_deflate_bgzf's device buffers and pinned staging, grown on demand, kept for the next call
    uint8_t* d_text = nullptr; size_t text_cap = 0;
    uint8_t* d_slots = nullptr; size_t slots_cap = 0;
    uint8_t* d_out = nullptr; size_t out_cap = 0;
    uint8_t* d_tok = nullptr; size_t tok_cap = 0;
    uint8_t* d_tab = nullptr; size_t tab_cap = 0;      // sizes (u32) + offsets (u64)
    uint64_t* h_off = nullptr; size_t h_cap = 0;       // pinned
    std::vector<uint32_t> crc;
    hipStream_t q = nullptr;       // a queue of its own: the call comes from a writer thread beside the caller's kernels
    std::mutex m;                  // one call at a time (the buffers are the state's)
};

struct trk_ctx {
    int device = 0;
    int n_cu = 256;
    // two in-order queues (trk_stream_select / trk_stream_wait): entry points enqueue on the selected one, and
    // each has its own finaliser scratch, so that e.g. statSTR's finaliser can run beside dumpSTR's call-filter pass
    hipStream_t streams[TRK_N_STREAMS] = {};
    hipEvent_t join_event[TRK_N_STREAMS] = {};
    hipEvent_t user_event[TRK_N_EVENTS] = {};
    bool user_event_set[TRK_N_EVENTS] = {};
    int cur = 0;
    hipStream_t s() const { return streams[t_queue >= 0 ? t_queue : cur]; }
    std::string err;
    // entry points may be called from two threads at once (the reader's helper thread on its own queue, trk_thread_queue):
    // the profile bookkeeping (event pool, pending brackets) and the error string are shared and go through this lock
    std::mutex book_m;
    std::mutex queue_m;       // creation of a queue on its first use (ensure_queue)
    hipEvent_t t_start[TRK_N_TIMERS] = {};
    hipEvent_t t_stop[TRK_N_TIMERS] = {};
    bool profiling = false;
    std::vector<ProfRec> prof_pending;
    std::vector<hipEvent_t> event_pool;
    int64_t prof_n[TRK_K_COUNT] = {};
    double prof_ms[TRK_K_COUNT] = {};
    int32_t* scratch_[TRK_N_STREAMS] = {};  // finaliser class-count scratch (per queue)
    size_t scratch_bytes_[TRK_N_STREAMS] = {};
    void* worklist_[TRK_N_STREAMS] = {};    // deferred HWE tests (count + items)
    size_t worklist_bytes_[TRK_N_STREAMS] = {};
    void* cf_ws_[TRK_N_STREAMS] = {};       // call-filter partial sample counters (per queue)
    size_t cf_ws_bytes_[TRK_N_STREAMS] = {};
    void* assoc_ws_[TRK_N_STREAMS] = {};   // per queue: scans on different queues run side by side    // associaTR scan workspace (Gram, partial records, class counts)
    size_t assoc_ws_bytes_[TRK_N_STREAMS] = {};
    ncclComm_t comm = nullptr;
    int rank = 0, n_ranks = 1;
    // the reserved pair of output planes (trk_reserve_pair): owned by the context, lent out by trk_dev_alloc_pair
    struct InflateState* inflate = nullptr;   // the reader's inflate hook served by this context (trk_inflate_hook)
    DeflateState* deflate = nullptr;          // trk_deflate_bgzf
    void* res_plane[2] = {nullptr, nullptr};
    size_t res_bytes = 0;
    bool res_lent[2] = {false, false};
    float res_tbps = 0.f;
};

static int fail(trk_ctx* ctx, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) {
        std::lock_guard<std::mutex> g(ctx->book_m);
        ctx->err = buf;
    } else {
        g_init_error = buf;
    }
    return code;
}

#define HIPCHK(ctx, expr)                                                                        \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) return fail(ctx, TRK_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

static hipEvent_t get_event(trk_ctx* ctx) {
    {
        std::lock_guard<std::mutex> g(ctx->book_m);
        if (!ctx->event_pool.empty()) {
            hipEvent_t e = ctx->event_pool.back();
            ctx->event_pool.pop_back();
            return e;
        }
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

struct ProfScope {
    trk_ctx* ctx;
    ProfRec rec;
    bool on;
    hipStream_t st = nullptr;
    ProfScope(trk_ctx* c, int kernel) : ctx(c), on(c->profiling) {
        if (!on) return;
        rec.kernel = kernel;
        rec.start = get_event(c);
        rec.stop = get_event(c);
        st = c->s();
        (void)hipEventRecord(rec.start, st);
    }
    ~ProfScope() {
        if (!on) return;
        (void)hipEventRecord(rec.stop, st);
        std::lock_guard<std::mutex> g(ctx->book_m);
        ctx->prof_pending.push_back(rec);
    }
    // ends the bracket of the first kernel of a launch sequence here and opens one for the next kernel
    void split(int kernel2) {
        if (!on) return;
        (void)hipEventRecord(rec.stop, st);
        {
            std::lock_guard<std::mutex> g(ctx->book_m);
            ctx->prof_pending.push_back(rec);
        }
        rec.kernel = kernel2;
        rec.start = get_event(ctx);
        rec.stop = get_event(ctx);
        (void)hipEventRecord(rec.start, st);
    }
};

static void drain_profile(trk_ctx* ctx) {
    std::vector<ProfRec> pend;
    {
        std::lock_guard<std::mutex> g(ctx->book_m);
        pend.swap(ctx->prof_pending);
    }
    std::vector<hipEvent_t> back;
    for (auto& r : pend) {
        (void)hipEventSynchronize(r.stop);
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.start, r.stop) == hipSuccess) {
            ctx->prof_n[r.kernel] += 1;
            ctx->prof_ms[r.kernel] += ms;
        }
        back.push_back(r.start);
        back.push_back(r.stop);
    }
    std::lock_guard<std::mutex> g(ctx->book_m);
    ctx->event_pool.insert(ctx->event_pool.end(), back.begin(), back.end());
}

extern "C" {

int trk_device_count(int* n) {
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) {
        *n = 0;
        return fail(nullptr, TRK_ERR_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *n = c;
    return TRK_OK;
}

static int ensure_queue(trk_ctx* ctx, int i) {
    if (ctx->streams[i]) return TRK_OK;
    std::lock_guard<std::mutex> g(ctx->queue_m);
    if (ctx->streams[i]) return TRK_OK;
    (void)hipSetDevice(ctx->device);
    hipStream_t q = nullptr;
    hipError_t e = hipStreamCreateWithFlags(&q, hipStreamNonBlocking);
    if (e == hipSuccess && !ctx->join_event[i]) e = hipEventCreateWithFlags(&ctx->join_event[i], hipEventDisableTiming);
    if (e != hipSuccess) {
        if (q) (void)hipStreamDestroy(q);
        return fail(ctx, TRK_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
    }
    ctx->streams[i] = q;
    return TRK_OK;
}

int trk_init(int device, trk_ctx** out) {
    if (!out) return fail(nullptr, TRK_ERR_ARG, "trk_init: out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(nullptr, TRK_ERR_HIP, "trk_init: no HIP device (%s)", hipGetErrorString(e));
    if (device < 0 || device >= n) return fail(nullptr, TRK_ERR_ARG, "trk_init: device %d out of range [0,%d)", device, n);
    e = hipSetDevice(device);
    if (e != hipSuccess) return fail(nullptr, TRK_ERR_HIP, "hipSetDevice: %s", hipGetErrorString(e));
    trk_ctx* ctx = new trk_ctx();
    ctx->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->n_cu = prop.multiProcessorCount;
    // queue 0 now, the others when they are first selected (trk_stream_select / trk_thread_queue / trk_stream_wait): a
    // queue costs ~10 ms to create, and a command line uses two of the four
    if (ensure_queue(ctx, 0) != TRK_OK) {
        std::string msg = ctx->err;
        delete ctx;
        return fail(nullptr, TRK_ERR_HIP, "%s", msg.c_str());
    }
    for (int i = 0; i < TRK_N_TIMERS; ++i) {
        (void)hipEventCreate(&ctx->t_start[i]);
        (void)hipEventCreate(&ctx->t_stop[i]);
    }
    for (int i = 0; i < TRK_N_EVENTS; ++i) (void)hipEventCreateWithFlags(&ctx->user_event[i], hipEventDisableTiming);
    *out = ctx;
    return TRK_OK;
}

void trk_free(trk_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    for (int i = 0; i < TRK_N_STREAMS; ++i)
        if (ctx->streams[i]) (void)hipStreamSynchronize(ctx->streams[i]);
    drain_profile(ctx);
    if (ctx->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(ctx->comm);
    for (int k = 0; k < 2; ++k)
        if (ctx->res_plane[k]) (void)hipFree(ctx->res_plane[k]);
    if (ctx->inflate) inflate_state_free(ctx->inflate);
    if (ctx->deflate) {
        DeflateState* d = ctx->deflate;
        for (uint8_t* p : {d->d_text, d->d_slots, d->d_out, d->d_tok, d->d_tab})
            if (p) (void)hipFree(p);
        if (d->h_off) (void)hipHostFree(d->h_off);
        if (d->q) (void)hipStreamDestroy(d->q);
        delete d;
    }
    for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
    for (int i = 0; i < TRK_N_TIMERS; ++i) {
        (void)hipEventDestroy(ctx->t_start[i]);
        (void)hipEventDestroy(ctx->t_stop[i]);
    }
    for (int i = 0; i < TRK_N_EVENTS; ++i)
        if (ctx->user_event[i]) (void)hipEventDestroy(ctx->user_event[i]);
    for (int i = 0; i < TRK_N_STREAMS; ++i) {
        if (ctx->scratch_[i]) (void)hipFree(ctx->scratch_[i]);
        if (ctx->worklist_[i]) (void)hipFree(ctx->worklist_[i]);
        if (ctx->cf_ws_[i]) (void)hipFree(ctx->cf_ws_[i]);
        if (ctx->join_event[i]) (void)hipEventDestroy(ctx->join_event[i]);
        if (ctx->streams[i]) (void)hipStreamDestroy(ctx->streams[i]);
    }
    for (int i = 0; i < TRK_N_STREAMS; ++i)
        if (ctx->assoc_ws_[i]) (void)hipFree(ctx->assoc_ws_[i]);
    delete ctx;
}

const char* trk_last_error(trk_ctx* ctx) { return ctx ? ctx->err.c_str() : g_init_error.c_str(); }

int trk_backend(trk_ctx*) { return 1; }

int trk_device_info(trk_ctx* ctx, char* name, int name_len, int* n_cu, uint64_t* hbm_bytes, char* arch,
                    int arch_len) {
    if (!ctx) return TRK_ERR_ARG;
    hipDeviceProp_t prop;
    HIPCHK(ctx, hipGetDeviceProperties(&prop, ctx->device));
    if (name && name_len > 0) snprintf(name, name_len, "%s", prop.name);
    if (arch && arch_len > 0) snprintf(arch, arch_len, "%s", prop.gcnArchName);
    if (n_cu) *n_cu = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (uint64_t)prop.totalGlobalMem;
    return TRK_OK;
}

// ---- memory ----------------------------------------------------------------------
int trk_dev_alloc(trk_ctx* ctx, size_t bytes, void** dptr) {
    if (!ctx || !dptr) return TRK_ERR_ARG;
    *dptr = nullptr;
    if (bytes == 0) bytes = 16;
    (void)hipSetDevice(ctx->device);
    hipError_t e = hipMalloc(dptr, bytes);
    if (e != hipSuccess) return fail(ctx, TRK_ERR_NOMEM, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
    return TRK_OK;
}
int trk_dev_free(trk_ctx* ctx, void* dptr) {
    if (!ctx) return TRK_ERR_ARG;
    if (!dptr) return TRK_OK;
    for (int i = 0; i < TRK_N_STREAMS; ++i)
        if (ctx->streams[i]) HIPCHK(ctx, hipStreamSynchronize(ctx->streams[i]));
    for (int k = 0; k < 2; ++k)
        if (dptr == ctx->res_plane[k]) {   // a plane of the reserved pair goes back to the context, not to the driver
            ctx->res_lent[k] = false;
            return TRK_OK;
        }
    HIPCHK(ctx, hipFree(dptr));
    return TRK_OK;
}
int trk_memcpy_h2d(trk_ctx* ctx, void* d, const void* h, size_t n) {
    if (!ctx) return TRK_ERR_ARG;
    if (n == 0) return TRK_OK;
    HIPCHK(ctx, hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, ctx->s()));
    HIPCHK(ctx, hipStreamSynchronize(ctx->s()));
    return TRK_OK;
}
int trk_memcpy_d2h(trk_ctx* ctx, void* h, const void* d, size_t n) {
    if (!ctx) return TRK_ERR_ARG;
    if (n == 0) return TRK_OK;
    HIPCHK(ctx, hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, ctx->s()));
    HIPCHK(ctx, hipStreamSynchronize(ctx->s()));
    return TRK_OK;
}
int trk_memcpy_d2d(trk_ctx* ctx, void* d, const void* s, size_t n) {
    if (!ctx) return TRK_ERR_ARG;
    if (n == 0) return TRK_OK;
    HIPCHK(ctx, hipMemcpyAsync(d, s, n, hipMemcpyDeviceToDevice, ctx->s()));
    return TRK_OK;
}
int trk_memset(trk_ctx* ctx, void* d, int v, size_t n) {
    if (!ctx) return TRK_ERR_ARG;
    if (n == 0) return TRK_OK;
    HIPCHK(ctx, hipMemsetAsync(d, v, n, ctx->s()));
    return TRK_OK;
}
int trk_sync(trk_ctx* ctx) {
    if (!ctx) return TRK_ERR_ARG;
    for (int i = 0; i < TRK_N_STREAMS; ++i)
        if (ctx->streams[i]) HIPCHK(ctx, hipStreamSynchronize(ctx->streams[i]));
    return TRK_OK;
}
int trk_host_alloc(trk_ctx* ctx, size_t bytes, void** hptr) {
    if (!ctx || !hptr) return TRK_ERR_ARG;
    *hptr = nullptr;
    if (bytes == 0) bytes = 16;
    (void)hipSetDevice(ctx->device);
    hipError_t e = hipHostMalloc(hptr, bytes, hipHostMallocDefault);
    if (e != hipSuccess) return fail(ctx, TRK_ERR_NOMEM, "hipHostMalloc(%zu): %s", bytes, hipGetErrorString(e));
    return TRK_OK;
}
int trk_host_free(trk_ctx* ctx, void* hptr) {
    if (!ctx) return TRK_ERR_ARG;
    if (!hptr) return TRK_OK;
    HIPCHK(ctx, hipHostFree(hptr));
    return TRK_OK;
}
int trk_memcpy_h2d_async(trk_ctx* ctx, void* d, const void* h, size_t n) {
    if (!ctx) return TRK_ERR_ARG;
    if (n == 0) return TRK_OK;
    HIPCHK(ctx, hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, ctx->s()));
    return TRK_OK;
}
int trk_memcpy_d2h_async(trk_ctx* ctx, void* h, const void* d, size_t n) {
    if (!ctx) return TRK_ERR_ARG;
    if (n == 0) return TRK_OK;
    HIPCHK(ctx, hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, ctx->s()));
    return TRK_OK;
}
int trk_queue_sync(trk_ctx* ctx, int queue) {
    if (!ctx) return TRK_ERR_ARG;
    if (queue < 0 || queue >= TRK_N_STREAMS) return fail(ctx, TRK_ERR_ARG, "queue %d outside [0, %d)", queue, TRK_N_STREAMS);
    if (ctx->streams[queue]) HIPCHK(ctx, hipStreamSynchronize(ctx->streams[queue]));
    return TRK_OK;
}
int trk_stream_select(trk_ctx* ctx, int queue) {
    if (!ctx) return TRK_ERR_ARG;
    if (queue < 0 || queue >= TRK_N_STREAMS) return fail(ctx, TRK_ERR_ARG, "queue %d outside [0, %d)", queue, TRK_N_STREAMS);
    if (const int rc = ensure_queue(ctx, queue)) return rc;
    ctx->cur = queue;
    return TRK_OK;
}
int trk_thread_queue(trk_ctx* ctx, int queue) {
    if (!ctx) return TRK_ERR_ARG;
    if (queue < -1 || queue >= TRK_N_STREAMS) return fail(ctx, TRK_ERR_ARG, "queue %d outside [-1, %d)", queue, TRK_N_STREAMS);
    if (queue >= 0)
        if (const int rc = ensure_queue(ctx, queue)) return rc;
    t_queue = queue;
    return TRK_OK;
}
int trk_stream_wait(trk_ctx* ctx, int waiter, int signal) {
    if (!ctx) return TRK_ERR_ARG;
    if (waiter < 0 || waiter >= TRK_N_STREAMS || signal < 0 || signal >= TRK_N_STREAMS)
        return fail(ctx, TRK_ERR_ARG, "queue outside [0, %d)", TRK_N_STREAMS);
    if (waiter == signal) return TRK_OK;
    (void)hipSetDevice(ctx->device);
    if (!ctx->streams[signal]) return TRK_OK;            // nothing was ever put on that queue
    if (const int rc = ensure_queue(ctx, waiter)) return rc;
    HIPCHK(ctx, hipEventRecord(ctx->join_event[signal], ctx->streams[signal]));
    HIPCHK(ctx, hipStreamWaitEvent(ctx->streams[waiter], ctx->join_event[signal], 0));
    return TRK_OK;
}

int trk_event_record(trk_ctx* ctx, int slot) {
    if (!ctx) return TRK_ERR_ARG;
    if (slot < 0 || slot >= TRK_N_EVENTS) return fail(ctx, TRK_ERR_ARG, "event slot %d outside [0, %d)", slot, TRK_N_EVENTS);
    HIPCHK(ctx, hipEventRecord(ctx->user_event[slot], ctx->s()));
    ctx->user_event_set[slot] = true;
    return TRK_OK;
}
int trk_event_wait(trk_ctx* ctx, int slot) {
    if (!ctx) return TRK_ERR_ARG;
    if (slot < 0 || slot >= TRK_N_EVENTS) return fail(ctx, TRK_ERR_ARG, "event slot %d outside [0, %d)", slot, TRK_N_EVENTS);
    if (!ctx->user_event_set[slot]) return TRK_OK;
    HIPCHK(ctx, hipStreamWaitEvent(ctx->s(), ctx->user_event[slot], 0));
    return TRK_OK;
}

// ---- timers / profiling ----------------------------------------------------------
int trk_timer_start(trk_ctx* ctx, int slot) {
    if (!ctx || slot < 0 || slot >= TRK_N_TIMERS) return TRK_ERR_ARG;
    HIPCHK(ctx, hipEventRecord(ctx->t_start[slot], ctx->s()));
    return TRK_OK;
}
int trk_timer_stop(trk_ctx* ctx, int slot) {
    if (!ctx || slot < 0 || slot >= TRK_N_TIMERS) return TRK_ERR_ARG;
    HIPCHK(ctx, hipEventRecord(ctx->t_stop[slot], ctx->s()));
    return TRK_OK;
}
int trk_timer_elapsed_ms(trk_ctx* ctx, int slot, float* ms) {
    if (!ctx || slot < 0 || slot >= TRK_N_TIMERS || !ms) return TRK_ERR_ARG;
    HIPCHK(ctx, hipEventSynchronize(ctx->t_stop[slot]));
    HIPCHK(ctx, hipEventElapsedTime(ms, ctx->t_start[slot], ctx->t_stop[slot]));
    return TRK_OK;
}
int trk_profile_enable(trk_ctx* ctx, int on) {
    if (!ctx) return TRK_ERR_ARG;
    ctx->profiling = on != 0;
    return TRK_OK;
}
int trk_profile_get(trk_ctx* ctx, int kernel, int64_t* n, double* ms) {
    if (!ctx || kernel < 0 || kernel >= TRK_K_COUNT) return TRK_ERR_ARG;
    drain_profile(ctx);
    if (n) *n = ctx->prof_n[kernel];
    if (ms) *ms = ctx->prof_ms[kernel];
    return TRK_OK;
}
int trk_profile_reset(trk_ctx* ctx) {
    if (!ctx) return TRK_ERR_ARG;
    drain_profile(ctx);
    for (int i = 0; i < TRK_K_COUNT; ++i) {
        ctx->prof_n[i] = 0;
        ctx->prof_ms[i] = 0.0;
    }
    return TRK_OK;
}

// ---- the hot path ----------------------------------------------------------------
// Device scratch of the selected queue (partial counters of the call-filter and qc passes), grown on demand: a growth
// synchronises that queue, so it happens before the launch and outside the profiling bracket.
static void* workspace_for_queue(trk_ctx* c, size_t bytes) {
    const int q = c->cur;
    if (bytes > c->cf_ws_bytes_[q]) {
        if (hipStreamSynchronize(c->streams[q]) != hipSuccess) return nullptr;
        if (c->cf_ws_[q]) (void)hipFree(c->cf_ws_[q]);
        c->cf_ws_[q] = nullptr;
        c->cf_ws_bytes_[q] = 0;
        const size_t want = bytes + bytes / 4;
        if (hipMalloc(&c->cf_ws_[q], want) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        c->cf_ws_bytes_[q] = want;
    }
    return c->cf_ws_[q];
}

static int check_batch(trk_ctx* ctx, const trk_batch* b, bool count_entry = false) {
    if (!b) return fail(ctx, TRK_ERR_ARG, "batch is NULL");
    if (b->row_stride != 0 && (!count_entry || b->row_stride < b->n_samples || (b->row_stride & 3)))
        return fail(ctx, TRK_ERR_ARG, "row_stride %d: column-range views are for trk_locus_stats, rows a multiple of "
                                      "four samples apart and at least n_samples long", b->row_stride);
    if (b->n_class_runs < 0 || (b->n_class_runs > 0 && (!b->class_runs || !b->group_bits)))
        return fail(ctx, TRK_ERR_ARG, "class_runs needs the run table and group_bits");
    if (b->n_class_runs > 0 && !count_entry)   // class-ordered columns are not the cohort's sample order
        return fail(ctx, TRK_ERR_ARG, "a class-ordered batch (class_runs) is for trk_locus_stats only");
    if (b->n_loci < 0 || b->n_samples < 0) return fail(ctx, TRK_ERR_ARG, "negative batch dimensions");
    if (b->ploidy < 1 || b->ploidy > TRK_MAX_PLOIDY)
        return fail(ctx, TRK_ERR_ARG, "ploidy %d outside [1,%d]", b->ploidy, TRK_MAX_PLOIDY);
    if (b->n_loci > 0 && (!b->gt && b->n_samples > 0)) return fail(ctx, TRK_ERR_ARG, "gt is NULL");
    if (b->n_loci > 0 && (!b->allele_off || !b->len_class || !b->str_class || !b->len_class_value))
        return fail(ctx, TRK_ERR_ARG, "allele tables are NULL");
    if (b->group_bits && (b->n_groups < 1 || b->n_groups > 8))
        return fail(ctx, TRK_ERR_ARG, "n_groups %d outside [1,8]", b->n_groups);
    if (((uintptr_t)b->gt & 15u) != 0) return fail(ctx, TRK_ERR_ARG, "gt must be 16-byte aligned");
    if (b->n_pad_samples < 0 || (b->n_pad_samples > 0 && b->n_pad_samples >= b->n_samples))
        return fail(ctx, TRK_ERR_ARG, "n_pad_samples %d outside [0, n_samples)", b->n_pad_samples);
    return TRK_OK;
}

static int ensure_fin_buffers(trk_ctx* ctx, int G, int64_t sumA, int n_loci);

int trk_locus_stats(trk_ctx* ctx, const trk_batch* in, const trk_stats_params* prm, trk_stats_out* out) {
    if (!ctx) return TRK_ERR_ARG;
    int rc = check_batch(ctx, in, true);
    if (rc) return rc;
    if (!out || !out->allele_count || !out->locus_int) return fail(ctx, TRK_ERR_ARG, "stats outputs are NULL");
    const bool count_only = prm && (prm->flags & TRK_STATS_COUNT_ONLY);
    if (!count_only && !out->locus_f64) return fail(ctx, TRK_ERR_ARG, "locus_f64 is NULL");
    if (in->n_loci == 0) return TRK_OK;
    (void)hipSetDevice(ctx->device);
    const int G = in->group_bits ? in->n_groups : 1;
    const int64_t sumA = in->n_alleles_total;
    int32_t* class_ws = nullptr;
    if (in->n_class_runs > 0)   // (grown before the bracket: a growth synchronises this queue; null -> per-call kernels)
        class_ws = static_cast<int32_t*>(workspace_for_queue(
            ctx, (size_t)in->n_class_runs * ((size_t)sumA + (size_t)in->n_loci * TRK_LI_COLS) * sizeof(int32_t)));
    if (!count_only && !(prm && (prm->flags & TRK_STATS_TWIN))) {
        // small batches: the finaliser is the count kernel's epilogue (two launches per pass instead of five)
        rc = ensure_fin_buffers(ctx, G, sumA, in->n_loci);
        if (rc) return rc;
        if (trk::launch_locus_stats_fused(*in, nullptr, nullptr, nullptr, nullptr, 0.0, nullptr, nullptr, 0)) {
            const double thr = prm ? prm->nalleles_thresh : 0.01;
            ProfScope ps(ctx, TRK_K_LOCUS_COUNT);
            hipError_t fe = hipSuccess;
            (void)trk::launch_locus_stats_fused(*in, out->allele_count, out->locus_int, out->locus_f64,
                                                ctx->worklist_[ctx->cur], thr, ctx->s(), &fe, 1);
            HIPCHK(ctx, fe);
            ps.split(TRK_K_LOCUS_FINALIZE);
            (void)trk::launch_locus_stats_fused(*in, out->allele_count, out->locus_int, out->locus_f64,
                                                ctx->worklist_[ctx->cur], thr, ctx->s(), &fe, 2);
            HIPCHK(ctx, fe);
            return TRK_OK;
        }
    }
    {
        ProfScope ps(ctx, TRK_K_LOCUS_COUNT);
        HIPCHK(ctx, trk::launch_locus_count(*in, in->max_alleles, out->allele_count, out->locus_int,
                                            ctx->n_cu, ctx->s(), prm && (prm->flags & TRK_STATS_TWIN), class_ws));
    }
    if (count_only) return TRK_OK;
    rc = ensure_fin_buffers(ctx, G, sumA, in->n_loci);
    if (rc) return rc;
    {
        ProfScope ps(ctx, TRK_K_LOCUS_FINALIZE);
        HIPCHK(ctx, trk::launch_locus_finalize(*in, out->allele_count, out->locus_int, out->locus_f64, ctx->scratch_[ctx->cur],
                                               ctx->worklist_[ctx->cur], prm ? prm->nalleles_thresh : 0.01, ctx->s()));
    }
    return TRK_OK;
}

static int ensure_fin_buffers(trk_ctx* ctx, int G, int64_t sumA, int n_loci) {
    size_t need = (size_t)G * 2 * (size_t)sumA * sizeof(int32_t) + 16;
    if (need > ctx->scratch_bytes_[ctx->cur]) {
        HIPCHK(ctx, hipStreamSynchronize(ctx->s()));
        if (ctx->scratch_[ctx->cur]) (void)hipFree(ctx->scratch_[ctx->cur]);
        ctx->scratch_[ctx->cur] = nullptr;
        ctx->scratch_bytes_[ctx->cur] = 0;
        hipError_t e = hipMalloc((void**)&ctx->scratch_[ctx->cur], need);
        if (e != hipSuccess) return fail(ctx, TRK_ERR_NOMEM, "scratch hipMalloc(%zu): %s", need, hipGetErrorString(e));
        ctx->scratch_bytes_[ctx->cur] = need;
    }
    size_t wneed = trk::finalize_worklist_bytes((int64_t)G * n_loci);
    if (wneed > ctx->worklist_bytes_[ctx->cur]) {
        HIPCHK(ctx, hipStreamSynchronize(ctx->s()));
        if (ctx->worklist_[ctx->cur]) (void)hipFree(ctx->worklist_[ctx->cur]);
        ctx->worklist_[ctx->cur] = nullptr;
        ctx->worklist_bytes_[ctx->cur] = 0;
        hipError_t e = hipMalloc(&ctx->worklist_[ctx->cur], wneed);
        if (e != hipSuccess) return fail(ctx, TRK_ERR_NOMEM, "worklist hipMalloc(%zu): %s", wneed, hipGetErrorString(e));
        ctx->worklist_bytes_[ctx->cur] = wneed;
    }
    return TRK_OK;
}

int trk_locus_finalize(trk_ctx* ctx, const trk_batch* in, const trk_stats_params* prm, trk_stats_out* out) {
    if (!ctx) return TRK_ERR_ARG;
    int rc = check_batch(ctx, in, true);   // (reads the allele tables and the counts only: any column layout)
    if (rc) return rc;
    if (!out || !out->allele_count || !out->locus_int || !out->locus_f64)
        return fail(ctx, TRK_ERR_ARG, "stats outputs are NULL");
    if (in->n_loci == 0) return TRK_OK;
    (void)hipSetDevice(ctx->device);
    const int G = in->group_bits ? in->n_groups : 1;
    rc = ensure_fin_buffers(ctx, G, in->n_alleles_total, in->n_loci);
    if (rc) return rc;
    ProfScope ps(ctx, TRK_K_LOCUS_FINALIZE);
    HIPCHK(ctx, trk::launch_locus_finalize(*in, out->allele_count, out->locus_int, out->locus_f64, ctx->scratch_[ctx->cur],
                                           ctx->worklist_[ctx->cur], prm ? prm->nalleles_thresh : 0.01, ctx->s()));
    return TRK_OK;
}

int trk_call_filters(trk_ctx* ctx, const trk_batch* in, const trk_plane* planes, int n_planes,
                     const trk_call_filter* filters, int n_filters, int dp_plane, trk_call_out* out) {
    if (!ctx) return TRK_ERR_ARG;
    int rc = check_batch(ctx, in);
    if (rc) return rc;
    if (n_planes < 0 || n_planes > TRK_MAX_PLANES) return fail(ctx, TRK_ERR_ARG, "n_planes %d > %d", n_planes, TRK_MAX_PLANES);
    if (n_filters < 0 || n_filters > TRK_MAX_FILTERS)
        return fail(ctx, TRK_ERR_ARG, "n_filters %d > %d", n_filters, TRK_MAX_FILTERS);
    if (dp_plane >= 0 && dp_plane < n_planes && planes && (planes[dp_plane].dtype & 0xff) == TRK_DT_F32 && out &&
        !out->sample_totaldp_f64)
        return fail(ctx, TRK_ERR_ARG, "a Float depth plane needs sample_totaldp_f64");
    if (!out || !out->sample_counters || !out->sample_totaldp || !out->sample_dp_missing || !out->error)
        return fail(ctx, TRK_ERR_ARG, "call-filter outputs are NULL");
    if (dp_plane >= n_planes) return fail(ctx, TRK_ERR_ARG, "dp_plane out of range");
    if ((out->delta_allele_count == nullptr) != (out->delta_locus_int == nullptr))
        return fail(ctx, TRK_ERR_ARG, "delta_allele_count and delta_locus_int must be given together");
    if (out->delta_allele_count && in->group_bits)
        return fail(ctx, TRK_ERR_ARG, "delta outputs are defined for ungrouped batches only");
    for (int i = 0; i < n_planes; ++i) {
        if (!planes[i].data || planes[i].ncol < 1) return fail(ctx, TRK_ERR_ARG, "plane %d is empty", i);
        if ((planes[i].dtype & ~TRK_DT_PLANAR) != TRK_DT_I32 && (planes[i].dtype & ~TRK_DT_PLANAR) != TRK_DT_F32)
            return fail(ctx, TRK_ERR_ARG, "plane %d has unknown dtype", i);
    }
    for (int k = 0; k < n_filters; ++k) {
        const trk_call_filter& f = filters[k];
        if (f.op < TRK_F_LT || f.op > TRK_F_AD_SUPPORT_LT) return fail(ctx, TRK_ERR_ARG, "filter %d: unknown op %d", k, f.op);
        if (f.plane_a < 0 || f.plane_a >= n_planes) return fail(ctx, TRK_ERR_ARG, "filter %d: plane_a out of range", k);
        const bool needs_b = f.op == TRK_F_RATIO_GT || f.op == TRK_F_CALLED_EQ || f.op == TRK_F_CALLED_SUM_EQ ||
                             f.op == TRK_F_CALLED_OUTSIDE_CI;
        if (needs_b && (f.plane_b < 0 || f.plane_b >= n_planes))
            return fail(ctx, TRK_ERR_ARG, "filter %d: plane_b out of range", k);
        if (f.op != TRK_F_CALLED_OUTSIDE_CI && f.op != TRK_F_AD_SUPPORT_LT &&
            (f.col_a < 0 || f.col_a >= planes[f.plane_a].ncol))
            return fail(ctx, TRK_ERR_ARG, "filter %d: col_a out of range", k);
        if ((f.op == TRK_F_CALLED_SUM_LT || f.op == TRK_F_CALLED_SUM_EQ) &&
            (f.col_a2 < 0 || f.col_a2 >= planes[f.plane_a].ncol))
            return fail(ctx, TRK_ERR_ARG, "filter %d: col_a2 out of range", k);
        if (needs_b && f.op != TRK_F_CALLED_OUTSIDE_CI && (f.col_b < 0 || f.col_b >= planes[f.plane_b].ncol))
            return fail(ctx, TRK_ERR_ARG, "filter %d: col_b out of range", k);
        if (f.op == TRK_F_CALLED_OUTSIDE_CI && planes[f.plane_b].ncol < 2 * planes[f.plane_a].ncol)
            return fail(ctx, TRK_ERR_ARG, "filter %d: REPCI plane needs 2 columns per REPCN column", k);
        if ((f.op == TRK_F_CALLED_EQ || f.op == TRK_F_CALLED_SUM_EQ || f.op == TRK_F_CALLED_OUTSIDE_CI ||
             f.op == TRK_F_AD_SUPPORT_LT) &&
            (planes[f.plane_a].dtype & 0xff) != TRK_DT_I32)
            return fail(ctx, TRK_ERR_ARG, "filter %d: integer plane required", k);
    }
    if (in->n_loci == 0 || in->n_samples == 0) return TRK_OK;
    (void)hipSetDevice(ctx->device);
    // grown outside the profiling bracket and before the launch (a growth synchronises this queue)
    // TRK_K_CALL_FILTER brackets the streaming kernel alone; the reduction of the per-workgroup partial counters that
    // follows it (k_cf_reduce) is TRK_K_CF_REDUCE
    ProfScope ps(ctx, TRK_K_CALL_FILTER);
    struct Hook { trk_ctx* ctx; ProfScope* ps; } hook{ctx, &ps};
    trk::Scratch sc{&hook,
                    [](void* user, size_t bytes) -> void* {
                        return workspace_for_queue(static_cast<Hook*>(user)->ctx, bytes);   // null: atomics instead
                    },
                    [](void* user) { static_cast<Hook*>(user)->ps->split(TRK_K_CF_REDUCE); }};
    HIPCHK(ctx, trk::launch_call_filter(*in, planes, n_planes, filters, n_filters, dp_plane, *out, ctx->n_cu,
                                        ctx->s(), sc));
    return TRK_OK;
}

int trk_locus_filters(trk_ctx* ctx, int32_t n_loci, const trk_stats_out* stats, const trk_locus_filter_spec* spec,
                      trk_locus_out* out) {
    if (!ctx) return TRK_ERR_ARG;
    if (!stats || !spec || !out || !out->locus_bits || !out->loc_counters || !stats->locus_int || !stats->locus_f64)
        return fail(ctx, TRK_ERR_ARG, "locus-filter arguments are NULL");
    if (spec->n_extern < 0 || spec->n_extern > 24) return fail(ctx, TRK_ERR_ARG, "n_extern outside [0,24]");
    if (n_loci <= 0) return TRK_OK;
    (void)hipSetDevice(ctx->device);
    ProfScope ps(ctx, TRK_K_LOCUS_FILTER);
    HIPCHK(ctx, trk::launch_locus_filter(n_loci, stats->locus_int, stats->locus_f64, *spec, out->locus_bits,
                                         out->loc_counters, ctx->s()));
    return TRK_OK;
}

int trk_synth_fill(trk_ctx* ctx, const trk_synth_spec* spec, int16_t* gt, int32_t* dp, float* q, int32_t* dstutter,
                   int32_t* dflankindel) {
    if (!ctx || !spec || !gt) return fail(ctx, TRK_ERR_ARG, "synth arguments are NULL");
    (void)hipSetDevice(ctx->device);
    ProfScope ps(ctx, TRK_K_SYNTH);
    HIPCHK(ctx, trk::launch_synth(*spec, gt, dp, q, dstutter, dflankindel, ctx->n_cu, ctx->s()));
    return TRK_OK;
}

int trk_synth_fill_gangstr(trk_ctx* ctx, const trk_synth_spec* spec, const int16_t* gt, const int32_t* dp,
                           const int32_t* allele_repcn, float* qexp, int32_t* repcn, int32_t* rc, int32_t* repci) {
    if (!ctx || !spec || !gt || !dp || !allele_repcn || !qexp || !repcn || !rc || !repci)
        return fail(ctx, TRK_ERR_ARG, "synth (gangstr) arguments are NULL");
    (void)hipSetDevice(ctx->device);
    ProfScope ps(ctx, TRK_K_SYNTH);
    HIPCHK(ctx, trk::launch_synth_gangstr(*spec, gt, dp, allele_repcn, qexp, repcn, rc, repci, ctx->n_cu, ctx->s()));
    return TRK_OK;
}

// ---- multi-GPU -------------------------------------------------------------------
int trk_comm_unique_id(uint8_t id[128]) {
    std::string err;
    if (!load_rccl(err)) return fail(nullptr, TRK_ERR_RCCL, "%s", err.c_str());
    ncclUniqueId uid;
    ncclResult_t r = g_rccl.GetUniqueId(&uid);
    if (r != 0) return fail(nullptr, TRK_ERR_RCCL, "ncclGetUniqueId failed (%d)", r);
    memcpy(id, uid.internal, 128);
    return TRK_OK;
}

int trk_comm_init(trk_ctx* ctx, int rank, int n_ranks, const uint8_t id[128]) {
    if (!ctx) return TRK_ERR_ARG;
    std::string err;
    if (!load_rccl(err)) return fail(ctx, TRK_ERR_RCCL, "%s", err.c_str());
    (void)hipSetDevice(ctx->device);
    ncclUniqueId uid;
    memcpy(uid.internal, id, 128);
    ncclResult_t r = g_rccl.CommInitRank(&ctx->comm, n_ranks, uid, rank);
    if (r != 0)
        return fail(ctx, TRK_ERR_RCCL, "ncclCommInitRank: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
    ctx->rank = rank;
    ctx->n_ranks = n_ranks;
    return TRK_OK;
}

int trk_allreduce_sum_i64(trk_ctx* ctx, int64_t* dev, size_t count) {
    if (!ctx) return TRK_ERR_ARG;
    if (ctx->n_ranks <= 1 && !ctx->comm) return TRK_OK;
    if (!ctx->comm) return fail(ctx, TRK_ERR_RCCL, "communicator not initialised");
    ncclResult_t r = g_rccl.AllReduce(dev, dev, count, trkNcclInt64, trkNcclSum, ctx->comm, ctx->s());
    if (r != 0) return fail(ctx, TRK_ERR_RCCL, "ncclAllReduce: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
    return TRK_OK;
}

int trk_allgather(trk_ctx* ctx, const void* send, void* recv, size_t bytes_per_rank) {
    if (!ctx) return TRK_ERR_ARG;
    if (ctx->n_ranks <= 1 && !ctx->comm) {
        if (send != recv) HIPCHK(ctx, hipMemcpyAsync(recv, send, bytes_per_rank, hipMemcpyDeviceToDevice, ctx->s()));
        return TRK_OK;
    }
    if (!ctx->comm) return fail(ctx, TRK_ERR_RCCL, "communicator not initialised");
    ncclResult_t r = g_rccl.AllGather(send, recv, bytes_per_rank, trkNcclUint8, ctx->comm, ctx->s());
    if (r != 0) return fail(ctx, TRK_ERR_RCCL, "ncclAllGather: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
    return TRK_OK;
}

int trk_exchange(trk_ctx* ctx, int64_t* sums, size_t n_sums, const void* send, void* recv, size_t bytes_per_rank) {
    if (!ctx) return TRK_ERR_ARG;
    const bool do_sum = sums && n_sums > 0, do_gather = send && recv && bytes_per_rank > 0;
    if (ctx->n_ranks <= 1 && !ctx->comm) {
        if (do_gather && send != recv)
            HIPCHK(ctx, hipMemcpyAsync(recv, send, bytes_per_rank, hipMemcpyDeviceToDevice, ctx->s()));
        return TRK_OK;
    }
    if (!ctx->comm) return fail(ctx, TRK_ERR_RCCL, "communicator not initialised");
    const bool group = do_sum && do_gather && g_rccl.GroupStart && g_rccl.GroupEnd;
    ncclResult_t r = 0;
    if (group) r = g_rccl.GroupStart();
    if (r == 0 && do_sum) r = g_rccl.AllReduce(sums, sums, n_sums, trkNcclInt64, trkNcclSum, ctx->comm, ctx->s());
    if (r == 0 && do_gather) r = g_rccl.AllGather(send, recv, bytes_per_rank, trkNcclUint8, ctx->comm, ctx->s());
    if (group) {
        const ncclResult_t r2 = g_rccl.GroupEnd();
        if (r == 0) r = r2;
    }
    if (r != 0) return fail(ctx, TRK_ERR_RCCL, "trk_exchange: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
    return TRK_OK;
}

// ---- scalar helpers --------------------------------------------------------------
// ---- associaTR scan -----------------------------------------------------------------------
int trk_assoc_scan(trk_ctx* ctx, const trk_batch* in, const trk_assoc_params* prm, trk_assoc_out* out) {
    if (!ctx) return TRK_ERR_ARG;
    int rc = check_batch(ctx, in);
    if (rc) return rc;
    if (!prm || !out) return fail(ctx, TRK_ERR_ARG, "assoc params/outputs are NULL");
    if (prm->n_vec < 1 || prm->n_vec > TRK_ASSOC_MAX_VEC_WIDE)
        return fail(ctx, TRK_ERR_ARG, "n_vec %d outside [1,%d]", prm->n_vec, TRK_ASSOC_MAX_VEC_WIDE);
    if (in->group_bits) return fail(ctx, TRK_ERR_ARG, "sample groups are not used by the association scan");
    if (in->n_loci == 0) return TRK_OK;
    if (!prm->vec || !prm->allele_len || !prm->rlen_class) return fail(ctx, TRK_ERR_ARG, "assoc inputs are NULL");
    if (!out->locus_int || !out->locus_f64 || !out->allele_count)
        return fail(ctx, TRK_ERR_ARG, "assoc outputs are NULL");
    (void)hipSetDevice(ctx->device);
    const size_t need = trk::assoc_workspace_bytes(*in, prm->n_vec);
    if (need > ctx->assoc_ws_bytes_[ctx->cur]) {
        HIPCHK(ctx, hipStreamSynchronize(ctx->s()));
        if (ctx->assoc_ws_[ctx->cur]) (void)hipFree(ctx->assoc_ws_[ctx->cur]);
        ctx->assoc_ws_[ctx->cur] = nullptr;
        ctx->assoc_ws_bytes_[ctx->cur] = 0;
        hipError_t e = hipMalloc(&ctx->assoc_ws_[ctx->cur], need);
        if (e != hipSuccess) return fail(ctx, TRK_ERR_NOMEM, "assoc workspace hipMalloc(%zu): %s", need, hipGetErrorString(e));
        ctx->assoc_ws_bytes_[ctx->cur] = need;
    }
    HIPCHK(ctx, trk::launch_assoc_prepare(*in, *prm, *out, ctx->assoc_ws_[ctx->cur], ctx->s()));
    {
        ProfScope ps(ctx, TRK_K_ASSOC_SCAN);
        HIPCHK(ctx, trk::launch_assoc_scan(*in, *prm, *out, ctx->assoc_ws_[ctx->cur], ctx->n_cu, ctx->s()));
    }
    {
        ProfScope ps(ctx, TRK_K_ASSOC_FINALIZE);
        HIPCHK(ctx, trk::launch_assoc_finalize(*in, *prm, *out, ctx->assoc_ws_[ctx->cur], ctx->s()));
    }
    return TRK_OK;
}

int trk_assoc_scan_dosage(trk_ctx* ctx, const trk_batch* in, const trk_assoc_params* prm, const trk_assoc_dosage* dos,
                          trk_assoc_out* out, double* class_sums, double* locus_sums) {
    if (!ctx) return TRK_ERR_ARG;
    int rc = check_batch(ctx, in);
    if (rc) return rc;
    if (!prm || !out || !dos) return fail(ctx, TRK_ERR_ARG, "assoc params/outputs are NULL");
    if (prm->n_vec < 1 || prm->n_vec > TRK_ASSOC_MAX_VEC_WIDE)
        return fail(ctx, TRK_ERR_ARG, "n_vec %d outside [1,%d]", prm->n_vec, TRK_ASSOC_MAX_VEC_WIDE);
    if (in->n_loci == 0) return TRK_OK;
    if (!prm->vec || !prm->allele_len || !prm->rlen_class) return fail(ctx, TRK_ERR_ARG, "assoc inputs are NULL");
    if (!dos->ap1 || !dos->ap2 || !dos->perm || !dos->dclass || !dos->dclass_value || !dos->best_class)
        return fail(ctx, TRK_ERR_ARG, "dosage inputs are NULL");
    if (!out->locus_int || !out->locus_f64 || !out->allele_count || !class_sums || !locus_sums)
        return fail(ctx, TRK_ERR_ARG, "assoc outputs are NULL");
    if (in->ploidy < 1) return fail(ctx, TRK_ERR_ARG, "ploidy");
    (void)hipSetDevice(ctx->device);
    trk_batch bb = *in;
    bb.max_alleles = 0;
    const size_t need = trk::assoc_workspace_bytes(bb, prm->n_vec);
    if (need > ctx->assoc_ws_bytes_[ctx->cur]) {
        HIPCHK(ctx, hipStreamSynchronize(ctx->s()));
        if (ctx->assoc_ws_[ctx->cur]) (void)hipFree(ctx->assoc_ws_[ctx->cur]);
        ctx->assoc_ws_[ctx->cur] = nullptr;
        ctx->assoc_ws_bytes_[ctx->cur] = 0;
        hipError_t e = hipMalloc(&ctx->assoc_ws_[ctx->cur], need);
        if (e != hipSuccess) return fail(ctx, TRK_ERR_NOMEM, "assoc workspace hipMalloc(%zu): %s", need, hipGetErrorString(e));
        ctx->assoc_ws_bytes_[ctx->cur] = need;
    }
    ProfScope ps(ctx, TRK_K_ASSOC_SCAN);
    HIPCHK(ctx, trk::launch_assoc_dosage(*in, *prm, *dos, *out, class_sums, locus_sums, ctx->assoc_ws_[ctx->cur], ctx->s()));
    return TRK_OK;
}

int trk_dosages(trk_ctx* ctx, const trk_batch* in, const double* allele_len, int dosage_type, const float* ap1,
                const float* ap2, int n_alt_cols, float* out, int32_t* locus_err) {
    if (!ctx) return TRK_ERR_ARG;
    int rc = check_batch(ctx, in);
    if (rc) return rc;
    if (dosage_type < 0 || dosage_type > 3) return fail(ctx, TRK_ERR_ARG, "dosage type %d", dosage_type);
    if (in->n_loci == 0) return TRK_OK;
    const bool beagle = dosage_type == TRK_DOS_BEAGLEAP || dosage_type == TRK_DOS_BEAGLEAP_NORM;
    if (!allele_len || !out || !locus_err || (beagle && (!ap1 || !ap2)))
        return fail(ctx, TRK_ERR_ARG, "dosage inputs/outputs are NULL");
    if (in->n_loci > 65535) return fail(ctx, TRK_ERR_ARG, "at most 65535 loci per dosage call");
    (void)hipSetDevice(ctx->device);
    HIPCHK(ctx, trk::launch_dosages(*in, allele_len, dosage_type, ap1, ap2, n_alt_cols, out, locus_err, ctx->s()));
    return TRK_OK;
}

int trk_qc_reduce(trk_ctx* ctx, const trk_batch* in, const trk_qc_params* prm, trk_qc_out* out) {
    if (!ctx) return TRK_ERR_ARG;
    int rc = check_batch(ctx, in);
    if (rc) return rc;
    if (!prm || !out || !out->sample_calls || !out->locus_calls)
        return fail(ctx, TRK_ERR_ARG, "qc reduce: parameters / call-count outputs are NULL");
    if (prm->quality && (!out->sample_qual_sum || !out->locus_qual_sum))
        return fail(ctx, TRK_ERR_ARG, "qc reduce: a quality plane needs sample_qual_sum and locus_qual_sum");
    (void)hipSetDevice(ctx->device);
    if (in->n_loci == 0 || in->n_samples == 0) {
        if (in->n_samples > 0) {
            HIPCHK(ctx, hipMemsetAsync(out->sample_calls, 0, (size_t)in->n_samples * 8, ctx->s()));
            if (prm->quality) {
                HIPCHK(ctx, hipMemsetAsync(out->sample_qual_sum, 0, (size_t)in->n_samples * 8, ctx->s()));
                if (out->sample_qual_n)
                    HIPCHK(ctx, hipMemsetAsync(out->sample_qual_n, 0, (size_t)in->n_samples * 8, ctx->s()));
            }
        }
        return TRK_OK;
    }
    void* ws = workspace_for_queue(ctx, trk::qc_workspace_bytes(*in, prm->quality, ctx->n_cu));
    if (!ws) return fail(ctx, TRK_ERR_HIP, "qc reduce: workspace allocation failed");
    HIPCHK(ctx, trk::launch_qc_reduce(*in, *prm, *out, ws, ctx->n_cu, ctx->s()));
    return TRK_OK;
}

int trk_parse_samples(trk_ctx* ctx, const trk_parse_in* in, trk_parse_out* out) {
    if (!ctx) return TRK_ERR_ARG;
    if (!in || !out) return fail(ctx, TRK_ERR_ARG, "parse_samples: arguments are NULL");
    if (in->n_records < 0 || in->n_samples < 0 || in->ploidy < 1 || in->n_planes < 0 || in->n_planes > TRK_PARSE_MAX_PLANES)
        return fail(ctx, TRK_ERR_ARG, "parse_samples: records %d, samples %d, ploidy %d, planes %d", in->n_records, in->n_samples,
                    in->ploidy, in->n_planes);
    if (in->n_records == 0 || in->n_samples == 0) return TRK_OK;
    if (!in->text || ((uintptr_t)in->text & 15u) || !in->smp_off || !in->line_end || !out->gt || !out->locus_ploidy || !out->flags)
        return fail(ctx, TRK_ERR_ARG, "parse_samples: text (16-byte aligned), offsets, gt, locus_ploidy and flags are required");
    for (int i = 0; i < in->n_planes; ++i) {
        if (!in->plane_idx[i] || !out->planes[i]) return fail(ctx, TRK_ERR_ARG, "parse_samples: plane %d is NULL", i);
        if (in->plane_kind[i] != TRK_PARSE_INT && in->plane_kind[i] != TRK_PARSE_FLOAT)
            return fail(ctx, TRK_ERR_ARG, "parse_samples: plane %d kind %d", i, in->plane_kind[i]);
    }
    (void)hipSetDevice(ctx->device);
    ProfScope ps(ctx, TRK_K_SYNTH);      // (the "generate the inputs" slot of the profile)
    HIPCHK(ctx, trk::launch_parse_samples(*in, *out, ctx->s()));
    return TRK_OK;
}

int trk_inflate_blocks(trk_ctx* ctx, const trk_inflate_in* in, const trk_inflate_out* out) {
    if (!ctx) return TRK_ERR_ARG;
    if (!in || !out) return fail(ctx, TRK_ERR_ARG, "inflate_blocks: arguments are NULL");
    if (in->n_blocks < 0 || in->n_comp_bytes < 0) return fail(ctx, TRK_ERR_ARG, "inflate_blocks: %d blocks, %lld bytes", in->n_blocks, (long long)in->n_comp_bytes);
    if (in->n_blocks == 0) return TRK_OK;
    if (!in->comp || !in->in_off || !in->in_len || !in->out_off || !in->out_len || !out->text || !out->flags)
        return fail(ctx, TRK_ERR_ARG, "inflate_blocks: comp, the four block tables, text and flags are required");
    (void)hipSetDevice(ctx->device);
    ProfScope ps(ctx, TRK_K_SYNTH);      // (the "generate the inputs" slot of the profile)
    HIPCHK(ctx, trk::launch_inflate(*in, *out, ctx->n_cu, ctx->s()));
    return TRK_OK;
}

// ---- the native reader's inflate hook, served by the device (include/trk_vcf.h: trk_vcf_set_inflate_hook) ----------
// A run of BGZF members goes to the device compressed, is inflated there (trk_inflate.hip) into a SEGMENT of text that
// stays in HBM, and the host gets back what it reads of a batch: the newlines and the heads of the lines.  The sample
// columns are never text on the host; trk_inflate_text copies a batch's span of the stream out of the segments for the
// parse / format kernels.  Called from the reader's thread (its own queue through trk_thread_queue); one reader per
// context at a time.
struct InfSeg {
    uint64_t abs;      // stream offset of the segment's first byte
    size_t n;          // bytes of text
    uint8_t* d;
    size_t cap;
};
// One run of members on its way through the device: submitted (compressed bytes and tables up, kernel launched on the
// hook's own queue, flags on their way down) and not yet collected.
struct InfRun {
    uint8_t* d_comp = nullptr;
    size_t comp_cap = 0;
    uint8_t* d_tab = nullptr;
    size_t tab_cap = 0;
    uint8_t* h_tab = nullptr;      // pinned: tables up, flags down
    size_t h_cap = 0;
    uint8_t* seg = nullptr;
    size_t seg_cap = 0;
    size_t nb = 0, total = 0, comp_bytes = 0, flag_off = 0;
    uint64_t abs_base = 0;
    std::vector<trk_vcf_iblock> blocks;
    hipEvent_t done = nullptr;
    double t_up = 0.0;
    std::chrono::steady_clock::time_point t_submit;
};
static void inf_run_free(InfRun& r) {
    if (r.d_comp) (void)hipFree(r.d_comp);
    if (r.d_tab) (void)hipFree(r.d_tab);
    if (r.h_tab) (void)hipHostFree(r.h_tab);
    if (r.done) (void)hipEventDestroy(r.done);
    r = InfRun{};
}

struct InflateState {
    trk_ctx* ctx = nullptr;
    std::deque<InfRun> runs;            // submitted, not collected (in order)
    std::vector<InfRun> run_pool;       // their buffers, to use again
    hipStream_t q_inf = nullptr;        // the inflate kernels' queue

    std::deque<InfSeg> segs;
    std::vector<std::pair<uint8_t*, size_t>> spare;     // segment buffers to use again
    uint8_t* d_ws = nullptr;       // line-index workspace
    size_t ws_cap = 0;
    uint8_t* h_stage = nullptr;    // pinned: tables up, results down
    size_t h_cap = 0;
    std::vector<uint64_t> nl_host;
    uint64_t n_blocks = 0, n_flagged = 0, n_text = 0, n_comp = 0, n_calls = 0, n_regrown = 0;
    int tail_byte = -1;            // the last byte of the text handed over so far (a '\r' there flags the next run's first newline)
};
static void inflate_state_free(InflateState* st) {
    for (auto& sg : st->segs) (void)hipFree(sg.d);
    for (auto& sp : st->spare) (void)hipFree(sp.first);
    if (st->q_inf) (void)hipStreamSynchronize(st->q_inf);
    for (auto& r : st->runs) {
        if (r.seg) (void)hipFree(r.seg);
        inf_run_free(r);
    }
    for (auto& r : st->run_pool) inf_run_free(r);
    if (st->q_inf) (void)hipStreamDestroy(st->q_inf);
    if (st->d_ws) (void)hipFree(st->d_ws);
    if (st->h_stage) (void)hipHostFree(st->h_stage);
    delete st;
}
static bool inf_grow_dev(uint8_t*& p, size_t& cap, size_t need) {
    if (cap >= need) return true;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    const size_t want = need + need / 4 + 4096;
    if (hipMalloc((void**)&p, want) != hipSuccess) { (void)hipGetLastError(); return false; }
    cap = want;
    return true;
}
static bool inf_grow_host(InflateState* st, size_t need) {
    if (st->h_cap >= need) return true;
    if (st->h_stage) (void)hipHostFree(st->h_stage);
    st->h_stage = nullptr;
    st->h_cap = 0;
    const size_t want = need + need / 4 + 65536;
    if (hipHostMalloc((void**)&st->h_stage, want, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return false; }
    st->h_cap = want;
    return true;
}
static uint8_t* inf_segment(InflateState* st, size_t need, size_t& cap) {
    size_t best = SIZE_MAX;
    for (size_t i = 0; i < st->spare.size(); ++i)
        if (st->spare[i].second >= need && (best == SIZE_MAX || st->spare[i].second < st->spare[best].second)) best = i;
    if (best != SIZE_MAX) {
        uint8_t* p = st->spare[best].first;
        cap = st->spare[best].second;
        st->spare.erase(st->spare.begin() + (long)best);
        return p;
    }
    uint8_t* p = nullptr;
    const size_t want = need + need / 8 + 4096;
    if (hipMalloc((void**)&p, want) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    cap = want;
    return p;
}
static void inf_release(InflateState* st, const InfSeg& sg) {
    if (st->spare.size() < 8) st->spare.emplace_back(sg.d, sg.cap);
    else (void)hipFree(sg.d);
}

static int inf_seed(void* user, const char* text, size_t n) {
    InflateState* st = static_cast<InflateState*>(user);
    trk_ctx* ctx = st->ctx;
    (void)hipSetDevice(ctx->device);
    for (auto& sg : st->segs) inf_release(st, sg);
    st->segs.clear();
    if (st->q_inf) (void)hipStreamSynchronize(st->q_inf);
    while (!st->runs.empty()) {            // (a reader that went away with runs in flight)
        InfRun r = std::move(st->runs.front());
        st->runs.pop_front();
        if (r.seg) st->spare.emplace_back(r.seg, r.seg_cap);
        r.seg = nullptr;
        st->run_pool.push_back(std::move(r));
    }
    st->tail_byte = n ? (int)(unsigned char)text[n - 1] : -1;
    if (n == 0) return 0;
    size_t cap = 0;
    uint8_t* d = inf_segment(st, n + 2048, cap);
    if (!d) return TRK_ERR_NOMEM;
    if (hipMemcpyAsync(d, text, n, hipMemcpyHostToDevice, ctx->s()) != hipSuccess || hipStreamSynchronize(ctx->s()) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(d);
        return TRK_ERR_HIP;
    }
    st->segs.push_back({0, n, d, cap});
    return 0;
}

// submit: the run's compressed bytes are copied before the call returns (the reader moves its buffer on), the kernel
// runs on the hook's own queue -- behind the kernel of the run before, beside whatever the reader's and the caller's
// queues do.
static int inf_submit(void* user, const unsigned char* comp, size_t comp_bytes, const trk_vcf_iblock* blocks, int n_blocks,
                      uint64_t abs_base, size_t total) {
    InflateState* st = static_cast<InflateState*>(user);
    trk_ctx* ctx = st->ctx;
    (void)hipSetDevice(ctx->device);
    ++st->n_calls;
    InfRun r;
    if (!st->run_pool.empty()) {
        r = std::move(st->run_pool.back());
        st->run_pool.pop_back();
    }
    auto bail = [&](int code) {
        (void)hipGetLastError();
        if (st->q_inf) (void)hipStreamSynchronize(st->q_inf);
        if (r.seg) st->spare.emplace_back(r.seg, r.seg_cap);
        r.seg = nullptr;
        st->run_pool.push_back(std::move(r));
        return code;
    };
    r.nb = n_blocks > 0 ? (size_t)n_blocks : 0;
    r.total = total;
    r.comp_bytes = comp_bytes;
    r.abs_base = abs_base;
    r.t_submit = std::chrono::steady_clock::now();
    if (total == 0 || r.nb == 0) {                  // (members without text: the end-of-file marker)
        r.nb = 0;
        r.total = 0;
        st->runs.push_back(std::move(r));
        return 0;
    }
    if (!st->q_inf) {
        // the kernels' queue at the LOWEST priority: a hardware queue of its own (queues of one priority share a few, in
        // order -- the index kernels of run k then sat behind the kernel of run k + 1 for its 11 ms), and whatever the
        // reader's and the caller's queues launch is dispatched ahead of members that still wait for a CU
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        if (hipStreamCreateWithPriority(&st->q_inf, hipStreamNonBlocking, least) != hipSuccess) return bail(TRK_ERR_HIP);
    }
    if (!r.done && hipEventCreateWithFlags(&r.done, hipEventDisableTiming) != hipSuccess) return bail(TRK_ERR_HIP);
    const size_t nb = r.nb;
    // tables: in_off, out_off (int64), in_len, out_len (int32), then the flags
    const size_t tab_bytes = nb * 24;
    r.flag_off = (tab_bytes + 15) & ~(size_t)15;
    if (!inf_grow_dev(r.d_comp, r.comp_cap, comp_bytes + 64) || !inf_grow_dev(r.d_tab, r.tab_cap, r.flag_off + nb + 16)) return bail(TRK_ERR_NOMEM);
    if (r.h_cap < r.flag_off + nb + 64) {
        if (r.h_tab) (void)hipHostFree(r.h_tab);
        r.h_tab = nullptr;
        r.h_cap = 0;
        const size_t want = (r.flag_off + nb + 64) * 5 / 4 + 4096;
        if (hipHostMalloc((void**)&r.h_tab, want, hipHostMallocDefault) != hipSuccess) return bail(TRK_ERR_NOMEM);
        r.h_cap = want;
    }
    r.seg = inf_segment(st, total + 2048, r.seg_cap);
    if (!r.seg) return bail(TRK_ERR_NOMEM);
    int64_t* t_in_off = reinterpret_cast<int64_t*>(r.h_tab);
    int64_t* t_out_off = t_in_off + nb;
    int32_t* t_in_len = reinterpret_cast<int32_t*>(t_out_off + nb);
    int32_t* t_out_len = t_in_len + nb;
    for (size_t i = 0; i < nb; ++i) {
        if (blocks[i].payload_off + blocks[i].payload_len > comp_bytes || blocks[i].dst + blocks[i].isize > total) return bail(TRK_ERR_ARG);
        t_in_off[i] = (int64_t)blocks[i].payload_off;
        t_out_off[i] = (int64_t)blocks[i].dst;
        t_in_len[i] = (int32_t)blocks[i].payload_len;
        t_out_len[i] = (int32_t)blocks[i].isize;
    }
    r.blocks.assign(blocks, blocks + nb);
    // (the upload on the calling thread's queue -- idle: collect waits for what it puts there -- and waited for here:
    // `comp` is the reader's to reuse when this returns, and waiting on the kernels' queue would wait for the run before)
    hipStream_t q_up = ctx->s();
    if (hipMemcpyAsync(r.d_comp, comp, comp_bytes, hipMemcpyHostToDevice, q_up) != hipSuccess) return bail(TRK_ERR_HIP);
    if (hipMemcpyAsync(r.d_tab, r.h_tab, tab_bytes, hipMemcpyHostToDevice, q_up) != hipSuccess) return bail(TRK_ERR_HIP);
    if (hipStreamSynchronize(q_up) != hipSuccess) return bail(TRK_ERR_HIP);
    r.t_up = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - r.t_submit).count();
    trk_inflate_in in = {};
    in.comp = r.d_comp;
    in.n_comp_bytes = (int64_t)comp_bytes;
    in.n_blocks = n_blocks;
    in.in_off = reinterpret_cast<const int64_t*>(r.d_tab);
    in.out_off = in.in_off + nb;
    in.in_len = reinterpret_cast<const int32_t*>(in.out_off + nb);
    in.out_len = in.in_len + nb;
    trk_inflate_out io = {r.seg, r.d_tab + r.flag_off};
    // three workgroups per CU, not the four that fit: a CU full of members (134 KB of its LDS, for the 5 ms a member takes)
    // has no room for a workgroup of the parse kernel (35 KB) or of the count kernels -- the batch before waited for members
    // to retire: statSTR 0.082-0.089 -> 0.076-0.084 s at 1 GB (profiles/r05_notes.md section 6)
    if (trk::launch_inflate(in, io, ctx->n_cu, st->q_inf, 3) != hipSuccess) return bail(TRK_ERR_HIP);
    if (hipMemsetAsync(r.seg + total, '\n', 2048, st->q_inf) != hipSuccess) return bail(TRK_ERR_HIP);     // (readable padding)
    if (hipMemcpyAsync(r.h_tab + r.flag_off, r.d_tab + r.flag_off, nb, hipMemcpyDeviceToHost, st->q_inf) != hipSuccess) return bail(TRK_ERR_HIP);
    if (hipEventRecord(r.done, st->q_inf) != hipSuccess) return bail(TRK_ERR_HIP);
    st->runs.push_back(std::move(r));
    return 0;
}

// collect: the oldest run submitted -- wait for its kernel, inflate what it flagged here, index the text and bring the
// newlines and the heads back (on the calling thread's queue)
static int inf_collect(void* user, char* out, int* line_state, const uint64_t** nl, size_t* n_nl) {
    InflateState* st = static_cast<InflateState*>(user);
    trk_ctx* ctx = st->ctx;
    (void)hipSetDevice(ctx->device);
    hipStream_t q = ctx->s();
    *nl = nullptr;
    *n_nl = 0;
    if (st->runs.empty()) return TRK_ERR_ARG;
    InfRun r = std::move(st->runs.front());
    st->runs.pop_front();
    const size_t nb = r.nb, total = r.total;
    if (total == 0 || nb == 0) {
        st->run_pool.push_back(std::move(r));
        return 0;
    }
    const bool timing = trk_opt("TRK_INFLATE_TIMING") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double, std::milli>(b - a).count();
    };
    uint8_t* seg = r.seg;
    const size_t seg_cap = r.seg_cap;
    auto bail = [&](int code) {
        (void)hipGetLastError();
        (void)hipStreamSynchronize(q);
        st->spare.emplace_back(seg, seg_cap);
        r.seg = nullptr;
        st->run_pool.push_back(std::move(r));
        return code;
    };
    const auto t1 = now();
    if (hipEventSynchronize(r.done) != hipSuccess) return bail(TRK_ERR_HIP);
    const auto t2 = now();
    const uint8_t* h_flags = r.h_tab + r.flag_off;
    // members the kernel left: inflated here (zlib; their compressed bytes come back from the device), their text copied in
    for (size_t i = 0; i < nb; ++i) {
        if (!h_flags[i]) continue;
        ++st->n_flagged;
        const trk_vcf_iblock& b = r.blocks[i];
        std::vector<unsigned char> cin(b.payload_len ? b.payload_len : 1), tmp(b.isize ? b.isize : 1);
        if (b.payload_len && hipMemcpy(cin.data(), r.d_comp + b.payload_off, b.payload_len, hipMemcpyDeviceToHost) != hipSuccess)
            return bail(TRK_ERR_HIP);
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        if (inflateInit2(&zs, -15) != Z_OK) return bail(TRK_ERR_HIP);
        zs.next_in = cin.data();
        zs.avail_in = b.payload_len;
        zs.next_out = tmp.data();
        zs.avail_out = b.isize;
        const int rc = inflate(&zs, Z_FINISH);
        const bool okz = rc == Z_STREAM_END && zs.total_out == b.isize;
        inflateEnd(&zs);
        if (!okz) return bail(TRK_ERR_ARG);                 // a member nobody can inflate: the file is corrupt
        if (b.isize && hipMemcpy(seg + b.dst, tmp.data(), b.isize, hipMemcpyHostToDevice) != hipSuccess) return bail(TRK_ERR_HIP);
    }
    // the line index and the heads.  Two steps: the newlines are COUNTED first and the tables sized from the count (a
    // guess -- a line per sixteen bytes -- only sizes the first try: text with shorter lines, blank lines or no VCF at
    // all, is indexed like any other, as the host's path reads it)
    const size_t n_tiles = (total + 16383) / 16384;
    size_t nl_cap = total / 16 + 1024;
    trk::LineIndexWs ws;
    uint32_t* scal = nullptr;
    // workspace: counts, scalars (n_nl, head_total, state, last byte), nl, head_off, head_len, pack_off, packed
    auto lay_out = [&]() -> bool {
        size_t o_counts = 0, o_scal = (o_counts + (n_tiles + 1) * 4 + 15) & ~(size_t)15, o_nl = o_scal + 64, o_hoff = o_nl + nl_cap * 8,
               o_hlen = o_hoff + (nl_cap + 1) * 8, o_poff = (o_hlen + (nl_cap + 1) * 4 + 15) & ~(size_t)15,
               o_pack = (o_poff + (nl_cap + 1) * 4 + 15) & ~(size_t)15, ws_bytes = o_pack + total + 64;
        if (!inf_grow_dev(st->d_ws, st->ws_cap, ws_bytes)) return false;
        ws.counts = reinterpret_cast<uint32_t*>(st->d_ws + o_counts);
        scal = reinterpret_cast<uint32_t*>(st->d_ws + o_scal);
        ws.n_nl = scal;
        ws.head_total = scal + 1;
        ws.state = reinterpret_cast<int32_t*>(scal + 2);
        ws.nl = reinterpret_cast<uint64_t*>(st->d_ws + o_nl);
        ws.nl_cap = (uint32_t)nl_cap;
        ws.head_off = reinterpret_cast<uint64_t*>(st->d_ws + o_hoff);
        ws.head_len = reinterpret_cast<uint32_t*>(st->d_ws + o_hlen);
        ws.pack_off = reinterpret_cast<uint32_t*>(st->d_ws + o_poff);
        ws.packed = st->d_ws + o_pack;
        ws.packed_cap = (uint32_t)std::min<size_t>(total + 64, 0xffffffffu);
        return true;
    };
    if (total >= 0xfffffff0u) return bail(TRK_ERR_ARG);             // (32-bit line tables; a run is a few hundred megabytes)
    if (!lay_out()) return bail(TRK_ERR_NOMEM);
    if (!inf_grow_host(st, 4096)) return bail(TRK_ERR_NOMEM);
    uint32_t* h_scal = reinterpret_cast<uint32_t*>(st->h_stage);
    for (int attempt = 0;; ++attempt) {
        if (trk::launch_line_count(seg, (int64_t)total, ws, q) != hipSuccess) return bail(TRK_ERR_HIP);
        if (hipMemcpyAsync(h_scal, scal, 4, hipMemcpyDeviceToHost, q) != hipSuccess || hipStreamSynchronize(q) != hipSuccess) return bail(TRK_ERR_HIP);
        if (h_scal[0] <= nl_cap) break;
        if (attempt) return bail(TRK_ERR_HIP);                       // (the count of one text cannot change)
        nl_cap = (size_t)h_scal[0] + 1024;                           // the workspace moves: counted again in the new one
        ++st->n_regrown;
        if (!lay_out()) return bail(TRK_ERR_NOMEM);
    }
    if (trk::launch_line_index(seg, (int64_t)total, *line_state, ws, q) != hipSuccess) return bail(TRK_ERR_HIP);
    if (hipMemcpyAsync(h_scal, scal, 16, hipMemcpyDeviceToHost, q) != hipSuccess || hipStreamSynchronize(q) != hipSuccess) return bail(TRK_ERR_HIP);
    const auto t3 = now();
    const uint32_t n_found = h_scal[0], head_total = h_scal[1];
    const int state = (int)h_scal[2];
    const int tail_now = (int)(int32_t)h_scal[3];      // (read here: the staging buffer may move below)
    if (n_found > nl_cap || head_total > ws.packed_cap) return bail(TRK_ERR_ARG);       // (cannot happen: sized from the count above)
    const size_t n_lines = (size_t)n_found + 1;
    const size_t r_nl = 0, r_hoff = r_nl + (size_t)n_found * 8, r_hlen = r_hoff + n_lines * 8, r_pack = (r_hlen + n_lines * 4 + 15) & ~(size_t)15,
                 r_bytes = r_pack + head_total;
    if (!inf_grow_host(st, r_bytes + 64)) return bail(TRK_ERR_NOMEM);
    hipError_t e = hipSuccess;
    if (n_found) e = hipMemcpyAsync(st->h_stage + r_nl, ws.nl, (size_t)n_found * 8, hipMemcpyDeviceToHost, q);
    if (e == hipSuccess) e = hipMemcpyAsync(st->h_stage + r_hoff, ws.head_off, n_lines * 8, hipMemcpyDeviceToHost, q);
    if (e == hipSuccess) e = hipMemcpyAsync(st->h_stage + r_hlen, ws.head_len, n_lines * 4, hipMemcpyDeviceToHost, q);
    if (e == hipSuccess && head_total) e = hipMemcpyAsync(st->h_stage + r_pack, ws.packed, head_total, hipMemcpyDeviceToHost, q);
    if (e == hipSuccess) e = hipStreamSynchronize(q);
    if (e != hipSuccess) return bail(TRK_ERR_HIP);
    const auto t4 = now();
    const uint64_t* h_nl = reinterpret_cast<const uint64_t*>(st->h_stage + r_nl);
    const uint64_t* h_hoff = reinterpret_cast<const uint64_t*>(st->h_stage + r_hoff);
    const uint32_t* h_hlen = reinterpret_cast<const uint32_t*>(st->h_stage + r_hlen);
    const uint8_t* h_pack = st->h_stage + r_pack;
    size_t at = 0;
    for (size_t i = 0; i < n_lines; ++i) {
        const size_t len = h_hlen[i];
        if (h_hoff[i] + len > total || at + len > head_total) return bail(TRK_ERR_ARG);
        if (len) memcpy(out + h_hoff[i], h_pack + at, len);
        at += len;
    }
    st->nl_host.assign(h_nl, h_nl + n_found);
    // a CRLF pair cut by the run's boundary: the '\r' was the last byte of the text before (the run before's, or the seed's)
    if (n_found && (st->nl_host[0] & ~(1ull << 63)) == 0 && st->tail_byte == '\r') st->nl_host[0] |= 1ull << 63;
    st->tail_byte = tail_now;
    *nl = st->nl_host.data();
    *n_nl = n_found;
    *line_state = state;
    st->segs.push_back({r.abs_base, total, seg, seg_cap});
    st->n_blocks += nb;
    st->n_text += total;
    st->n_comp += r.comp_bytes;
    if (timing)
        fprintf(stderr, "[trk inflate] %zu members, %.1f MB -> %.1f MB: upload %.2f ms, submitted %.2f ms ago, waited %.2f for the kernel, "
                        "index %.2f, results down %.2f, heads %.2f\n", nb, r.comp_bytes / 1e6, total / 1e6, r.t_up, ms(r.t_submit, t1),
                ms(t1, t2), ms(t2, t3), ms(t3, t4), ms(t4, now()));
    r.seg = nullptr;
    st->run_pool.push_back(std::move(r));
    return 0;
}

// the two halves in one call (a reader that keeps one run in flight)
static int inf_inflate(void* user, const unsigned char* comp, size_t comp_bytes, const trk_vcf_iblock* blocks, int n_blocks,
                       uint64_t abs_base, size_t total, char* out, int* line_state, const uint64_t** nl, size_t* n_nl) {
    const int rc = inf_submit(user, comp, comp_bytes, blocks, n_blocks, abs_base, total);
    if (rc != 0) return rc;
    return inf_collect(user, out, line_state, nl, n_nl);
}

int trk_inflate_hook(trk_ctx* ctx, void** user, void** seed_fn, void** inflate_fn) {
    if (!ctx || !user || !seed_fn || !inflate_fn) return TRK_ERR_ARG;
    if (!ctx->inflate) {
        ctx->inflate = new InflateState();
        ctx->inflate->ctx = ctx;
    }
    *user = ctx->inflate;
    *seed_fn = reinterpret_cast<void*>(&inf_seed);
    *inflate_fn = reinterpret_cast<void*>(&inf_inflate);
    return TRK_OK;
}

int trk_inflate_hook_async(trk_ctx* ctx, void** submit_fn, void** collect_fn) {
    if (!ctx || !submit_fn || !collect_fn) return TRK_ERR_ARG;
    *submit_fn = reinterpret_cast<void*>(&inf_submit);
    *collect_fn = reinterpret_cast<void*>(&inf_collect);
    return TRK_OK;
}

int trk_inflate_text(trk_ctx* ctx, uint64_t abs_from, int64_t n_bytes, void* dst, uint64_t release_before) {
    if (!ctx || !ctx->inflate) return TRK_ERR_ARG;
    if (n_bytes < 0 || (n_bytes > 0 && !dst)) return fail(ctx, TRK_ERR_ARG, "inflate_text: %lld bytes", (long long)n_bytes);
    InflateState* st = ctx->inflate;
    (void)hipSetDevice(ctx->device);
    hipStream_t q = ctx->s();
    const uint64_t to = abs_from + (uint64_t)n_bytes;
    uint64_t covered = abs_from;          // the segments are consecutive: everything below `covered` has been copied
    for (const InfSeg& sg : st->segs) {
        const uint64_t a = std::max<uint64_t>(sg.abs, abs_from), b = std::min<uint64_t>(sg.abs + sg.n, to);
        if (a >= b) continue;
        if (a != covered) return fail(ctx, TRK_ERR_ARG, "inflate_text: bytes %llu .. %llu of the stream are no longer (or not yet) on the device",
                                      (unsigned long long)covered, (unsigned long long)a);
        HIPCHK(ctx, hipMemcpyAsync(static_cast<uint8_t*>(dst) + (a - abs_from), sg.d + (a - sg.abs), (size_t)(b - a), hipMemcpyDeviceToDevice, q));
        covered = b;
    }
    if (covered < to) {
        // beyond the inflated text: the newline the reader appends to a last line without one
        if (to - covered > 16) return fail(ctx, TRK_ERR_ARG, "inflate_text: %llu bytes beyond the inflated text", (unsigned long long)(to - covered));
        HIPCHK(ctx, hipMemsetAsync(static_cast<uint8_t*>(dst) + (covered - abs_from), '\n', (size_t)(to - covered), q));
    }
    while (!st->segs.empty() && st->segs.front().abs + st->segs.front().n <= release_before) {
        HIPCHK(ctx, hipStreamSynchronize(q));          // (its bytes may still be on their way out)
        inf_release(st, st->segs.front());
        st->segs.pop_front();
    }
    return TRK_OK;
}

int trk_inflate_stats(trk_ctx* ctx, uint64_t out[5]) {
    if (!ctx || !out) return TRK_ERR_ARG;
    for (int i = 0; i < 5; ++i) out[i] = 0;
    if (ctx->inflate) {
        out[0] = ctx->inflate->n_blocks;
        out[1] = ctx->inflate->n_flagged;
        out[2] = ctx->inflate->n_text;
        out[3] = ctx->inflate->n_comp;
        out[4] = ctx->inflate->n_calls;
    }
    return TRK_OK;
}

// ---- BGZF members deflated on the device (include/trk.h) ----
size_t trk_deflate_bound(size_t n) { return ((n + TRK_DEFLATE_MEMBER - 1) / TRK_DEFLATE_MEMBER + 1) * (size_t)(TRK_DEFLATE_MEMBER + 64 + 26); }

int trk_deflate_bgzf(trk_ctx* ctx, const void* host_text, size_t n, void* host_out, size_t out_cap, size_t* out_bytes) {
    if (!ctx) return TRK_ERR_ARG;
    if (out_bytes) *out_bytes = 0;
    if ((!host_text && n) || !host_out || !out_bytes) return fail(ctx, TRK_ERR_ARG, "deflate_bgzf: arguments");
    if (out_cap < trk_deflate_bound(n)) return fail(ctx, TRK_ERR_ARG, "deflate_bgzf: out_cap below trk_deflate_bound(%zu)", n);
    if (n == 0) return TRK_OK;
    (void)hipSetDevice(ctx->device);
    {
        std::lock_guard<std::mutex> g0(ctx->queue_m);
        if (!ctx->deflate) ctx->deflate = new DeflateState();
    }
    DeflateState* d = ctx->deflate;
    std::lock_guard<std::mutex> g(d->m);
    if (!d->q && hipStreamCreateWithFlags(&d->q, hipStreamNonBlocking) != hipSuccess) return fail(ctx, TRK_ERR_HIP, "deflate_bgzf: queue");
    hipStream_t q = d->q;
    const size_t nm = (n + TRK_DEFLATE_MEMBER - 1) / TRK_DEFLATE_MEMBER;
    auto grow = [&](uint8_t*& p, size_t& cap, size_t need) {
        if (cap >= need) return true;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        const size_t want = need + need / 4 + 4096;
        if (hipMalloc((void**)&p, want) != hipSuccess) { (void)hipGetLastError(); return false; }
        cap = want;
        return true;
    };
    const size_t tab_bytes = ((nm * 4 + 15) & ~(size_t)15) + (nm + 1) * 8;
    if (!grow(d->d_text, d->text_cap, n + 512) || !grow(d->d_slots, d->slots_cap, nm * trk::deflate_slot_bytes()) ||
        !grow(d->d_out, d->out_cap, nm * (trk::deflate_slot_bytes() + 26)) || !grow(d->d_tok, d->tok_cap, trk::deflate_tok_bytes(ctx->n_cu, (int)nm)) ||
        !grow(d->d_tab, d->tab_cap, tab_bytes))
        return fail(ctx, TRK_ERR_NOMEM, "deflate_bgzf: device buffers for %zu members", nm);
    if (d->h_cap < (nm + 1) * 8) {
        if (d->h_off) (void)hipHostFree(d->h_off);
        d->h_off = nullptr;
        d->h_cap = 0;
        const size_t want = (nm + 1) * 8 * 2 + 4096;
        if (hipHostMalloc((void**)&d->h_off, want, hipHostMallocDefault) != hipSuccess) return fail(ctx, TRK_ERR_NOMEM, "deflate_bgzf: staging");
        d->h_cap = want;
    }
    uint32_t* d_sizes = reinterpret_cast<uint32_t*>(d->d_tab);
    uint64_t* d_off = reinterpret_cast<uint64_t*>(d->d_tab + ((nm * 4 + 15) & ~(size_t)15));
    HIPCHK(ctx, hipMemcpyAsync(d->d_text, host_text, n, hipMemcpyHostToDevice, q));
    HIPCHK(ctx, trk::launch_deflate(d->d_text, (int64_t)n, d->d_slots, d_sizes, reinterpret_cast<uint32_t*>(d->d_tok), d_off, d->d_out,
                                    ctx->n_cu, q));
    HIPCHK(ctx, hipMemcpyAsync(d->h_off, d_off, (nm + 1) * 8, hipMemcpyDeviceToHost, q));
    // the members' checksums while the device works (the text is the host's)
    d->crc.resize(nm);
    trk_member_crc32(host_text, n, TRK_DEFLATE_MEMBER, d->crc.data());
    HIPCHK(ctx, hipStreamSynchronize(q));
    const uint64_t total = d->h_off[nm];
    if (total > out_cap) return fail(ctx, TRK_ERR_ARG, "deflate_bgzf: %llu bytes of members for a buffer of %zu", (unsigned long long)total, out_cap);
    HIPCHK(ctx, hipMemcpyAsync(host_out, d->d_out, total, hipMemcpyDeviceToHost, q));
    HIPCHK(ctx, hipStreamSynchronize(q));
    unsigned char* o = static_cast<unsigned char*>(host_out);
    for (size_t m = 0; m < nm; ++m) {
        unsigned char* t = o + d->h_off[m + 1] - 8;
        const uint32_t c = d->crc[m];
        t[0] = (unsigned char)c; t[1] = (unsigned char)(c >> 8); t[2] = (unsigned char)(c >> 16); t[3] = (unsigned char)(c >> 24);
    }
    *out_bytes = (size_t)total;
    return TRK_OK;
}

int trk_format_samples(trk_ctx* ctx, const trk_format_in* in, trk_format_out* out, int pass) {
    if (!ctx) return TRK_ERR_ARG;
    if (!in || !out || (pass != 1 && pass != 2)) return fail(ctx, TRK_ERR_ARG, "format_samples: arguments");
    if (in->n_records < 0 || in->n_samples < 0 || in->n_filters < 0 || in->n_filters > TRK_FORMAT_MAX_FILTERS)
        return fail(ctx, TRK_ERR_ARG, "format_samples: records %d, samples %d, filters %d", in->n_records, in->n_samples, in->n_filters);
    if (in->n_records == 0 || in->n_samples == 0) return TRK_OK;
    if (!in->text || ((uintptr_t)in->text & 15u) || !in->smp_off || !in->line_end || !in->field_kind || !in->n_fields ||
        !in->ploidy || !in->mask8 || !out->rec_len || !out->flags || in->mask_stride < in->n_samples ||
        in->plane_stride < in->n_samples)
        return fail(ctx, TRK_ERR_ARG, "format_samples: inputs / outputs are NULL or too narrow");
    for (int i = 0; i < in->n_filters; ++i)
        if (!in->filter_plane[i] || !memchr(in->filter_name[i], 0, sizeof in->filter_name[i]))
            return fail(ctx, TRK_ERR_ARG, "format_samples: filter %d", i);
    if (pass == 2 && (!out->out || !out->out_off)) return fail(ctx, TRK_ERR_ARG, "format_samples: pass 2 needs out and out_off");
    (void)hipSetDevice(ctx->device);
    HIPCHK(ctx, trk::launch_format_samples(*in, *out, pass, ctx->s()));
    return TRK_OK;
}

int trk_planarize(trk_ctx* ctx, const void* src, void* dst, int64_t n_cells, int32_t ncol) {
    if (!ctx) return TRK_ERR_ARG;
    if (!src || !dst || n_cells < 0 || ncol < 1) return fail(ctx, TRK_ERR_ARG, "planarize arguments");
    if (n_cells == 0) return TRK_OK;
    (void)hipSetDevice(ctx->device);
    HIPCHK(ctx, trk::launch_planarize(src, dst, n_cells, ncol, ctx->s()));
    return TRK_OK;
}

int trk_pad_rows(trk_ctx* ctx, const void* src, void* dst, int64_t n_rows, int32_t row_words, int32_t pad_words,
                 uint32_t fill) {
    if (!ctx) return TRK_ERR_ARG;
    if (n_rows < 0 || row_words < 0 || pad_words < 0) return fail(ctx, TRK_ERR_ARG, "pad_rows arguments");
    if (n_rows == 0 || row_words + pad_words == 0) return TRK_OK;
    if (!src || !dst || src == dst) return fail(ctx, TRK_ERR_ARG, "pad_rows: NULL or aliased arrays");
    (void)hipSetDevice(ctx->device);
    HIPCHK(ctx, trk::launch_pad_rows(src, dst, n_rows, row_words, pad_words, fill, ctx->n_cu, ctx->s()));
    return TRK_OK;
}

int trk_permute_columns(trk_ctx* ctx, const int16_t* src, int16_t* dst, const int32_t* col, int64_t n_loci,
                        int32_t n_src, int32_t n_dst, int32_t ploidy) {
    if (!ctx) return TRK_ERR_ARG;
    if (n_loci < 0 || n_src < 0 || n_dst < 0 || ploidy < 1 || ploidy > TRK_MAX_PLOIDY)
        return fail(ctx, TRK_ERR_ARG, "permute_columns arguments");
    if (n_loci == 0 || n_dst == 0) return TRK_OK;
    if (!src || !dst || !col) return fail(ctx, TRK_ERR_ARG, "permute_columns: NULL array");
    (void)hipSetDevice(ctx->device);
    HIPCHK(ctx, trk::launch_permute_columns(src, dst, col, n_loci, n_src, n_dst, ploidy, ctx->n_cu, ctx->s()));
    return TRK_OK;
}

int trk_stream_probe(trk_ctx* ctx, const void* in0, const void* in1, const void* in2, void* out0, void* out1,
                     int64_t n_loci, int64_t n_samples, int32_t reps, float* avg_ms) {
    if (!ctx) return TRK_ERR_ARG;
    if (!in0 || !in1 || !in2 || !out0 || !out1 || !avg_ms || n_loci < 1 || n_samples < 4 || n_samples % 4 || reps < 1)
        return fail(ctx, TRK_ERR_ARG, "stream probe arguments");
    (void)hipSetDevice(ctx->device);
    const void* in[3] = {in0, in1, in2};
    void* out[2] = {out0, out1};
    hipEvent_t e0, e1;
    HIPCHK(ctx, hipEventCreate(&e0));
    HIPCHK(ctx, hipEventCreate(&e1));
    hipError_t e = trk::launch_stream_probe(in, 3, out, 2, n_loci, n_samples, ctx->n_cu, ctx->s());   // warm-up
    if (e == hipSuccess) e = hipEventRecord(e0, ctx->s());
    for (int r = 0; r < reps && e == hipSuccess; ++r)
        e = trk::launch_stream_probe(in, 3, out, 2, n_loci, n_samples, ctx->n_cu, ctx->s());
    if (e == hipSuccess) e = hipEventRecord(e1, ctx->s());
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (e != hipSuccess) return fail(ctx, TRK_ERR_HIP, "trk_stream_probe: %s", hipGetErrorString(e));
    *avg_ms = ms / (float)reps;
    return TRK_OK;
}

// write-only probe of a pair of output planes (the two-output half of the call-filter pass's stream): ms per launch
static hipError_t probe_pair_ms(trk_ctx* ctx, void* a, void* b, int64_t n_loci, int64_t n_samples, float* ms_out) {
    void* out[2] = {a, b};
    hipEvent_t e0, e1;
    hipError_t e = hipEventCreate(&e0);
    if (e != hipSuccess) return e;
    e = hipEventCreate(&e1);
    if (e != hipSuccess) { (void)hipEventDestroy(e0); return e; }
    const int reps = 2;
    e = trk::launch_stream_probe(nullptr, 0, out, 2, n_loci, n_samples, ctx->n_cu, ctx->s());   // warm-up (first touch)
    if (e == hipSuccess) e = hipEventRecord(e0, ctx->s());
    for (int r = 0; r < reps && e == hipSuccess; ++r)
        e = trk::launch_stream_probe(nullptr, 0, out, 2, n_loci, n_samples, ctx->n_cu, ctx->s());
    if (e == hipSuccess) e = hipEventRecord(e1, ctx->s());
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *ms_out = ms / (float)reps;
    return e;
}

int trk_reserve_pair(trk_ctx* ctx, size_t bytes_each, trk_pair_info* info) {
    if (!ctx) return TRK_ERR_ARG;
    trk_pair_info pi = {};
    pi.have_a = pi.have_b = -1;
    if (ctx->res_plane[0]) return fail(ctx, TRK_ERR_ARG, "trk_reserve_pair: a pair is reserved already");
    bytes_each = (bytes_each + 4095) & ~(size_t)4095;
    if (bytes_each < ((size_t)1 << 20)) return fail(ctx, TRK_ERR_ARG, "trk_reserve_pair: at least 1 MB per plane");
    (void)hipSetDevice(ctx->device);
    const auto t0 = std::chrono::steady_clock::now();
    // Planes one at a time, each timed with every plane taken before it, until a pair is on the fast level: at the start
    // of a process the placement class changes within the first few allocations (profiles/r05_class_probe.txt: a fast
    // partner for plane 0 among planes 1 ... 4 in seven of eight fresh processes) and the memory is empty, so up to
    // four spare planes cost nothing but 5 ms of probes each.  The best pair stays, the rest goes back.
    constexpr int MAXP = 8;
    void* p[MAXP] = {};
    const int64_t S = 8192, Lp = (int64_t)(bytes_each / ((size_t)S * 4u));   // the probe's shape: rows of 8192 samples
    const double gbytes = 2.0 * (double)Lp * (double)S * 4.0 * 1e-9;
    int n = 0, n_probes = 0, keep_a = 0, keep_b = 1;
    float best = 0.f;
    hipError_t e = hipSuccess;
    const bool can_probe = Lp >= 1 && bytes_each >= ((size_t)1 << 28);      // (below 256 MB no levels can be told apart)
    while (n < MAXP) {
        if (hipMalloc(&p[n], bytes_each) != hipSuccess) {
            (void)hipGetLastError();
            p[n] = nullptr;
            break;
        }
        ++n;
        if (n < 2) continue;
        if (!can_probe) break;
        bool fast = false;
        for (int j = 0; j < n - 1 && e == hipSuccess; ++j) {
            float ms = 0.f;
            e = probe_pair_ms(ctx, p[j], p[n - 1], Lp, S, &ms);
            if (e != hipSuccess) break;
            if (n_probes < TRK_PAIR_MAX_PROBES) pi.probe_ms[n_probes] = ms;
            ++n_probes;
            if (best == 0.f || ms < best) { best = ms; keep_a = j; keep_b = n - 1; }
        }
        if (e != hipSuccess) break;
        fast = gbytes / (double)best >= TRK_PAIR_FAST_TBPS;
        if (fast) break;
    }
    if (e != hipSuccess || n < 2) {
        for (int k = 0; k < MAXP; ++k) if (p[k]) (void)hipFree(p[k]);
        if (e != hipSuccess) return fail(ctx, TRK_ERR_HIP, "trk_reserve_pair probe: %s", hipGetErrorString(e));
        return fail(ctx, TRK_ERR_NOMEM, "trk_reserve_pair: hipMalloc(%zu)", bytes_each);
    }
    const float tbps = (can_probe && best > 0.f) ? (float)(gbytes / (double)best) : 0.f;
    if (can_probe && tbps < (float)TRK_PAIR_FAST_TBPS) {
        // no fast pair among MAXP planes: a pair that trk_dev_alloc_pair would never lend is not kept either (ADVICE r05:
        // 2 x bytes_each held for the life of the context for nothing) -- everything goes back, the caller is told
        for (int k = 0; k < MAXP; ++k) if (p[k]) (void)hipFree(p[k]);
        pi.n_probed = n_probes < TRK_PAIR_MAX_PROBES ? n_probes : TRK_PAIR_MAX_PROBES;
        pi.kept_ms = best;
        pi.placed = 0;
        pi.peak_extra_bytes = (uint64_t)n * (uint64_t)bytes_each;
        pi.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (info) *info = pi;
        return TRK_OK;
    }
    for (int k = 0; k < MAXP; ++k)
        if (p[k] && k != keep_a && k != keep_b) (void)hipFree(p[k]);
    ctx->res_plane[0] = p[keep_a];
    ctx->res_plane[1] = p[keep_b];
    ctx->res_bytes = bytes_each;
    ctx->res_lent[0] = ctx->res_lent[1] = false;
    ctx->res_tbps = tbps;
    pi.n_probed = n_probes < TRK_PAIR_MAX_PROBES ? n_probes : TRK_PAIR_MAX_PROBES;
    pi.kept_ms = best;
    pi.placed = ctx->res_tbps >= (float)TRK_PAIR_FAST_TBPS ? 1 : 0;
    pi.peak_extra_bytes = (uint64_t)(n > 2 ? n - 2 : 0) * (uint64_t)bytes_each;
    pi.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (info) *info = pi;
    return TRK_OK;
}

int trk_dev_alloc_pair(trk_ctx* ctx, size_t bytes_each, int64_t n_loci, int64_t n_samples, int32_t max_spare,
                       void* const* have, int32_t n_have, void** a, void** b, trk_pair_info* info) {
    if (!ctx || !a || !b || n_have < 0 || (n_have > 0 && !have)) return TRK_ERR_ARG;
    *a = *b = nullptr;
    trk_pair_info pi = {};
    pi.have_a = pi.have_b = -1;
    if (n_loci < 1 || n_samples < 4 || n_samples % 4 || bytes_each < (size_t)n_loci * (size_t)n_samples * 4u)
        return fail(ctx, TRK_ERR_ARG, "trk_dev_alloc_pair: planes of [n_loci, n_samples] 4-byte cells, n_samples %% 4 == 0");
    if (max_spare < 0) max_spare = 0;
    (void)hipSetDevice(ctx->device);
    const auto t0 = std::chrono::steady_clock::now();
    hipError_t e = hipSuccess;
    // the reserved pair (trk_reserve_pair: the process's first two allocations), when it is free and large enough:
    // lent out as it is, timed once with the caller's shape for the record
    if (ctx->res_plane[0] && !ctx->res_lent[0] && !ctx->res_lent[1] && bytes_each <= ctx->res_bytes &&
        (ctx->res_tbps >= (float)TRK_PAIR_FAST_TBPS || ctx->res_bytes < ((size_t)1 << 28))) {    // (a pair that is not fast is not lent: the search below)
        *a = ctx->res_plane[0];
        *b = ctx->res_plane[1];
        ctx->res_lent[0] = ctx->res_lent[1] = true;
        e = probe_pair_ms(ctx, *a, *b, n_loci, n_samples, &pi.probe_ms[0]);
        if (e != hipSuccess) {
            ctx->res_lent[0] = ctx->res_lent[1] = false;
            *a = *b = nullptr;
            return fail(ctx, TRK_ERR_HIP, "trk_dev_alloc_pair probe: %s", hipGetErrorString(e));
        }
        pi.n_probed = 1;
        pi.kept_ms = pi.probe_ms[0];
        pi.reserved = 1;
        const double gb = 2.0 * (double)n_loci * (double)n_samples * 4.0 * 1e-9;
        pi.placed = (gb / (double)pi.kept_ms >= TRK_PAIR_FAST_TBPS || ctx->res_tbps >= (float)TRK_PAIR_FAST_TBPS) ? 1 : 0;
        pi.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (info) *info = pi;
        return TRK_OK;
    }
    // the first plane: one the caller holds already (a pooled buffer), else a fresh allocation
    if (n_have > 0) {
        *a = have[0];
        pi.have_a = 0;
    } else {
        e = hipMalloc(a, bytes_each);
        if (e != hipSuccess) return fail(ctx, TRK_ERR_NOMEM, "hipMalloc(%zu): %s", bytes_each, hipGetErrorString(e));
    }
    // Candidates for the second plane, one at a time, each timed together with the first (a pair of planes is on one
    // of two levels for as long as the allocations live -- profiles/r03_notes.md section 22 -- and the level shows in
    // the write-only half of the stream: 7.0 against 5.4-5.9 TB/s): first the other planes the caller holds, then
    // fresh allocations.  At most `max_spare` fresh planes beyond the ones returned exist at any time; the search stops
    // at the first pair that is clearly on the fast level.  (A candidate handed back to the driver comes back as the
    // next allocation: every candidate is held until the search ends, which is why their number is the memory bound.)
    struct Cand { void* p; int have; };
    Cand cand[TRK_PAIR_MAX_PROBES] = {};
    float ms[TRK_PAIR_MAX_PROBES] = {};
    int n = 0, best = -1, n_fresh = 0, n_jumps = 0;
    const int fresh_cap = 1 + max_spare;
    // TRK_PLACE_JUMP_GB: the spacer sizes of the jumps, a comma-separated list (default "16,16"; "0": no jumps; at
    // most four).  The classes are regions of the device's memory that the driver fills in turn (r04_notes section 4):
    // a process whose first 50 GB are one class needs a long step -- bench.py asks for "16,64,150" and reports what the
    // search took.
    size_t jumps[4] = {(size_t)16 << 30, (size_t)16 << 30, 0, 0}, peak_jump = 0;
    int max_jumps = 2;
    if (const char* ev = trk_opt("TRK_PLACE_JUMP_GB")) {
        max_jumps = 0;
        for (const char* q = ev; *q && max_jumps < 4;) {
            const double gb = atof(q);
            if (gb > 0) jumps[max_jumps++] = (size_t)(gb * 1073741824.0);
            while (*q && *q != ',') ++q;
            if (*q == ',') ++q;
        }
    }
    // planes below 1 GB (a strong-scaling shard's): the neighbouring candidates only.  At 0.5 GB the candidates of one
    // process came out within 4 % of each other six times out of six (profiles/r04_bench_final.json, extras.strong_shard)
    // and a jump's spacers are 300 times the plane
    if (bytes_each < ((size_t)1 << 30) && !trk_opt("TRK_PLACE_JUMP_SMALL")) max_jumps = 0;
    size_t jump_bytes = max_jumps > 0 ? jumps[0] : 0;
    const double gbytes = 2.0 * (double)n_loci * (double)n_samples * 4.0 * 1e-9;
    int rc = TRK_OK;
    while (n < TRK_PAIR_MAX_PROBES) {
        Cand c = {nullptr, -1};
        if (1 + n < n_have) {
            c.p = have[1 + n];
            c.have = 1 + n;
        } else if (n_fresh < fresh_cap) {
            e = hipMalloc(&c.p, bytes_each);
            if (e != hipSuccess) {                    // out of memory for a spare: keep what there is
                (void)hipGetLastError();
                if (n == 0) rc = fail(ctx, TRK_ERR_NOMEM, "hipMalloc(%zu): %s", bytes_each, hipGetErrorString(e));
                break;
            }
            ++n_fresh;
        } else if (n_jumps < max_jumps && jump_bytes > 0) {
            // every neighbour is on the slow level (the driver hands a whole region out plane by plane): give the
            // losing spares back, step `jump_bytes` ahead behind a spacer that is never touched, take ONE candidate
            // there, return the spacer.  Transient: the spacer + that plane.
            for (int k = 0; k < n; ++k)
                if (k != best && cand[k].p && cand[k].have < 0) {
                    (void)hipFree(cand[k].p);
                    cand[k].p = nullptr;
                }
            jump_bytes = jumps[n_jumps];
            void* spacer = nullptr;
            if (hipMalloc(&spacer, jump_bytes) != hipSuccess) {
                (void)hipGetLastError();
                break;
            }
            e = hipMalloc(&c.p, bytes_each);
            (void)hipFree(spacer);
            if (e != hipSuccess) {
                (void)hipGetLastError();
                break;
            }
            ++n_jumps;
            if (jump_bytes + bytes_each > peak_jump) peak_jump = jump_bytes + bytes_each;
        } else {
            break;
        }
        cand[n] = c;
        e = probe_pair_ms(ctx, *a, c.p, n_loci, n_samples, &ms[n]);
        ++n;
        if (e != hipSuccess) { rc = fail(ctx, TRK_ERR_HIP, "trk_dev_alloc_pair probe: %s", hipGetErrorString(e)); break; }
        if (best < 0 || ms[n - 1] < ms[best]) best = n - 1;
        float worst = ms[0];
        for (int k = 1; k < n; ++k) worst = ms[k] > worst ? ms[k] : worst;
        const double tbps = gbytes / (double)ms[best];            // GB / ms = TB/s
        if (worst >= 1.06f * ms[best] || tbps >= TRK_PAIR_FAST_TBPS) { pi.placed = 1; break; }
    }
    pi.peak_extra_bytes = (uint64_t)(n_fresh > 1 ? n_fresh - 1 : 0) * (uint64_t)bytes_each;
    if (peak_jump > pi.peak_extra_bytes) pi.peak_extra_bytes = peak_jump;
    pi.n_jumps = n_jumps;
    if (rc == TRK_OK && best >= 0) {
        *b = cand[best].p;
        pi.have_b = cand[best].have;
        pi.kept_ms = ms[best];
        cand[best].p = nullptr;
    }
    for (int k = 0; k < n; ++k)
        if (cand[k].p && cand[k].have < 0) (void)hipFree(cand[k].p);      // (the caller's own planes stay the caller's)
    if (rc != TRK_OK) {
        if (*b && pi.have_b < 0) (void)hipFree(*b);
        if (pi.have_a < 0) (void)hipFree(*a);
        *a = *b = nullptr;
        pi.have_a = pi.have_b = -1;
    }
    pi.n_probed = n;
    for (int k = 0; k < n; ++k) pi.probe_ms[k] = ms[k];
    pi.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (info) *info = pi;
    return rc;
}

int trk_device_clocks(trk_ctx* ctx, int32_t* sclk_khz, int32_t* mclk_khz, int32_t* mem_bus_bits) {
    if (!ctx) return TRK_ERR_ARG;
    int v = 0;
    if (sclk_khz) { HIPCHK(ctx, hipDeviceGetAttribute(&v, hipDeviceAttributeClockRate, ctx->device)); *sclk_khz = v; }
    if (mclk_khz) { HIPCHK(ctx, hipDeviceGetAttribute(&v, hipDeviceAttributeMemoryClockRate, ctx->device)); *mclk_khz = v; }
    if (mem_bus_bits) { HIPCHK(ctx, hipDeviceGetAttribute(&v, hipDeviceAttributeMemoryBusWidth, ctx->device)); *mem_bus_bits = v; }
    return TRK_OK;
}

double trk_student_t_two_sided(double t, double df) { return trkmath::student_t_two_sided(t, df); }

double trk_binomtest_two_sided(int64_t k, int64_t n, double p) {
    if (n < 1 || k < 0 || k > n || !(p >= 0.0 && p <= 1.0)) return std::nan("");
    return trkmath::binomtest_two_sided(k, n, p);
}
double trk_binom_pmf(int64_t k, int64_t n, double p) { return trkmath::binom_pmf(k, n, p); }

int trk_binomtest_batch(trk_ctx* ctx, const int64_t* k, const int64_t* n, const double* p, int64_t count, double* out,
                        int32_t lanes) {
    if (!ctx) return TRK_ERR_ARG;
    if (count < 0 || (count > 0 && (!k || !n || !p || !out)) || (lanes != 1 && lanes != 2))
        return fail(ctx, TRK_ERR_ARG, "binomtest arguments");
    if (count == 0) return TRK_OK;
    (void)hipSetDevice(ctx->device);
    const size_t col = (size_t)count * 8;
    unsigned char* dev = nullptr;
    hipError_t e = hipMalloc(&dev, 4 * col);
    if (e != hipSuccess) return fail(ctx, TRK_ERR_NOMEM, "binomtest hipMalloc(%zu): %s", 4 * col, hipGetErrorString(e));
    int rc = TRK_OK;
    if ((e = hipMemcpyAsync(dev, k, col, hipMemcpyHostToDevice, ctx->s())) == hipSuccess &&
        (e = hipMemcpyAsync(dev + col, n, col, hipMemcpyHostToDevice, ctx->s())) == hipSuccess &&
        (e = hipMemcpyAsync(dev + 2 * col, p, col, hipMemcpyHostToDevice, ctx->s())) == hipSuccess &&
        (e = trk::launch_binomtest_batch(reinterpret_cast<int64_t*>(dev), reinterpret_cast<int64_t*>(dev + col),
                                         reinterpret_cast<double*>(dev + 2 * col), count,
                                         reinterpret_cast<double*>(dev + 3 * col), lanes, ctx->s())) == hipSuccess &&
        (e = hipMemcpyAsync(out, dev + 3 * col, col, hipMemcpyDeviceToHost, ctx->s())) == hipSuccess)
        e = hipStreamSynchronize(ctx->s());
    if (e != hipSuccess) rc = fail(ctx, TRK_ERR_HIP, "trk_binomtest_batch: %s", hipGetErrorString(e));
    (void)hipFree(dev);
    return rc;
}

}  // extern "C"
