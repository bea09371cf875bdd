// fmhip_comm.hip — the data-parallel step of libfmhip.so: rows sharded over one process per GPU, the
// packed gradient summed across ranks with RCCL (xGMI on an MI355X node) INSIDE the library.
//
// The reference's learner does its own reduction inside `learn` — `error.reduce(_+_)`
// (S/fm/lib/ALS.scala:153), `collectAsMap` (:34, :139) — and the driver just calls
// `fm = fml.learn(fm, dataset)` (S/fm/impl/FactorizationMachines.scala:45).  Same here: a JVM-side
// `HipSGD.learn` on N GPUs calls fmhip_dp_epoch on every rank; the exchange is not its business.
//
// RCCL is bound at run time (dlopen of librccl.so.1), not at link time: single-GPU users never load it,
// and inside a process that already carries a copy (PyTorch bundles one under the same soname) the
// loader hands back that copy instead of a second runtime.
//
// Schedule of one step (two streams, no host synchronisation):
//   compute stream  forward | backward(cold ids) | backward(hot ids) + statistics | wait 1 | apply(cold rows) | wait 2 | apply(hot rows, w0)
//   comm stream     rows    |                    | all-reduce(cold slice)         | all-reduce(hot slice + scalars)
// (a slice = the G_V, G_w and G_b rows of a feature interval, one grouped call; the row count |B| is exchanged at the
// start of the step so that an interval can be applied as soon as ITS slice has arrived, beside the later slices)
// The CSC stream is sorted by feature id, so the backward can deliver the gradient rows of an interval of
// ids at a time; the cold interval is nearly all of the gradient's volume and under half of the work.
// More cuts give a deeper pipeline (the collective of interval i runs beside the backward of interval i+1).
//
// FMHIP_EXCHANGE_SHARDED — the same exchange with the update sharded as well:
//   compute stream  forward | backward(int n-1) | backward(int n-2) | ... | backward(int 0) + statistics | wait(comm stream)
//   comm stream     rows    |                   | RS(n-1) upd(n-1) AG(n-1) | RS(n-2) upd(n-2) AG(n-2) ... | RS(0) upd(0) AG(0)
// RS(i) = reduce-scatter of interval i's G_V rows (+ all-reduce of its G_w / G_b entries, 1/Kp of the bytes, one grouped
// call); upd(i) = ONE launch: this rank's 1/world share of the interval's V rows updated, all of its w stepped, the other
// shares' gradient rows zeroed; AG(i) = all-gather of the updated V rows into every replica.  RS + AG moves the bytes of
// the all-reduce it replaces; the update and the zeroing — 4x the model's bytes on EVERY rank in the dense mode — shrink
// to 1/world + 1x.  All three pieces of an interval sit on the comm stream, in order: no event hops between them (on
// streams of their own they cost the host ~20 API calls per interval and the host, not the GPU, paced the step —
// profiles/r03_experiments.md).
#include "fmhip_internal.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_select.hpp>

#include <algorithm>
#include <cstring>
#include <mutex>

using namespace fmhip;
using namespace fmhip::host;

static_assert(sizeof(ncclUniqueId) == FMHIP_UNIQUE_ID_BYTES, "fmhip.h and RCCL disagree on the unique-id size");

namespace {

struct Rccl {
    void *handle = nullptr;
    std::string why;    // why loading failed
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclReduceScatter) ReduceScatter = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
};

Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {getenv("FMHIP_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            if (!n || !*n) continue;
            r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (r.handle) break;
            r.why = dlerror();
        }
        if (!r.handle) return;
        bool ok = true;
        auto sym = [&](const char *name) {
            void *p = dlsym(r.handle, name);
            if (!p) { ok = false; r.why = std::string("missing symbol ") + name; }
            return p;
        };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
        r.ReduceScatter = reinterpret_cast<decltype(r.ReduceScatter)>(sym("ncclReduceScatter"));
        r.Broadcast = reinterpret_cast<decltype(r.Broadcast)>(sym("ncclBroadcast"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
        if (!ok) {
            dlclose(r.handle);
            r.handle = nullptr;
        }
    });
    return r;
}

int need_rccl() {
    Rccl &r = rccl();
    if (!r.handle) return fail(FMHIP_ERR_COMM, "RCCL could not be loaded (librccl.so.1): %s", r.why.c_str());
    return FMHIP_OK;
}

#define NCCL_TRY(expr)                                                                                      \
    do {                                                                                                    \
        ncclResult_t _r = (expr);                                                                           \
        if (_r != ncclSuccess)                                                                              \
            return fail(FMHIP_ERR_COMM, "%s failed: %s (%s:%d)", #expr, rccl().GetErrorString(_r), __FILE__, __LINE__); \
    } while (0)

constexpr int kMaxCuts = 7;      // == FMHIP_DP_MAX_CUTS
static_assert(kMaxCuts == FMHIP_DP_MAX_CUTS, "fmhip.h and fmhip_comm.hip disagree on the number of cuts");

// Stand-in for a collective's duration on the comm stream (fmhip_comm_emulate): one wave spins on the
// constant-rate clock (100 MHz) until `ticks` have passed.  Bounded by construction; occupies one wave of one CU.
__global__ void k_comm_delay(uint64_t ticks) {
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

// The same stand-in with the collective's FOOTPRINT on this GPU (fmhip_comm_emulate_load): `gridDim.x` workgroups stream the
// payload through HBM `rounds` times — read, write back unchanged (nobody else touches a slice while it is "exchanged") — at a
// pace that makes the whole take `ticks`: a ring all-reduce reads and writes about 2(N-1)/N x the payload twice on every rank
// and keeps a few dozen workgroups resident while it does, which is what takes CU slots and memory bandwidth from the backward
// running beside it.  Bounded: the chunk count is finite and every wait ends by the clock.
typedef float f32x4_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_comm_traffic(f32x4_t *buf, size_t n4, uint64_t ticks, int rounds) {
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    const size_t per = (n4 + gridDim.x - 1) / gridDim.x;
    const size_t lo = (size_t)blockIdx.x * per < n4 ? (size_t)blockIdx.x * per : n4;
    const size_t hi = lo + per < n4 ? lo + per : n4;
    constexpr size_t kChunk = 256 * 16;                     // float4 per workgroup and chunk: 64 KB
    const size_t chunks = ((hi - lo + kChunk - 1) / kChunk) * (size_t)rounds;
    size_t done = 0;
    for (int r = 0; r < rounds; ++r)
        for (size_t base = lo; base < hi; base += kChunk) {
            const size_t end = base + kChunk < hi ? base + kChunk : hi;
            for (size_t i = base + threadIdx.x; i < end; i += 256) {
                const f32x4_t v = __builtin_nontemporal_load(buf + i);
                __builtin_nontemporal_store(v, buf + i);
            }
            ++done;
            const uint64_t due = chunks ? ticks * done / chunks : ticks;
            while (__builtin_amdgcn_s_memrealtime() - t0 < due) __builtin_amdgcn_s_sleep(8);
        }
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

// ---- fmhip_comm_selftest: known patterns through every collective kind the step uses
// pattern value of element i on rank r (small integers: every sum below is exact in fp32)
__device__ __forceinline__ float st_pat(int r, size_t i) { return (float)((r + 1) * (int)(1 + i % 7)); }

// what = 0: this rank's contribution to a sum (SUM_F32, REDUCE_SCATTER_F32); 1: its own segment of an all-gather of floats
// (the others' poisoned); 2: the same for int32
__global__ __launch_bounds__(256) void k_st_fill(void *buf, size_t n, size_t seg, int rank, int what) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (what == 0) { static_cast<float *>(buf)[i] = st_pat(rank, i); return; }
    const bool mine = i / seg == (size_t)rank;
    if (what == 1) static_cast<float *>(buf)[i] = mine ? (float)rank + 0.5f + (float)(i % 3) : -1.0f;
    else static_cast<int32_t *>(buf)[i] = mine ? (int32_t)(rank * 100000 + (int32_t)(i % seg)) : -1;
}

// counts the elements of [lo, hi) that differ from what the collective must have left there
__global__ __launch_bounds__(256) void k_st_check(const void *buf, size_t lo, size_t hi, size_t seg, int world, int what, unsigned *bad) {
    const size_t i = lo + (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= hi) return;
    bool ok;
    if (what == 0) {
        ok = static_cast<const float *>(buf)[i] == (float)(world * (world + 1) / 2 * (int)(1 + i % 7));
    } else if (what == 1) {
        ok = static_cast<const float *>(buf)[i] == (float)(i / seg) + 0.5f + (float)(i % 3);
    } else {
        ok = static_cast<const int32_t *>(buf)[i] == (int32_t)((i / seg) * 100000 + i % seg);
    }
    if (!ok) atomicAdd(bad, 1u);
}

__global__ void k_st_i64(int64_t *p, int rank) {
    p[0] = rank; p[1] = -(int64_t)rank; p[2] = 5;        // MAX -> world - 1, 0, 5
    p[4] = 10 * (int64_t)rank + 3; p[5] = 77 - rank;     // BCAST0 -> 3, 77
}
__global__ void k_st_i64_check(const int64_t *p, int world, unsigned *bad_max, unsigned *bad_bcast) {
    if (p[0] != world - 1 || p[1] != 0 || p[2] != 5) atomicAdd(bad_max, 1u);
    if (p[4] != 3 || p[5] != 77) atomicAdd(bad_bcast, 1u);
}

// the row count of this rank's mini-batch travels as a kernel argument (a host buffer would have to outlive the
// asynchronous copy, and the host runs steps ahead of the stream)
__global__ void k_set_float(float *p, float v) { *p = v; }

// ---- touched-rows exchange (fmhip_dp_exchange): the kernels around the collectives
// this rank's segment of the id table: the batch's distinct features, the hot block's, then -1 up to `cap`
__global__ __launch_bounds__(256) void k_fill_ids(int32_t *dst, const int32_t *feat, int32_t n_feat, const int32_t *hot, int32_t n_hot,
                                                  int64_t cap) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= cap) return;
    int32_t v = -1;
    if (i < n_feat) v = feat[i];
    else if (i < (int64_t)n_feat + n_hot) v = hot[i - n_feat];
    dst[i] = v;
}

// pos[i] = the position of ids[i] in the sorted, duplicate-free table u[0..n_u) (ids < 0: -1).  Every id is in the table by
// construction (the table is the union of every rank's ids, this rank's among them).
__global__ __launch_bounds__(256) void k_positions(const int32_t *ids, int32_t n, const int32_t *u, int32_t n_u, int32_t *pos) {
    const int32_t i = (int32_t)(blockIdx.x * 256 + threadIdx.x);
    if (i >= n) return;
    const int32_t id = ids[i];
    if (id < 0) { pos[i] = -1; return; }
    int32_t lo = 0, hi = n_u;
    while (lo < hi) {
        const int32_t mid = (lo + hi) >> 1;
        if (u[mid] < id) lo = mid + 1;
        else hi = mid;
    }
    pos[i] = lo;
}

// pos[i] = lower bound of cuts[i] in u[0..*n_u): where the feature interval that starts at cuts[i] begins in the compact gradient
__global__ void k_cut_positions(const int32_t *cuts, int32_t n, const int32_t *u, const int32_t *n_u, int32_t *pos) {
    const int32_t i = (int32_t)threadIdx.x;
    if (i >= n) return;
    int32_t lo = 0, hi = *n_u;
    while (lo < hi) {
        const int32_t mid = (lo + hi) >> 1;
        if (u[mid] < cuts[i]) lo = mid + 1;
        else hi = mid;
    }
    pos[i] = lo;
}

// The event set of ONE profiled step.  Sets are created in fmhip_comm_profile_begin (a pool), never inside a step: an
// event creation costs the host tens of microseconds, and the pass that uses them is the one that measures what the
// exchange leaves exposed.
constexpr int kProfColl = 2 * (kMaxCuts + 1) + 2;     // sharded mode: a reduce-scatter and an all-gather per interval
constexpr int kProfSteps = 32;                        // steps a profiling pass can record (later ones go unrecorded)
struct CommProf {
    hipEvent_t wait_a = nullptr, wait_b = nullptr;   // compute stream: around its wait for the last collective
    hipEvent_t top_a = nullptr, top_b = nullptr;     // pipelined mode: around the wait for the PREVIOUS step's top slice (has_top)
    bool has_top = false;
    hipEvent_t c0[kProfColl] = {}, c1[kProfColl] = {};         // comm stream: around each collective
    hipEvent_t a0[kMaxCuts + 1] = {}, a1[kMaxCuts + 1] = {};   // compute stream: around each interval's update
    int n_coll = 0, n_apply = 0;
    bool used = false;
};

}  // namespace

struct fmhip_comm {
    int device = 0, rank = 0, world = 1;
    ncclComm_t comm = nullptr;                // RCCL communicator, or ...
    fmhip_collective_fn ext = nullptr;        // ... the caller's own transport (fmhip_comm_create_external)
    void *ext_ctx = nullptr;
    int exchange = FMHIP_EXCHANGE_DENSE;
    // touched-rows exchange (fmhip_dp_exchange).  The mini-batches of a dataset are fixed, so the rows a lock-step step
    // touches on ANY rank — the union of every rank's batch t — are known when the plan is made: fmhip_dp_plan forms every
    // step's union once (all-gather of the ids, sort, unique) together with the position of each of this rank's columns in
    // it, and a step's gradient is written straight into a COMPACT buffer with one row per union feature (GradView).  A
    // step then has no id exchange, no sort, no pack / unpack and no host read-back: backward -> all-reduce of the compact
    // buffer (its size is known on the host) -> rows-only update.
    struct TStep {
        int32_t *uni = nullptr;               // device: the union's feature ids, ascending (a leading -1 = padding)
        int32_t n_u = 0;
        int32_t *cdst = nullptr;              // device: per compressed column of this rank's batch t, its row in the compact buffer
        int32_t *hot_pos = nullptr;           // device: the same for the slots of the dense hot block (-1 = unused)
        int32_t cut_pos[kMaxCuts + 1] = {};   // where the plan's cuts (c->cuts, ascending) fall in the union: an interval of ids = a slice of rows
    };
    std::vector<TStep> tsteps;                // one per lock-step step of an epoch
    int64_t t_cursor = 0;                     // the next step of the planned schedule
    int64_t cap = 0;                          // id slots per rank in the plan's all-gathers: the largest per-batch id count of any rank
    float *cg = nullptr;                      // device: the compact gradient [ scalars (kGradHead) | G_w | G_b | pad | G_V rows ]
    size_t cg_floats = 0;
    int msg_kp = 0;                           // the padded factor count the plan was made for
    const void *planned_data = nullptr;       // ... and the dataset
    hipStream_t cs = nullptr;                 // the collectives' stream
    hipEvent_t ev_ready[kMaxCuts + 1] = {};   // compute stream: interval i of the gradient is final
    hipEvent_t ev_done[kMaxCuts + 1] = {};    // comm stream: interval i's slice has been exchanged
    hipEvent_t ev_rows = nullptr;             // compute stream: this rank's row count is in place
    float *rows_dev = nullptr;                // device float: the step's global row count |B| (exchanged first)
    std::vector<int64_t> cuts;                // ascending feature ids in (0, n+1) cutting the backward into intervals (empty: one collective)
    int64_t *scratch = nullptr;               // device int64[kMaxCuts + 1] for the small control collectives
    double emu_bytes_per_us = 0.0;            // > 0: every collective is followed by a delay of bytes / this (fmhip_comm_emulate)
    int emu_wgs = 0;                          // > 0: the delay is spent by this many workgroups streaming the payload (fmhip_comm_emulate_load)
    bool profiling = false;
    std::vector<CommProf> prof;               // the pool of event sets (fmhip_comm_profile_begin)
    size_t prof_next = 0;                     // sets handed out since _begin
    int64_t prof_bytes = 0;
    // sharded update (FMHIP_EXCHANGE_SHARDED)
    hipEvent_t ev_gathered = nullptr;         // comm stream: behind the last all-gather of a step
    int emu_ranks = 0;                        // > 0: one real rank plays rank 0 of this many (fmhip_comm_emulate_ranks)
    // agreed by fmhip_dp_plan over all ranks: the largest mini-batch of any rank (rows), so that every size check of a
    // step passes or fails on every rank alike
    int64_t plan_max_rows = -1;
    int64_t plan_steps = 0;                   // ... and the largest batch count: the lock-step steps of an epoch
};

namespace {

void destroy_events(CommProf &p) {
    for (hipEvent_t e : {p.wait_a, p.wait_b, p.top_a, p.top_b})
        if (e) (void)hipEventDestroy(e);
    for (int i = 0; i < kProfColl; ++i)
        for (hipEvent_t e : {p.c0[i], p.c1[i]})
            if (e) (void)hipEventDestroy(e);
    for (int i = 0; i <= kMaxCuts; ++i)
        for (hipEvent_t e : {p.a0[i], p.a1[i]})
            if (e) (void)hipEventDestroy(e);
    p = CommProf();
}

int create_events(CommProf &p) {
    hipError_t e = hipEventCreate(&p.wait_a);
    if (e == hipSuccess) e = hipEventCreate(&p.wait_b);
    if (e == hipSuccess) e = hipEventCreate(&p.top_a);
    if (e == hipSuccess) e = hipEventCreate(&p.top_b);
    for (int i = 0; i < kProfColl && e == hipSuccess; ++i) {
        e = hipEventCreate(&p.c0[i]);
        if (e == hipSuccess) e = hipEventCreate(&p.c1[i]);
    }
    for (int i = 0; i <= kMaxCuts && e == hipSuccess; ++i) {
        e = hipEventCreate(&p.a0[i]);
        if (e == hipSuccess) e = hipEventCreate(&p.a1[i]);
    }
    if (e != hipSuccess) return fail(FMHIP_ERR_HIP, "profiling events: %s", hipGetErrorString(e));
    return FMHIP_OK;
}

// the event set of the step being enqueued (NULL: not profiling, or the pool is used up)
CommProf *next_prof(fmhip_comm_t c) {
    if (!c->profiling || c->prof_next >= c->prof.size()) return nullptr;
    CommProf *p = &c->prof[c->prof_next++];
    p->used = true;
    p->has_top = false;
    p->n_coll = p->n_apply = 0;
    return p;
}

// one collective on `s`: through RCCL, or handed to the caller's transport
int collective(fmhip_comm_t c, void *buf, size_t count, int kind, hipStream_t s) {
    if (c->ext) {
        const int rc = c->ext(c->ext_ctx, buf, count, kind, reinterpret_cast<void *>(s));
        if (rc != 0) return fail(FMHIP_ERR_COMM, "the caller's collective (kind %d, %zu elements) returned %d", kind, count, rc);
        return FMHIP_OK;
    }
    switch (kind) {
        case FMHIP_COLL_SUM_F32: NCCL_TRY(rccl().AllReduce(buf, buf, count, ncclFloat, ncclSum, c->comm, s)); break;
        case FMHIP_COLL_MAX_I64: NCCL_TRY(rccl().AllReduce(buf, buf, count, ncclInt64, ncclMax, c->comm, s)); break;
        case FMHIP_COLL_BCAST0_I64: NCCL_TRY(rccl().Broadcast(buf, buf, count, ncclInt64, 0, c->comm, s)); break;
        case FMHIP_COLL_ALLGATHER_I32:     // in place: rank r's `count` elements already sit at buf + r * count
            NCCL_TRY(rccl().AllGather(static_cast<int32_t *>(buf) + (size_t)c->rank * count, buf, count, ncclInt32, c->comm, s));
            break;
        case FMHIP_COLL_REDUCE_SCATTER_F32:  // in place: rank r receives the sum of segment r where it already lies
            NCCL_TRY(rccl().ReduceScatter(buf, static_cast<float *>(buf) + (size_t)c->rank * count, count, ncclFloat, ncclSum, c->comm, s));
            break;
        case FMHIP_COLL_ALLGATHER_F32:
            NCCL_TRY(rccl().AllGather(static_cast<float *>(buf) + (size_t)c->rank * count, buf, count, ncclFloat, c->comm, s));
            break;
        default: return fail(FMHIP_ERR_INVALID, "unknown collective kind %d", kind);
    }
    return FMHIP_OK;
}

// holds stream `s` for the time `bytes` would take at the emulated payload rate (fmhip_comm_emulate)
// buf / floats: the payload itself (16-byte aligned), for the emulation with a footprint; rounds: passes over it (2 = an
// all-reduce, 1 = a reduce-scatter or an all-gather)
int emu_delay(fmhip_comm_t c, double bytes, hipStream_t s, float *buf = nullptr, size_t floats = 0, int rounds = 2) {
    if (c->emu_bytes_per_us <= 0.0 || bytes <= 0.0) return FMHIP_OK;
    const uint64_t ticks = (uint64_t)(bytes / c->emu_bytes_per_us * 100.0);
    if (c->emu_wgs > 0 && buf && floats >= 4 && (reinterpret_cast<uintptr_t>(buf) & 15u) == 0)
        hipLaunchKernelGGL(k_comm_traffic, dim3((unsigned)c->emu_wgs), dim3(256), 0, s, reinterpret_cast<f32x4_t *>(buf), floats / 4, ticks, rounds);
    else
        hipLaunchKernelGGL(k_comm_delay, dim3(1), dim3(64), 0, s, ticks);
    HIP_TRY(hipGetLastError());
    return FMHIP_OK;
}

int check_comm(fmhip_model_t m, fmhip_comm_t c) {
    if (!m || !c) return fail(FMHIP_ERR_INVALID, "model or communicator is NULL");
    if (m->device != c->device) return fail(FMHIP_ERR_INVALID, "model on device %d, communicator on device %d", m->device, c->device);
    return set_device(m->device);
}

struct Region {
    float *p;
    size_t n;
};

// The all-reduce of up to three regions of the packed gradient (one grouped call) on the comm stream, behind `after`
// (an event of the compute stream); `done` is recorded behind it.
int reduce_regions(fmhip_model_t m, fmhip_comm_t c, const Region *reg, int n_reg, hipEvent_t after, hipEvent_t done, CommProf *pr) {
    HIP_TRY(hipEventRecord(after, m->stream));
    HIP_TRY(hipStreamWaitEvent(c->cs, after, 0));
    int pi = -1;
    if (pr && pr->n_coll < kProfColl) {
        pi = pr->n_coll++;
        HIP_TRY(hipEventRecord(pr->c0[pi], c->cs));
    }
    size_t bytes = 0;
    const bool group = n_reg > 1 && !c->ext;      // RCCL: the regions of an interval travel as one grouped call
    if (group) NCCL_TRY(rccl().GroupStart());
    for (int i = 0; i < n_reg; ++i) {
        if (!reg[i].n) continue;
        TRY(collective(c, reg[i].p, reg[i].n, FMHIP_COLL_SUM_F32, c->cs));
        bytes += reg[i].n * sizeof(float);
    }
    if (group) NCCL_TRY(rccl().GroupEnd());
    if (c->emu_wgs > 0) {
        // a grouped call is ONE kernel on the wire's stream: the whole call's duration, its footprint on the largest region (the
        // G_V rows: 32 of every 34 floats).  (One stand-in per region — three launches, two of them a few microseconds long —
        // put ~16 us of launch latency into every slice that a grouped RCCL call does not have: r05_experiments.md section 7c)
        int big = 0;
        for (int i = 1; i < n_reg; ++i)
            if (reg[i].n > reg[big].n) big = i;
        TRY(emu_delay(c, (double)bytes, c->cs, reg[big].p, reg[big].n));
    } else {
        TRY(emu_delay(c, (double)bytes, c->cs));
    }
    if (pi >= 0) HIP_TRY(hipEventRecord(pr->c1[pi], c->cs));
    HIP_TRY(hipEventRecord(done, c->cs));
    c->prof_bytes += c->profiling ? (int64_t)bytes : 0;
    return FMHIP_OK;
}

void free_touched(fmhip_comm_t c) {
    for (auto &t : c->tsteps)
        for (void *p : {(void *)t.uni, (void *)t.cdst, (void *)t.hot_pos})
            if (p) (void)hipFree(p);
    c->tsteps.clear();
    if (c->cg) (void)hipFree(c->cg);
    c->cg = nullptr;
    c->cg_floats = 0;
    c->planned_data = nullptr;
    c->t_cursor = 0;
}

// compact-gradient layout for a union of n_u rows: offsets of G_w, G_b, G_V and the total, in floats
struct CompactLayout {
    size_t n_up, gw, gb, gv, total;
    CompactLayout(int32_t n_u, int kp) {
        n_up = ((size_t)n_u + 3) & ~(size_t)3;
        gw = (size_t)kGradHead;
        gb = gw + n_up;
        gv = (gb + n_up + 31) / 32 * 32;
        total = gv + n_up * (size_t)kp;
    }
};

// small control collectives (a count, a cut) through a device scratch word
int control_i64(fmhip_model_t m, fmhip_comm_t c, int64_t *value, int count, bool broadcast_from_0) {
    HIP_TRY(hipMemcpyAsync(c->scratch, value, count * sizeof(int64_t), hipMemcpyHostToDevice, m->stream));
    TRY(collective(c, c->scratch, (size_t)count, broadcast_from_0 ? FMHIP_COLL_BCAST0_I64 : FMHIP_COLL_MAX_I64, m->stream));
    HIP_TRY(hipMemcpyAsync(value, c->scratch, count * sizeof(int64_t), hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    return FMHIP_OK;
}

// The plan of the touched-rows exchange (collective): for every position t < steps of the lock-step schedule, the union of the
// rows the ranks' batches t touch, where this rank's columns lie in it, and where the plan's cuts fall in it.  Plan-time
// work: one all-gather, one sort and one small read-back per position — what every STEP used to pay.
// A failure that only THIS rank sees (an allocation, a sort) must not leave the peers inside the next all-gather: the rank
// keeps taking part in the collectives, skips its own work, and all ranks agree on the outcome at the end.
int plan_touched(fmhip_model_t m, fmhip_dataset_t d, fmhip_comm_t c, int64_t cap, int64_t steps) {
    free_touched(c);
    c->cap = cap;
    c->msg_kp = m->Kp;
    const size_t n = (size_t)c->world * (size_t)cap;
    // sized alike on every rank (cap, world and steps are agreed values): passes or fails everywhere
    if (n > (size_t)INT32_MAX) return fail(FMHIP_ERR_UNSUPPORTED, "%zu id slots exceed the touched-rows exchange's 2^31 limit", n);
    DevBuf<int32_t> ids, sorted, uniq, n_uniq, dcuts, dpos;
    DevBuf<uint8_t> tmp;
    size_t tmp_bytes = 0;
    const int n_cuts = (int)c->cuts.size();
    auto scratch = [&]() -> int {
        size_t ta = 0, tb = 0;
        int32_t *kq = nullptr;
        HIP_TRY(rocprim::radix_sort_keys(nullptr, ta, kq, kq, n, 0, 32, m->stream));
        HIP_TRY(rocprim::unique(nullptr, tb, kq, kq, kq, n, rocprim::equal_to<int32_t>(), m->stream));
        tmp_bytes = std::max(ta, tb);
        TRY(ids.alloc(std::max<size_t>(n, 1)));
        TRY(sorted.alloc(std::max<size_t>(n, 1)));
        TRY(uniq.alloc(std::max<size_t>(n, 1)));
        TRY(n_uniq.alloc(1));
        TRY(tmp.alloc(tmp_bytes + 16));
        TRY(dcuts.alloc(kMaxCuts + 1));
        TRY(dpos.alloc(kMaxCuts + 1));
        int32_t hc[kMaxCuts + 1] = {};
        for (int i = 0; i < n_cuts; ++i) hc[i] = (int32_t)c->cuts[(size_t)i];
        HIP_TRY(hipMemcpyAsync(dcuts.p, hc, sizeof hc, hipMemcpyHostToDevice, m->stream));
        HIP_TRY(hipStreamSynchronize(m->stream));
        return FMHIP_OK;
    };
    int64_t bad = scratch() != FMHIP_OK;
    std::string why = bad ? fmhip_last_error() : "";
    {
        // without the id table there is nothing to all-gather into: agreed BEFORE the first collective that needs it
        int64_t flag = bad;
        TRY(control_i64(m, c, &flag, 1, false));
        if (flag) return fail(FMHIP_ERR_NOMEM, "%s", why.empty() ? "another rank could not allocate the touched-rows plan's scratch" : why.c_str());
    }
    const int64_t nb = (int64_t)d->batches.size();
    const int32_t n_hot = d->hot_pages * kHotT;
    int32_t max_nu = 0;
    c->tsteps.resize((size_t)steps);
    for (int64_t t = 0; t < steps; ++t) {
        auto &ts = c->tsteps[(size_t)t];
        const bool live = t < nb;
        const int32_t *feat = live ? d->cfeat.p + d->batches[(size_t)t].col_off : nullptr;
        const int32_t n_feat = live ? d->batches[(size_t)t].n_cols : 0;
        hipLaunchKernelGGL(k_fill_ids, dim3((unsigned)((cap + 255) / 256)), dim3(256), 0, m->stream, ids.p + (size_t)c->rank * cap, feat, n_feat,
                           live ? d->d_hot_ids.p : nullptr, live ? n_hot : 0, cap);
        // the collective itself: every rank, every position, whatever happened locally (a transport error IS collective)
        TRY(collective(c, ids.p, (size_t)cap, FMHIP_COLL_ALLGATHER_I32, m->stream));
        if (bad) continue;
        auto local = [&]() -> int {
            HIP_TRY(hipGetLastError());
            size_t tbytes = tmp_bytes;
            HIP_TRY(rocprim::radix_sort_keys(tmp.p, tbytes, ids.p, sorted.p, n, 0, 32, m->stream));
            tbytes = tmp_bytes;
            HIP_TRY(rocprim::unique(tmp.p, tbytes, sorted.p, uniq.p, n_uniq.p, n, rocprim::equal_to<int32_t>(), m->stream));
            if (n_cuts) {
                hipLaunchKernelGGL(k_cut_positions, dim3(1), dim3(64), 0, m->stream, dcuts.p, n_cuts, uniq.p, n_uniq.p, dpos.p);
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipMemcpyAsync(ts.cut_pos, dpos.p, (size_t)n_cuts * sizeof(int32_t), hipMemcpyDeviceToHost, m->stream));
            }
            HIP_TRY(hipMemcpyAsync(&ts.n_u, n_uniq.p, sizeof(int32_t), hipMemcpyDeviceToHost, m->stream));
            HIP_TRY(hipStreamSynchronize(m->stream));
            max_nu = std::max(max_nu, ts.n_u);
            HIP_TRY(hipMalloc(reinterpret_cast<void **>(&ts.uni), std::max<size_t>((size_t)ts.n_u, 1) * sizeof(int32_t)));
            HIP_TRY(hipMemcpyAsync(ts.uni, uniq.p, (size_t)ts.n_u * sizeof(int32_t), hipMemcpyDeviceToDevice, m->stream));
            if (live) {
                HIP_TRY(hipMalloc(reinterpret_cast<void **>(&ts.cdst), std::max<size_t>((size_t)n_feat, 1) * sizeof(int32_t)));
                HIP_TRY(hipMalloc(reinterpret_cast<void **>(&ts.hot_pos), std::max<size_t>((size_t)n_hot, 1) * sizeof(int32_t)));
                if (n_feat) {
                    hipLaunchKernelGGL(k_positions, dim3((unsigned)((n_feat + 255) / 256)), dim3(256), 0, m->stream, feat, n_feat, ts.uni, ts.n_u, ts.cdst);
                    HIP_TRY(hipGetLastError());
                }
                if (n_hot) {
                    hipLaunchKernelGGL(k_positions, dim3((unsigned)((n_hot + 255) / 256)), dim3(256), 0, m->stream, d->d_hot_ids.p, n_hot, ts.uni, ts.n_u,
                                       ts.hot_pos);
                    HIP_TRY(hipGetLastError());
                }
            }
            return FMHIP_OK;
        };
        if (local() != FMHIP_OK) {
            bad = 1;
            why = fmhip_last_error();
        }
    }
    if (!bad) {
        auto finish = [&]() -> int {
            c->cg_floats = CompactLayout(max_nu, m->Kp).total;
            HIP_TRY(hipMalloc(reinterpret_cast<void **>(&c->cg), c->cg_floats * sizeof(float)));
            HIP_TRY(hipMemsetAsync(c->cg, 0, c->cg_floats * sizeof(float), m->stream));
            HIP_TRY(hipStreamSynchronize(m->stream));
            return FMHIP_OK;
        };
        if (finish() != FMHIP_OK) {
            bad = 1;
            why = fmhip_last_error();
        }
    }
    TRY(control_i64(m, c, &bad, 1, false));       // every rank leaves with the same verdict
    if (bad) {
        free_touched(c);
        return fail(FMHIP_ERR_HIP, "%s", why.empty() ? "the touched-rows plan failed on another rank" : why.c_str());
    }
    c->planned_data = d;
    return FMHIP_OK;
}

// One data-parallel step that exchanges only the gradient rows some rank touched (models far wider than a global batch:
// C5's 2^25 x 64 table moves 8.9 GB per dense all-reduce and a few percent of that here).  Position t of the planned schedule:
//   compute stream  forward | backward(cold ids) -> compact rows | backward(next) ... + statistics | wait 1 | update(slice 1) | ...
//   comm stream     |B|     |                                    | all-reduce(slice 1)            | all-reduce(slice 2) ...
// — the dense step's schedule with a COMPACT gradient: row j of the buffer belongs to feature U_t[j], U_t ascending, so the
// feature interval [cut_i, cut_i+1) is the contiguous slice of rows [cut_pos_i, cut_pos_i+1) and its all-reduce runs beside
// the backward of the next interval; an interval gets the rows-only update (lazy weight decay: the tables' scale moves with
// the last slice) as soon as its slice has arrived.  Nothing is sized on the device: no read-back, no host synchronisation.
// Replicas stay bit-identical: same U, same sums, same update.
int dp_step_touched(fmhip_model_t m, fmhip_dataset_t d, int64_t batch, fmhip_comm_t c, double eta, double reg0, double regw, double regv,
                    int64_t position) {
    const bool live = batch >= 0;
    // (no plan at all is a state every rank shares — the plan is collective — so this returns on all of them alike; a model
    // of another width than the plan's can only be THIS rank's mistake: local_checks turns it into a zero contribution,
    // sized by the plan's width, whose update is skipped)
    if (c->tsteps.empty() || !c->cg)
        return fail(FMHIP_ERR_INVALID, "the touched-rows exchange is not planned: call fmhip_dp_plan (every rank)");
    const bool foreign = c->msg_kp != m->Kp;
    if (foreign && live) return fail(FMHIP_ERR_INVALID, "the touched-rows exchange was planned for rows of %d floats, this model has %d", c->msg_kp, m->Kp);
    if (!lazy_decay_ok(m, eta, regw, regv))
        return fail(FMHIP_ERR_UNSUPPORTED, "the touched-rows exchange needs weight decay that fits the tables' scale (0.5 <= 1 - eta*reg <= 1)");
    const bool packed_dirty = m->grad_dirty;      // this step neither writes nor cleans the model's packed gradient
    const int64_t t = position >= 0 ? position : c->t_cursor;
    if (t >= (int64_t)c->tsteps.size())
        return fail(FMHIP_ERR_INVALID, "position %lld outside the planned schedule of %zu steps", (long long)t, c->tsteps.size());
    const auto &ts = c->tsteps[(size_t)t];
    const CompactLayout L(ts.n_u, c->msg_kp);
    const GradView view{c->cg, c->cg + L.gw, c->cg + L.gb, c->cg + L.gv, ts.cdst, ts.hot_pos};
    const float my_rows = live ? (float)d->batches[(size_t)batch].rows : 0.f;
    hipLaunchKernelGGL(k_set_float, dim3(1), dim3(1), 0, m->stream, c->rows_dev, my_rows);
    HIP_TRY(hipGetLastError());
    // one interval (no cuts): nothing can overlap, so both collectives run on the compute stream itself — no event hops
    // between the streams, which cost a one-rank step ~40 us of idle time (profiles/r04_experiments.md)
    const bool serial = c->cuts.empty();
    if (serial) {
        TRY(collective(c, c->rows_dev, 1, FMHIP_COLL_SUM_F32, m->stream));
    } else {
        HIP_TRY(hipEventRecord(c->ev_rows, m->stream));
        HIP_TRY(hipStreamWaitEvent(c->cs, c->ev_rows, 0));
        TRY(collective(c, c->rows_dev, 1, FMHIP_COLL_SUM_F32, c->cs));
    }
    if (live) {
        // the packed gradient is not written by this step: a pending memset of it (8.9 GB at C5's width) would be wasted
        m->grad_dirty = false;
        const int rc = step_forward(m, d, batch);
        m->grad_dirty = packed_dirty;
        TRY(rc);
    } else {
        // out of rows: the rows of the compact buffer are clean after every update; the scalars in front of them keep the
        // last step's (already exchanged) sums and must not travel again
        HIP_TRY(hipMemsetAsync(c->cg, 0, (size_t)kGradHead * sizeof(float), m->stream));
        m->last_nnz = m->last_rows = 0;
    }
    CommProf *pr = next_prof(c);
    // intervals of feature ids [edge[i], edge[i+1]) = rows [pe[i], pe[i+1]) of the compact buffer, from the top down
    std::vector<int64_t> edge{0};
    std::vector<int64_t> pe{0};
    for (size_t i = 0; i < c->cuts.size(); ++i) {
        const int64_t x = c->cuts[i];
        if (x > edge.back() && x < m->n1) {
            edge.push_back(x);
            pe.push_back(ts.cut_pos[i]);
        }
    }
    edge.push_back(m->n1);
    pe.push_back(ts.n_u);
    const int n_int = (int)edge.size() - 1;
    for (int i = n_int - 1; i >= 0; --i) {
        const bool last = i == 0;
        if (live) {
            m->view = &view;
            m->grad_dirty = false;
            const int rc = step_backward(m, d, batch, edge[(size_t)i], n_int == 1 ? INT64_MAX : edge[(size_t)i + 1], last, nullptr);
            m->view = nullptr;
            m->grad_dirty = packed_dirty;
            TRY(rc);
        }
        const size_t plo = (size_t)pe[(size_t)i], phi = (size_t)pe[(size_t)i + 1];
        if (serial) {
            int pi = -1;
            if (pr && pr->n_coll < kProfColl) {
                pi = pr->n_coll++;
                HIP_TRY(hipEventRecord(pr->c0[pi], m->stream));
            }
            TRY(collective(c, c->cg, L.total, FMHIP_COLL_SUM_F32, m->stream));
            TRY(emu_delay(c, (double)L.total * sizeof(float), m->stream, c->cg, L.total));
            if (pi >= 0) HIP_TRY(hipEventRecord(pr->c1[pi], m->stream));
            if (c->profiling) c->prof_bytes += (int64_t)(L.total * sizeof(float));
        } else if (n_int == 1) {
            const Region whole[1] = {{c->cg, L.total}};
            TRY(reduce_regions(m, c, whole, 1, c->ev_ready[i], c->ev_done[i], pr));
        } else {
            // the interval's G_V rows, G_w and G_b entries; the lowest interval's G_w region starts at the scalars in front of it
            const Region reg[3] = {{view.GV + plo * (size_t)c->msg_kp, (phi - plo) * (size_t)c->msg_kp},
                                   {last ? c->cg : view.Gw + plo, (phi - plo) + (last ? (size_t)kGradHead : 0)},
                                   {view.Gb + plo, phi - plo}};
            TRY(reduce_regions(m, c, reg, 3, c->ev_ready[i], c->ev_done[i], pr));
        }
    }
    m->bw_next_hi = -1;
    if (pr) HIP_TRY(hipEventRecord(pr->wait_a, m->stream));
    int64_t pend_hi = -1;
    for (int i = n_int - 1; i >= 0; --i) {
        const int64_t plo = pe[(size_t)i], phi = pe[(size_t)i + 1];
        if (pend_hi < 0) pend_hi = phi;
        const bool last = i == 0;
        if (!serial) HIP_TRY(hipStreamWaitEvent(m->stream, c->ev_done[i], 0));
        if (foreign) {
            // this rank only kept the peers' collectives company: the compact buffer must be clean for the next step, its
            // rows mean nothing to a model of another width
            if (last) HIP_TRY(hipMemsetAsync(c->cg, 0, L.total * sizeof(float), m->stream));
            continue;
        }
        if (!last && (pend_hi - plo) * 8 < ts.n_u) continue;      // small slices share the next one's launch
        if (last)       // the step's global sums where fmhip_step_stats / fmhip_dp_epoch read them
            HIP_TRY(hipMemcpyAsync(m->scal(), c->cg, (size_t)kScalars * sizeof(float), hipMemcpyDeviceToDevice, m->stream));
        if (pr) HIP_TRY(hipEventRecord(pr->a0[pr->n_apply++], m->stream));
        TRY(step_apply_rows(m, eta, reg0, regw, regv, ts.uni, (int32_t)(pend_hi - plo), c->rows_dev, &view, plo, last));
        if (pr) HIP_TRY(hipEventRecord(pr->a1[pr->n_apply - 1], m->stream));
        pend_hi = -1;
    }
    if (pr) HIP_TRY(hipEventRecord(pr->wait_b, m->stream));
    m->grad_dirty = packed_dirty;
    c->t_cursor = (t + 1) % (int64_t)c->tsteps.size();
    return FMHIP_OK;
}

int dp_step_dense(fmhip_model_t m, fmhip_dataset_t d, int64_t batch, fmhip_comm_t c, double eta, double reg0, double regw, double regv) {
    const bool live = batch >= 0;
    // |B| first: every interval's update divides by the GLOBAL row count, so it is exchanged on its own (4 bytes,
    // hidden under the forward) instead of waiting for the head in the last message
    const float my_rows = live ? (float)d->batches[(size_t)batch].rows : 0.f;
    hipLaunchKernelGGL(k_set_float, dim3(1), dim3(1), 0, m->stream, c->rows_dev, my_rows);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(c->ev_rows, m->stream));
    HIP_TRY(hipStreamWaitEvent(c->cs, c->ev_rows, 0));
    TRY(collective(c, c->rows_dev, 1, FMHIP_COLL_SUM_F32, c->cs));
    if (live) {
        TRY(step_forward(m, d, batch));
    } else {
        // out of rows: contribute zeros (row count 0 included)
        HIP_TRY(hipMemsetAsync(m->grad, 0, m->grad_floats() * sizeof(float), m->stream));
        m->grad_dirty = true;
        m->last_nnz = m->last_rows = 0;
    }
    CommProf *pr = next_prof(c);
    // intervals [edge[i], edge[i+1]) from the top down; the lowest one carries the statistics scalars
    const std::vector<int64_t> edge = interval_edges(c->cuts, m->n1, 0);
    const int n_int = (int)edge.size() - 1;
    for (int i = n_int - 1; i >= 0; --i) {
        const int64_t lo = edge[(size_t)i], hi = edge[(size_t)i + 1];
        const bool last = i == 0;
        if (live) TRY(step_backward(m, d, batch, lo, n_int == 1 ? INT64_MAX : hi, last, nullptr));
        if (n_int == 1) {
            const Region whole[1] = {{m->grad, m->grad_floats()}};
            TRY(reduce_regions(m, c, whole, 1, c->ev_ready[i], c->ev_done[i], pr));
        } else {
            // G_V rows, G_w and G_b of the interval; the lowest interval's G_w region starts at the scalars in front of it
            const Region reg[3] = {{m->GV() + (size_t)lo * m->Kp, (size_t)(hi - lo) * m->Kp},
                                   {last ? m->grad : m->Gw() + lo, (size_t)(hi - lo) + (last ? (size_t)kGradHead : 0)},
                                   {m->Gb() + lo, (size_t)(hi - lo)}};
            TRY(reduce_regions(m, c, reg, 3, c->ev_ready[i], c->ev_done[i], pr));
        }
    }
    m->bw_next_hi = -1;
    if (pr) HIP_TRY(hipEventRecord(pr->wait_a, m->stream));
    // update every interval as its slice arrives (identical on all ranks: replicas stay bit-identical); intervals of
    // less than an eighth of the rows wait for the next one and share its launch (a launch costs more than they do)
    int64_t pend_hi = -1;
    for (int i = n_int - 1; i >= 0; --i) {
        const int64_t lo = edge[(size_t)i], hi = edge[(size_t)i + 1];
        if (pend_hi < 0) pend_hi = hi;
        const bool last = i == 0;
        HIP_TRY(hipStreamWaitEvent(m->stream, c->ev_done[i], 0));
        if (!last && (pend_hi - lo) * 8 < m->n1) continue;
        if (pr) HIP_TRY(hipEventRecord(pr->a0[pr->n_apply++], m->stream));
        TRY(step_apply_interval(m, eta, reg0, regw, regv, lo, pend_hi, c->rows_dev, last));
        if (pr) HIP_TRY(hipEventRecord(pr->a1[pr->n_apply - 1], m->stream));
        pend_hi = -1;
    }
    if (pr) HIP_TRY(hipEventRecord(pr->wait_b, m->stream));
    return FMHIP_OK;
}

// ---- FMHIP_EXCHANGE_PIPELINED: the dense exchange with the NEXT step's forward under the coldest slice
// The dense schedule leaves the wire idle while the forward runs (no gradient row exists before every row's residual does) and
// the GPU idle while the last slices travel.  Here the COLDEST interval (most of the bytes, a few per cent of the nonzeros) is
// walked and sent LAST, the others before it from the second-coldest down to feature 0 and updated as they arrive — and while
// the coldest slice travels the forward of the NEXT position runs over every feature below the top cut (pass A of the two-pass
// forward, fm_kernels.h; the rows' entries are partitioned at that cut).  When the slice has arrived its rows are updated, pass B adds the cold
// features' terms and finishes the rows, and the next backward starts.  Same sums, same update as the dense mode; the forward's
// fp32 sums in another order.  The run knows its positions (an epoch, fmhip_dp_steps); a single fmhip_dp_step is the same step
// without the overlap.
//   compute  A(t0) | B | bwd T-1 | .. bwd 0 | bwd T | wait, apply T-1 .. 0 | A(t1) | wait T apply T | B | bwd T-1 ...
//   comm           |B|        | slice T-1 | .. slice 0    | slice T ............|       |B|   | slice T-1 ..
int dp_run_pipelined(fmhip_model_t m, fmhip_dataset_t d, fmhip_comm_t c, const int64_t *batches, int64_t n, double eta, double reg0, double regw,
                     double regv) {
    const std::vector<int64_t> edge = interval_edges(c->cuts, m->n1, 0);
    const int n_int = (int)edge.size() - 1, T = n_int - 1;
    if (n_int < 2 || m->Kp > 64) {      // nothing to pipeline: the dense step (the same collectives on every rank: cuts and width are agreed)
        for (int64_t t = 0; t < n; ++t) TRY(dp_step_dense(m, d, batches[t], c, eta, reg0, regw, regv));
        return FMHIP_OK;
    }
    bool any_live = false;
    for (int64_t t = 0; t < n; ++t) any_live = any_live || batches[t] >= 0;
    // the rows' entries partitioned at the top cut, in a copy of the stream that only the two-pass forward reads (the dataset's
    // own streams never move: other threads may be scoring or training on it) — local work, no collective inside: if it fails
    // here, this rank keeps in step with zeros, as after any other local failure, and reports afterwards: no peer is left
    // waiting in a collective.  The partition is held for the length of the run (PartUse): nobody re-makes it for another cut
    // under this run's launches.
    struct PartUse {
        fmhip_dataset_t d = nullptr;
        ~PartUse() {
            if (!d) return;
            std::lock_guard<std::mutex> lock(d->part_mu);
            --d->part_users;
        }
    } part_use;
    std::vector<int64_t> zeros;
    std::string part_err;
    int part_rc = FMHIP_OK;
    if (any_live) {
        std::lock_guard<std::mutex> lock(d->part_mu);
        part_rc = partition_rows_locked(d, edge[(size_t)T]);
        if (part_rc == FMHIP_OK) {
            ++d->part_users;
            part_use.d = d;
        } else {
            part_err = fmhip_last_error();
            zeros.assign((size_t)n, -1);
            batches = zeros.data();
        }
    }
    TRY(fold_scales(m));                 // pass A of the next step runs before the last interval's update: the tables stay at scale 1
    for (int64_t t = 0; t < n; ++t) {
        const int64_t b = batches[t];
        const bool live = b >= 0;
        float *rows_t = c->rows_dev + (t & 1);          // |B| of this step (the previous step's is still needed by its last update)
        CommProf *pr = next_prof(c);
        if (t == 0 && live) TRY(step_forward_pass(m, d, b, 0));
        if (t > 0) {
            // the coldest slice of the previous step: arrived -> its rows, and w0 (the statistics came with slice 0)
            if (pr) HIP_TRY(hipEventRecord(pr->top_a, m->stream));
            HIP_TRY(hipStreamWaitEvent(m->stream, c->ev_done[T], 0));
            if (pr) { HIP_TRY(hipEventRecord(pr->top_b, m->stream)); pr->has_top = true; }
            TRY(step_apply_interval(m, eta, reg0, regw, regv, edge[(size_t)T], edge[(size_t)T + 1], c->rows_dev + ((t - 1) & 1), true));
        }
        hipLaunchKernelGGL(k_set_float, dim3(1), dim3(1), 0, m->stream, rows_t, live ? (float)d->batches[(size_t)b].rows : 0.f);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(c->ev_rows, m->stream));
        HIP_TRY(hipStreamWaitEvent(c->cs, c->ev_rows, 0));
        TRY(collective(c, rows_t, 1, FMHIP_COLL_SUM_F32, c->cs));
        if (live) {
            TRY(step_forward_pass(m, d, b, 1));
        } else {
            HIP_TRY(hipMemsetAsync(m->grad, 0, m->grad_floats() * sizeof(float), m->stream));
            m->grad_dirty = true;
            m->last_nnz = m->last_rows = 0;
        }
        // the order of the intervals: the second-coldest first (little work, many bytes: the wire starts early), down to feature
        // 0, the coldest last (it travels beside the next position's pass A).  Who walks a straddling range: whoever comes first
        for (int o = 0; o < n_int; ++o) {
            const int i = o < T ? T - 1 - o : T;
            const int64_t lo = edge[(size_t)i], hi = edge[(size_t)i + 1];
            const bool head = i == 0;                    // the lowest interval carries the statistics scalars
            const int own = i == T ? 0 : (i == T - 1 ? (kOwnLower | kOwnUpper) : kOwnLower);
            if (live) TRY(step_backward(m, d, b, lo, hi, head, nullptr, nullptr, own));
            const Region reg[3] = {{m->GV() + (size_t)lo * m->Kp, (size_t)(hi - lo) * m->Kp},
                                   {head ? m->grad : m->Gw() + lo, (size_t)(hi - lo) + (head ? (size_t)kGradHead : 0)},
                                   {m->Gb() + lo, (size_t)(hi - lo)}};
            TRY(reduce_regions(m, c, reg, 3, c->ev_ready[i], c->ev_done[i], pr));
        }
        m->bw_next_hi = -1;
        if (pr) HIP_TRY(hipEventRecord(pr->wait_a, m->stream));
        // every interval but the top one is updated as its slice arrives (small ones share the next one's launch)
        int64_t pend_hi = -1;
        for (int i = T - 1; i >= 0; --i) {
            const int64_t lo = edge[(size_t)i], hi = edge[(size_t)i + 1];
            if (pend_hi < 0) pend_hi = hi;
            HIP_TRY(hipStreamWaitEvent(m->stream, c->ev_done[i], 0));
            if (i > 0 && (pend_hi - lo) * 8 < m->n1) continue;
            if (pr) HIP_TRY(hipEventRecord(pr->a0[pr->n_apply++], m->stream));
            TRY(step_apply_interval(m, eta, reg0, regw, regv, lo, pend_hi, rows_t, false));
            if (pr) HIP_TRY(hipEventRecord(pr->a1[pr->n_apply - 1], m->stream));
            pend_hi = -1;
        }
        if (pr) HIP_TRY(hipEventRecord(pr->wait_b, m->stream));
        if (t + 1 < n) {
            // ... and the next position's pass A beside the top slice (its rows are the only ones not final yet)
            if (batches[t + 1] >= 0) TRY(step_forward_pass(m, d, batches[t + 1], 0));
        } else {
            HIP_TRY(hipStreamWaitEvent(m->stream, c->ev_done[T], 0));
            TRY(step_apply_interval(m, eta, reg0, regw, regv, edge[(size_t)T], edge[(size_t)T + 1], rows_t, true));
        }
    }
    if (part_rc != FMHIP_OK) return fail(part_rc, "%s (this rank contributed zeros to the run)", part_err.c_str());
    return FMHIP_OK;
}

// the shares of [0, n+1) for `world` ranks: every interval edge is a multiple of world, the top one rounded UP (into the zero
// rows kept behind the tables, fmhip_model::kSlackRows) — fmhip_host.h: shard_top, interval_edges, shard_share
inline int64_t shard_top(fmhip_model_t m, int W) { return fmhip::host::shard_top(m->n1, W); }

// One data-parallel step with the update sharded over the ranks (FMHIP_EXCHANGE_SHARDED; the schedule is drawn at the
// top of this file).  Per feature interval, from the cold ids down: backward -> reduce-scatter of its G_V rows (rank r
// receives the summed rows of share r) grouped with the all-reduce of its G_w / G_b entries -> rank r updates ITS
// rows of V and all of the interval's w, zeroes the other shares' gradient rows -> all-gather of the updated rows in
// place into V.  One writer per V row: the replicas are identical whatever order the transport sums in.
int dp_step_sharded(fmhip_model_t m, fmhip_dataset_t d, int64_t batch, fmhip_comm_t c, double eta, double reg0, double regw, double regv) {
    const bool live = batch >= 0;
    const int W = c->emu_ranks > 0 ? c->emu_ranks : c->world;        // shares per interval
    const int R = c->emu_ranks > 0 ? 0 : c->rank;                    // ... and which one is this rank's
    const int64_t top = shard_top(m, W);
    if (top > m->n1p + (m->grad == m->grad_own.p ? (int64_t)fmhip_model::kSlackRows : 0))
        return fail(FMHIP_ERR_UNSUPPORTED, "the sharded exchange needs %lld rows for %d equal shares of %lld: more ranks than the tables' "
                                           "slack allows (%d), or a caller-owned gradient buffer (fmhip_grad_bind) with n+1 not a multiple of world",
                    (long long)top, W, (long long)m->n1, fmhip_model::kSlackRows);
    const float my_rows = live ? (float)d->batches[(size_t)batch].rows : 0.f;
    hipLaunchKernelGGL(k_set_float, dim3(1), dim3(1), 0, m->stream, c->rows_dev, my_rows);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(c->ev_rows, m->stream));
    HIP_TRY(hipStreamWaitEvent(c->cs, c->ev_rows, 0));
    TRY(collective(c, c->rows_dev, 1, FMHIP_COLL_SUM_F32, c->cs));
    if (live) {
        TRY(step_forward(m, d, batch));
    } else {
        // out of rows: contribute zeros.  After a sharded step every row of G is clean (own share: the update; the other
        // shares: the zeroing); only the statistics scalars in front keep the last step's sums
        if (m->grad_dirty) HIP_TRY(hipMemsetAsync(m->grad, 0, m->grad_floats() * sizeof(float), m->stream));
        else HIP_TRY(hipMemsetAsync(m->grad, 0, (size_t)kGradHead * sizeof(float), m->stream));
        m->grad_dirty = true;
        m->last_nnz = m->last_rows = 0;
    }
    CommProf *pr = next_prof(c);
    const std::vector<int64_t> edge = interval_edges(c->cuts, m->n1, W);      // (the plan rounds already; a plan made for another world may not have)
    const int n_int = (int)edge.size() - 1;
    const size_t kp = (size_t)m->Kp;
    const double half = 0.5;            // emulated durations: a reduce-scatter or an all-gather is half an all-reduce of the same bytes
    for (int i = n_int - 1; i >= 0; --i) {
        const int64_t lo = edge[(size_t)i], hi = edge[(size_t)i + 1];
        const bool last = i == 0;
        if (live) TRY(step_backward(m, d, batch, lo, n_int == 1 ? INT64_MAX : hi, last, nullptr));
        const Share sh = shard_share(lo, hi, i == n_int - 1, m->n1, W, R);      // the top interval reaches into the slack rows
        const int64_t hi_r = sh.hi_r, chunk = sh.chunk, vlo = sh.vlo, vhi = sh.vhi;
        // one real rank playing `emu_ranks`: the collectives run on its own share (in place), the delays are the interval's
        const size_t count = (size_t)chunk * kp, at = (size_t)(c->emu_ranks > 0 ? vlo : lo) * kp;
        float *gw = last ? m->grad : m->Gw() + lo;
        const size_t n_gw = (size_t)(hi - lo) + (last ? (size_t)kGradHead : 0), n_gb = (size_t)(hi - lo);
        const double gv_bytes = (double)(hi_r - lo) * kp * sizeof(float), head_bytes = (double)(n_gw + n_gb) * sizeof(float);
        // Everything behind the backward is queued on the comm stream, in order — no event hops between the pieces:
        //   reduce-scatter of G_V (+ all-reduce of G_w, G_b: one grouped call) -> this rank's share of the update, the other
        //   shares' gradient rows zeroed (one launch) -> all-gather of the updated V rows
        HIP_TRY(hipEventRecord(c->ev_ready[i], m->stream));
        HIP_TRY(hipStreamWaitEvent(c->cs, c->ev_ready[i], 0));
        int pi = -1;
        if (pr && pr->n_coll < kProfColl) {
            pi = pr->n_coll++;
            HIP_TRY(hipEventRecord(pr->c0[pi], c->cs));
        }
        if (!c->ext) NCCL_TRY(rccl().GroupStart());
        TRY(collective(c, m->GV() + at, count, FMHIP_COLL_REDUCE_SCATTER_F32, c->cs));
        TRY(collective(c, gw, n_gw, FMHIP_COLL_SUM_F32, c->cs));
        TRY(collective(c, m->Gb() + lo, n_gb, FMHIP_COLL_SUM_F32, c->cs));
        if (!c->ext) NCCL_TRY(rccl().GroupEnd());
        TRY(emu_delay(c, half * gv_bytes + head_bytes, c->cs, m->GV() + (size_t)lo * kp, (size_t)(hi_r - lo) * kp, 1));
        if (pi >= 0) HIP_TRY(hipEventRecord(pr->c1[pi], c->cs));
        TRY(step_apply_shard(m, eta, reg0, regw, regv, lo, hi, hi_r, vlo, vhi, c->rows_dev, last, c->cs));
        pi = -1;
        if (pr && pr->n_coll < kProfColl) {
            pi = pr->n_coll++;
            HIP_TRY(hipEventRecord(pr->c0[pi], c->cs));
        }
        TRY(collective(c, m->V.p + at, count, FMHIP_COLL_ALLGATHER_F32, c->cs));
        TRY(emu_delay(c, half * gv_bytes, c->cs, m->V.p + (size_t)lo * kp, (size_t)(hi_r - lo) * kp, 1));
        if (pi >= 0) HIP_TRY(hipEventRecord(pr->c1[pi], c->cs));
        if (c->profiling) c->prof_bytes += (int64_t)(gv_bytes + head_bytes);
    }
    HIP_TRY(hipEventRecord(c->ev_gathered, c->cs));
    m->bw_next_hi = -1;
    // the next forward reads V and the next backward writes G: everything queued on the comm stream must have landed
    if (pr) HIP_TRY(hipEventRecord(pr->wait_a, m->stream));
    HIP_TRY(hipStreamWaitEvent(m->stream, c->ev_gathered, 0));
    if (pr) HIP_TRY(hipEventRecord(pr->wait_b, m->stream));
    m->grad_dirty = false;
    return FMHIP_OK;
}

// A step's size checks that depend on THIS rank's batch only.  A failure here must not leave the peers waiting in a
// collective: the caller runs the step with a zero contribution (as a rank that has run out of rows does) and
// reports the error afterwards.
int local_checks(fmhip_model_t m, fmhip_dataset_t d, int64_t batch, fmhip_comm_t c, bool in_schedule = true) {
    if (c->exchange == FMHIP_EXCHANGE_TOUCHED && !c->tsteps.empty() && c->msg_kp != m->Kp)
        return fail(FMHIP_ERR_INVALID, "the touched-rows exchange was planned for rows of %d floats, this model has %d: call fmhip_dp_plan "
                                       "with this model (every rank)", c->msg_kp, m->Kp);
    if (batch < 0) return FMHIP_OK;
    const auto &bm = d->batches[(size_t)batch];
    if (c->exchange != FMHIP_EXCHANGE_TOUCHED && d->rb_rows != 0 && !c->cuts.empty())
        return fail(FMHIP_ERR_INVALID, "the communicator's plan cuts the backward, but this dataset's transposes are row-blocked: "
                                       "call fmhip_dp_plan with this dataset (every rank)");
    // the global row count travels as one fp32 sum: exact below 2^24 rows per global batch.  fmhip_dp_plan agreed on the
    // largest batch of any rank; a batch beyond it belongs to a dataset the plan has not seen
    if (c->plan_max_rows >= 0 ? bm.rows > c->plan_max_rows : (double)bm.rows * c->world >= 16777216.0)
        return fail(FMHIP_ERR_INVALID, "batch %lld has %lld rows, the plan covers batches of up to %lld (a global batch must stay below 2^24 "
                                       "rows): call fmhip_dp_plan with this dataset (every rank)",
                    (long long)batch, (long long)bm.rows, (long long)c->plan_max_rows);
    if (c->exchange == FMHIP_EXCHANGE_TOUCHED) {
        // the unions were formed for the lock-step schedule "position t = every rank's batch t" of ONE dataset
        if (c->planned_data != d)
            return fail(FMHIP_ERR_INVALID, "the touched-rows exchange was planned for another dataset: call fmhip_dp_plan with this one (every rank)");
        // fmhip_dp_step: a rank without rows follows the cursor, so the ranks with rows must too; any other order goes
        // through fmhip_dp_step_at, where EVERY rank names the position
        if (in_schedule && batch != c->t_cursor)
            return fail(FMHIP_ERR_INVALID, "fmhip_dp_step walks the touched-rows plan in order: step %lld takes batch %lld (or -1 on a rank "
                                           "without it), not batch %lld; another order: fmhip_dp_step_at / fmhip_dp_epoch_order (every rank)",
                        (long long)c->t_cursor, (long long)c->t_cursor, (long long)batch);
    }
    return FMHIP_OK;
}

int dp_step_mode(fmhip_model_t m, fmhip_dataset_t d, int64_t batch, fmhip_comm_t c, double eta, double reg0, double regw, double regv,
                 int64_t position) {
    switch (c->exchange) {
        case FMHIP_EXCHANGE_TOUCHED: return dp_step_touched(m, d, batch, c, eta, reg0, regw, regv, position);
        case FMHIP_EXCHANGE_SHARDED: return dp_step_sharded(m, d, batch, c, eta, reg0, regw, regv);
        case FMHIP_EXCHANGE_PIPELINED: return dp_run_pipelined(m, d, c, &batch, 1, eta, reg0, regw, regv);
        default: return dp_step_dense(m, d, batch, c, eta, reg0, regw, regv);
    }
}

// position >= 0: the step's place in the lock-step schedule, named by every rank alike (fmhip_dp_step_at; the touched-rows
// exchange picks that position's union); -1: the next one in order (fmhip_dp_step)
int dp_step(fmhip_model_t m, fmhip_dataset_t d, int64_t batch, fmhip_comm_t c, double eta, double reg0, double regw, double regv,
            int64_t position = -1) {
    const int pre = local_checks(m, d, batch, c, position < 0);
    if (pre == FMHIP_OK) return dp_step_mode(m, d, batch, c, eta, reg0, regw, regv, position);
    const std::string why = fmhip_last_error();
    const int rc = dp_step_mode(m, d, -1, c, eta, reg0, regw, regv, position);     // keep in step with the peers: contribute zeros
    if (rc != FMHIP_OK) return rc;
    return fail(pre, "%s (this rank contributed zeros to the step)", why.c_str());
}

// `n` positions of the lock-step schedule in one call.  The pipelined mode overlaps consecutive steps; the other modes step one
// by one.  A position whose batch fails a local check contributes zeros (as in dp_step) and the error is reported afterwards.
int dp_run(fmhip_model_t m, fmhip_dataset_t d, fmhip_comm_t c, const int64_t *positions, int64_t n, double eta, double reg0, double regw, double regv) {
    const int64_t nb = (int64_t)d->batches.size();
    if (c->exchange != FMHIP_EXCHANGE_PIPELINED) {
        // a position whose batch fails a check only THIS rank can see contributes zeros and the run goes on — stopping here
        // would leave the peers alone in the next position's collectives; the first such error is reported at
        // the end.  A failure of the step itself (HIP, the transport) is no local matter and ends the run at once.
        int pre = FMHIP_OK;
        std::string why;
        for (int64_t j = 0; j < n; ++j) {
            int64_t b = positions[j] < nb ? positions[j] : -1;
            const int rc = local_checks(m, d, b, c, false);
            if (rc != FMHIP_OK) {
                if (pre == FMHIP_OK) { pre = rc; why = fmhip_last_error(); }
                b = -1;
            }
            TRY(dp_step_mode(m, d, b, c, eta, reg0, regw, regv, positions[j]));
        }
        if (pre != FMHIP_OK) return fail(pre, "%s (this rank contributed zeros to those steps)", why.c_str());
        return FMHIP_OK;
    }
    std::vector<int64_t> batches((size_t)n);
    int pre = FMHIP_OK;
    std::string why;
    for (int64_t j = 0; j < n; ++j) {
        batches[(size_t)j] = positions[j] < nb ? positions[j] : -1;
        const int rc = local_checks(m, d, batches[(size_t)j], c, false);
        if (rc != FMHIP_OK) {
            if (pre == FMHIP_OK) { pre = rc; why = fmhip_last_error(); }
            batches[(size_t)j] = -1;
        }
    }
    TRY(dp_run_pipelined(m, d, c, batches.data(), n, eta, reg0, regw, regv));
    if (pre != FMHIP_OK) return fail(pre, "%s (this rank contributed zeros to those steps)", why.c_str());
    return FMHIP_OK;
}

}  // namespace

extern "C" {

int fmhip_comm_unique_id(void *id) {
    if (!id) return fail(FMHIP_ERR_INVALID, "id is NULL");
    TRY(need_rccl());
    ncclUniqueId u;
    NCCL_TRY(rccl().GetUniqueId(&u));
    memcpy(id, &u, sizeof u);
    return FMHIP_OK;
}

static int comm_resources(fmhip_comm *c) {
    hipError_t e = hipStreamCreateWithFlags(&c->cs, hipStreamNonBlocking);
    for (int i = 0; i <= kMaxCuts && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&c->ev_ready[i], hipEventDisableTiming);
    for (int i = 0; i <= kMaxCuts && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&c->ev_done[i], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_rows, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_gathered, hipEventDisableTiming);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&c->scratch), (kMaxCuts + 1) * sizeof(int64_t));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&c->rows_dev), 32 * sizeof(float));
    if (e != hipSuccess) return fail(FMHIP_ERR_HIP, "communicator resources: %s", hipGetErrorString(e));
    return FMHIP_OK;
}

int fmhip_comm_create(fmhip_model_t m, const void *id, int rank, int world, fmhip_comm_t *out) {
    if (!out) return fail(FMHIP_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!m || !id) return fail(FMHIP_ERR_INVALID, "model or id is NULL");
    if (world < 1 || rank < 0 || rank >= world) return fail(FMHIP_ERR_INVALID, "rank %d outside a world of %d", rank, world);
    TRY(need_rccl());
    TRY(set_device(m->device));
    fmhip_comm *c = new (std::nothrow) fmhip_comm();
    if (!c) return fail(FMHIP_ERR_NOMEM, "out of host memory");
    c->device = m->device;
    c->rank = rank;
    c->world = world;
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    ncclResult_t r = rccl().CommInitRank(&c->comm, world, u, rank);
    if (r != ncclSuccess) {
        delete c;
        return fail(FMHIP_ERR_COMM, "ncclCommInitRank(rank %d of %d) failed: %s", rank, world, rccl().GetErrorString(r));
    }
    const int rc = comm_resources(c);
    if (rc != FMHIP_OK) {
        fmhip_comm_destroy(c);
        return rc;
    }
    *out = c;
    return FMHIP_OK;
}

int fmhip_comm_create_external(fmhip_model_t m, int rank, int world, fmhip_collective_fn fn, void *ctx, fmhip_comm_t *out) {
    if (!out) return fail(FMHIP_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!m || !fn) return fail(FMHIP_ERR_INVALID, "model or collective function is NULL");
    if (world < 1 || rank < 0 || rank >= world) return fail(FMHIP_ERR_INVALID, "rank %d outside a world of %d", rank, world);
    TRY(set_device(m->device));
    fmhip_comm *c = new (std::nothrow) fmhip_comm();
    if (!c) return fail(FMHIP_ERR_NOMEM, "out of host memory");
    c->device = m->device;
    c->rank = rank;
    c->world = world;
    c->ext = fn;
    c->ext_ctx = ctx;
    const int rc = comm_resources(c);
    if (rc != FMHIP_OK) {
        fmhip_comm_destroy(c);
        return rc;
    }
    *out = c;
    return FMHIP_OK;
}

// what a host-staged transport needs and a JVM / ctypes caller cannot reach by itself
int fmhip_stream_wait(void *hip_stream) {
    HIP_TRY(hipStreamSynchronize(reinterpret_cast<hipStream_t>(hip_stream)));
    return FMHIP_OK;
}

int fmhip_device_read(void *host_dst, const void *device_src, size_t bytes, void *hip_stream) {
    if (bytes && (!host_dst || !device_src)) return fail(FMHIP_ERR_INVALID, "NULL buffer");
    hipStream_t s = reinterpret_cast<hipStream_t>(hip_stream);
    HIP_TRY(hipMemcpyAsync(host_dst, device_src, bytes, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return FMHIP_OK;
}

int fmhip_device_write(void *device_dst, const void *host_src, size_t bytes, void *hip_stream) {
    if (bytes && (!device_dst || !host_src)) return fail(FMHIP_ERR_INVALID, "NULL buffer");
    hipStream_t s = reinterpret_cast<hipStream_t>(hip_stream);
    HIP_TRY(hipMemcpyAsync(device_dst, host_src, bytes, hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));
    return FMHIP_OK;
}

int fmhip_comm_destroy(fmhip_comm_t c) {
    if (!c) return FMHIP_OK;
    (void)hipSetDevice(c->device);
    if (c->cs) (void)hipStreamSynchronize(c->cs);
    for (auto &p : c->prof) destroy_events(p);
    if (c->ev_gathered) (void)hipEventDestroy(c->ev_gathered);
    free_touched(c);
    if (c->comm) (void)rccl().CommDestroy(c->comm);
    for (hipEvent_t e : c->ev_ready)
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->ev_done)
        if (e) (void)hipEventDestroy(e);
    if (c->ev_rows) (void)hipEventDestroy(c->ev_rows);
    if (c->rows_dev) (void)hipFree(c->rows_dev);
    if (c->cs) (void)hipStreamDestroy(c->cs);
    if (c->scratch) (void)hipFree(c->scratch);
    delete c;
    return FMHIP_OK;
}

// Known patterns through every collective kind a step or a plan issues, on the communicator's own stream and with the
// same calls (the grouped three-region all-reduce of a gradient slice included): a binding that moves the wrong
// elements, counts bytes for elements or scatters to the wrong segment shows up HERE, on every rank alike, and not as
// a model that quietly diverges.  With one rank every collective is the identity and the test still runs the calls.
int fmhip_comm_selftest(fmhip_comm_t c, int *failed_kinds) {
    if (failed_kinds) *failed_kinds = 0;
    if (!c) return fail(FMHIP_ERR_INVALID, "communicator is NULL");
    TRY(set_device(c->device));
    const int W = c->world;
    constexpr size_t kSeg = 4099;                         // elements per rank's segment (odd on purpose)
    const size_t n = kSeg * (size_t)W;
    constexpr int kKinds = 6;
    float *fb = nullptr;
    int64_t *ib = nullptr;
    unsigned *bad = nullptr;
    // regions of the grouped all-reduce: one word, a long run, a short odd run (as G_b's head, G_V rows, G_w of an interval)
    const size_t r0 = 0, n0 = 1, r1 = 32, n1 = n - 32 - 64, r2 = n - 37, n2 = 37;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&fb), 3 * n * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&ib), 8 * sizeof(int64_t));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&bad), 8 * sizeof(unsigned));
    int rc = e == hipSuccess ? FMHIP_OK : fail(FMHIP_ERR_HIP, "self-test buffers: %s", hipGetErrorString(e));
    unsigned h_bad[8] = {};
    auto body = [&]() -> int {
        hipStream_t s = c->cs;
        const dim3 b(256), g((unsigned)((n + 255) / 256));
        float *sum = fb, *rs = fb + n, *ag = fb + 2 * n;
        HIP_TRY(hipMemsetAsync(bad, 0, 8 * sizeof(unsigned), s));
        hipLaunchKernelGGL(k_st_fill, g, b, 0, s, sum, n, kSeg, c->rank, 0);
        hipLaunchKernelGGL(k_st_fill, g, b, 0, s, rs, n, kSeg, c->rank, 0);
        hipLaunchKernelGGL(k_st_fill, g, b, 0, s, ag, n, kSeg, c->rank, 1);
        hipLaunchKernelGGL(k_st_i64, dim3(1), dim3(1), 0, s, ib, c->rank);
        HIP_TRY(hipGetLastError());
        // 0: SUM_F32, three regions in one group (exactly reduce_regions' calls)
        const bool group = !c->ext;
        if (group) NCCL_TRY(rccl().GroupStart());
        TRY(collective(c, sum + r0, n0, FMHIP_COLL_SUM_F32, s));
        TRY(collective(c, sum + r1, n1, FMHIP_COLL_SUM_F32, s));
        TRY(collective(c, sum + r2, n2, FMHIP_COLL_SUM_F32, s));
        if (group) NCCL_TRY(rccl().GroupEnd());
        TRY(collective(c, ib, 3, FMHIP_COLL_MAX_I64, s));
        TRY(collective(c, ib + 4, 2, FMHIP_COLL_BCAST0_I64, s));
        TRY(collective(c, rs, kSeg, FMHIP_COLL_REDUCE_SCATTER_F32, s));
        TRY(collective(c, ag, kSeg, FMHIP_COLL_ALLGATHER_F32, s));
        for (auto r : {std::pair<size_t, size_t>(r0, n0), std::pair<size_t, size_t>(r1, n1), std::pair<size_t, size_t>(r2, n2)})
            hipLaunchKernelGGL(k_st_check, dim3((unsigned)((r.second + 255) / 256)), b, 0, s, sum, r.first, r.first + r.second, kSeg, W, 0,
                               bad + FMHIP_COLL_SUM_F32);
        hipLaunchKernelGGL(k_st_i64_check, dim3(1), dim3(1), 0, s, ib, W, bad + FMHIP_COLL_MAX_I64, bad + FMHIP_COLL_BCAST0_I64);
        hipLaunchKernelGGL(k_st_check, dim3((unsigned)((kSeg + 255) / 256)), b, 0, s, rs, kSeg * c->rank, kSeg * (c->rank + 1), kSeg, W, 0,
                           bad + FMHIP_COLL_REDUCE_SCATTER_F32);
        hipLaunchKernelGGL(k_st_check, g, b, 0, s, ag, (size_t)0, n, kSeg, W, 1, bad + FMHIP_COLL_ALLGATHER_F32);
        // 3: ALLGATHER_I32 reuses the all-gather buffer once the float check has read it (same stream)
        hipLaunchKernelGGL(k_st_fill, g, b, 0, s, ag, n, kSeg, c->rank, 2);
        HIP_TRY(hipGetLastError());
        TRY(collective(c, ag, kSeg, FMHIP_COLL_ALLGATHER_I32, s));
        hipLaunchKernelGGL(k_st_check, g, b, 0, s, ag, (size_t)0, n, kSeg, W, 2, bad + FMHIP_COLL_ALLGATHER_I32);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(h_bad, bad, 8 * sizeof(unsigned), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        // every rank learns every rank's verdict: the mask travels through MAX_I64 bit by bit (a broken MAX shows in its own bit
        // on the rank that saw it; that rank still reports)
        int64_t mask[kKinds];
        for (int k = 0; k < kKinds; ++k) mask[k] = h_bad[k] ? 1 : 0;
        HIP_TRY(hipMemcpyAsync(ib, mask, sizeof mask, hipMemcpyHostToDevice, s));
        TRY(collective(c, ib, kKinds, FMHIP_COLL_MAX_I64, s));
        HIP_TRY(hipMemcpyAsync(mask, ib, sizeof mask, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        for (int k = 0; k < kKinds; ++k)
            if (mask[k] != 0 || h_bad[k]) h_bad[k] = h_bad[k] ? h_bad[k] : 1u;
        return FMHIP_OK;
    };
    if (rc == FMHIP_OK) rc = body();
    if (fb) (void)hipFree(fb);
    if (ib) (void)hipFree(ib);
    if (bad) (void)hipFree(bad);
    if (rc != FMHIP_OK) return rc;
    int failed = 0;
    for (int k = 0; k < kKinds; ++k) failed |= h_bad[k] ? 1 << k : 0;
    if (failed_kinds) *failed_kinds = failed;
    if (failed)
        return fail(FMHIP_ERR_COMM, "communicator self-test: wrong results from collective kinds 0x%x on some rank (rank %d of %d saw %u / %u / %u / %u / %u / %u wrong elements)",
                    failed, c->rank, W, h_bad[0], h_bad[1], h_bad[2], h_bad[3], h_bad[4], h_bad[5]);
    return FMHIP_OK;
}

int fmhip_comm_emulate(fmhip_comm_t c, double payload_gb_per_s) {
    if (!c) return fail(FMHIP_ERR_INVALID, "communicator is NULL");
    if (payload_gb_per_s < 0.0) return fail(FMHIP_ERR_INVALID, "negative rate");
    c->emu_bytes_per_us = payload_gb_per_s * 1e3;      // GB/s = 1e3 bytes per microsecond
    return FMHIP_OK;
}

int fmhip_comm_emulate_load(fmhip_comm_t c, int workgroups) {
    if (!c) return fail(FMHIP_ERR_INVALID, "communicator is NULL");
    if (workgroups < 0 || workgroups > 1024) return fail(FMHIP_ERR_INVALID, "workgroups must be 0..1024");
    c->emu_wgs = workgroups;
    return FMHIP_OK;
}

int fmhip_comm_emulate_ranks(fmhip_comm_t c, int ranks) {
    if (!c) return fail(FMHIP_ERR_INVALID, "communicator is NULL");
    if (ranks < 0 || ranks > fmhip_model::kSlackRows) return fail(FMHIP_ERR_INVALID, "ranks must be 0..%d", fmhip_model::kSlackRows);
    if (ranks > 0 && c->world != 1) return fail(FMHIP_ERR_INVALID, "emulated ranks are for one-rank communicators (this one has %d)", c->world);
    c->emu_ranks = ranks;
    return FMHIP_OK;
}

int fmhip_dp_exchange(fmhip_comm_t c, int mode) {
    if (!c) return fail(FMHIP_ERR_INVALID, "communicator is NULL");
    if (mode != FMHIP_EXCHANGE_DENSE && mode != FMHIP_EXCHANGE_TOUCHED && mode != FMHIP_EXCHANGE_SHARDED && mode != FMHIP_EXCHANGE_PIPELINED)
        return fail(FMHIP_ERR_INVALID, "unknown exchange mode %d", mode);
    if (mode == FMHIP_EXCHANGE_SHARDED && c->world > fmhip_model::kSlackRows)
        return fail(FMHIP_ERR_UNSUPPORTED, "the sharded exchange supports up to %d ranks", fmhip_model::kSlackRows);
    c->exchange = mode;
    return FMHIP_OK;
}

int fmhip_dp_exchange_info(fmhip_comm_t c, int *mode, int64_t *id_slots_per_rank, double *mean_union_rows) {
    if (!c) return fail(FMHIP_ERR_INVALID, "communicator is NULL");
    if (mode) *mode = c->exchange;
    if (id_slots_per_rank) *id_slots_per_rank = c->cap;
    if (mean_union_rows) {
        double sum = 0.0;
        for (const auto &t : c->tsteps) sum += t.n_u;
        *mean_union_rows = c->tsteps.empty() ? 0.0 : sum / (double)c->tsteps.size();       // over the planned steps
    }
    return FMHIP_OK;
}

int fmhip_comm_info(fmhip_comm_t c, int *rank, int *world) {
    if (!c) return fail(FMHIP_ERR_INVALID, "communicator is NULL");
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    return FMHIP_OK;
}

int fmhip_dp_plan(fmhip_model_t m, fmhip_dataset_t d, fmhip_comm_t c, int n_fractions, const double *upper_fractions, int64_t *cuts_out) {
    WriteLock lock(m);       // the step changes the model; the communicator is used by the thread that owns its model
    TRY(check_comm(m, c));
    TRY(check_train(m, d));
    if (n_fractions < 0 || n_fractions > kMaxCuts || (n_fractions > 0 && !upper_fractions))
        return fail(FMHIP_ERR_INVALID, "n_fractions must be 0..%d with an array of as many fractions", kMaxCuts);
    int64_t cuts[kMaxCuts + 1] = {};
    if (c->rank == 0 && n_fractions > 0 && d->rb_rows == 0) {
        // stored nonzeros per feature over this rank's batches (the sparse streams: the dense hot block's
        // features do not depend on the interval), then for every fraction the id above which that share lies
        std::vector<int32_t> cnt((size_t)m->n1, 0);
        for (size_t b = 0; b < d->batches.size(); ++b) {
            const auto &bm = d->batches[b];
            const int32_t *hf = d->h_cfeat.data() + bm.col_off, *hp = d->h_cptr.data() + bm.col_off + b;
            for (int32_t s = 0; s < bm.n_cols; ++s) cnt[(size_t)hf[s]] += hp[s + 1] - hp[s];
        }
        choose_cuts(cnt.data(), m->n1, n_fractions, upper_fractions, cuts);
    }
    // What every rank must agree on before a step can be sized (one max-reduce): the largest mini-batch of any rank —
    // its global row count travels as one fp32 sum, exact below 2^24 —, whether some rank's transposes are row-blocked
    // (it cannot cut its backward: then nobody does, same collectives everywhere), the touched-rows table's width, and
    // whether some rank cannot hold the sharded exchange's equal shares.  All ranks pass or fail together.
    int64_t agree[5] = {0, d->rb_rows != 0, 1, 0, (int64_t)d->batches.size()};
    for (const auto &bm : d->batches) {
        agree[0] = std::max<int64_t>(agree[0], bm.rows);
        agree[2] = std::max<int64_t>(agree[2], (int64_t)bm.n_cols + d->hot_pages * kHotT);
    }
    const int W = c->emu_ranks > 0 ? c->emu_ranks : c->world;
    if (c->exchange == FMHIP_EXCHANGE_SHARDED)
        agree[3] = shard_top(m, W) > m->n1p + (m->grad == m->grad_own.p ? (int64_t)fmhip_model::kSlackRows : 0);
    TRY(control_i64(m, c, agree, 5, false));
    if ((double)agree[0] * c->world >= 16777216.0)
        return fail(FMHIP_ERR_INVALID, "a global batch of %lld x %d rows exceeds 2^24 (the summed row count travels as one fp32 word): "
                                       "use smaller batches", (long long)agree[0], c->world);
    if (agree[3])
        return fail(FMHIP_ERR_UNSUPPORTED, "the sharded exchange cannot cut %lld rows into %d equal shares on some rank (a caller-owned "
                                           "gradient buffer, fmhip_grad_bind, needs n+1 to be a multiple of world)", (long long)m->n1, W);
    c->plan_max_rows = agree[0];
    c->plan_steps = agree[4];
    const int64_t blocked = agree[1];
    if (c->exchange == FMHIP_EXCHANGE_TOUCHED && blocked)
        return fail(FMHIP_ERR_UNSUPPORTED, "the touched-rows exchange needs transposes without row blocks (some rank's dataset has them)");
    TRY(control_i64(m, c, cuts, kMaxCuts + 1, true));
    c->cuts.clear();
    for (int i = 0; i < n_fractions && !blocked; ++i) {
        if (c->exchange == FMHIP_EXCHANGE_SHARDED) cuts[i] = cuts[i] / W * W;      // equal shares: edges at multiples of world
        if (cuts[i] > 0 && cuts[i] < m->n1) c->cuts.push_back(cuts[i]);
    }
    std::sort(c->cuts.begin(), c->cuts.end());
    c->cuts.erase(std::unique(c->cuts.begin(), c->cuts.end()), c->cuts.end());
    // the touched-rows plan: id slots per rank = the largest number of rows any batch of any rank touches; positions = the
    // largest batch count; it places the cuts chosen above in every position's union
    if (c->exchange == FMHIP_EXCHANGE_TOUCHED) TRY(plan_touched(m, d, c, agree[2], agree[4]));
    if (cuts_out)
        for (int i = 0; i < n_fractions; ++i) cuts_out[i] = i < (int)c->cuts.size() ? c->cuts[c->cuts.size() - 1 - (size_t)i] : 0;
    return FMHIP_OK;
}

int fmhip_dp_step(fmhip_model_t m, fmhip_dataset_t d, int64_t batch, fmhip_comm_t c, double eta, double reg0, double regw,
                  double regv) {
    WriteLock lock(m);       // the step changes the model; the communicator is used by the thread that owns its model
    TRY(check_comm(m, c));
    TRY(check_train(m, d));
    if (batch >= 0) TRY(check_batch(d, batch));
    return dp_step(m, d, batch, c, eta, reg0, regw, regv);
}

int fmhip_dp_step_at(fmhip_model_t m, fmhip_dataset_t d, int64_t position, fmhip_comm_t c, double eta, double reg0, double regw,
                     double regv) {
    WriteLock lock(m);       // the step changes the model; the communicator is used by the thread that owns its model
    TRY(check_comm(m, c));
    TRY(check_train(m, d));
    if (position < 0) return fail(FMHIP_ERR_INVALID, "position must be >= 0");
    return dp_step(m, d, position < (int64_t)d->batches.size() ? position : -1, c, eta, reg0, regw, regv, position);
}

int fmhip_dp_steps(fmhip_model_t m, fmhip_dataset_t d, const int64_t *positions, int64_t n, fmhip_comm_t c, double eta, double reg0, double regw,
                   double regv) {
    WriteLock lock(m);
    TRY(check_comm(m, c));
    TRY(check_train(m, d));
    if (n < 0 || (n > 0 && !positions)) return fail(FMHIP_ERR_INVALID, "positions is NULL or n < 0");
    for (int64_t j = 0; j < n; ++j)
        if (positions[j] < 0) return fail(FMHIP_ERR_INVALID, "positions[%lld] = %lld: a position is >= 0", (long long)j, (long long)positions[j]);
    return dp_run(m, d, c, positions, n, eta, reg0, regw, regv);
}

static int dp_epoch(fmhip_model_t m, fmhip_dataset_t d, fmhip_comm_t c, double eta, double reg0, double regw, double regv,
                    const int64_t *order, int64_t n_order, fmhip_stats *stats) {
    TRY(check_comm(m, c));
    TRY(check_train(m, d));
    const int64_t nb = (int64_t)d->batches.size();
    // every rank takes the same number of steps — the largest local batch count — and every rank learns whether SOME
    // rank's dataset does not fit the plan, or whether the ranks disagree about taking an order at all (then all of them stop
    // here, none inside a collective)
    // ... and whether every rank holds the SAME valid order: a permutation check that only one rank fails, or two ranks walking
    // different permutations, would otherwise end inside the first collective — the order's validity and a hash of
    // its content travel in the same max-reduce (max of h and of -h: equal on all ranks iff max == -max of the negatives)
    int64_t order_bad = 0, order_hash = 0;
    if (order) {
        std::vector<char> seen((size_t)std::max<int64_t>(n_order, 0), 0);
        uint64_t h = 1469598103934665603ull;
        for (int64_t j = 0; j < n_order; ++j) {
            if (order[j] < 0 || order[j] >= n_order || seen[(size_t)order[j]]) { order_bad = j + 1; break; }
            seen[(size_t)order[j]] = 1;
            h = (h ^ (uint64_t)order[j]) * 1099511628211ull;
        }
        order_hash = (int64_t)(h >> 2);        // (62 bits: its negative exists)
    }
    int64_t agree[7] = {nb, 0, order ? n_order : -1, order ? -n_order : 1, order_bad, order_hash, -order_hash};
    for (int64_t j = 0; j < nb && !agree[1]; ++j) agree[1] = local_checks(m, d, j, c, false) != FMHIP_OK;
    if (!agree[1] && c->exchange == FMHIP_EXCHANGE_TOUCHED && !lazy_decay_ok(m, eta, regw, regv)) {
        agree[1] = 1;
        (void)fail(FMHIP_ERR_UNSUPPORTED, "the touched-rows exchange needs weight decay that fits the tables' scale (0.5 <= 1 - eta*reg <= 1)");
    }
    const std::string why = agree[1] ? fmhip_last_error() : "";
    TRY(control_i64(m, c, agree, 7, false));
    if (agree[1])
        return fail(FMHIP_ERR_INVALID, "%s", why.empty() ? "another rank's dataset does not fit the communicator's plan: call fmhip_dp_plan "
                                                           "with the datasets of this epoch (every rank)" : why.c_str());
    const int64_t steps = agree[0];
    if (agree[2] != -agree[3] || (order && n_order != steps))
        return fail(FMHIP_ERR_INVALID, "fmhip_dp_epoch_order: every rank passes the same order of the epoch's %lld positions (this rank: %lld "
                                       "entries; the ranks' counts range from %lld to %lld)", (long long)steps, (long long)(order ? n_order : -1),
                    (long long)-agree[3], (long long)agree[2]);
    if (agree[4]) {
        if (order_bad)
            return fail(FMHIP_ERR_INVALID, "order[%lld] = %lld: the order must be a permutation of [0, %lld)", (long long)(order_bad - 1),
                        (long long)order[order_bad - 1], (long long)steps);
        return fail(FMHIP_ERR_INVALID, "fmhip_dp_epoch_order: another rank's order is not a permutation of [0, %lld)", (long long)steps);
    }
    if (agree[5] != -agree[6])
        return fail(FMHIP_ERR_INVALID, "fmhip_dp_epoch_order: the ranks passed DIFFERENT orders (every rank passes the same permutation of the epoch's %lld positions)",
                    (long long)steps);
    if (c->exchange == FMHIP_EXCHANGE_TOUCHED) {
        if (steps != (int64_t)c->tsteps.size())
            return fail(FMHIP_ERR_INVALID, "the touched-rows plan covers %zu steps, this epoch has %lld: call fmhip_dp_plan with this dataset (every rank)",
                        c->tsteps.size(), (long long)steps);
        c->t_cursor = 0;
    }
    {
        std::vector<int64_t> pos((size_t)steps);
        for (int64_t j = 0; j < steps; ++j) pos[(size_t)j] = order ? order[j] : j;
        TRY(dp_run(m, d, c, pos.data(), steps, eta, reg0, regw, regv));
    }
    if (stats) {
        memset(stats, 0, sizeof *stats);
        TRY(read_scal(m, stats));      // all-reduced: the global batch's sums
        stats->nnz = m->last_nnz;
        stats->steps = steps;
    }
    return FMHIP_OK;
}

int fmhip_dp_epoch(fmhip_model_t m, fmhip_dataset_t d, fmhip_comm_t c, double eta, double reg0, double regw, double regv,
                   fmhip_stats *stats) {
    WriteLock lock(m);       // the step changes the model; the communicator is used by the thread that owns its model
    return dp_epoch(m, d, c, eta, reg0, regw, regv, nullptr, 0, stats);
}

int fmhip_dp_epoch_order(fmhip_model_t m, fmhip_dataset_t d, fmhip_comm_t c, double eta, double reg0, double regw, double regv,
                         const int64_t *order, int64_t n_order, fmhip_stats *stats) {
    WriteLock lock(m);
    if (!order || n_order < 0) return dp_epoch(m, d, c, eta, reg0, regw, regv, nullptr, 0, stats);
    return dp_epoch(m, d, c, eta, reg0, regw, regv, order, n_order, stats);
}

int fmhip_dp_plan_info(fmhip_comm_t c, int64_t *steps, int64_t *max_batch_rows) {
    if (!c) return fail(FMHIP_ERR_INVALID, "communicator is NULL");
    if (c->plan_max_rows < 0) return fail(FMHIP_ERR_INVALID, "no plan yet: call fmhip_dp_plan (every rank)");
    if (steps) *steps = c->plan_steps;
    if (max_batch_rows) *max_batch_rows = c->plan_max_rows;
    return FMHIP_OK;
}

int fmhip_comm_profile_begin(fmhip_comm_t c) {
    if (!c) return fail(FMHIP_ERR_INVALID, "communicator is NULL");
    TRY(set_device(c->device));
    if (c->prof.empty()) {
        c->prof.resize(kProfSteps);
        for (auto &p : c->prof) {
            const int rc = create_events(p);
            if (rc != FMHIP_OK) {
                for (auto &q : c->prof) destroy_events(q);
                c->prof.clear();
                return rc;
            }
        }
    }
    for (auto &p : c->prof) p.used = false;
    c->prof_next = 0;
    c->prof_bytes = 0;
    c->profiling = true;
    return FMHIP_OK;
}

int fmhip_comm_profile_end(fmhip_comm_t c, fmhip_comm_profile *p) {
    if (!c || !p) return fail(FMHIP_ERR_INVALID, "NULL argument");
    TRY(set_device(c->device));
    c->profiling = false;
    memset(p, 0, sizeof *p);
    HIP_TRY(hipDeviceSynchronize());
    for (auto &r : c->prof) {
        if (!r.used) continue;
        float ms = 0.f;
        // what the compute stream spent between its last backward and the end of the step, minus the updates themselves
        // (dense mode; the sharded mode's updates run on a stream of their own and what is left of them counts as exposed)
        if (hipEventElapsedTime(&ms, r.wait_a, r.wait_b) == hipSuccess) p->exposed_ms += ms;
        if (r.has_top && hipEventElapsedTime(&ms, r.top_a, r.top_b) == hipSuccess) p->exposed_ms += ms;   // pipelined: the previous step's top slice
        for (int i = 0; i < r.n_apply; ++i)
            if (hipEventElapsedTime(&ms, r.a0[i], r.a1[i]) == hipSuccess) p->exposed_ms -= ms;
        for (int i = 0; i < r.n_coll; ++i)
            if (hipEventElapsedTime(&ms, r.c0[i], r.c1[i]) == hipSuccess) p->comm_ms += ms;
        p->steps += 1;
        r.used = false;
    }
    p->bytes = c->prof_bytes;
    c->prof_next = 0;
    return FMHIP_OK;
}

int fmhip_shard_rows(int64_t n_rows, const int64_t *row_ptr, int world, int rank, int64_t *lo, int64_t *hi) {
    if (!row_ptr || !lo || !hi) return fail(FMHIP_ERR_INVALID, "NULL argument");
    if (n_rows < 0 || world < 1 || rank < 0 || rank >= world) return fail(FMHIP_ERR_INVALID, "bad shard request (rows %lld, rank %d of %d)", (long long)n_rows, rank, world);
    shard_bounds(n_rows, row_ptr, world, rank, lo, hi);
    return FMHIP_OK;
}

}  // extern "C"
