// Multi-GPU evaluation inside ONE process: the keyframes of a problem sharded over the GPUs of a node, one iba_handle, one
// issuing thread and one RCCL communicator per device (ncclCommInitAll). The reference's only parallel strategy is the frame
// loop (`#pragma omp parallel for` over keyframes with critical-section sums: iba_global.cpp:193, 239, 318; iba_func.cpp:203;
// iba_local.cpp:162): here every device evaluates its frames into a partial block of B x 64 doubles, ONE
// ncclAllReduce(sum, f64) over xGMI adds the blocks in place on every device, and device 0's copy is finalised on the host
// (iba_finalize_*). No point, keypoint or tree ever crosses a link; the message is 32 KB at B = 64, latency-bound.
//
// Host side of a call: the candidate block (Sim3Exp, SE3Exp(-x), their duals) is computed ONCE on the calling thread; every
// device has a worker thread pinned to it (hipSetDevice once, at start) that copies the block into its handle's pinned ring,
// issues the launch chain and the collective on its own stream and waits for that stream — the devices are issued
// concurrently, the caller's current HIP device is never touched. Workers spin for a short while after a job (an optimiser
// calls back within microseconds) and sleep on a condition variable otherwise.
//
// RCCL is loaded lazily (dlopen) the first time a communicator is needed: the core library depends on libamdhip64 only.
// IBA_GROUP_REDUCE_HOST sums the blocks on the host in rank order instead (no RCCL; the same device may then appear more
// than once in the device list, which is how the n > 1 logic is exercised on a one-GPU box).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // types and NCCL_VERSION_CODE only: no symbol of librccl is linked

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/iba_mi355x.h"
#include "iba_internal.hpp"
#include "iba_lm.hpp"
#include "iba_mads.hpp"
#include "iba_workers.hpp"

using namespace iba;

namespace {

// ---- librccl, loaded on first use ----
struct Rccl {
    void* lib = nullptr;
    std::string path, err;
    int runtime_version = 0;
    ncclResult_t (*GetVersion)(int*) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;   // optional: frees a communicator whose collective can no longer complete
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok() const { return lib != nullptr; }
};
Rccl& rccl() {
    static Rccl R;
    static std::once_flag once;
    std::call_once(once, []() {
        // A copy the process has already mapped (torch ships its own librccl.so, SONAME librccl.so.1) is used in preference to a
        // second one. A fresh copy is loaded RTLD_LOCAL: /opt/rocm's librccl drags in /opt/rocm's librocm_smi64, and were their
        // symbols global, a torch imported LATER would run the static initialisers of its own bundled copy on the first copy's
        // objects (measured: "double free or corruption" in a std::map destructor of librocm_smi64 at exit).
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        void* lib = nullptr;
        for (const char* n : names) if ((lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
        if (!lib) for (const char* n : names) if ((lib = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
        if (!lib) { const char* e = dlerror(); R.err = std::string("librccl not found: ") + (e ? e : "dlopen failed"); return; }
#define IBA_SYM(field, name) R.field = (decltype(R.field))dlsym(lib, name); if (!R.field) { R.err = std::string("librccl lacks ") + name; dlclose(lib); return; }
        IBA_SYM(GetVersion, "ncclGetVersion") IBA_SYM(CommInitAll, "ncclCommInitAll") IBA_SYM(CommDestroy, "ncclCommDestroy") IBA_SYM(CommCount, "ncclCommCount")
        IBA_SYM(AllReduce, "ncclAllReduce") IBA_SYM(GetErrorString, "ncclGetErrorString")
#undef IBA_SYM
        R.CommAbort = (decltype(R.CommAbort))dlsym(lib, "ncclCommAbort");
        Dl_info info;
        if (dladdr((void*)R.AllReduce, &info) && info.dli_fname) R.path = info.dli_fname;
        (void)R.GetVersion(&R.runtime_version);
        R.lib = lib;
    });
    return R;
}

}  // namespace

struct iba_group {
    int n = 0;
    bool host_reduce = false;
    std::vector<int> dev;
    std::vector<iba_handle*> h;
    std::vector<ncclComm_t> comm;
    std::vector<hipStream_t> st;
    std::vector<double*> d_part;          // per device: kMaxChain x stride doubles
    std::vector<double*> h_parts;         // per device, pinned (host reduction); h_part = h_parts[0] otherwise
    std::vector<double*> h_parts_dev;     // the same blocks as the devices address them: with the host reduction the last kernel of a
                                          // device's chain writes its sums there (no copy behind it)
    std::vector<int32_t> f_begin, f_end;
    double* h_part = nullptr;             // the summed block, host
    std::vector<double> h_sum;            // host reduction: the sum in rank order
    std::vector<Cand> cands;              // candidate block of the current chunk, computed once per call
    iba_params params{};
    int stride = 64;
    std::string err;
    double last_issue_us = 0.0;           // host time of the last chunk: candidate block + hand-over to the workers + wait
    double last_enqueue_us = 0.0;         // of which: until the LAST device's launch chain and collective were enqueued (the host issue time)
    std::vector<double> enq_us;           // per worker
    // ---- one worker thread per device (iba_workers.hpp) ----
    WorkerPool pool;
    std::atomic<int> jets_ready{0};       // the derivative half of `cands` is complete (set by the calling thread while the workers' kernels run)
    // failure handling: a worker that fails BEFORE the collective is seen by its peers at the barrier between launch chain and
    // collective (nobody enters the all-reduce); a failure AFTER it (the collective did not enqueue on one rank, the bounded wait
    // ran out) raises the pool's abort flag: every worker abandons its wait, aborts its communicator and the group is `broken`
    // (every later call fails at once with IBA_ERR_STATE; destroy does not wait for the streams).
    std::atomic<bool> broken{false};
    // IBA_GROUP_TIMEOUT_MS (default 20 s): bound of a worker's wait for its stream (a collective that never completes). A group that exceeds
    // it is marked broken for good, so the bound must sit far above a healthy call: an evaluation takes 0.1 .. 10 ms; the FIRST call of a
    // group (cold device, RCCL's lazy channel set-up, a profiler attached) may take seconds and gets four times the bound.
    double wait_timeout_ms = 20000.0;
    std::atomic<bool> warm{false};        // a call has completed on this group
    int reserved_batch = IBA_MAX_BATCH;   // the handles' work buffers hold this many candidates (grown, on every device's own thread, before a larger chain is issued)
    int debug_fail_rank = -1, debug_fail_phase = 1; bool debug_fail_armed = false;   // IBA_DEBUG_FAIL_RANK / _PHASE: one injected failure (tests)
};

namespace {
thread_local std::string g_group_create_error;

iba_status gfail(iba_group* g, iba_status s, const std::string& m) { if (g) g->err = m; else g_group_create_error = m; return s; }

// runs fn(i) on every device's worker, concurrently (and `meanwhile`, if any, on the calling thread once the workers are off);
// the first failure is reported
iba_status run_all(iba_group* g, std::function<iba_status(int)> fn, const std::function<void()>& meanwhile = nullptr) {
    const int bad = g->pool.run_all(std::move(fn), meanwhile);
    if (bad >= 0) return gfail(g, g->pool.status_of(bad), std::string("device ") + std::to_string(g->dev[bad]) + ": " + g->pool.error_of(bad));
    return IBA_OK;
}

// The worker polls its stream (the rule of the single handle's wait_stream) — and never blocks inside the runtime: a collective
// whose peer never arrives would keep hipStreamSynchronize forever. Past 2 ms the poll yields between queries; it gives up when a
// peer has raised the abort flag or when the bound (IBA_GROUP_TIMEOUT_MS, default 20 s) has run out. 1: done, 0: abandoned.
int poll_stream(iba_group* g, hipStream_t st, hipError_t* err) {
    const auto t0 = std::chrono::steady_clock::now();
    int polls = 0;
    *err = hipSuccess;
    for (;;) {
        const hipError_t q = hipStreamQuery(st);
        if (q == hipSuccess) return 1;
        if (q != hipErrorNotReady) { *err = q; return 1; }
        if ((++polls & 15) == 0) {
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (g->pool.aborted()) return 0;
            if (ms > g->wait_timeout_ms * (g->warm.load(std::memory_order_relaxed) ? 1.0 : 4.0)) { g->pool.raise_abort(); return 0; }
            if (ms > 2.0) std::this_thread::yield();
        }
    }
}

iba_status wfail(iba_group* g, int i, iba_status s, const std::string& m, bool secondary = false) { return g->pool.fail(i, s, m, secondary); }
#define W_HIP(g, i, expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return wfail(g, i, IBA_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); } while (0)
#define W_IBA(g, i, expr) do { iba_status _s = (expr); if (_s != IBA_OK) return wfail(g, i, _s, iba_last_error((g)->h[i])); } while (0)

// contiguous frame ranges balanced by scan points: the cut before rank r is the first frame boundary at or beyond r / n of
// the points (the rule of shard_frames in the Python plumbing, so that both launch styles partition a problem identically)
void shard(const iba_problem_desc* d, int n, std::vector<int32_t>& b, std::vector<int32_t>& e) {
    const int F = d->n_frames;
    const double tot = F > 0 ? (double)(d->pt_offset[F] - d->pt_offset[0]) : 0.0;
    std::vector<int32_t> cuts(n + 1, 0);
    for (int r = 0; r <= n; ++r) {
        const double target = tot * r / n;
        int i = 0;
        while (i < F && (double)(d->pt_offset[i] - d->pt_offset[0]) < target) ++i;
        cuts[r] = i;
    }
    cuts[0] = 0; cuts[n] = F;
    for (int r = 1; r <= n; ++r) cuts[r] = std::max(cuts[r], cuts[r - 1]);
    b.assign(cuts.begin(), cuts.begin() + n); e.assign(cuts.begin() + 1, cuts.end());
}

// One chunk (Bc <= group_chain(g) candidates) on every device: launch chain -> sum over the devices -> the summed block in
// g->h_part. Per device, on its own thread: pinned copy of the candidate block, kernels, ONE collective, stream drained.
iba_status eval_chunk(iba_group* g, const double* x, int Bc, EvalKind kind) {
    const auto t0 = std::chrono::steady_clock::now();
    // the values first (6 us for 64 candidates): the devices start on them; the derivatives (38 us, read by the factor kernel alone)
    // are computed on this thread while the workers issue and the kernels run
    const bool late_jets = kind == kEvalNormal || kind == kEvalFull;
    if (Bc > g->reserved_batch) {   // a larger chain than any before: the work buffers grow now, not inside the concurrent launch chains
        const iba_status rs = run_all(g, [g, Bc](int i) -> iba_status { W_IBA(g, i, reserve_batch(g->h[i], Bc)); return IBA_OK; });
        if (rs != IBA_OK) return rs;
        g->reserved_batch = Bc;
    }
    make_cands_host(x, Bc, g->cands.data(), kind == kEvalFactors);
    g->jets_ready.store(0, std::memory_order_release);
    const size_t bytes = sizeof(double) * (size_t)Bc * g->stride;
    if (g->broken.load(std::memory_order_acquire)) return gfail(g, IBA_ERR_STATE, "the group is broken: an earlier call abandoned a collective (destroy the group)");
    iba_status s = run_all(g, [g, Bc, kind, bytes, t0, late_jets](int i) -> iba_status {
        // phase 1: this device's launch chain. Whatever happens here, the worker goes on to the barrier: its peers must learn of a
        // failure BEFORE any of them enters the collective.
        iba_status mine = eval_partial_cands(g->h[i], g->cands.data(), Bc, kind, g->host_reduce ? g->h_parts_dev[i] : g->d_part[i], g->st[i], late_jets ? &g->jets_ready : nullptr);
        if (mine != IBA_OK) (void)wfail(g, i, mine, iba_last_error(g->h[i]));
        if (g->debug_fail_armed && g->debug_fail_rank == i && g->debug_fail_phase == 1 && mine == IBA_OK) mine = wfail(g, i, IBA_ERR_STATE, "injected failure before the collective (IBA_DEBUG_FAIL_RANK)");
        const bool all_ok = g->pool.meet(i, mine == IBA_OK);
        if (!all_ok) {   // nobody enters the collective: drain what this device has enqueued and report
            (void)hipStreamSynchronize(g->st[i]);
            return mine != IBA_OK ? mine : wfail(g, i, IBA_ERR_STATE, "another device of the group failed before the collective; this device's launch chain was drained", true);
        }
        // phase 2: the one collective of the call
        if (!g->host_reduce) {
            ncclResult_t r = ncclSuccess;
            if (g->debug_fail_armed && g->debug_fail_rank == i && g->debug_fail_phase == 2) r = ncclInternalError;   // (injected: this rank never enqueues its all-reduce)
            else r = rccl().AllReduce(g->d_part[i], g->d_part[i], (size_t)Bc * g->stride, ncclDouble, ncclSum, g->comm[i], g->st[i]);
            if (r != ncclSuccess) {
                g->pool.raise_abort();   // the peers are (or will be) waiting for a collective that cannot complete
                if (rccl().CommAbort && g->comm[i]) { (void)rccl().CommAbort(g->comm[i]); g->comm[i] = nullptr; }
                g->broken.store(true, std::memory_order_release);
                return wfail(g, i, IBA_ERR_HIP, std::string("ncclAllReduce: ") + rccl().GetErrorString(r));
            }
            if (i == 0) W_HIP(g, i, hipMemcpyAsync(g->h_parts[0], g->d_part[0], bytes, hipMemcpyDeviceToHost, g->st[0]));
        }
        g->enq_us[i] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();   // everything of this device is enqueued
        hipError_t qe = hipSuccess;
        if (!poll_stream(g, g->st[i], &qe)) {   // abandoned: a peer failed after the barrier, or the bounded wait ran out
            if (!g->host_reduce && rccl().CommAbort && g->comm[i]) { (void)rccl().CommAbort(g->comm[i]); g->comm[i] = nullptr; }
            g->broken.store(true, std::memory_order_release);
            return wfail(g, i, IBA_ERR_STATE, "the collective of this call was abandoned (a peer failed or the wait ran out): the communicator was aborted", true);
        }
        if (qe != hipSuccess) return wfail(g, i, IBA_ERR_HIP, std::string("hipStreamQuery: ") + hipGetErrorString(qe));
        return IBA_OK;
    }, [g, x, Bc, late_jets]() {
        if (late_jets) make_cands_jets_host(x, Bc, g->cands.data());
        g->jets_ready.store(1, std::memory_order_release);   // always: a worker may be waiting on it
    });
    g->debug_fail_armed = false;   // one injected failure per arming
    if (s != IBA_OK) return s;
    g->last_enqueue_us = *std::max_element(g->enq_us.begin(), g->enq_us.end());
    if (g->host_reduce && g->n > 1) {   // rank order: bitwise reproducible whatever the timing
        const size_t m = (size_t)Bc * g->stride;
        for (size_t k = 0; k < m; ++k) { double a = g->h_parts[0][k]; for (int i = 1; i < g->n; ++i) a += g->h_parts[i][k]; g->h_sum[k] = a; }
        g->h_part = g->h_sum.data();
    } else g->h_part = g->h_parts[0];
    g->last_issue_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    g->warm.store(true, std::memory_order_relaxed);
    return IBA_OK;
}

void stop_workers(iba_group* g) { g->pool.stop(); }
// candidates one chain of the group takes: the smallest of its handles' (all equal: same options, same parameters)
int group_chain(const iba_group* g) { int c = kMaxChain; for (iba_handle* h : g->h) if (h) c = std::min(c, chain_capacity(h)); return std::max(1, c); }
}  // namespace

extern "C" {

const char* iba_group_last_error(const iba_group* g) { return g ? g->err.c_str() : g_group_create_error.c_str(); }
int32_t iba_group_size(const iba_group* g) { return g ? g->n : 0; }

iba_status iba_group_frame_range(const iba_group* g, int32_t rank, int32_t* frame_begin, int32_t* frame_end) {
    if (!g || rank < 0 || rank >= g->n || !frame_begin || !frame_end) return IBA_ERR_INVALID_ARG;
    *frame_begin = g->f_begin[rank]; *frame_end = g->f_end[rank];
    return IBA_OK;
}

void iba_group_destroy(iba_group* g) {
    if (!g) return;
    if (g->pool.size() > 0) {
        // every device's teardown on its own thread (and device)
        (void)run_all(g, [g](int i) -> iba_status {
            const bool broken = g->broken.load(std::memory_order_acquire);   // a stream of a broken group may hold a collective that never completes
            if (g->st[i] && !broken) (void)hipStreamSynchronize(g->st[i]);
            if (g->comm[i]) { if (broken && rccl().CommAbort) (void)rccl().CommAbort(g->comm[i]); else if (!broken) (void)rccl().CommDestroy(g->comm[i]); }
            // A broken group's stream may still hold the abandoned chain: iba_destroy, hipFree and hipHostFree synchronise with the device
            // and would block for ever behind a collective that never completes (ncclCommAbort is optional — dlsym may not find it — and
            // need not retire the kernel), and the chain may still be reading the buffers. Wait a bounded time for the stream to drain;
            // if it does not, LEAK the handle and the buffers of this device and say so.
            bool retired = true;
            if (broken && g->st[i]) {
                const auto t0 = std::chrono::steady_clock::now();
                hipError_t q;
                while ((q = hipStreamQuery(g->st[i])) == hipErrorNotReady && std::chrono::steady_clock::now() - t0 < std::chrono::milliseconds(2000)) std::this_thread::sleep_for(std::chrono::milliseconds(1));
                retired = q == hipSuccess;
            }
            if (!retired) {
                std::fprintf(stderr, "[iba] iba_group_destroy: device %d still runs an abandoned collective; its handle and buffers are leaked (not freed) so that teardown cannot hang\n", g->dev[i]);
                return IBA_OK;
            }
            if (g->h[i]) iba_destroy(g->h[i]);
            if (g->d_part[i]) (void)hipFree(g->d_part[i]);
            if (g->h_parts[i]) (void)hipHostFree(g->h_parts[i]);
            if (g->st[i]) (void)hipStreamDestroy(g->st[i]);
            return IBA_OK;
        });
        stop_workers(g);
    }
    delete g;
}

iba_status iba_group_create_ex(const iba_problem_desc* desc, const iba_params* params, const int32_t* devices, int32_t n_devices, int32_t flags, iba_group** out) {
    g_group_create_error.clear();
    if (!desc || !params || !devices || !out || n_devices < 1) return gfail(nullptr, IBA_ERR_INVALID_ARG, "bad arguments");
    *out = nullptr;
    const bool host_reduce = (flags & IBA_GROUP_REDUCE_HOST) != 0;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return gfail(nullptr, IBA_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
    for (int i = 0; i < n_devices; ++i) {
        if (devices[i] < 0 || devices[i] >= ndev) return gfail(nullptr, IBA_ERR_NO_DEVICE, "device index " + std::to_string(devices[i]) + " out of range: " + std::to_string(ndev) + " device(s) visible");
        if (!host_reduce) for (int j = 0; j < i; ++j) if (devices[j] == devices[i]) return gfail(nullptr, IBA_ERR_INVALID_ARG, "a device may appear once in an RCCL group (IBA_GROUP_REDUCE_HOST lifts this)");
    }
    if (!host_reduce && !rccl().ok()) return gfail(nullptr, IBA_ERR_UNSUPPORTED, rccl().err);
    iba_group* g = new iba_group;
    g->n = n_devices; g->params = *params; g->stride = iba_partial_stride(); g->host_reduce = host_reduce;
    g->dev.assign(devices, devices + n_devices);
    g->h.assign(n_devices, nullptr); g->comm.assign(n_devices, nullptr); g->st.assign(n_devices, nullptr); g->d_part.assign(n_devices, nullptr); g->h_parts.assign(n_devices, nullptr); g->h_parts_dev.assign(n_devices, nullptr);
    g->enq_us.assign(n_devices, 0.0);
    if (const char* e = debug_env("IBA_GROUP_TIMEOUT_MS")) g->wait_timeout_ms = std::max(1.0, std::atof(e));
    if (const char* e = debug_env("IBA_DEBUG_FAIL_RANK")) { g->debug_fail_rank = std::atoi(e); g->debug_fail_armed = true; if (const char* ph = debug_env("IBA_DEBUG_FAIL_PHASE")) g->debug_fail_phase = std::atoi(ph); }
    g->cands.resize(kMaxChain); g->h_sum.resize((size_t)kMaxChain * g->stride);
    shard(desc, n_devices, g->f_begin, g->f_end);
    g->pool.start(n_devices, [g](int i) { (void)hipSetDevice(g->dev[i]); });   // once: every HIP call of a worker targets its device
    // the handles (static index builds, uploads, plane memo) are created concurrently, one per worker
    iba_status s = run_all(g, [g, desc, params](int i) -> iba_status {
        const iba_status cs = iba_create(desc, params, g->dev[i], g->f_begin[i], g->f_end[i], &g->h[i]);
        if (cs != IBA_OK) return wfail(g, i, cs, std::string("iba_create: ") + iba_last_error(nullptr));
        W_HIP(g, i, hipSetDevice(g->dev[i]));
        W_HIP(g, i, hipStreamCreateWithFlags(&g->st[i], hipStreamNonBlocking));
        W_HIP(g, i, hipMalloc((void**)&g->d_part[i], sizeof(double) * (size_t)kMaxChain * g->stride));
        W_HIP(g, i, hipHostMalloc((void**)&g->h_parts[i], sizeof(double) * (size_t)kMaxChain * g->stride, hipHostMallocMapped));
        W_HIP(g, i, hipHostGetDevicePointer((void**)&g->h_parts_dev[i], g->h_parts[i], 0));
        W_IBA(g, i, reserve_batch(g->h[i], std::min((int)IBA_MAX_BATCH, chain_capacity(g->h[i]))));   // no allocation inside an evaluation (several threads are inside HIP then)
        return IBA_OK;
    });
    if (s == IBA_OK && !host_reduce) {
        int cur = -1; (void)hipGetDevice(&cur);   // the caller's current device survives the communicator setup
        const ncclResult_t r = rccl().CommInitAll(g->comm.data(), n_devices, g->dev.data());
        if (cur >= 0) (void)hipSetDevice(cur);
        if (r != ncclSuccess) s = gfail(g, IBA_ERR_HIP, std::string("ncclCommInitAll: ") + rccl().GetErrorString(r));
    }
    if (s != IBA_OK) { g_group_create_error = g->err; iba_group_destroy(g); return s; }
    *out = g;
    return IBA_OK;
}

iba_status iba_group_create(const iba_problem_desc* desc, const iba_params* params, const int32_t* devices, int32_t n_devices, iba_group** out) {
    return iba_group_create_ex(desc, params, devices, n_devices, 0, out);
}

int32_t iba_group_comm_ranks(const iba_group* g) {   // what RCCL itself says about the communicator (0: host reduction)
    if (!g || g->host_reduce || !g->comm[0]) return 0;
    int c = 0;
    return rccl().CommCount(g->comm[0], &c) == ncclSuccess ? c : -1;
}

double iba_group_last_issue_us(const iba_group* g) { return g ? g->last_issue_us : 0.0; }
double iba_group_last_enqueue_us(const iba_group* g) { return g ? g->last_enqueue_us : 0.0; }

iba_status iba_group_set_params(iba_group* g, const iba_params* p) {
    if (!g || !p) return IBA_ERR_INVALID_ARG;
    const iba_status s = run_all(g, [g, p](int i) -> iba_status { W_IBA(g, i, iba_set_params(g->h[i], p)); W_IBA(g, i, reserve_batch(g->h[i], std::min(g->reserved_batch, chain_capacity(g->h[i])))); return IBA_OK; });
    if (s != IBA_OK) return s;
    g->params = *p;
    return IBA_OK;
}

iba_status iba_group_eval_cost(iba_group* g, const double* x, int32_t B, iba_cost_out* out) {
    if (!g || !x || !out || B < 1) return gfail(g, IBA_ERR_INVALID_ARG, "bad arguments");
    const int cap = group_chain(g);
    for (int b0 = 0; b0 < B; b0 += cap) {   // larger batches run as consecutive chains
        const int Bc = std::min(cap, B - b0);
        iba_status s = eval_chunk(g, x + 7 * b0, Bc, kEvalCost); if (s != IBA_OK) return s;
        s = iba_finalize_cost(&g->params, g->h_part, Bc, out + b0); if (s != IBA_OK) return s;
    }
    return IBA_OK;
}

iba_status iba_group_eval_bbo(iba_group* g, const double* x, int32_t B, double he_threshold, double valid_rate, iba_bbo* out) {
    if (!g || !out || B < 1) return gfail(g, IBA_ERR_INVALID_ARG, "bad arguments");
    std::vector<iba_cost_out> c((size_t)B);
    const iba_status s = iba_group_eval_cost(g, x, B, c.data()); if (s != IBA_OK) return s;
    for (int b = 0; b < B; ++b) {   // iba_global.cpp:386-392
        out[b].f = c[b].f1 * g->params.err_weight[0] + c[b].f2 * g->params.err_weight[1];
        out[b].c1 = c[b].C - he_threshold; out[b].c2 = -c[b].C - he_threshold;
        out[b].c3 = valid_rate - static_cast<double>(c[b].valid_cnt_3d_2d) / (c[b].cnt_3d_2d + 1);
    }
    return IBA_OK;
}

iba_status iba_group_eval_full(iba_group* g, const double* x, int32_t B, iba_cost_out* cost, iba_normal_out* normal) {
    if (!g || !x || !cost || !normal || B < 1) return gfail(g, IBA_ERR_INVALID_ARG, "bad arguments");
    const int cap = group_chain(g);
    for (int b0 = 0; b0 < B; b0 += cap) {
        const int Bc = std::min(cap, B - b0);
        iba_status s = eval_chunk(g, x + 7 * b0, Bc, kEvalFull); if (s != IBA_OK) return s;
        s = iba_finalize_cost(&g->params, g->h_part, Bc, cost + b0); if (s != IBA_OK) return s;
        s = iba_finalize_normal(&g->params, g->h_part, Bc, normal + b0); if (s != IBA_OK) return s;
    }
    return IBA_OK;
}

iba_status iba_group_eval_normal(iba_group* g, const double* x, int32_t B, iba_normal_out* normal) {
    if (!g || !x || !normal || B < 1) return gfail(g, IBA_ERR_INVALID_ARG, "bad arguments");
    const int cap = group_chain(g);
    for (int b0 = 0; b0 < B; b0 += cap) {
        const int Bc = std::min(cap, B - b0);
        iba_status s = eval_chunk(g, x + 7 * b0, Bc, kEvalNormal); if (s != IBA_OK) return s;
        s = iba_finalize_normal(&g->params, g->h_part, Bc, normal + b0); if (s != IBA_OK) return s;
    }
    return IBA_OK;
}

iba_status iba_group_build_problem(iba_group* g, const double* x_assoc) {
    if (!g || !x_assoc) return gfail(g, IBA_ERR_INVALID_ARG, "bad arguments");
    make_cands_host(x_assoc, 1, g->cands.data());
    return run_all(g, [g](int i) -> iba_status { W_IBA(g, i, build_problem_cands(g->h[i], g->cands.data())); return IBA_OK; });
}

iba_status iba_group_eval_factors(iba_group* g, const double* x, int32_t B, iba_normal_out* normal) {
    if (!g || !x || !normal || B < 1) return gfail(g, IBA_ERR_INVALID_ARG, "bad arguments");
    const int cap = group_chain(g);
    for (int b0 = 0; b0 < B; b0 += cap) {
        const int Bc = std::min(cap, B - b0);
        iba_status s = eval_chunk(g, x + 7 * b0, Bc, kEvalFactors); if (s != IBA_OK) return s;
        s = iba_finalize_normal(&g->params, g->h_part, Bc, normal + b0); if (s != IBA_OK) return s;
    }
    return IBA_OK;
}

iba_status iba_group_calibrate_lm(iba_group* g, const double* x0, const iba_lm_options* opt, iba_lm_result* res) {
    if (!g || !x0 || !res) return gfail(g, IBA_ERR_INVALID_ARG, "null argument");
    LmOptions o;
    if (opt) {
        o.max_outer_iterations = opt->max_outer_iterations; o.max_inner_iterations = opt->max_inner_iterations; o.min_diff = opt->min_diff;
        o.function_tolerance = opt->function_tolerance; o.gradient_tolerance = opt->gradient_tolerance; o.parameter_tolerance = opt->parameter_tolerance;
        o.initial_trust_region_radius = opt->initial_trust_region_radius;
    }
    iba_status st = IBA_OK;
    LmResult r;
    const bool ok = calibrate_lm(x0, o,
        [&](const double* x) { st = iba_group_build_problem(g, x); return st == IBA_OK; },
        [&](const double* x, double* H, double* gr, double& cost) {
            iba_normal_out n; st = iba_group_eval_factors(g, x, 1, &n);
            if (st != IBA_OK) return false;
            std::memcpy(H, n.H, sizeof(n.H)); std::memcpy(gr, n.b, sizeof(n.b)); cost = n.cost;
            return true;
        }, r);
    if (!ok) return st == IBA_OK ? IBA_ERR_STATE : st;
    std::memcpy(res->x, r.x, sizeof(r.x));
    res->outer_iterations = r.outer_iterations; res->inner_iterations = r.inner_iterations; res->evaluations = r.evaluations; res->converged = r.converged;
    res->initial_cost = r.initial_cost; res->final_cost = r.final_cost;
    return IBA_OK;
}

iba_status iba_group_calibrate_mads(iba_group* g, const double* x0, const iba_mads_options* opt, iba_mads_result* res) {
    if (!g || !x0 || !res) return gfail(g, IBA_ERR_INVALID_ARG, "null argument");
    iba_mads_options dflt;
    if (!opt) { iba_default_mads_options(x0, &dflt); opt = &dflt; }
    if (opt->max_bb_eval < 1 || !(opt->min_mesh > 0)) return gfail(g, IBA_ERR_INVALID_ARG, "bad MADS options");
    for (int i = 0; i < 7; ++i) if (!(opt->lb[i] <= opt->ub[i]) || !(opt->init_frame[i] > 0)) return gfail(g, IBA_ERR_INVALID_ARG, "bad MADS options (bounds, frame sizes)");
    MadsOptions o;
    o.max_bb_eval = opt->max_bb_eval; o.min_mesh = opt->min_mesh; o.seed = opt->seed;
    o.bases_per_poll = std::max(1, std::min(4, opt->bases_per_poll)); o.speculative = opt->speculative != 0; o.max_batch = IBA_MAX_BATCH; o.vns_max_idle = std::max(0, opt->vns_max_idle);
    for (int i = 0; i < 7; ++i) { o.lb[i] = opt->lb[i]; o.ub[i] = opt->ub[i]; o.init_frame[i] = opt->init_frame[i]; }
    iba_status st = IBA_OK;
    MadsResult r;
    const bool ok = mads_minimize(x0, o, [&](const double* X, int B, MadsPoint* out) {
        iba_bbo bbo[IBA_MAX_BATCH];
        st = iba_group_eval_bbo(g, X, B, opt->he_threshold, opt->valid_rate, bbo);
        if (st != IBA_OK) return false;
        for (int b = 0; b < B; ++b) { out[b].f = bbo[b].f; out[b].c[0] = bbo[b].c1; out[b].c[1] = bbo[b].c2; out[b].c[2] = bbo[b].c3; }
        return true;
    }, r);
    if (!ok) return st == IBA_OK ? IBA_ERR_STATE : st;
    std::memcpy(res->x, r.best.x, sizeof(res->x));
    res->f = r.best.f; res->c1 = r.best.c[0]; res->c2 = r.best.c[1]; res->c3 = r.best.c[2];
    res->feasible = r.feasible; res->evaluations = r.evaluations; res->iterations = r.iterations; res->batches = r.batches;
    res->cache_hits = r.cache_hits; res->restarts = r.restarts; res->stop_reason = r.stop_reason;
    return IBA_OK;
}

// For a caller that runs one process per GPU and owns the communicator (MPI / torchrun style): the one collective of the
// path on the caller's ncclComm_t, so that nothing but this header is needed on the caller's side.
iba_status iba_comm_allreduce(void* nccl_comm, void* d_partials, int32_t B, void* stream) {
    if (!nccl_comm || !d_partials || B < 1) return IBA_ERR_INVALID_ARG;
    if (!rccl().ok()) return IBA_ERR_UNSUPPORTED;
    const ncclResult_t r = rccl().AllReduce(d_partials, d_partials, (size_t)B * (size_t)iba_partial_stride(), ncclDouble, ncclSum, (ncclComm_t)nccl_comm, (hipStream_t)stream);
    return r == ncclSuccess ? IBA_OK : IBA_ERR_HIP;
}

// a communicator over the given devices of this process, for callers of iba_comm_allreduce that have no RCCL headers of
// their own (and for its test): comms[i] belongs to devices[i]
iba_status iba_comm_init_all(void** comms, const int32_t* devices, int32_t n) {
    if (!comms || !devices || n < 1) return IBA_ERR_INVALID_ARG;
    if (!rccl().ok()) return IBA_ERR_UNSUPPORTED;
    std::vector<ncclComm_t> c((size_t)n, nullptr); std::vector<int> d(devices, devices + n);
    if (rccl().CommInitAll(c.data(), n, d.data()) != ncclSuccess) return IBA_ERR_HIP;
    for (int i = 0; i < n; ++i) comms[i] = (void*)c[i];
    return IBA_OK;
}
int32_t iba_comm_count(void* nccl_comm) {
    int c = 0;
    if (!nccl_comm || !rccl().ok() || rccl().CommCount((ncclComm_t)nccl_comm, &c) != ncclSuccess) return -1;
    return c;
}
iba_status iba_comm_destroy(void* nccl_comm) {
    if (!nccl_comm) return IBA_ERR_INVALID_ARG;
    if (!rccl().ok()) return IBA_ERR_UNSUPPORTED;
    return rccl().CommDestroy((ncclComm_t)nccl_comm) == ncclSuccess ? IBA_OK : IBA_ERR_HIP;
}

// which librccl this process runs, and whether it is the version the library was compiled against:
// "path=<file> runtime=<code> header=<code> match=<0|1>"
iba_status iba_rccl_info(char* buf, int32_t cap, int32_t* runtime_version, int32_t* header_version) {
    if (header_version) *header_version = NCCL_VERSION_CODE;
    if (runtime_version) *runtime_version = 0;
    if (!rccl().ok()) { if (buf && cap > 0) std::snprintf(buf, (size_t)cap, "%s", rccl().err.c_str()); return IBA_ERR_UNSUPPORTED; }
    if (runtime_version) *runtime_version = rccl().runtime_version;
    if (buf && cap > 0) std::snprintf(buf, (size_t)cap, "path=%s runtime=%d header=%d match=%d", rccl().path.c_str(), rccl().runtime_version, (int)NCCL_VERSION_CODE, rccl().runtime_version == NCCL_VERSION_CODE ? 1 : 0);
    return IBA_OK;
}

}  // extern "C"
