// dq_runtime.h -- host runtime shared by the translation units of libdq_sufsort_hip.so: error reporting, the per-call
// snapshot of the DQ_* flags, per-kernel hipEvent timers, device contexts (stream + workspace + pinned areas) and their
// leases, and the entry points one unit calls in another.  Host code only (no kernels): C++17 inline variables give
// every unit the same state.
//   dq_sorter_i32.hip / dq_sorter_i64.hip   the suffix sorter (dq_sorter_impl.h) for 32- / 64-bit indices
//   dq_diff.hip                             match search, Diff.Create / Patch.Apply, the many-new-files index
//   dq_abi.hip                              the C ABI (include/dq_sufsort.h), the batch pipeline, profile getters
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cctype>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <mutex>
#include <string>
#include <memory>
#include <thread>
#include <vector>

#include <sched.h>

#include "../../include/dq_sufsort.h"

namespace dq {

constexpr int kSmallMaxN = 8192;          // largest text the single-workgroup sorter takes (dq_small.h)

// ------------------------------------------------------------------ errors
inline thread_local std::string t_err;
inline thread_local int64_t t_info[3] = {0, 0, 0};
// the last Diff.Create / index diff on this thread (dq_last_diff_info): Search calls of the loop, windows, positions
// asked again exactly, launches of the device's anchor scan that were given back to the host loop, workgroups of its grid
inline thread_local int64_t t_diff_info[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};

// the last dq_sufsort_hip_batch_i32 on this thread (dq_last_batch_info): inputs through the pipelines, microseconds the
// copy-in / sort / copy-out stages were busy (summed over the device shares), wall microseconds of the slowest share,
// device shares whose host threads were bound to their device's NUMA node
inline thread_local int64_t t_batch_info[6] = {0, 0, 0, 0, 0, 0};

inline int fail(int code, const char *what, hipError_t e = hipSuccess)
{
    char buf[512];
    if (e != hipSuccess)
        snprintf(buf, sizeof buf, "%s: %s (%d)", what, hipGetErrorString(e), (int)e);
    else
        snprintf(buf, sizeof buf, "%s", what);
    t_err = buf;
    return code;
}

// ------------------------------------------------------------------ fault injection (tests of the error paths)
// DQ_FAULT = alloc:K | hip:K | spin, read once per outermost call on this thread (EnvScope) and, like every DQ_* flag but
// DQ_TRACE, only under DQ_DEBUG_FLAGS=1:
//   alloc:K   the K-th device / pinned allocation of the call fails as if the device were out of memory   -> DQ_ERR_OOM
//   hip:K     the K-th checked HIP call of the call (copies, memsets, launches, event work) fails          -> DQ_ERR_HIP
//   spin      every bounded device spin gives up at its first empty poll: the look-back of radix_rank_kernel /
//             seg_fused_kernel (-> DQ_ERR_HIP) and the answer exchange of anchor_scan_kernel (-> the host loop)
// What the tests then check: the error code and message, nothing written to the caller's output, the next call on the
// same thread correct, dq_sufsort_hip_release leaving no allocation behind (failure detection).
struct FaultPlan {
    int alloc_at = 0, hip_at = 0;       // 0: off
    int alloc_seen = 0, hip_seen = 0;
    bool spin = false;
};
inline thread_local FaultPlan t_fault;
inline bool fault_alloc() { return t_fault.alloc_at > 0 && ++t_fault.alloc_seen == t_fault.alloc_at; }
inline bool fault_hip() { return t_fault.hip_at > 0 && ++t_fault.hip_seen == t_fault.hip_at; }

// every device / pinned allocation of the library goes through these two (DQ_FAULT=alloc:K counts them)
inline hipError_t dq_malloc(void **p, size_t bytes) { return fault_alloc() ? hipErrorOutOfMemory : hipMalloc(p, bytes); }
inline hipError_t dq_host_malloc(void **p, size_t bytes, unsigned flags)
{
    return fault_alloc() ? hipErrorOutOfMemory : hipHostMalloc(p, bytes, flags);
}

#define HIP_TRY(expr)                                                                   \
    do {                                                                                \
        if (t_fault.hip_at && fault_hip())                                              \
            return fail(DQ_ERR_HIP, "injected fault (DQ_FAULT=hip) instead of " #expr); \
        hipError_t e_ = (expr);                                                         \
        if (e_ != hipSuccess)                                                           \
            return fail(e_ == hipErrorOutOfMemory ? DQ_ERR_OOM : DQ_ERR_HIP, #expr, e_); \
    } while (0)

// ------------------------------------------------------------------ DQ_* flags
// Every entry point reads the DQ_* environment flags through env(): the first lookup of a name inside a call asks
// the process environment, later ones get the same answer -- a call sees ONE consistent set of flags, each variable
// is read once per call and thread, and nothing on the per-kernel path touches the environment.  (The tests flip
// flags between calls, so the answers are not kept beyond the outermost call on this thread.)
struct EnvCache {
    struct Entry { const char *name; bool set; std::string val; };
    static constexpr int kMax = 64;
    Entry e[kMax];
    int count = 0, depth = 0;
    bool debug_flags = false;           // DQ_DEBUG_FLAGS as the outermost call on this thread found it
};
inline thread_local EnvCache t_env;

// The adaptive choices are compiled in; the DQ_* overrides (forced paths of the tests, experiment knobs, fault
// injection) are honoured only in a process that sets DQ_DEBUG_FLAGS=1 -- a stray DQ_PACKED in a production environment
// changes nothing.  Exempt: DQ_TRACE (prints, decides nothing) and DQ_HIP_DEVICE (which device "-1" means).
inline bool env_gated(const char *name)
{
    return strcmp(name, "DQ_TRACE") != 0 && strcmp(name, "DQ_HIP_DEVICE") != 0 && strcmp(name, "DQ_DEBUG_FLAGS") != 0 &&
           strcmp(name, "DQ_NUMA_BIND") != 0;
}

inline const char *env(const char *name)
{
    EnvCache &c = t_env;
    for (int i = 0; i < c.count; ++i)
        if (c.e[i].name == name || strcmp(c.e[i].name, name) == 0) return c.e[i].set ? c.e[i].val.c_str() : nullptr;
    const char *v = getenv(name);
    if (v && env_gated(name)) {
        const char *g = c.depth > 0 ? (c.debug_flags ? "1" : nullptr) : getenv("DQ_DEBUG_FLAGS");
        if (!g || atoi(g) == 0) v = nullptr;
    }
    if (c.depth == 0 || c.count == EnvCache::kMax) return v;          // outside an entry point: nothing is kept
    EnvCache::Entry &x = c.e[c.count++];
    x.name = name; x.set = v != nullptr; x.val = v ? v : "";
    return x.set ? x.val.c_str() : nullptr;
}

struct EnvScope {
    EnvScope()
    {
        if (t_env.depth++ != 0) return;
        t_env.count = 0;
        const char *g = getenv("DQ_DEBUG_FLAGS");
        t_env.debug_flags = g && atoi(g) != 0;
        t_fault = FaultPlan{};
        if (const char *f = env("DQ_FAULT")) {
            if (strncmp(f, "alloc:", 6) == 0) t_fault.alloc_at = std::max(1, atoi(f + 6));
            else if (strncmp(f, "hip:", 4) == 0) t_fault.hip_at = std::max(1, atoi(f + 4));
            else if (strcmp(f, "spin") == 0) t_fault.spin = true;
        }
    }
    ~EnvScope() { if (--t_env.depth == 0) t_fault = FaultPlan{}; }
};

// ------------------------------------------------------------------ profiling
struct KernelStat { int64_t launches = 0; double ms = 0; int64_t elems = 0; int64_t bytes = 0; };
inline std::mutex g_prof_mu;
inline KernelStat g_prof[DQ_K_COUNT];
inline std::atomic<int> g_prof_on{0};

inline const char *const kKernelNames[DQ_K_COUNT] = {
    "text_hist_kernel", "radix_hist_kernel", "radix_rank_kernel", "seg_fused_kernel",
    "tie_seam_kernel", "tie_collect_kernel", "small_group_finish_kernel", "small_group_round_kernel",
    "isa_update_kernel", "isa_from_pairs_kernel", "key2_from_pairs_kernel", "gather_key2_kernel",
    "gather_text_key_kernel", "isa_from_sa_kernel", "small_sufsort_kernel", "bucket_sort_kernel",
    "match_search_kernel", "pair_chain_kernels", "mid_group_round_kernel", "runlen_kernels",
    "split_pass_kernel", "bucket_finish_kernel", "split_round0_aux_kernels"};

struct ProfRec { int cat; hipEvent_t a, b; int64_t elems, bytes; };

// ------------------------------------------------------------------ per-device context
struct DeviceCtx {
    std::mutex mu;
    int dev = -1;
    int ncu = 0;                        // compute units of the device (grid of the persistent kernels)
    hipStream_t stream = nullptr;
    char *ws = nullptr;
    size_t ws_bytes = 0;
    int64_t *pinned = nullptr;          // 8 KiB pinned: readback area [0, 4 KiB), upload staging [4 KiB, 8 KiB)
    uint8_t *pinned_io = nullptr;       // short texts: text in / SA out, read and written by the kernel itself
    hipEvent_t readback = nullptr;      // "the pinned readback has landed" (work queued behind it keeps running)
    std::vector<ProfRec> pending;
    std::vector<hipEvent_t> pool;
    // batch pipeline (dq_sufsort_hip_batch_i32): device slots and streams, kept between calls
    std::mutex batch_mu;                // one batch at a time per device
    uint8_t *bslot_text[3] = {nullptr, nullptr, nullptr};
    int32_t *bslot_sa[3] = {nullptr, nullptr, nullptr};
    size_t bslot_cap = 0;               // bytes of text each slot holds
    hipStream_t b_in = nullptr, b_sort = nullptr, b_out = nullptr;
    // Diff.Create (dq_bsdiff_create / dq_bsdiff_index_diff): one diff at a time per device; its device scratch
    // (new file + mailbox; for the one-shot form also old file, suffix array and prefix table) and the pinned
    // answer windows are kept between calls -- hipMalloc / hipHostMalloc / hipFree are synchronous driver calls
    std::mutex diff_mu;
    char *diff_dev = nullptr;           // per-diff scratch
    size_t diff_dev_bytes = 0;
    char *diff_idx = nullptr;           // index buffers of the one-shot form
    size_t diff_idx_bytes = 0;
    char *diff_pinned = nullptr;        // fixed size (SearchWindows)
    // the persistent grid of the device's anchor scan (dq_anchor_scan.h) must be resident as a whole: workgroups the
    // device holds at once (occupancy x compute units; -1: not asked yet), and how many diffs still skip the device
    // scan after a launch whose workgroups waited in vain for each other (a device kept full by other work)
    int scan_groups_cap = -1;           // (a grid of kAsGroups workgroups, the widest: 64 KB of LDS each)
    int scan_groups_cap_narrow = -1;    // (grids of 32 workgroups, 16 KB of LDS each)
    int scan_skip = 0;
    // several grids on one new file: how many slots of each one's pinned list may not read "pending" any more (all of
    // them before the first use)
    std::shared_ptr<void> scan_pool;    // emitter threads' buffers of the device scan's chains (dq_diff.hip: ScanPool), kept between diffs
    unsigned long long scan_seq = 0;    // launches of the device scan so far (a chain writes its launch's number behind its result)
    int64_t scan_dirty[16] = {1 << 16, 1 << 16, 1 << 16, 1 << 16, 1 << 16, 1 << 16, 1 << 16, 1 << 16,
                              1 << 16, 1 << 16, 1 << 16, 1 << 16, 1 << 16, 1 << 16, 1 << 16, 1 << 16};
};
constexpr int kMaxDevices = 64;
// A device has several contexts ("slots": stream + workspace + pinned areas each).  Texts of up to kSlotSmallN bytes
// take whichever slot is free, so that the threads of a host sharing one provider (the reference's benchmark keeps
// static singletons, SuffixSortingBenchmarks.cs:59-61) overlap their sorts instead of queueing behind one mutex;
// anything larger, the match search, the batch pipeline and the diffs use slot 0 (a large sort fills the device anyway).
constexpr int kCtxSlots = 4;
constexpr int64_t kSlotSmallN = 4ll << 20;
constexpr size_t kSmallTextArea = kSmallMaxN + 64;
constexpr size_t kSmallIoBytes = kSmallTextArea + (size_t)kSmallMaxN * 8;
struct DeviceState {
    DeviceCtx slot[kCtxSlots];
    std::atomic<unsigned> next{0};
};
inline DeviceState g_dev[kMaxDevices];
inline DeviceCtx &ctx0(int dev) { return g_dev[dev].slot[0]; }

// holds one slot of a device for the duration of a sort
struct SlotLease {
    DeviceCtx *c = nullptr;
    SlotLease(int dev, int64_t n)
    {
        DeviceState &d = g_dev[dev];
        if (n > kSlotSmallN) { c = &d.slot[0]; c->mu.lock(); return; }
        for (int k = 1; k < kCtxSlots && !c; ++k)
            if (d.slot[k].mu.try_lock()) c = &d.slot[k];
        if (!c && d.slot[0].mu.try_lock()) c = &d.slot[0];
        if (!c) {
            c = &d.slot[1 + d.next.fetch_add(1u, std::memory_order_relaxed) % (unsigned)(kCtxSlots - 1)];
            c->mu.lock();
        }
    }
    ~SlotLease() { c->mu.unlock(); }
    SlotLease(const SlotLease &) = delete;
    SlotLease &operator=(const SlotLease &) = delete;
};

inline int init_ctx(DeviceCtx &c, int dev)
{
    HIP_TRY(hipSetDevice(dev));
    if (c.dev == dev) return DQ_OK;
    // c.dev is published only once every resource exists: a failure half way (e.g. pinned memory
    // exhausted) frees what was made and leaves the context unbuilt, so the next call retries
    hipError_t e = hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = dq_host_malloc((void **)&c.pinned, 8192, hipHostMallocDefault);
    if (e == hipSuccess) e = dq_host_malloc((void **)&c.pinned_io, kSmallIoBytes, hipHostMallocDefault);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c.readback, hipEventDisableTiming);
    if (e != hipSuccess) {
        if (c.readback) (void)hipEventDestroy(c.readback);
        if (c.pinned_io) (void)hipHostFree(c.pinned_io);
        if (c.pinned) (void)hipHostFree(c.pinned);
        if (c.stream) (void)hipStreamDestroy(c.stream);
        c.readback = nullptr; c.pinned_io = nullptr; c.pinned = nullptr; c.stream = nullptr;
        return fail(e == hipErrorOutOfMemory ? DQ_ERR_OOM : DQ_ERR_HIP, "device context setup", e);
    }
    c.dev = dev;
    return DQ_OK;
}

// A sort that failed half way leaves timing events queued in c.pending: hand them back to the pool
// (after the stream has drained, so none is still being recorded).
inline void drop_pending(DeviceCtx &c, hipStream_t st)
{
    (void)hipStreamSynchronize(st);
    for (ProfRec &r : c.pending) {
        if (r.a) c.pool.push_back(r.a);
        if (r.b) c.pool.push_back(r.b);
    }
    c.pending.clear();
}

inline int ensure_ws(DeviceCtx &c, size_t bytes)
{
    if (c.ws_bytes >= bytes) return DQ_OK;
    if (c.ws) { (void)hipFree(c.ws); c.ws = nullptr; c.ws_bytes = 0; }
    hipError_t e = dq_malloc((void **)&c.ws, bytes);
    if (e != hipSuccess) return fail(DQ_ERR_OOM, "hipMalloc(workspace)", e);
    c.ws_bytes = bytes;
    return DQ_OK;
}

struct Launcher {
    DeviceCtx &c;
    hipStream_t st;
    int prof;                         // 0 off, 1 every kernel, 2 only radix_rank_kernel, 100 + c only category c
    bool active = false;
    int begin(int cat, int64_t elems, int64_t bytes)
    {
        active = prof == 1 || (prof == 2 && cat == DQ_K_RADIX_RANK) || prof == 100 + cat;
        if (!active) return DQ_OK;
        c.pending.push_back(ProfRec{cat, nullptr, nullptr, elems, bytes});      // queued first: an error below leaks nothing
        ProfRec &r = c.pending.back();
        for (hipEvent_t *ev : {&r.a, &r.b}) {
            if (!c.pool.empty()) { *ev = c.pool.back(); c.pool.pop_back(); }
            else HIP_TRY(hipEventCreate(ev));
        }
        HIP_TRY(hipEventRecord(r.a, st));
        return DQ_OK;
    }
    int end()
    {
        if (!active) return DQ_OK;
        HIP_TRY(hipEventRecord(c.pending.back().b, st));
        return DQ_OK;
    }
};

inline int flush_profile(DeviceCtx &c)
{
    if (c.pending.empty()) return DQ_OK;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (ProfRec &r : c.pending) {
        float ms = 0;
        HIP_TRY(hipEventSynchronize(r.b));
        HIP_TRY(hipEventElapsedTime(&ms, r.a, r.b));
        KernelStat &s = g_prof[r.cat];
        s.launches += 1; s.ms += ms; s.elems += r.elems; s.bytes += r.bytes;
        c.pool.push_back(r.a); c.pool.push_back(r.b);
    }
    c.pending.clear();
    return DQ_OK;
}

#define LAUNCH(L, cat, elems, bytes, ...)                 \
    do {                                                  \
        int rc_ = (L).begin(cat, elems, bytes);           \
        if (rc_ != DQ_OK) return rc_;                     \
        __VA_ARGS__;                                      \
        HIP_TRY(hipGetLastError());                       \
        rc_ = (L).end();                                  \
        if (rc_ != DQ_OK) return rc_;                     \
    } while (0)


// ------------------------------------------------------------------ workspace carving
inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

inline int bit_length(uint64_t x) { return x == 0 ? 1 : 64 - __builtin_clzll(x); }

// ------------------------------------------------------------------ short texts: one launch
// Largest n the single-workgroup sorter takes (DQ_SMALL_N=0 sends everything down the
// device-wide pipeline; the tests use that to keep the pipeline covered on the fixtures).
inline int64_t small_limit()
{
    if (const char *v = env("DQ_SMALL_N")) return std::min<int64_t>(std::max(0, atoi(v)), kSmallMaxN);
    return kSmallMaxN;
}

inline int resolve_device(int32_t device, int *out)
{
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) return fail(DQ_ERR_NO_DEVICE, "no HIP device available", e);
    if (device < 0) {
        const char *v = env("DQ_HIP_DEVICE");
        device = v ? atoi(v) : 0;
    }
    if (device < 0 || device >= count || device >= kMaxDevices)
        return fail(DQ_ERR_BAD_ARGS, "device ordinal out of range");
    *out = device;
    return DQ_OK;
}

// ------------------------------------------------------------------ NUMA placement of a device's host threads
// The batch pipeline's stage threads copy through pageable host memory: 8 devices x 50+ GB/s of staged copies meet in
// host memory, and a thread on the far socket pays the inter-socket link both ways.  Each device's threads are
// therefore bound to the CPUs of the NUMA node its PCIe function hangs off:
//   hipDeviceGetPCIBusId -> /sys/bus/pci/devices/<domain:bus:dev.fn>/numa_node -> /sys/devices/system/node/node<k>/cpulist
// Nothing happens where any of these is absent or says -1 (single-socket hosts, containers without sysfs), or under
// DQ_NUMA_BIND=0.  Only threads the library itself starts are bound -- never the caller's.
inline int device_numa_node(int dev)
{
    static std::mutex mu;
    static int cache[kMaxDevices];
    static bool known[kMaxDevices];
    if (dev < 0 || dev >= kMaxDevices) return -1;
    std::lock_guard<std::mutex> lk(mu);
    if (known[dev]) return cache[dev];
    int node = -1;
    char bdf[64] = {0};
    if (hipDeviceGetPCIBusId(bdf, (int)sizeof bdf, dev) == hipSuccess && bdf[0]) {
        for (char *p = bdf; *p; ++p) *p = (char)tolower((unsigned char)*p);
        char path[160];
        snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bdf);
        if (FILE *f = fopen(path, "r")) {
            if (fscanf(f, "%d", &node) != 1) node = -1;
            fclose(f);
        }
    }
    known[dev] = true;
    cache[dev] = node;
    return node;
}

// the CPUs of a NUMA node ("0-31,64-95"); false: unknown
inline bool numa_node_cpus(int node, cpu_set_t *set)
{
    if (node < 0) return false;
    char path[96];
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    FILE *f = fopen(path, "r");
    if (!f) return false;
    char buf[4096] = {0};
    const bool got = fgets(buf, sizeof buf, f) != nullptr;
    fclose(f);
    if (!got) return false;
    CPU_ZERO(set);
    int any = 0;
    for (char *p = buf; *p;) {
        char *end = nullptr;
        const long a = strtol(p, &end, 10);
        if (end == p) break;
        long b = a;
        p = end;
        if (*p == '-') { b = strtol(p + 1, &end, 10); if (end == p + 1) break; p = end; }
        for (long c = a; c <= b && c < CPU_SETSIZE; ++c) { if (c >= 0) { CPU_SET((int)c, set); ++any; } }
        while (*p == ',' || *p == ' ' || *p == '\n') ++p;
    }
    return any > 0;
}

// binds the CALLING thread (one the library started) to the device's NUMA node; true if it did
inline bool bind_this_thread_to_device(int dev)
{
    if (const char *v = env("DQ_NUMA_BIND")) if (atoi(v) == 0) return false;
    cpu_set_t want, have;
    if (!numa_node_cpus(device_numa_node(dev), &want)) return false;
    // (never widen what the process was given: a container's cpuset, taskset)
    if (sched_getaffinity(0, sizeof have, &have) != 0) return false;
    cpu_set_t both;
    CPU_AND(&both, &want, &have);
    if (CPU_COUNT(&both) == 0) return false;
    return sched_setaffinity(0, sizeof both, &both) == 0;
}

struct JoinAll {                        // joins whatever was started, also when leaving by exception
    std::vector<std::thread> v;
    ~JoinAll() { for (std::thread &t : v) if (t.joinable()) t.join(); }
};

// ------------------------------------------------------------------ entry points across translation units
// the suffix sorter (dq_sorter_impl.h; instantiated for int32_t in dq_sorter_i32.hip, int64_t in dq_sorter_i64.hip)
// What a caller inside the library knows about its text (bzip2's block transform, dq_bz2.h / dq_diff.hip):
//   doubled     the text is some block twice (n even, text[i] == text[i + n / 2]): the pairs (i, i + n / 2) leave the
//               list as soon as they are what is left of a tie group (twin_mark_kernel)
//   run_period  a good part of the text lies in stretches that repeat with this period (<= 8): run lengths with that
//               period and the run-order round up front (dq_runs.h), as for texts with long runs of one byte
struct SortHints {
    bool doubled = false;
    int run_period = 0;
};
template <typename IdxT> int sufsort_host(const uint8_t *text, int64_t n, IdxT *sa, int32_t device, SortHints hints = SortHints());
template <typename IdxT> int sufsort_dev(const void *d_text, int64_t n, void *d_sa, int32_t device, void *stream);
template <typename IdxT> int64_t sufsort_workspace_bytes(int64_t n);
extern template int sufsort_host<int32_t>(const uint8_t *, int64_t, int32_t *, int32_t, SortHints);
extern template int sufsort_host<int64_t>(const uint8_t *, int64_t, int64_t *, int32_t, SortHints);
extern template int sufsort_dev<int32_t>(const void *, int64_t, void *, int32_t, void *);
extern template int sufsort_dev<int64_t>(const void *, int64_t, void *, int32_t, void *);
extern template int64_t sufsort_workspace_bytes<int32_t>(int64_t);
extern template int64_t sufsort_workspace_bytes<int64_t>(int64_t);

// match search + BSDIFF40 (dq_diff.hip)
int match_search_dev_i32(const void *d_old, int64_t n, const void *d_sa, const void *d_new, int64_t m, const int64_t *d_scans,
                         int64_t scan0, int64_t count, int64_t cap, void *d_pos, void *d_len, int32_t device, void *stream);
int match_search_dev_i64(const void *d_old, int64_t n, const void *d_sa, const void *d_new, int64_t m, const int64_t *d_scans,
                         int64_t scan0, int64_t count, int64_t cap, void *d_pos, void *d_len, int32_t device, void *stream);
int match_search_host_i32(const uint8_t *old, int64_t n, const int32_t *sa, const uint8_t *nw, int64_t m, const int64_t *scans,
                          int64_t scan0, int64_t count, int64_t cap, int32_t *pos, int32_t *len, int32_t device);
int match_search_host_i64(const uint8_t *old, int64_t n, const int64_t *sa, const uint8_t *nw, int64_t m, const int64_t *scans,
                          int64_t scan0, int64_t count, int64_t cap, int64_t *pos, int64_t *len, int32_t device);
int bsdiff_scan_raw(const uint8_t *old, int64_t n, const uint8_t *nw, int64_t m, int32_t device, std::vector<int64_t> &ctrl,
                    std::vector<uint8_t> &diff, std::vector<uint8_t> &extra, int64_t stats[3]);
int bsdiff_create_host(const uint8_t *old, int64_t n, const uint8_t *nw, int64_t m, int32_t device, std::vector<uint8_t> &patch);
int bspatch_apply_host(const uint8_t *old, int64_t n, const uint8_t *patch, int64_t plen, uint8_t *out, int64_t cap, int64_t *out_len);
int diff_index_new(const uint8_t *old, int64_t n, int32_t device, const void *d_old, const void *d_sa, void **index_out);
int diff_index_clone(const void *index, int32_t device, void **index_out);
int diff_index_buffers(const void *index, const void **d_old, const void **d_sa, int64_t *n);
int diff_index_diff(const void *index, const uint8_t *nw, int64_t m, std::vector<uint8_t> &patch);
void diff_index_delete(void *index);

}  // namespace dq
