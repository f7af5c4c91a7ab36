// dq_abi.hip -- the C ABI of libdq_sufsort_hip.so (include/dq_sufsort.h) and the batch pipeline behind
// dq_sufsort_hip_batch_i32.  Host code only: the kernels live in dq_sorter_i32/i64.hip and dq_diff.hip.
//
// This library contains no CPU sorting path: if HIP is unusable the entry points fail.
#include "dq_runtime.h"

namespace dq {
namespace {

// ------------------------------------------------------------------ batch: one device's share, pipelined
// Three stages on three host threads and three streams, kBatchSlots device buffers in flight:
//   copy-in   text j -> slot          (pageable host memory: the copy blocks its thread, not the others)
//   sort      slot's text -> slot's SA (device-resident sorter; one sort at a time per device anyway)
//   copy-out  slot's SA -> sas[j]
// so the PCIe transfers of neighbouring inputs overlap the sort.  Inputs that need the
// short-text path or that are larger than the slot size go through the plain host entry point.
constexpr int kBatchSlots = 3;


// what one device share reports back (dq_last_batch_info): inputs through its pipeline, busy microseconds per stage
struct ShareStats { int64_t piped = 0, in_us = 0, sort_us = 0, out_us = 0, wall_us = 0, bound = 0; };
inline int64_t us_since(std::chrono::steady_clock::time_point t0)
{
    return std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
}

int batch_on_device(int device, const std::vector<int> &jobs, const uint8_t *const *texts, const int64_t *lens,
                    int32_t *const *sas, std::string *err, ShareStats *stats)
{
    const auto t_share = std::chrono::steady_clock::now();
    // this thread was started by dq_sufsort_hip_batch_i32 for this device: it and the three stage threads below run on
    // the CPUs of the device's NUMA node (dq_runtime.h; nothing happens where the platform does not say)
    stats->bound = bind_this_thread_to_device(device) ? 1 : 0;
    struct WallAtExit { ShareStats *s; std::chrono::steady_clock::time_point t0; ~WallAtExit() { s->wall_us = us_since(t0); } } wall_guard{stats, t_share};
    auto plain = [&](int j) -> int {
        int rc = sufsort_host<int32_t>(texts[j], lens[j], sas[j], device);
        if (rc != DQ_OK) *err = t_err;
        return rc;
    };
    const int64_t direct = std::max<int64_t>(small_limit(), 2);             // these bypass the pipeline
    int64_t cap = 0;
    int big = 0;
    for (int j : jobs)
        if (lens[j] > direct) { cap = std::max(cap, lens[j]); ++big; }
    if (big < 3 || cap > (1ll << 30)) {                       // nothing to overlap / slots would be huge
        for (int j : jobs) { int rc = plain(j); if (rc != DQ_OK) return rc; }
        return DQ_OK;
    }
    if (hipSetDevice(device) != hipSuccess) { *err = "hipSetDevice failed"; return DQ_ERR_HIP; }
    // the three device slots and streams live in the device context: allocated once, grown on demand
    DeviceCtx &bc = ctx0(device);
    std::lock_guard<std::mutex> batch_lock(bc.batch_mu);
    struct Slot { uint8_t *text = nullptr; int32_t *sa = nullptr; int job = -1; };
    Slot slots[kBatchSlots];
    {
        bool ok = true;
        for (hipStream_t *st : {&bc.b_in, &bc.b_sort, &bc.b_out})
            if (!*st) ok = ok && hipStreamCreateWithFlags(st, hipStreamNonBlocking) == hipSuccess;
        if (ok && bc.bslot_cap < (size_t)cap) {
            for (int k = 0; k < kBatchSlots; ++k) {
                if (bc.bslot_text[k]) (void)hipFree(bc.bslot_text[k]);
                if (bc.bslot_sa[k]) (void)hipFree(bc.bslot_sa[k]);
                bc.bslot_text[k] = nullptr; bc.bslot_sa[k] = nullptr;
            }
            bc.bslot_cap = 0;
            for (int k = 0; k < kBatchSlots; ++k)
                ok = ok && dq_malloc((void **)&bc.bslot_text[k], (size_t)cap + 64) == hipSuccess &&
                     dq_malloc((void **)&bc.bslot_sa[k], (size_t)cap * sizeof(int32_t)) == hipSuccess;
            if (ok) bc.bslot_cap = (size_t)cap;
        }
        if (!ok) { *err = "batch slot allocation failed"; return DQ_ERR_OOM; }
        for (int k = 0; k < kBatchSlots; ++k) { slots[k].text = bc.bslot_text[k]; slots[k].sa = bc.bslot_sa[k]; }
    }
    hipStream_t s_in = bc.b_in, s_sort = bc.b_sort, s_out = bc.b_out;

    // slot hand-over: free -> filled (text on the device) -> sorted (SA on the device) -> free
    std::mutex mu;
    std::condition_variable cv;
    std::vector<int> filled, sorted, freeq;
    for (int k = 0; k < kBatchSlots; ++k) freeq.push_back(k);
    bool in_done = false, sort_done = false;
    std::atomic<int> failed{DQ_OK};
    std::string errs[3];
    auto take = [&](std::vector<int> &q, const bool *producer_done) -> int {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return !q.empty() || (producer_done && *producer_done) || failed.load() != DQ_OK; });
        if (q.empty()) return -1;
        const int k = q.front();
        q.erase(q.begin());
        return k;
    };
    auto give = [&](std::vector<int> &q, int k) { { std::lock_guard<std::mutex> lk(mu); q.push_back(k); } cv.notify_all(); };
    auto fail_stage = [&](int stage, int rc, const std::string &what) {
        {   // under the mutex: a waiter between its predicate check and its block must not miss this
            std::lock_guard<std::mutex> lk(mu);
            errs[stage] = what;
            int expect = DQ_OK;
            failed.compare_exchange_strong(expect, rc);
        }
        cv.notify_all();
    };

    // (busy time per stage: what tells a share that waits for the GPU from one that waits for host memory / PCIe)
    int64_t busy_in = 0, busy_sort = 0, busy_out = 0, piped = 0;
    auto stage_in = [&]() {
        (void)hipSetDevice(device);
        (void)bind_this_thread_to_device(device);
        for (int j : jobs) {
            if (lens[j] <= direct) continue;                                // handled after the pipeline
            const int k = take(freeq, nullptr);
            if (k < 0 || failed.load() != DQ_OK) break;
            slots[k].job = j;
            const auto t0 = std::chrono::steady_clock::now();
            hipError_t e = hipMemcpyAsync(slots[k].text, texts[j], (size_t)lens[j], hipMemcpyHostToDevice, s_in);
            if (e == hipSuccess) e = hipStreamSynchronize(s_in);
            busy_in += us_since(t0);
            if (e != hipSuccess) { fail_stage(0, DQ_ERR_HIP, std::string("batch copy-in: ") + hipGetErrorString(e)); break; }
            ++piped;
            give(filled, k);
        }
        { std::lock_guard<std::mutex> lk(mu); in_done = true; }
        cv.notify_all();
    };
    auto stage_sort = [&]() {
        (void)hipSetDevice(device);
        (void)bind_this_thread_to_device(device);
        for (;;) {
            const int k = take(filled, &in_done);
            if (k < 0 || failed.load() != DQ_OK) break;
            const int j = slots[k].job;
            const auto t0 = std::chrono::steady_clock::now();
            int rc = sufsort_dev<int32_t>(slots[k].text, lens[j], slots[k].sa, device, s_sort);
            busy_sort += us_since(t0);
            if (rc != DQ_OK) { fail_stage(1, rc, t_err); break; }
            give(sorted, k);
        }
        { std::lock_guard<std::mutex> lk(mu); sort_done = true; }
        cv.notify_all();
    };
    auto stage_out = [&]() {
        (void)hipSetDevice(device);
        (void)bind_this_thread_to_device(device);
        for (;;) {
            const int k = take(sorted, &sort_done);
            if (k < 0 || failed.load() != DQ_OK) break;
            const int j = slots[k].job;
            const auto t0 = std::chrono::steady_clock::now();
            hipError_t e = hipMemcpyAsync(sas[j], slots[k].sa, (size_t)lens[j] * sizeof(int32_t), hipMemcpyDeviceToHost, s_out);
            if (e == hipSuccess) e = hipStreamSynchronize(s_out);
            busy_out += us_since(t0);
            if (e != hipSuccess) { fail_stage(2, DQ_ERR_HIP, std::string("batch copy-out: ") + hipGetErrorString(e)); break; }
            give(freeq, k);
        }
    };
    {
        // a thread that cannot be started (std::system_error) fails the batch instead of terminating:
        // the stages already running are woken through fail_stage and joined
        JoinAll stages;
        try {
            stages.v.emplace_back(stage_in);
            stages.v.emplace_back(stage_sort);
            stages.v.emplace_back(stage_out);
        } catch (const std::exception &e) {
            fail_stage(0, DQ_ERR_OOM, std::string("batch: cannot start a pipeline thread: ") + e.what());
        }
    }
    stats->piped = piped; stats->in_us = busy_in; stats->sort_us = busy_sort; stats->out_us = busy_out;    // (the stages are joined)
    if (failed.load() != DQ_OK) {
        for (const std::string &e : errs) if (!e.empty()) { *err = e; break; }
        return failed.load();
    }
    for (int j : jobs)
        if (lens[j] <= direct) { int rc = plain(j); if (rc != DQ_OK) return rc; }
    return DQ_OK;
}

}  // namespace
}  // namespace dq

using namespace dq;

// ====================================================================== C ABI
extern "C" {

int32_t dq_abi_version(void) { return DQ_ABI_VERSION; }

int32_t dq_device_count(void)
{
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) return 0;
    return count;
}

const char *dq_last_error(void) { return t_err.c_str(); }

int32_t dq_sufsort_hip_i32(const uint8_t *text, int64_t n, int32_t *sa, int32_t device)
{
    EnvScope flags;
    return sufsort_host<int32_t>(text, n, sa, device);
}

int32_t dq_sufsort_hip_i64(const uint8_t *text, int64_t n, int64_t *sa, int32_t device)
{
    EnvScope flags;
    return sufsort_host<int64_t>(text, n, sa, device);
}

int32_t dq_sufsort_hip_dev_i32(const void *d_text, int64_t n, void *d_sa, int32_t device, void *stream)
{
    EnvScope flags;
    return sufsort_dev<int32_t>(d_text, n, d_sa, device, stream);
}

int32_t dq_sufsort_hip_dev_i64(const void *d_text, int64_t n, void *d_sa, int32_t device, void *stream)
{
    EnvScope flags;
    return sufsort_dev<int64_t>(d_text, n, d_sa, device, stream);
}

int32_t dq_sufsort_hip_batch_i32(int32_t count, const uint8_t *const *texts, const int64_t *lens,
                                 int32_t *const *sas, int32_t ndev, const int32_t *devs)
{
    EnvScope flags;
    if (count < 0 || ndev <= 0 || (count > 0 && (!texts || !lens || !sas)))
        return fail(DQ_ERR_BAD_ARGS, "bad batch arguments");
    if (count == 0) return DQ_OK;
    try {
    // longest-processing-time-first assignment of inputs to devices
    std::vector<int> order(count);
    for (int i = 0; i < count; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return lens[a] > lens[b]; });
    std::vector<std::vector<int>> share(ndev);
    std::vector<int64_t> load(ndev, 0);
    for (int j : order) {
        int best = 0;
        for (int d = 1; d < ndev; ++d)
            if (load[d] < load[best]) best = d;
        share[best].push_back(j);
        load[best] += lens[j];
    }
    std::vector<int> rcs(ndev, DQ_OK);
    std::vector<std::string> errs(ndev);
    std::vector<ShareStats> stats(ndev);
    for (int64_t &x : t_batch_info) x = 0;
    {
        JoinAll threads;
        for (int d = 0; d < ndev; ++d) {
            threads.v.emplace_back([&, d]() {
                const int device = devs ? devs[d] : d;
                try {
                    rcs[d] = batch_on_device(device, share[d], texts, lens, sas, &errs[d], &stats[d]);
                } catch (const std::exception &e) {
                    rcs[d] = DQ_ERR_OOM;
                    errs[d] = std::string("batch: ") + e.what();
                }
            });
        }
    }
    for (const ShareStats &s : stats) {
        t_batch_info[0] += s.piped; t_batch_info[1] += s.in_us; t_batch_info[2] += s.sort_us; t_batch_info[3] += s.out_us;
        t_batch_info[4] = std::max(t_batch_info[4], s.wall_us);
        t_batch_info[5] += s.bound;
    }
    for (int d = 0; d < ndev; ++d)
        if (rcs[d] != DQ_OK) { t_err = errs[d]; return rcs[d]; }
    return DQ_OK;
    } catch (const std::bad_alloc &) {             // nothing may propagate through the C ABI
        return fail(DQ_ERR_OOM, "batch: host allocation failed");
    } catch (const std::exception &e) {            // std::system_error from std::thread, ...
        return fail(DQ_ERR_HIP, e.what());
    }
}

int32_t dq_bsdiff_search_dev_i32(const void *d_old, int64_t n, const void *d_sa, const void *d_new, int64_t m,
                                 const int64_t *d_scans, int64_t scan0, int64_t count, int64_t cap, void *d_pos,
                                 void *d_len, int32_t device, void *stream)
{
    EnvScope flags;
    return match_search_dev_i32(d_old, n, d_sa, d_new, m, d_scans, scan0, count, cap, d_pos, d_len, device, stream);
}

int32_t dq_bsdiff_search_dev_i64(const void *d_old, int64_t n, const void *d_sa, const void *d_new, int64_t m,
                                 const int64_t *d_scans, int64_t scan0, int64_t count, int64_t cap, void *d_pos,
                                 void *d_len, int32_t device, void *stream)
{
    EnvScope flags;
    return match_search_dev_i64(d_old, n, d_sa, d_new, m, d_scans, scan0, count, cap, d_pos, d_len, device, stream);
}

int32_t dq_bsdiff_search_i32(const uint8_t *old_data, int64_t n, const int32_t *sa, const uint8_t *new_data, int64_t m,
                             const int64_t *scans, int64_t scan0, int64_t count, int64_t cap, int32_t *pos, int32_t *len,
                             int32_t device)
{
    EnvScope flags;
    return match_search_host_i32(old_data, n, sa, new_data, m, scans, scan0, count, cap, pos, len, device);
}

int32_t dq_bsdiff_search_i64(const uint8_t *old_data, int64_t n, const int64_t *sa, const uint8_t *new_data, int64_t m,
                             const int64_t *scans, int64_t scan0, int64_t count, int64_t cap, int64_t *pos, int64_t *len,
                             int32_t device)
{
    EnvScope flags;
    return match_search_host_i64(old_data, n, sa, new_data, m, scans, scan0, count, cap, pos, len, device);
}

int32_t dq_bsdiff_scan_i32(const uint8_t *old_data, int64_t n, const uint8_t *new_data, int64_t m, int64_t *ctrl,
                           int64_t ctrl_cap, int64_t *nctrl, uint8_t *diff, int64_t *ndiff, uint8_t *extra, int64_t *nextra,
                           int64_t *stats, int32_t device)
{
    EnvScope flags;
    try {
        std::vector<int64_t> r_ctrl;
        std::vector<uint8_t> r_diff, r_extra;
        int64_t st[3] = {0, 0, 0};
        const int rc = bsdiff_scan_raw(old_data, n, new_data, m, device, r_ctrl, r_diff, r_extra, st);
        if (rc != DQ_OK) return rc;
        const int64_t triples = (int64_t)r_ctrl.size() / 3;
        if (triples > ctrl_cap) return fail(DQ_ERR_BAD_ARGS, "control buffer too small");
        for (int64_t i = 0; i < 3 * triples; ++i) ctrl[i] = r_ctrl[(size_t)i];
        if (!r_diff.empty()) memcpy(diff, r_diff.data(), r_diff.size());
        if (!r_extra.empty()) memcpy(extra, r_extra.data(), r_extra.size());
        *nctrl = triples; *ndiff = (int64_t)r_diff.size(); *nextra = (int64_t)r_extra.size();
        if (stats) { stats[0] = st[0]; stats[1] = st[1]; stats[2] = st[2]; }
        return DQ_OK;
    } catch (const std::bad_alloc &) {
        return fail(DQ_ERR_OOM, "bsdiff: host allocation failed");
    } catch (const std::exception &e) {            // nothing may propagate through the C ABI
        return fail(DQ_ERR_HIP, e.what());
    }
}

int32_t dq_bsdiff_create(const uint8_t *old_data, int64_t n, const uint8_t *new_data, int64_t m, uint8_t *patch,
                         int64_t cap, int64_t *patch_len, int32_t device)
{
    EnvScope flags;
    try {
        std::vector<uint8_t> v;
        const int rc = bsdiff_create_host(old_data, n, new_data, m, device, v);
        if (rc != DQ_OK) return rc;
        if (patch_len) *patch_len = (int64_t)v.size();
        if ((int64_t)v.size() > cap || !patch) return fail(DQ_ERR_BAD_ARGS, "patch buffer too small (see dq_bsdiff_patch_bound)");
        memcpy(patch, v.data(), v.size());
        return DQ_OK;
    } catch (const std::bad_alloc &) {
        return fail(DQ_ERR_OOM, "bsdiff: host allocation failed");
    } catch (const std::exception &e) {            // nothing may propagate through the C ABI
        return fail(DQ_ERR_HIP, e.what());
    }
}

int32_t dq_bsdiff_index_create(const uint8_t *old_data, int64_t n, const void *d_old, const void *d_sa, int32_t device,
                               void **index_out)
{
    EnvScope flags;
    if (!index_out) return fail(DQ_ERR_BAD_ARGS, "null index pointer");
    *index_out = nullptr;
    try {
        const int rc = diff_index_new(old_data, n, device, d_old, d_sa, index_out);
        if (rc != DQ_OK) return rc;
        return DQ_OK;
    } catch (const std::bad_alloc &) {
        return fail(DQ_ERR_OOM, "bsdiff: host allocation failed");
    } catch (const std::exception &e) {
        return fail(DQ_ERR_HIP, e.what());
    }
}

int32_t dq_bsdiff_index_clone(const void *index, int32_t device, void **index_out)
{
    EnvScope flags;
    if (!index || !index_out) return fail(DQ_ERR_BAD_ARGS, "null index");
    *index_out = nullptr;
    try {
        return diff_index_clone(index, device, index_out);
    } catch (const std::bad_alloc &) {
        return fail(DQ_ERR_OOM, "bsdiff: host allocation failed");
    } catch (const std::exception &e) {
        return fail(DQ_ERR_HIP, e.what());
    }
}

int32_t dq_bsdiff_index_buffers(const void *index, const void **d_old, const void **d_sa, int64_t *n)
{
    if (!index) return fail(DQ_ERR_BAD_ARGS, "null index");
    return diff_index_buffers(index, d_old, d_sa, n);
}

int32_t dq_bsdiff_index_diff(const void *index, const uint8_t *new_data, int64_t m, uint8_t *patch, int64_t cap,
                             int64_t *patch_len)
{
    EnvScope flags;
    if (!index) return fail(DQ_ERR_BAD_ARGS, "null index");
    try {
        std::vector<uint8_t> v;
        const int rc = diff_index_diff(index, new_data, m, v);
        if (rc != DQ_OK) return rc;
        if (patch_len) *patch_len = (int64_t)v.size();
        if ((int64_t)v.size() > cap || !patch) return fail(DQ_ERR_BAD_ARGS, "patch buffer too small (see dq_bsdiff_patch_bound)");
        memcpy(patch, v.data(), v.size());
        return DQ_OK;
    } catch (const std::bad_alloc &) {
        return fail(DQ_ERR_OOM, "bsdiff: host allocation failed");
    } catch (const std::exception &e) {
        return fail(DQ_ERR_HIP, e.what());
    }
}

void dq_bsdiff_index_free(void *index)
{
    if (!index) return;
    diff_index_delete(index);
}

int64_t dq_bsdiff_patch_bound(int64_t n, int64_t m)
{
    if (n < 0 || m < 0) return -1;
    // three bzip2 streams: 24 bytes of control per triple (at most m + 1 triples), m diff + extra bytes in total;
    // bzip2 never grows its input by more than 1 % + 600 bytes per stream
    const int64_t raw = 24 * (m + 1) + m;
    return 32 /* BSDIFF40 header, Diff.cs:54-70 */ + raw + raw / 100 + 3 * 600 + 64;
}

int32_t dq_bspatch_apply(const uint8_t *old_data, int64_t n, const uint8_t *patch, int64_t patch_len, uint8_t *out,
                         int64_t cap, int64_t *out_len)
{
    try {
        return bspatch_apply_host(old_data, n, patch, patch_len, out, cap, out_len);
    } catch (const std::bad_alloc &) {
        return fail(DQ_ERR_OOM, "bspatch: host allocation failed");
    } catch (const std::exception &e) {
        return fail(DQ_ERR_HIP, e.what());
    }
}

int64_t dq_sufsort_hip_workspace_bytes(int64_t n, int32_t index_bytes)
{
    if (n < 0) return -1;
    if (index_bytes == 4) return sufsort_workspace_bytes<int32_t>(n);
    if (index_bytes == 8) return sufsort_workspace_bytes<int64_t>(n);
    return -1;
}

void dq_sufsort_hip_release(void)
{
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) count = 0;
    for (int d = 0; d < kMaxDevices && d < count; ++d) {
        DeviceCtx &c0 = ctx0(d);
        // lock order everywhere: batch_mu, then diff_mu, then a slot's mu
        std::lock_guard<std::mutex> bl(c0.batch_mu);
        std::lock_guard<std::mutex> dl(c0.diff_mu);
        bool any = c0.diff_dev || c0.diff_idx || c0.diff_pinned || c0.bslot_cap;
        for (int k = 0; k < kCtxSlots; ++k) {
            std::lock_guard<std::mutex> lk(g_dev[d].slot[k].mu);       // (a first use of the slot may be publishing dev right now)
            any = any || g_dev[d].slot[k].dev >= 0;
        }
        if (!any || hipSetDevice(d) != hipSuccess) continue;
        for (int k = 0; k < 3; ++k) {
            if (c0.bslot_text[k]) (void)hipFree(c0.bslot_text[k]);
            if (c0.bslot_sa[k]) (void)hipFree(c0.bslot_sa[k]);
            c0.bslot_text[k] = nullptr; c0.bslot_sa[k] = nullptr;
        }
        c0.bslot_cap = 0;
        for (hipStream_t *st : {&c0.b_in, &c0.b_sort, &c0.b_out}) { if (*st) (void)hipStreamDestroy(*st); *st = nullptr; }
        for (int64_t &d : c0.scan_dirty) d = 1 << 16;
        c0.scan_pool.reset();
        if (c0.diff_dev) (void)hipFree(c0.diff_dev);
        if (c0.diff_idx) (void)hipFree(c0.diff_idx);
        if (c0.diff_pinned) (void)hipHostFree(c0.diff_pinned);
        c0.diff_dev = nullptr; c0.diff_idx = nullptr; c0.diff_pinned = nullptr;
        c0.diff_dev_bytes = 0; c0.diff_idx_bytes = 0;
        for (int k = 0; k < kCtxSlots; ++k) {
            DeviceCtx &c = g_dev[d].slot[k];
            std::lock_guard<std::mutex> lk(c.mu);
            if (c.ws) (void)hipFree(c.ws);
            c.ws = nullptr; c.ws_bytes = 0;
            for (hipEvent_t e : c.pool) (void)hipEventDestroy(e);
            c.pool.clear();
            if (c.pinned) (void)hipHostFree(c.pinned);
            c.pinned = nullptr;
            if (c.pinned_io) (void)hipHostFree(c.pinned_io);
            c.pinned_io = nullptr;
            if (c.readback) (void)hipEventDestroy(c.readback);
            c.readback = nullptr;
            if (c.stream) (void)hipStreamDestroy(c.stream);
            c.stream = nullptr;
            c.dev = -1;
        }
    }
}

int32_t dq_profile_enable(int32_t on)
{
    g_prof_on.store((on == 2 || (on >= 100 && on < 100 + DQ_K_COUNT)) ? on : (on ? 1 : 0));
    return DQ_OK;
}

void dq_profile_reset(void)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto &s : g_prof) s = KernelStat{};
}

int32_t dq_profile_get(int32_t category, int64_t *launches, double *total_ms, int64_t *elements,
                       int64_t *alg_bytes)
{
    if (category < 0 || category >= DQ_K_COUNT) return DQ_ERR_BAD_ARGS;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    const KernelStat &s = g_prof[category];
    if (launches) *launches = s.launches;
    if (total_ms) *total_ms = s.ms;
    if (elements) *elements = s.elems;
    if (alg_bytes) *alg_bytes = s.bytes;
    return DQ_OK;
}

int32_t dq_profile_category_count(void) { return DQ_K_COUNT; }

const char *dq_profile_kernel_name(int32_t category)
{
    return (category >= 0 && category < DQ_K_COUNT) ? kKernelNames[category] : "";
}

int32_t dq_last_sort_info(int64_t *rounds, int64_t *initial_active, int64_t *sum_active)
{
    if (rounds) *rounds = t_info[0];
    if (initial_active) *initial_active = t_info[1];
    if (sum_active) *sum_active = t_info[2];
    return DQ_OK;
}

int32_t dq_last_batch_info(int64_t *info, int32_t count)
{
    if (!info || count < 0) return fail(DQ_ERR_BAD_ARGS, "bad arguments");
    for (int32_t k = 0; k < count; ++k) info[k] = k < 6 ? t_batch_info[k] : 0;
    return DQ_OK;
}

int32_t dq_device_numa_node(int32_t device)
{
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return -1;
    return device_numa_node(device);
}

int32_t dq_last_diff_info(int64_t *info, int32_t count)
{
    if (!info || count < 0) return fail(DQ_ERR_BAD_ARGS, "bad arguments");
    for (int32_t k = 0; k < count; ++k) info[k] = k < 9 ? t_diff_info[k] : 0;
    return DQ_OK;
}

}  // extern "C"