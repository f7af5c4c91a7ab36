// dq_diff.hip -- the consumers of the suffix array: match search on the device-resident SA (Diff.cs:267-298), Diff.Create
// (Diff.cs:27-253: scan loop over windows of device answers, BSDIFF40 framing with the own bzip2 codec), Patch.Apply
// (Patch.cs:52-168), and the one-old-file-many-new-files index.  The sorter is called through dq_runtime.h.
#include <sys/mman.h>

#include "dq_runtime.h"
#include "dq_match_search.h"
#include "dq_anchor_scan.h"
#include "dq_bz2.h"
#include "dq_bsdiff.h"
#include "dq_bspatch.h"

namespace dq {
namespace {

// ------------------------------------------------------------------ match search (Diff.cs:267-298) on the device
template <typename IdxT>
int match_search_dev(const void *d_old, int64_t n, const void *d_sa, const void *d_new, int64_t m,
                     const int64_t *d_scans, int64_t scan0, int64_t count, int64_t cap, void *d_pos, void *d_len,
                     int32_t device, void *stream, const void *d_ptab = nullptr, int pk = 0, int exact_first = 0)
{
    if (n < 0 || m < 0 || count < 0 || cap < 0) return fail(DQ_ERR_BAD_ARGS, "negative length");
    if ((n > 0 && (!d_old || !d_sa)) || (m > 0 && !d_new) || (count > 0 && (!d_pos || !d_len)))
        return fail(DQ_ERR_BAD_ARGS, "null buffer");
    if (!d_scans && (scan0 < 0 || scan0 + count > m + 1)) return fail(DQ_ERR_BAD_ARGS, "scan range outside the new data");
    if (sizeof(IdxT) == 4 && (n > 0x7fffffffLL || m > 0x7fffffffLL))
        return fail(DQ_ERR_TOO_LARGE, "n or m exceeds 2^31-1; use the i64 entry point");
    int dev = 0;
    int rc = resolve_device(device, &dev);
    if (rc != DQ_OK) return rc;
    if (count == 0) return DQ_OK;
    DeviceCtx &c = ctx0(dev);
    std::lock_guard<std::mutex> lk(c.mu);
    rc = init_ctx(c, dev);
    if (rc != DQ_OK) return rc;
    hipStream_t st = stream ? (hipStream_t)stream : c.stream;
    Launcher L{c, st, g_prof_on.load()};
    // per query: ~log2(n) probes of one SA entry and one 64-byte sector of old, + the match itself
    const int64_t probes = bit_length((uint64_t)std::max<int64_t>(n, 1));
    // (DQ_SEARCH_WAVE=1: consecutive positions through the one-wave-per-position kernel of the scan-loop driver, so
    // that the tests can compare its answers one by one; position 0 is answered exactly whatever the cap)
    const bool wave = env("DQ_SEARCH_WAVE") && !d_scans && count <= 4096;
    // (DQ_SEARCH_PTAB = 2 | 3: the search starts from a prefix table of that many bytes, as the scan-loop driver's
    // windows do -- built here for the call, so that the tests can compare the answers of both kernels with it)
    struct TmpTab { void *p = nullptr; ~TmpTab() { if (p) (void)hipFree(p); } } tmp_tab;
    if (!d_ptab && env("DQ_SEARCH_PTAB") && n > 0) {
        pk = atoi(env("DQ_SEARCH_PTAB")) >= 3 ? 3 : 2;
        const int64_t total = (1ll << (8 * pk)) + 1;
        HIP_TRY(dq_malloc(&tmp_tab.p, (size_t)total * sizeof(IdxT)));
        hipLaunchKernelGGL(prefix_bounds_kernel<IdxT>, dim3((unsigned)((total + kBlock - 1) / kBlock)), dim3(kBlock), 0, st,
                           (const uint8_t *)d_old, n, (const IdxT *)d_sa, pk, (IdxT *)tmp_tab.p, (const IdxT *)nullptr);
        HIP_TRY(hipGetLastError());
        d_ptab = tmp_tab.p;
    }
    auto launch = [&]() -> int {
        if (wave) {
            constexpr int kPer = kMsThreads / kWave;
            LAUNCH(L, DQ_K_MATCH_SEARCH, count, count * 4 * ((int64_t)sizeof(IdxT) + 64) * 64,
                   hipLaunchKernelGGL(match_search_wave_kernel<IdxT>, dim3((unsigned)((count + kPer - 1) / kPer)),
                                      dim3(kMsThreads), 0, st, (const uint8_t *)d_old, n, (const IdxT *)d_sa,
                                      (const uint8_t *)d_new, m, scan0, count, cap, (IdxT *)d_pos, (IdxT *)d_len,
                                      (const IdxT *)d_ptab, pk, 0));
            return DQ_OK;
        }
        LAUNCH(L, DQ_K_MATCH_SEARCH, count, count * probes * ((int64_t)sizeof(IdxT) + 64),
               hipLaunchKernelGGL(match_search_kernel<IdxT>, dim3((unsigned)((count + kMsThreads - 1) / kMsThreads)),
                                  dim3(kMsThreads), 0, st, (const uint8_t *)d_old, n, (const IdxT *)d_sa,
                                  (const uint8_t *)d_new, m, d_scans, scan0, count, cap, (IdxT *)d_pos, (IdxT *)d_len,
                                  (const IdxT *)d_ptab, pk, exact_first));
        return DQ_OK;
    };
    rc = launch();
    if (rc != DQ_OK) { drop_pending(c, st); return rc; }
    HIP_TRY(hipStreamSynchronize(st));
    return flush_profile(c);
}

// host buffers in / out: what a P/Invoke caller without device memory of its own uses (and the tests)
template <typename IdxT>
int match_search_host(const uint8_t *old, int64_t n, const IdxT *sa, const uint8_t *nw, int64_t m, const int64_t *scans,
                      int64_t scan0, int64_t count, int64_t cap, IdxT *pos, IdxT *len, int32_t device)
{
    if (n < 0 || m < 0 || count < 0) return fail(DQ_ERR_BAD_ARGS, "negative length");
    if ((n > 0 && (!old || !sa)) || (m > 0 && !nw) || (count > 0 && (!pos || !len))) return fail(DQ_ERR_BAD_ARGS, "null buffer");
    // host-resident scan positions are checked here (a position outside [0, m] would be a device read out of bounds);
    // the device forms take them as they are (include/dq_sufsort.h says so)
    if (scans)
        for (int64_t q = 0; q < count; ++q)
            if (scans[q] < 0 || scans[q] > m) return fail(DQ_ERR_BAD_ARGS, "scan position outside the new data");
    int dev = 0;
    int rc = resolve_device(device, &dev);
    if (rc != DQ_OK) return rc;
    if (count == 0) return DQ_OK;
    HIP_TRY(hipSetDevice(dev));
    char *base = nullptr;
    const size_t b_old = align_up((size_t)n + 16), b_sa = align_up((size_t)n * sizeof(IdxT) + 16), b_new = align_up((size_t)m + 16);
    const size_t b_sc = scans ? align_up((size_t)count * 8) : 0, b_out = align_up((size_t)count * sizeof(IdxT));
    hipError_t e = dq_malloc((void **)&base, b_old + b_sa + b_new + b_sc + 2 * b_out);
    if (e != hipSuccess) return fail(DQ_ERR_OOM, "hipMalloc(match search buffers)", e);
    char *d_old = base, *d_sa = d_old + b_old, *d_new = d_sa + b_sa, *d_sc = d_new + b_new, *d_pos = d_sc + b_sc,
         *d_len = d_pos + b_out;
    auto done = [&](int code) { (void)hipFree(base); return code; };
    if (n > 0) {
        if (hipMemcpy(d_old, old, (size_t)n, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(d_sa, sa, (size_t)n * sizeof(IdxT), hipMemcpyHostToDevice) != hipSuccess)
            return done(fail(DQ_ERR_HIP, "match search: copy-in failed"));
    }
    if (m > 0 && hipMemcpy(d_new, nw, (size_t)m, hipMemcpyHostToDevice) != hipSuccess)
        return done(fail(DQ_ERR_HIP, "match search: copy-in failed"));
    if (scans && hipMemcpy(d_sc, scans, (size_t)count * 8, hipMemcpyHostToDevice) != hipSuccess)
        return done(fail(DQ_ERR_HIP, "match search: copy-in failed"));
    rc = match_search_dev<IdxT>(d_old, n, d_sa, d_new, m, scans ? (const int64_t *)d_sc : nullptr, scan0, count, cap, d_pos,
                                d_len, dev, nullptr);
    if (rc != DQ_OK) return done(rc);
    if (hipMemcpy(pos, d_pos, (size_t)count * sizeof(IdxT), hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(len, d_len, (size_t)count * sizeof(IdxT), hipMemcpyDeviceToHost) != hipSuccess)
        return done(fail(DQ_ERR_HIP, "match search: copy-out failed"));
    return done(DQ_OK);
}

// ------------------------------------------------------------------ BSDIFF40: Diff.Create / Patch.Apply (dq_bsdiff.h)
// Answers of the match search for a window of scan positions ahead of the scan loop.  Windows start small after a
// jump and double while the loop consumes them to the end (a region where old and new differ: one Search per
// byte, the regime the device is for: 4.2 M searches in 31 ms against 5.4 s on one host core; the one-query-per-lane
// kernel).  The cap is low: the positions of a window that lie inside the next long match would each cost `cap` byte
// comparisons for nothing (the loop leaves the window with its next jump).  Between nearly identical files the
// loop hops from match to match and every launch is a dependent round trip (~1 per edit): windows of up to 2048
// positions go to the one-wave-per-position kernel (65-ary search; the position the loop stands on and the probable
// start of the next long match answered exactly), whose answers are polled in pinned memory, whose second stage
// answers the window behind the predicted jump, and which answers exactly throughout while positions keep coming
// back capped (dq_match_search.h; DESIGN.md section 2c has the measurements).
struct SearchWindows {
    const void *d_old, *d_sa, *d_new;
    int64_t n, m;
    int device;
    // kMaxWindow + 2 entries each in PINNED HOST memory that the kernel writes directly (no copy back: between
    // similar files the loop is a chain of dependent round trips, and two small hipMemcpy cost more than the kernel)
    int32_t *h_pos = nullptr, *h_len = nullptr;
    uint64_t *h_packed = nullptr;                        // pinned: (len << 32 | pos) of the wave windows, polled by the loop
    void *d_mail = nullptr;                              // device: mailbox of the window kernel's second stage
    static constexpr int64_t kSecond = 1024;             // slots of the predicted next window (second <= kSecond are used)
    int64_t second = 128;                                // positions of the predicted next window
    int64_t min_window = 128;                            // first window after a jump
    bool walk_on = true;                                 // second stage without a winner: the positions behind the window
    bool no_resume = false;                              // DQ_NO_RESUME: capped first positions searched again from the top
    static constexpr int64_t kSecondMaxFirst = 1024;     // ... behind first stages of up to this many positions
    int64_t sec_region = 0, predicted = 0;               // slot region (offset into h_packed) of the pending second stage
    bool sec_pending = false, no_second = false;
    int64_t last_capped = -2, capped_streak = 0;         // consecutive positions that came back capped
    bool from_capped = false;
    unsigned long long ticket = 0, done_total = 0;       // of the launches with a second stage (the mailbox is never reset)
    const void *d_ptab = nullptr;                        // prefix table (prefix_bounds_kernel), or none
    int pk = 0;
    int64_t w0 = -1, wc = 0, next_size = 128;
    int64_t windows = 0, exact = 0;
    static constexpr int64_t kMinWindow = 128, kMaxWindow = 65536, kCap = 64, kWaveWindow = 2048;
    static constexpr uint64_t kPending = 0x8000000080000000ull;   // (no answer looks like this: len >= -1)

    // wait for one pinned slot to leave the "pending" state (bounded polling, then the ordinary stream wait)
    int await_slot(const uint64_t *slot, hipStream_t st, uint64_t *value)
    {
        const auto t0 = std::chrono::steady_clock::now();
        for (uint32_t spins = 0;; ++spins) {
            const uint64_t v = __atomic_load_n(slot, __ATOMIC_ACQUIRE);
            if (v != kPending) { *value = v; return DQ_OK; }
            if ((spins & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) break;
        }
        HIP_TRY(hipStreamSynchronize(st));               // (a slow window -- megabytes of equal text)
        *value = __atomic_load_n(slot, __ATOMIC_ACQUIRE);
        if (*value == kPending) return fail(DQ_ERR_HIP, "match search: a window position was left unanswered");
        return DQ_OK;
    }

    int refill(int64_t scan)
    {
        int dev = 0;
        int rc = resolve_device(device, &dev);
        if (rc != DQ_OK) return rc;
        DeviceCtx &c = ctx0(dev);
        // the window the device was asked to answer ahead (second stage of the previous launch): is it this one?
        if (sec_pending) {
            sec_pending = false;
            const uint64_t *reg = h_packed + sec_region;
            uint64_t hdr = 0;
            rc = await_slot(&reg[0], c.stream, &hdr);
            if (rc != DQ_OK) return rc;
            if (hdr != kMsSkipped && (int64_t)hdr == scan) {
                int64_t got = 0;
                for (; got < second; ++got) {
                    uint64_t v = 0;
                    rc = await_slot(&reg[1 + got], c.stream, &v);
                    if (rc != DQ_OK) return rc;
                    if (v == kMsSkipped) break;
                    h_pos[got] = (int32_t)(uint32_t)v;
                    h_len[got] = (int32_t)(uint32_t)(v >> 32);
                }
                if (got > 0) {
                    w0 = scan;
                    wc = got;
                    next_size = min_window;
                    ++windows;
                    ++predicted;
                    return DQ_OK;
                }
            }
        }
        // (the loop jumped: whatever made positions come back capped in a row is behind it)
        if (!from_capped && !(w0 >= 0 && scan == w0 + wc)) capped_streak = 0;
        from_capped = false;
        // the previous window was used up to its end: the loop is walking byte by byte -> a larger one
        next_size = (w0 >= 0 && scan == w0 + wc) ? std::min(next_size * 2, kMaxWindow) : min_window;
        const int64_t count = std::min(next_size, m - scan);
        if (count <= kWaveWindow && !env("DQ_NO_WAVE_WINDOWS")) {
            // short windows (the loop is hopping from match to match: every launch is a dependent round trip): one WAVE
            // per position, 65-ary search; the position the loop stands on exactly, the ones behind it with the cap
            std::lock_guard<std::mutex> lk(c.mu);
            rc = init_ctx(c, dev);
            if (rc != DQ_OK) return rc;
            Launcher L{c, c.stream, g_prof_on.load()};
            constexpr int kPer = kMsThreads / kWave;
            const bool poll_now = h_packed != nullptr && !L.prof;
            // second stage: the window the loop will want after its next jump (dq_match_search.h), windows of up to 1024 positions.
            // Its answers are looked at when the loop gets there, not now; two slot regions take turns, so that a
            // region is written by one launch at a time (the launch in between has answered: the older one is over).
            const int64_t count2 = (poll_now && d_mail && count <= kSecondMaxFirst && !no_second && ticket < (1ull << 20) - 2) ? second : 0;
            uint64_t *reg2 = nullptr;
            if (count2) {
                ++ticket;
                done_total += (unsigned long long)count;
                sec_region = kWaveWindow + (int64_t)(ticket & 1) * (kSecond + 1);
                reg2 = h_packed + sec_region;
                for (int64_t i = 0; i < second + 1; ++i) reg2[i] = kPending;
            }
            auto launch = [&]() -> int {
                LAUNCH(L, DQ_K_MATCH_SEARCH, count, count * 4 * (4 + 64) * 64,
                       hipLaunchKernelGGL(match_search_wave_kernel<int32_t>, dim3((unsigned)((count + count2 + kPer - 1) / kPer)),
                                          dim3(kMsThreads), 0, c.stream, (const uint8_t *)d_old, n, (const int32_t *)d_sa,
                                          (const uint8_t *)d_new, m, scan, count, capped_streak >= 2 ? (int64_t)0 : kCap, h_pos, h_len,
                                          (const int32_t *)d_ptab, pk,
                                          poll_now ? h_packed : (uint64_t *)nullptr, count2, reg2,
                                          count2 ? reinterpret_cast<unsigned long long *>(d_mail) : (unsigned long long *)nullptr,
                                          ticket, done_total, walk_on ? 1 : 0, no_resume ? 1 : 0));
                return DQ_OK;
            };
            if (poll_now) for (int64_t i = 0; i < count; ++i) h_packed[i] = kPending;
            rc = launch();
            if (rc != DQ_OK) { drop_pending(c, c.stream); return rc; }
            if (poll_now) {
                for (int64_t i = 0; i < count; ++i) {
                    uint64_t v = 0;
                    rc = await_slot(&h_packed[i], c.stream, &v);
                    if (rc != DQ_OK) return rc;
                    h_pos[i] = (int32_t)(uint32_t)v;
                    h_len[i] = (int32_t)(uint32_t)(v >> 32);
                }
                sec_pending = count2 > 0;
            } else {
                HIP_TRY(hipStreamSynchronize(c.stream));
            }
            rc = flush_profile(c);
        } else {
            rc = match_search_dev<int32_t>(d_old, n, d_sa, d_new, m, nullptr, scan, count, kCap, h_pos, h_len, device, nullptr,
                                           d_ptab, pk, /*exact_first=*/1);
        }
        if (rc != DQ_OK) return rc;                      // (the answers are there)
        w0 = scan;
        wc = count;
        ++windows;
        return DQ_OK;
    }
    int operator()(int64_t scan, int64_t *pos, int64_t *len)
    {
        if (scan < w0 || scan >= w0 + wc) {
            const int rc = refill(scan);
            if (rc != DQ_OK) return rc;
        }
        int64_t p = h_pos[(size_t)(scan - w0)], l = h_len[(size_t)(scan - w0)];
        if (l < 0) {
            // undecided within the cap (the loop has reached the next long match): a new window from here, whose first
            // position is answered exactly -- and whose other positions are there if the match turns out not to be taken.
            // When that happens at one position after the other (the loop is walking through text that matches far
            // everywhere -- periodic data, runs -- without jumping), the windows are answered exactly throughout:
            // one launch per 128 positions instead of one per position.
            capped_streak = (scan == last_capped + 1) ? capped_streak + 1 : 1;
            last_capped = scan;
            w0 = -1;
            from_capped = true;
            int rc = refill(scan);
            if (rc != DQ_OK) return rc;
            p = h_pos[0];
            l = h_len[0];
            ++exact;
            if (l < 0) {                                 // (a long window: its exact position may not be this one)
                rc = match_search_dev<int32_t>(d_old, n, d_sa, d_new, m, nullptr, scan, 1, 0, h_pos + kMaxWindow,
                                               h_len + kMaxWindow, device, nullptr, d_ptab, pk);
                if (rc != DQ_OK) return rc;
                p = h_pos[kMaxWindow];
                l = h_len[kMaxWindow];
                h_pos[0] = (int32_t)p;
                h_len[0] = (int32_t)l;
            }
        }
        *pos = p;
        *len = l;
        return DQ_OK;
    }
};


// ---- "one old file, many new files": the suffix array of old (Diff.cs:89-90) is what a diff costs before its scan loop,
// and it depends on old alone.  A DiffIndex holds (old, suffix array, prefix table of the match search) on the
// device; any number of new files are diffed against it (dq_bsdiff_index_*; the reference pays the sort once per
// Diff.Create call).  The buffers are either the index's own (built here) or the caller's (a rank that received
// text + suffix array by RCCL broadcast, deltaq_amd/batch.py: diff_many_distributed).
struct DiffIndex {
    int dev = 0;
    int64_t n = 0;
    const uint8_t *old = nullptr;       // host copy the scan loop walks: the caller's, valid while the index lives
    char *own = nullptr;                // device allocation of this index (old + SA if built here, prefix table)
    bool own_cached = false;            // ... which is the device context's cached one-shot buffer (not freed)
    const char *d_old = nullptr, *d_sa = nullptr;
    const char *d_tab = nullptr;
    int pk = 0;
};

constexpr size_t kDiffWindowBytes = 1 << 20;               // SearchWindows' part of the pinned area (asserted where it is laid out)
constexpr size_t kDiffPinnedBytes = kDiffWindowBytes + kAnchorPinned;      // + lists, counts and control blocks of the device scan's chains

size_t diff_tab_bytes(int64_t n, int *pk_out)
{
    // prefix table of the match search: 3 bytes (64 MiB of entries) for old files from 4 MiB, 2 bytes from 64 KiB
    const int pk = n >= (4 << 20) ? 3 : n >= (1 << 16) ? 2 : 0;
    *pk_out = pk;
    // (+ the 2-byte table the 3-byte one is built from, behind it)
    return pk ? align_up(((size_t)1 << (8 * pk)) * 4 + 16) + (pk == 3 ? align_up(((size_t)1 << 16) * 4 + 16) : 0) : 0;
}

int grow_cached(char **buf, size_t *have, size_t want, const char *what)
{
    if (*have >= want) return DQ_OK;
    if (*buf) { (void)hipFree(*buf); *buf = nullptr; *have = 0; }
    hipError_t e = dq_malloc((void **)buf, want);
    if (e != hipSuccess) return fail(DQ_ERR_OOM, what, e);
    *have = want;
    return DQ_OK;
}

// d_old_in / d_sa_in: device-resident text and suffix array of the caller (both or neither).  cached: build into the
// device context's reusable buffer (the one-shot dq_bsdiff_create; the caller holds diff_mu).
int diff_index_build(const uint8_t *old, int64_t n, int32_t device, const void *d_old_in, const void *d_sa_in, bool cached,
                     DiffIndex *ix)
{
    if (n < 0 || (n > 0 && !old)) return fail(DQ_ERR_BAD_ARGS, "bad arguments");
    if ((d_old_in == nullptr) != (d_sa_in == nullptr)) return fail(DQ_ERR_BAD_ARGS, "device text and suffix array go together");
    if (n > 0x7fffffffLL) return fail(DQ_ERR_TOO_LARGE, "the BSDIFF40 path takes files below 2 GiB (int indices, as the reference)");
    int dev = 0;
    int rc = resolve_device(device, &dev);
    if (rc != DQ_OK) return rc;
    HIP_TRY(hipSetDevice(dev));
    ix->dev = dev; ix->n = n; ix->old = old;
    int pk = 0;
    const size_t b_tab = diff_tab_bytes(n, &pk);
    const size_t b_old = d_old_in ? 0 : align_up((size_t)n + 16), b_sa = d_old_in ? 0 : align_up((size_t)n * 4 + 16);
    const size_t total = b_old + b_sa + b_tab;
    if (total > 0) {
        if (cached) {
            DeviceCtx &c = ctx0(dev);
            rc = grow_cached(&c.diff_idx, &c.diff_idx_bytes, total, "hipMalloc(bsdiff index)");
            if (rc != DQ_OK) return rc;
            ix->own = c.diff_idx;
            ix->own_cached = true;
        } else {
            hipError_t e = dq_malloc((void **)&ix->own, total);
            if (e != hipSuccess) return fail(DQ_ERR_OOM, "hipMalloc(bsdiff index)", e);
        }
    }
    if (d_old_in) {
        ix->d_old = (const char *)d_old_in;
        ix->d_sa = (const char *)d_sa_in;
    } else {
        ix->d_old = ix->own;
        ix->d_sa = ix->own + b_old;
        if (n > 0) HIP_TRY(hipMemcpy(ix->own, old, (size_t)n, hipMemcpyHostToDevice));
        rc = sufsort_dev<int32_t>(ix->d_old, n, const_cast<char *>(ix->d_sa), dev, nullptr);     // Diff.cs:90; the SA never leaves the device
        if (rc != DQ_OK) return rc;
    }
    ix->pk = pk;
    if (pk) {
        char *tab = ix->own + b_old + b_sa;
        const int64_t total_e = (1ll << (8 * pk)) + 1;
        const int32_t *coarse = nullptr;
        if (pk == 3) {                                       // the 2-byte table first: it bounds every search of the 3-byte one
            int32_t *two = reinterpret_cast<int32_t *>(tab + align_up(((size_t)1 << 24) * 4 + 16));
            hipLaunchKernelGGL(prefix_bounds_kernel<int32_t>, dim3((unsigned)(((1 << 16) + 1 + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                               nullptr, (const uint8_t *)ix->d_old, n, (const int32_t *)ix->d_sa, 2, two, (const int32_t *)nullptr);
            HIP_TRY(hipGetLastError());
            coarse = two;
        }
        hipLaunchKernelGGL(prefix_bounds_kernel<int32_t>, dim3((unsigned)((total_e + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                           nullptr, (const uint8_t *)ix->d_old, n, (const int32_t *)ix->d_sa, pk, (int32_t *)tab, coarse);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipDeviceSynchronize());
        ix->d_tab = tab;
    }
    return DQ_OK;
}

// One more copy of an index on `device` (the same device or another one of the node): text, suffix array and prefix
// table are copied device to device -- over xGMI when the devices differ -- instead of being computed again.  This is
// the exchange step of the many-files path for a host that has no collective library in its process (the C# shim):
// xGMI is point to point, so the copies to 7 other devices of a node, issued from 7 threads, travel over 7 different
// links at once -- what a broadcast over those links does, without a communicator.
int diff_index_copy(const DiffIndex &src, int32_t device, DiffIndex *ix)
{
    int dev = 0;
    int rc = resolve_device(device, &dev);
    if (rc != DQ_OK) return rc;
    HIP_TRY(hipSetDevice(dev));
    ix->dev = dev; ix->n = src.n; ix->old = src.old; ix->pk = src.pk;
    int pk = 0;
    const size_t b_tab = diff_tab_bytes(src.n, &pk);
    const size_t b_old = align_up((size_t)src.n + 16), b_sa = align_up((size_t)src.n * 4 + 16);
    if (pk != src.pk) return fail(DQ_ERR_BAD_ARGS, "index to copy is inconsistent");
    const hipError_t e = dq_malloc((void **)&ix->own, b_old + b_sa + b_tab);
    if (e != hipSuccess) return fail(DQ_ERR_OOM, "hipMalloc(bsdiff index copy)", e);
    ix->d_old = ix->own;
    ix->d_sa = ix->own + b_old;
    ix->d_tab = pk ? ix->own + b_old + b_sa : nullptr;
    if (dev != src.dev) {
        // direct peer access where the platform has it (xGMI inside a node); hipMemcpyPeer stages through the host without
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, dev, src.dev) == hipSuccess && can) {
            const hipError_t pe = hipDeviceEnablePeerAccess(src.dev, 0);
            if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
        }
    }
    auto copy = [&](const char *to, const char *from, size_t bytes) -> hipError_t {
        if (bytes == 0) return hipSuccess;
        return dev == src.dev ? hipMemcpy(const_cast<char *>(to), from, bytes, hipMemcpyDeviceToDevice)
                              : hipMemcpyPeer(const_cast<char *>(to), dev, from, src.dev, bytes);
    };
    HIP_TRY(copy(ix->d_old, src.d_old, (size_t)src.n));
    HIP_TRY(copy(ix->d_sa, src.d_sa, (size_t)src.n * 4));
    if (pk) HIP_TRY(copy(ix->d_tab, src.d_tab, (((size_t)1 << (8 * pk)) + 1) * 4));
    HIP_TRY(hipDeviceSynchronize());
    return DQ_OK;
}

void diff_index_drop(DiffIndex *ix)
{
    if (ix->own && !ix->own_cached) { (void)hipSetDevice(ix->dev); (void)hipFree(ix->own); }
    ix->own = nullptr;
}

// The suffix sorter as bzip2's block transform calls it (several encoder threads at once, each call leasing its own
// device context); the first error is kept for the thread that collects the stream.
struct BlockSorter {
    int dev = 0;
    std::atomic<int> rc{DQ_OK};
    std::mutex mu;
    std::string err;
    int sort(const uint8_t *t, int64_t n2, int32_t *sa)
    {
        const auto t0 = std::chrono::steady_clock::now();
        SortHints hints;
        hints.doubled = true;
        // bzip2's run-length pre-pass turns a long run into a stretch of period 5 (four bytes + a count of 251): the diff
        // stream of two similar files is little else.  One look at the block: where 1/8 of it lies in such stretches, the
        // sorter is told (they tie thousands of suffixes per phase until the doubling has walked through them).
        {
            const int64_t nb = n2 / 2;
            int64_t words = 0;
            for (int64_t i = 0; i + 13 <= nb; i += 8) {
                uint64_t a, b;
                memcpy(&a, t + i, 8);
                memcpy(&b, t + i + 5, 8);
                words += a == b;
            }
            if (words * 64 >= nb && nb >= (1 << 15)) hints.run_period = 5;
        }
        const int r = sufsort_host<int32_t>(t, n2, sa, dev, hints);
        if (env("DQ_TRACE"))
            fprintf(stderr, "[dq] bzip2 block transform: suffix array of %lld bytes in %.3f ms\n", (long long)n2,
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        if (r == DQ_OK) return 0;
        int expect = DQ_OK;
        if (rc.compare_exchange_strong(expect, r)) {
            std::lock_guard<std::mutex> lk(mu);
            err = t_err;                                  // (thread-local on the worker: carried over)
        }
        return -2;
    }
    bz2::DoubledSorter fn() { return [this](const uint8_t *t, int64_t n2, int32_t *sa) { return sort(t, n2, sa); }; }
};

// Large host buffers that are written once from front to back (the diff / extra streams of a large pair, the chain
// emitters' own streams): transparent huge pages where the host hands them out on request ("madvise" mode: 4 KiB pages
// otherwise, and a 128 MiB diff stream is 32 768 page faults on the thread that strings the chains' streams together).
inline void advise_huge(const void *p, size_t bytes)
{
#if defined(MADV_HUGEPAGE)
    if (bytes < ((size_t)4 << 20)) return;
    const uintptr_t a = ((uintptr_t)p + 4095) & ~(uintptr_t)4095, e = ((uintptr_t)p + bytes) & ~(uintptr_t)4095;
    if (e > a) (void)madvise(reinterpret_cast<void *>(a), e - a, MADV_HUGEPAGE);
#else
    (void)p; (void)bytes;
#endif
}

// Framing that follows the scan: while the device searches for anchors and the emitter appends to the diff and extra
// streams, one thread per stream runs bzip2's run-length pre-pass and block CRCs over what is final and sends full
// blocks to their encoders (bz2::StreamEncoder).  What is left behind the scan is the last block of each stream.
// The emitter's thread calls start / complete / abandon; finish(k) may be called from any one thread per stream.
struct PatchFramer {
    explicit PatchFramer(int dev) { sorter.dev = dev; }
    PatchFramer(const PatchFramer &) = delete;
    PatchFramer &operator=(const PatchFramer &) = delete;
    ~PatchFramer() { abandon(); }

    // raw.diff / raw.extra get their final capacity here (both stay below m bytes), so that they never move
    bool start(bsdiff::RawStreams &raw, int64_t m)
    {
        try {
            raw.diff.reserve((size_t)m);
            raw.extra.reserve((size_t)m);
            advise_huge(raw.diff.data(), raw.diff.capacity());
            advise_huge(raw.extra.data(), raw.extra.capacity());
            base[0] = raw.diff.data();
            base[1] = raw.extra.data();
            for (int k = 0; k < 2; ++k) {
                final_len[k].store(0);
                failed[k] = false;
                enc[k].reset(new bz2::StreamEncoder(sorter.fn()));
            }
            state.store(0);
            for (int k = 0; k < 2; ++k) th[k] = std::thread([this, k] { follow(k); });
        } catch (const std::exception &) {
            abandon();
            return false;
        }
        running = true;
        return true;
    }
    void complete() { state.store(1, std::memory_order_release); }
    void abandon()
    {
        state.store(2, std::memory_order_release);
        for (std::thread &t : th) if (t.joinable()) t.join();
        for (auto &e : enc) e.reset();
        running = false;
    }
    bool ready() const { return running && state.load() == 1; }

    // stream k (0 diff, 1 extra) as a bzip2 stream
    int finish(int k, std::vector<uint8_t> &out)
    {
        if (th[k].joinable()) th[k].join();
        if (failed[k]) return fail(DQ_ERR_OOM, "bsdiff: out of memory while framing a stream");
        const int rc = enc[k]->finish(out);
        if (rc == -2) {
            std::lock_guard<std::mutex> lk(sorter.mu);
            t_err = sorter.err;
            return sorter.rc.load();
        }
        if (rc != 0) return fail(DQ_ERR_HIP, "bzip2 block transform failed");
        return DQ_OK;
    }

    std::atomic<size_t> final_len[2];                     // what the emitter has finished of diff / extra

private:
    void follow(int k)
    {
        size_t seen = 0;
        for (;;) {
            const int s = state.load(std::memory_order_acquire);           // (before the length: complete() comes after the last one)
            if (s == 2) return;
            const size_t upto = final_len[k].load(std::memory_order_acquire);
            const auto t0 = std::chrono::steady_clock::now();
            try {
                enc[k]->feed(base[k], upto, s == 1);
            } catch (...) {
                failed[k] = true;
                return;
            }
            busy_ms[k] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (s == 1) return;
            if (upto - seen < (64u << 10)) std::this_thread::sleep_for(std::chrono::microseconds(200));
            seen = upto;
        }
    }

    BlockSorter sorter;
    std::unique_ptr<bz2::StreamEncoder> enc[2];
    std::thread th[2];
    const uint8_t *base[2] = {nullptr, nullptr};
    std::atomic<int> state{0};                            // 0 the streams are growing, 1 complete, 2 given up
    bool failed[2] = {false, false};
    bool running = false;
public:
    double busy_ms[2] = {0, 0};                           // time inside feed() (DQ_TRACE prints it)
};

// Step 1 of the scan loop on the device (dq_anchor_scan.h): persistent launches walk the new file and leave the
// (cursor, hit_pos) pair of every control triple; steps 2 and 3 run here on those pairs.  A list has room for
// kAnchorRecs entries per launch -- a new file that needs more (text with a short match every few bytes) continues from
// the state the kernel left.
//
// Several grids per file ("chains").  A window costs ~15 us whatever its width -- the latency of a search and an exchange
// -- and the chain of windows is the whole cost of a diff between similar files, so up to kScanChains grids walk the file
// at once (one launch, each grid on its own lists and answer buffers): the first from the front (the chain this thread
// FOLLOWS: its entries go to the emitter), the others from equally spaced places under a shift no alignment can have.
// Where the followed chain ends an iteration at the place and under the shift where another chain ended one, that chain's
// entries ARE what the followed one would write from there on (dq_anchor_scan.h), and this thread follows it instead
// ("joins").  Every grid leaves by itself: kScanExtra iteration ends behind the start of the next chain, or -- in the
// middle of an iteration -- after kScanLaneBudget one-lane-per-position windows (a long differing stretch: the followed
// chain then takes it with the whole grid, alone, until that iteration ends).  Nothing is decided by a guess: a chain that
// is never joined is dropped, and the followed chain is launched again from where it stood.
constexpr unsigned long long kAnchorPending = ~0ull;
constexpr int kScanChains = 16;                           // grids on one new file (DQ_SCAN_CHAINS) and workgroups of each when there are
constexpr int kScanChainGroups = 32;                      // several (DQ_SCAN_GROUPS).  16 MiB pairs, Diff.Create in ms -- random bytes with
                                                          // 2000 edits / text with 2000 / random with 20 000 small edits / 4 MiB of
                                                          // unrelated bytes: one grid of 128 workgroups 38.8 / 67.1 / 261.5 / 83.2,
                                                          // 4 x 64: 19.8 / 29.9 / 76.7 / 87.1, 6 x 40: 17.9 / 25.6 / 55.2 / 82.9,
                                                          // 8 x 32: 17.1 / 23.1 / 46.1 / 85.7 (profiles/r06/r06s_chains_variants.log)
constexpr int64_t kScanMinSegment = 128ll << 10;          // bytes of new per grid below which no further one is started (DQ_SCAN_MIN_SEG):
                                                          // 1 MiB / 128 KiB -- 1 MiB pair with 128 edits 4.2 / 2.9 ms (text 6.4 / 4.0),
                                                          // 2 MiB 5.0 / 2.8 (text 11.2 / 4.7), 4 MiB 5.8 / 4.6 (text 9.3 / 6.6)
constexpr int64_t kScanExtra = 8;                         // iteration ends a grid walks on into the next one's part
constexpr int64_t kScanLaneBudget = 2;                    // one-lane-per-position windows after which a grid that is not alone leaves
static_assert(kDiffPinnedBytes >= kDiffWindowBytes + kAnchorPinned, "pinned area of the chains");

// Steps 2 and 3 for the entries of ONE speculative chain, on a thread of its own, into streams of its own: what
// TripleEmitter::take writes for an entry is a function of the entry and of `prev` (the end of the last forward extension),
// so from the first entry on at which this emitter's `prev` equals the followed emitter's, its output IS the followed
// one's and is copied instead of computed (the extensions walk every byte of both files: 5 of the 6 ms the emitter takes
// for a 16 MiB pair, and behind 8 grids it was the longest thing left).  Again nothing rests on a guess: until the two
// states have been seen equal the thread that follows the chains computes the entries itself.
struct ChainEmitter {
    struct Mark { int64_t entry; bsdiff::TripleEmitter::Anchor prev; size_t ctrl, diff, extra; };     // after an emitted entry: state, stream lengths
    bsdiff::RawStreams priv;
    std::vector<Mark> marks;                              // (full capacity reserved: read by the following thread while this one appends)
    std::atomic<int64_t> n_marks{0}, seen{0};             // marks published; list entries this thread is through with
    std::atomic<int64_t> nent{-1};                        // entries of the chain's launch once it is over (from the following thread)
    std::atomic<int> stop{0}, failed{0};
    bsdiff::TripleEmitter::Anchor first;                  // the state it began with
    std::thread th;

    void run(const uint8_t *old, int64_t n, const uint8_t *nw, int64_t m, const unsigned long long *ring, int64_t start)
    {
        try {
            bsdiff::TripleEmitter em(old, n, nw, m, priv);
            em.prev = first;
            (void)start;
            for (int64_t i = 0; !stop.load(std::memory_order_relaxed);) {
                const int64_t over = nent.load(std::memory_order_acquire);
                if (over >= 0 && i >= over) return;
                if (i >= kAnchorRecs) return;
                const unsigned long long v = __atomic_load_n(&ring[i], __ATOMIC_ACQUIRE);
                if (v == ~0ull) {
                    for (int q = 0; q < 64; ++q) __builtin_ia32_pause();
                    std::this_thread::yield();
                    continue;
                }
                if (!(v & kAsSilent)) {
                    // (a triple adds at most the bytes between the last extension and this anchor to either stream; the
                    // vectors may not move under the thread that reads them: out of room means out of this thread's job)
                    const size_t span = (size_t)((int64_t)(v >> 32) - em.prev.at) + 16;
                    if (priv.diff.size() + span > priv.diff.capacity() || priv.extra.size() + span > priv.extra.capacity() ||
                        priv.ctrl.size() + 24 > priv.ctrl.capacity() || marks.size() + 1 > marks.capacity()) {
                        failed.store(1, std::memory_order_release);
                        return;
                    }
                    em.take((int64_t)(v >> 32), (int64_t)(uint32_t)v);
                    marks.push_back(Mark{i, em.prev, priv.ctrl.size(), priv.diff.size(), priv.extra.size()});
                    n_marks.store((int64_t)marks.size(), std::memory_order_release);
                }
                ++i;
                seen.store(i, std::memory_order_release);
            }
        } catch (...) {
            failed.store(1, std::memory_order_release);
        }
    }
    void halt()
    {
        stop.store(1, std::memory_order_relaxed);
        if (th.joinable()) th.join();
    }
    // for another chain: room for `bytes` of either stream (the buffers are kept from diff to diff -- fresh ones cost their
    // page faults on the way in and 1.6 ms of munmap on the way out of a 16 MiB pair -- unless they have grown large)
    void reset(int64_t start, size_t bytes)
    {
        halt();
        stop.store(0); failed.store(0); n_marks.store(0); seen.store(0); nent.store(-1);
        marks.clear(); priv.ctrl.clear(); priv.diff.clear(); priv.extra.clear();
        first.at = start; first.in_old = 0;
        marks.reserve(1 << 14);
        priv.ctrl.reserve((size_t)24 << 14);
        priv.diff.reserve(bytes);
        priv.extra.reserve(bytes);
        advise_huge(priv.diff.data(), priv.diff.capacity());
        advise_huge(priv.extra.data(), priv.extra.capacity());
    }
    // (large ones go back to the system, off the caller's path: unmapping the 16 x 20 MB of a 128 MiB pair took 18 ms)
    void trim()
    {
        halt();
        if (priv.diff.capacity() + priv.extra.capacity() > ((size_t)16 << 20)) {
            try {
                auto *gone = new std::pair<std::vector<uint8_t>, std::vector<uint8_t>>();
                gone->first.swap(priv.diff);
                gone->second.swap(priv.extra);
                std::thread([gone] { delete gone; }).detach();
            } catch (...) {
                std::vector<uint8_t>().swap(priv.diff);
                std::vector<uint8_t>().swap(priv.extra);
            }
        }
    }
    ~ChainEmitter() { halt(); }
};
struct ScanPool { std::unique_ptr<ChainEmitter> em[kScanMaxChains]; };

struct ScanChain {
    AnchorCtl *d_ctl = nullptr, *h_up = nullptr;
    const AnchorCtl *h_out = nullptr;                     // pinned: what the chain's launch left ...
    const unsigned long long *h_landed = nullptr;         // ... and, behind it, that launch's number
    unsigned long long *ring = nullptr, *cum = nullptr;
    int64_t *dirty = nullptr;                             // list slots that may not read "pending" (kept in the device context)
    AnchorCtl st{};                                       // what the next launch starts from / what the last one left
    int64_t start = 0;                                    // first position of the chain
    bool alive = false;                                   // followed, or its entries may still be joined
    bool running = false;                                 // its part of a launch has not said it is over
    unsigned long long seq = 0;                           // that launch's number
    int groups = 0;
    int64_t taken = 0;                                    // entries of the current launch read (followed or stepped over)
    int64_t nent = 0;                                     // entries of the launch once it is over
    int64_t shift = 0;                                    // shift in force behind the entries read so far
    ChainEmitter *em = nullptr;                           // its own emitter thread (speculative chains of a launch; kept in the device context)
    int64_t mark_at = 0;                                  // marks of it whose entries the following thread has passed
};

int scan_on_device(const DiffIndex &ix, DeviceCtx &c, char *d_new, char *scratch, char *pinned_chains, const uint8_t *nw,
                   int64_t m, bsdiff::RawStreams &raw, bool *retry_on_host, PatchFramer *framer)
{
    *retry_on_host = false;
    bsdiff::TripleEmitter em(ix.old, ix.n, nw, m, raw);
    // (short files: two more threads cost more than the framing they would hide)
    const int64_t follow_min = env("DQ_FRAME_FOLLOW_MIN") ? atoll(env("DQ_FRAME_FOLLOW_MIN")) : (int64_t)256 << 10;
    if (framer && m >= follow_min && framer->start(raw, m)) em.progress = framer->final_len;
    std::lock_guard<std::mutex> lk(c.mu);                 // (the device context's stream and pinned areas)
    int rc = init_ctx(c, ix.dev);
    if (rc != DQ_OK) return rc;
    const bool trace = env("DQ_TRACE") != nullptr;
    // The grids are persistent and their workgroups wait for each other's answers: all of them must be on the device at
    // once.  What the device holds (occupancy of this kernel x compute units; a partitioned or smaller part holds
    // fewer) bounds them; below 8 workgroups, or for a while after a launch whose workgroups waited in vain (a device
    // kept full by other streams or processes -- every such launch costs its spin bound), the host loop over windows
    // takes the file instead.
    if (c.scan_groups_cap < 0) {
        int per_cu = 0, per_cu_narrow = 0, ncu = 0;
        // (the widest grid's workgroups ask for a little more than 64 KB of dynamic LDS)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&anchor_scan_kernel<int32_t, 1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)as_agp_bytes(kAsMaxGroups)));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&anchor_scan_kernel<int32_t, 2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)as_agp_bytes(kAsMaxGroups)));
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, anchor_scan_kernel<int32_t, 1>, kAsThreads, as_agp_bytes(kAsMaxGroups)) != hipSuccess) per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_narrow, anchor_scan_kernel<int32_t, 2>, kAsThreads, as_agp_bytes(kScanChainGroups)) != hipSuccess)
            per_cu_narrow = 0;
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, ix.dev) != hipSuccess) ncu = 0;
        c.scan_groups_cap = per_cu > 0 && ncu > 0 ? per_cu * ncu : kAsGroups;       // (unknown: as before)
        c.scan_groups_cap_narrow = per_cu_narrow > 0 && ncu > 0 ? per_cu_narrow * ncu : c.scan_groups_cap;
        if (trace) fprintf(stderr, "[dq] anchor scan: %d workgroups per compute unit (%d of a grid of %d) x %d compute units resident\n", per_cu, per_cu_narrow,
                           kScanChainGroups, ncu);
    }
    int cap = c.scan_groups_cap;
    if (const char *v = env("DQ_SCAN_GROUPS_CAP")) cap = std::min(cap, std::max(0, atoi(v)));           // (tests: a small device)
    const int asked = env("DQ_SCAN_GROUPS") ? std::max(8, std::min(kAsMaxGroups, atoi(env("DQ_SCAN_GROUPS")))) : 0;
    const int groups_alone = std::min(asked ? asked : kAsGroups, cap);                 // a grid that is alone on the file
    const int groups_chain = std::min(asked ? asked : kScanChainGroups, cap);          // one of several
    // (workgroups of narrow grids the device holds at once: more than of the widest, their LDS is a quarter)
    int cap_chains = groups_chain <= kScanChainGroups ? std::max(cap, c.scan_groups_cap_narrow) : cap;
    if (const char *v = env("DQ_SCAN_GROUPS_CAP")) cap_chains = std::min(cap_chains, std::max(0, atoi(v)));
    int chains_max = env("DQ_SCAN_CHAINS") ? std::max(1, std::min(kScanMaxChains, atoi(env("DQ_SCAN_CHAINS")))) : kScanChains;
    if (groups_chain >= 8) chains_max = std::min(chains_max, cap_chains / groups_chain);
    const int64_t min_seg = env("DQ_SCAN_MIN_SEG") ? std::max<int64_t>(64, atoll(env("DQ_SCAN_MIN_SEG"))) : kScanMinSegment;
    const int64_t extra_ends = env("DQ_SCAN_EXTRA") ? std::max<int64_t>(1, atoll(env("DQ_SCAN_EXTRA"))) : kScanExtra;
    const int64_t lane_budget = env("DQ_SCAN_LANE_BUDGET") ? std::max<int64_t>(1, atoll(env("DQ_SCAN_LANE_BUDGET"))) : kScanLaneBudget;
    t_diff_info[4] = chains_max > 1 && m >= 2 * min_seg ? groups_chain : groups_alone;
    if (groups_alone < 8 || c.scan_skip > 0) {
        if (c.scan_skip > 0) --c.scan_skip;
        // (not an error: the host loop takes the file and the call succeeds -- dq_last_error() must not be left saying
        // otherwise behind a DQ_OK, so nothing goes through fail(); dq_last_diff_info counts the file, skipped ones too)
        if (trace)
            fprintf(stderr, "[dq] %s\n", groups_alone < 8 ? "anchor scan: the device holds fewer than 8 of its workgroups" : "anchor scan: skipped after a starved launch");
        *retry_on_host = true;
        return DQ_ERR_HIP;
    }
    ScanChain ch[kScanMaxChains];
    for (int k = 0; k < kScanMaxChains; ++k) {
        ScanChain &x = ch[k];
        char *hp = pinned_chains + kAsSlotAt + (size_t)k * kAsSlotBytes;
        x.d_ctl = reinterpret_cast<AnchorCtl *>(scratch + (size_t)k * 256);
        x.h_up = reinterpret_cast<AnchorCtl *>(pinned_chains + (size_t)k * 256);
        x.ring = reinterpret_cast<unsigned long long *>(hp);
        x.cum = x.ring + kAnchorRecs;
        x.h_out = reinterpret_cast<const AnchorCtl *>(x.ring + 2 * kAnchorRecs);
        x.h_landed = x.ring + 2 * kAnchorRecs + 31;
        x.dirty = &c.scan_dirty[k];
    }
    struct EmitterGuard {                                 // (no emitter thread outlives this call, however it is left)
        DeviceCtx &c;
        ~EmitterGuard()
        {
            if (!c.scan_pool) return;
            ScanPool &pool = *static_cast<ScanPool *>(c.scan_pool.get());
            for (auto &e : pool.em) if (e) e->trim();
        }
    } emitter_guard{c};
    double emit_ms = 0;                                   // (DQ_TRACE: time inside the emitter)
    double emit_phase_ms[3] = {0, 0, 0};
    if (trace) em.phase_ms = emit_phase_ms;
    const auto t_host0 = std::chrono::steady_clock::now();
    auto host_ms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_host0).count(); };
    // Whatever way this function is left while a launch is out -- a failed copy, an exception out of the emitter -- the
    // kernel must be off the stream before anybody refills the lists or the answer buffers: its chains are told to stop
    // (the error word every spin looks at), the stream drains, the timing events go back to their pool.
    bool launch_out = false;                              // a launch may still be on the stream
    struct LaunchGuard {
        DeviceCtx &c; ScanChain *ch; bool &out;
        ~LaunchGuard()
        {
            if (!out) return;
            hipStream_t side = nullptr;
            if (hipStreamCreateWithFlags(&side, hipStreamNonBlocking) == hipSuccess) {
                static const unsigned int one = 1;
                for (int k = 0; k < kScanMaxChains; ++k)
                    if (ch[k].running) (void)hipMemcpyAsync(&ch[k].d_ctl->error, &one, sizeof(one), hipMemcpyHostToDevice, side);
                (void)hipStreamSynchronize(side);
                (void)hipStreamDestroy(side);
            }
            for (int k = 0; k < kScanMaxChains; ++k) if (ch[k].running) *ch[k].dirty = kAnchorRecs;
            drop_pending(c, c.stream);                    // (synchronises c.stream first)
        }
    } guard{c, ch, launch_out};

    // Search counts along the path this thread followed: the last chain's own count + (count at the entry a chain was
    // left at - count at the entry its successor was joined at) over the joins; the counts beside the entries are read
    // once the launch that wrote them has said it is over.
    struct Join { int from, to; int64_t from_at, to_at; bool have_from, have_to; unsigned long long from_v, to_v; };
    std::vector<Join> joins;
    auto settle = [&](int k) {
        for (Join &j : joins) {
            if (j.from == k && !j.have_from) { j.from_v = ch[k].cum[j.from_at]; j.have_from = true; }
            if (j.to == k && !j.have_to) { j.to_v = ch[k].cum[j.to_at]; j.have_to = true; }
        }
    };
    bool gave_up = false;
    unsigned long long t_first = 0;                       // (DQ_TRACE: device clock of the first chain that came back)
    // is chain k's part of its launch over?  (x.st, x.nent are what it left)  1 yes, 0 not yet, < 0 error
    auto landed = [&](int k, bool wait) -> int {
        ScanChain &x = ch[k];
        if (!x.running) return 1;
        for (uint32_t spins = 0;;) {
            if (__atomic_load_n(x.h_landed, __ATOMIC_ACQUIRE) == x.seq) break;
            if (!wait) return 0;
            if ((++spins & 4095u) == 0) {                  // (a launch that died leaves no word: ask the stream now and then)
                const hipError_t q = hipStreamQuery(c.stream);
                if (q == hipSuccess) {
                    if (__atomic_load_n(x.h_landed, __ATOMIC_ACQUIRE) == x.seq) break;
                    return fail(DQ_ERR_HIP, "anchor scan: a launch ended without its result");
                }
                if (q != hipErrorNotReady) return fail(DQ_ERR_HIP, "anchor scan: stream query failed", q);
                std::this_thread::yield();
            } else {
                __builtin_ia32_pause();
            }
        }
        x.running = false;
        AnchorCtl back;
        std::memcpy(&back, x.h_out, sizeof(back));
        if (back.error) { gave_up = true; if (x.em) x.em->stop.store(1); return 1; }
        if ((int64_t)back.nrec < x.taken || (int64_t)back.nrec > kAnchorRecs) return fail(DQ_ERR_HIP, "anchor scan: bad record count");
        x.st = back;
        x.nent = (int64_t)back.nrec;
        if (x.em) x.em->nent.store(x.nent, std::memory_order_release);
        *x.dirty = x.nent;
        raw.windows += (int64_t)back.windows;
        raw.exact += (int64_t)back.stops;
        settle(k);
        if (trace && (t_first == 0 || back.t_begin < t_first)) t_first = back.t_begin;
        if (trace)
            fprintf(stderr, "[dq] anchor scan, chain %d from %lld, %d workgroups: over at %.3f ms (%lld of %lld entries read), %llu windows, %llu stop points, "
                    "left at %lld%s; on the device from %.3f to %.3f ms; workgroup 0: search %.2f ms, waiting for answers %.2f ms, evaluation %.2f ms, "
                    "stop points %.2f ms\n", k, (long long)x.start, x.groups, host_ms(), (long long)x.taken, (long long)x.nent, back.windows, back.stops,
                    (long long)(back.mid ? back.i : back.cursor), back.done ? " (end of file)" : back.mid ? " (in the middle of an iteration)" : "",
                    (double)(back.t_begin - t_first) * 1e-5, (double)(back.t_end - t_first) * 1e-5, back.t_search * 1e-5, back.t_wait * 1e-5,
                    back.t_eval * 1e-5, back.t_stop * 1e-5);
        return 1;
    };
    // every chain of the launch that is out over, the stream drained (before the buffers of any chain are touched again)
    auto drain = [&]() -> int {
        if (!launch_out) return DQ_OK;
        for (int k = 0; k < kScanMaxChains; ++k) {
            const int r = landed(k, true);
            if (r < 0) return r;
        }
        HIP_TRY(hipStreamSynchronize(c.stream));
        launch_out = false;
        return flush_profile(c);
    };
    // ---- one launch: the chains in `slots`, each from its x.st ----
    auto launch = [&](const std::vector<int> &slots, int groups) -> int {
        AnchorLaunch ln{};
        ln.chains = (int)slots.size();
        ln.groups = groups;
        ln.seq = ++c.scan_seq;
        Launcher L{c, c.stream, g_prof_on.load()};
        for (size_t q = 0; q < slots.size(); ++q) {
            ScanChain &x = ch[slots[q]];
            ln.slot[q] = slots[q];
            AnchorCtl up = x.st;
            up.nrec = 0; up.done = 0; up.windows = 0; up.stops = 0; up.error = 0;
            up.t_search = up.t_wait = up.t_eval = up.t_stop = 0;
            up.pad = (trace ? 1u : 0u) | ((unsigned)(env("DQ_SCAN_POLL_SLEEP") ? std::max(1, std::min(32, atoi(env("DQ_SCAN_POLL_SLEEP")))) : 16) << 8);
            // (the tests: a spin bound of 2^k polls -- DQ_FAULT=spin: 2 --, and workgroup k - 1 as the straggler of every window)
            if (t_fault.spin) up.pad |= 1u << 16;
            else if (const char *v = env("DQ_SCAN_SPIN_LOG2")) up.pad |= (unsigned)std::max(1, std::min(24, atoi(v))) << 16;
            if (const char *v = env("DQ_SCAN_SLOW_GROUP")) up.pad |= (unsigned)std::max(0, std::min(255, atoi(v))) << 24;
            *x.h_up = up;
            const int64_t refill = std::min<int64_t>(*x.dirty, kAnchorRecs);
            for (int64_t r = 0; r < refill; ++r) x.ring[r] = kAnchorPending;
            *x.dirty = kAnchorRecs;                       // (until the chain has said how many it wrote)
        }
        std::atomic_thread_fence(std::memory_order_seq_cst);
        HIP_TRY(hipMemcpyAsync(scratch, pinned_chains, (size_t)kScanMaxChains * 256, hipMemcpyHostToDevice, c.stream));
        HIP_TRY(hipMemsetAsync(scratch + kAsFinishedAt, 0, (size_t)kScanMaxChains * 2048, c.stream));
        for (int s : slots)                                // (no answer word carries a window's tag yet)
            HIP_TRY(hipMemsetAsync(scratch + kAsAnswersAt + (size_t)s * kAnchorAnswers, 0xff, kAnchorAnswers, c.stream));
        // (all its registers where the launch fits the device with one workgroup per compute unit)
        if (groups * ln.chains <= cap)
            LAUNCH(L, DQ_K_MATCH_SEARCH, m, m * 2,
                   hipLaunchKernelGGL((anchor_scan_kernel<int32_t, 1>), dim3(groups * ln.chains), dim3(kAsThreads), as_agp_bytes(groups), c.stream,
                                      (const uint8_t *)ix.d_old, ix.n, (const int32_t *)ix.d_sa, (const uint8_t *)d_new, m,
                                      (const int32_t *)ix.d_tab, ix.pk, scratch, pinned_chains, ln));
        else
            LAUNCH(L, DQ_K_MATCH_SEARCH, m, m * 2,
                   hipLaunchKernelGGL((anchor_scan_kernel<int32_t, 2>), dim3(groups * ln.chains), dim3(kAsThreads), as_agp_bytes(groups), c.stream,
                                      (const uint8_t *)ix.d_old, ix.n, (const int32_t *)ix.d_sa, (const uint8_t *)d_new, m,
                                      (const int32_t *)ix.d_tab, ix.pk, scratch, pinned_chains, ln));
        launch_out = true;
        for (int s : slots) {
            ScanChain &x = ch[s];
            x.running = true; x.seq = ln.seq; x.groups = groups; x.taken = 0; x.nent = 0;
        }
        return DQ_OK;
    };
    auto entry_cursor = [](unsigned long long v) -> int64_t { return (int64_t)((v & ~kAsSilent) >> 32); };

    int cur = 0;                                          // the chain this thread follows
    ch[0].alive = true;                                   // (from the loop's initial state: all zero)
    int serial_log2 = 0;                                  // iteration ends the next launch that is alone on purpose walks: 2^this
    bool serial_next = false;
    int64_t n_joins = 0, n_launches = 0, n_dropped = 0, n_adopted = 0;
    // emitters of the speculative chains on threads of their own (DQ_SCAN_PAR_EMIT=0: everything on this thread)
    const bool par_emit = env("DQ_SCAN_PAR_EMIT") ? atoi(env("DQ_SCAN_PAR_EMIT")) != 0 : true;
    bool want_adopt = false;                              // the followed chain has an emitter whose state has not been seen equal to em's yet
    bool adopting = false;                                // ... it has: its output is copied
    // Launch the followed chain (again) from its state -- and, when no other chain is left and enough of the file is,
    // new chains over the rest of it.
    auto relaunch = [&]() -> int {
        rc = drain();                                     // (one launch at a time: chains that were left behind end by themselves)
        if (rc != DQ_OK || gave_up) return rc;
        ScanChain &t = ch[cur];
        const int64_t pos = t.st.mid ? t.st.i : t.st.cursor + t.st.hit_len;
        bool others = false;
        for (int k = 0; k < kScanMaxChains; ++k) others = others || (k != cur && ch[k].alive);
        int spawn = 0;
        if (!others && !serial_next && chains_max > 1 && groups_chain >= 8 && m - pos >= 2 * min_seg)
            spawn = (int)std::min<int64_t>(chains_max, (m - pos) / min_seg);
        t.st.lane_budget = 0; t.st.extra = 0; t.st.stop_at = 0;
        std::vector<int> slots{cur};
        if (spawn > 1) {
            const int64_t seg = (m - pos) / spawn;
            int slot = 0;
            for (int j = 1; j < spawn; ++j) {
                while (slot == cur) ++slot;
                ScanChain &x = ch[slot];
                x.st = AnchorCtl{};
                x.start = pos + j * seg;
                x.st.cursor = x.start; x.st.shift = ix.n;  // (agree() is false everywhere under it: no alignment has this shift)
                x.shift = ix.n;
                x.alive = true;
                x.st.lane_budget = lane_budget;
                if (slots.size() > 1) { ch[slots.back()].st.stop_at = x.start; ch[slots.back()].st.extra = extra_ends; }
                slots.push_back(slot);
                ++slot;
            }
            t.st.stop_at = ch[slots[1]].start; t.st.extra = extra_ends; t.st.lane_budget = lane_budget;
        } else if (serial_next) {
            t.st.extra = (int64_t)1 << serial_log2;       // (stop_at 0: every iteration end counts)
        } else if (others) {
            // as far as the start of the next chain that is still worth reaching; a long differing stretch is left to a
            // launch of its own
            int64_t next_start = -1;
            for (int k = 0; k < kScanMaxChains; ++k)
                if (k != cur && ch[k].alive && ch[k].start > pos && (next_start < 0 || ch[k].start < next_start)) next_start = ch[k].start;
            if (next_start >= 0) { t.st.stop_at = next_start; t.st.extra = extra_ends; }
            t.st.lane_budget = lane_budget;
        }
        serial_next = false;
        for (int sl : slots) {                             // (their lists are about to be refilled: the threads that read them end first)
            if (ch[sl].em) ch[sl].em->halt();
            ch[sl].em = nullptr;
        }
        adopting = false; want_adopt = false;             // (the followed chain's new entries are computed here)
        rc = launch(slots, slots.size() > 1 ? groups_chain : groups_alone);
        if (rc != DQ_OK) return rc;
        n_launches += (int64_t)slots.size();
        if (par_emit) {
            try {
                if (!c.scan_pool) c.scan_pool = std::make_shared<ScanPool>();
                ScanPool &pool = *static_cast<ScanPool *>(c.scan_pool.get());
                for (size_t q = 1; q < slots.size(); ++q) {
                    ScanChain &x = ch[slots[q]];
                    if (!pool.em[slots[q]]) pool.em[slots[q]].reset(new ChainEmitter);
                    ChainEmitter &e = *pool.em[slots[q]];
                    // (its own part of the file and a quarter more; a chain that walks further -- nothing joined it for a
                    // long time -- leaves the rest to the thread that follows it)
                    const int64_t upto = q + 1 < slots.size() ? ch[slots[q + 1]].start : m;
                    const int64_t part = upto - x.start;
                    e.reset(x.start, (size_t)std::min<int64_t>(m - x.start, part + part / 4 + (64 << 10)) + 64);
                    x.mark_at = 0;
                    e.th = std::thread([&e, &ix, nw, m, ring = x.ring, start = x.start] { e.run(ix.old, ix.n, nw, m, ring, start); });
                    x.em = &e;
                }
            } catch (const std::exception &) {
                for (int sl : slots) {                     // (no memory or no thread: everything is computed here, as without them)
                    if (ch[sl].em) ch[sl].em->halt();
                    ch[sl].em = nullptr;
                }
            }
        }
        return DQ_OK;
    };
    // The followed chain has just ended an iteration at c (silent: without a triple; under shift s): is that where
    // another chain ended one under the same shift?  (Entries of that chain in front of c are stepped over for good:
    // the followed chain's ends only grow.)
    auto try_join = [&](int64_t cpos, bool silent, int64_t s) -> int {
        for (;;) {
            int best = -1;
            for (int k = 0; k < kScanMaxChains; ++k)
                if (k != cur && ch[k].alive && ch[k].start <= cpos && (best < 0 || ch[k].start > ch[best].start)) best = k;
            if (best < 0) return 0;
            ScanChain &x = ch[best];
            bool again = false;
            for (uint32_t idle = 0;;) {
                if ((x.running || x.taken < x.nent) && x.taken < kAnchorRecs) {
                    const unsigned long long v = __atomic_load_n(&x.ring[x.taken], __ATOMIC_ACQUIRE);
                    if (v != kAnchorPending) {
                        idle = 0;
                        const int64_t cj = entry_cursor(v);
                        const bool sj = (v & kAsSilent) != 0;
                        if (cj < cpos) {
                            if (!sj) x.shift = (int64_t)(uint32_t)v - cj;
                            ++x.taken;
                            continue;
                        }
                        if (cj == cpos && sj == silent && (!silent || x.shift == s)) {
                            if (!sj) x.shift = (int64_t)(uint32_t)v - cj;
                            joins.push_back(Join{cur, best, ch[cur].taken - 1, x.taken, false, false, 0, 0});
                            ++x.taken;
                            if (!ch[cur].running) settle(cur);
                            if (!x.running) settle(best);
                            ch[cur].alive = false;        // (its grid leaves by itself a few iterations on)
                            if (ch[cur].em) ch[cur].em->stop.store(1, std::memory_order_relaxed);
                            adopting = false;
                            want_adopt = x.em != nullptr;
                            cur = best;
                            ++n_joins;
                            serial_log2 = 0;
                            if (trace)
                                fprintf(stderr, "[dq] anchor scan: chain %d joined at %lld (its entry %lld), %.3f ms (emitter %.2f ms so far; its own emitter: %s, %lld entries seen, %lld marks)\n",
                                        best, (long long)cpos, (long long)x.taken - 1, host_ms(), emit_ms, !x.em ? "none" : x.em->failed.load() ? "failed" : "running",
                                        x.em ? (long long)x.em->seen.load() : 0ll, x.em ? (long long)x.em->n_marks.load() : 0ll);
                            return 1;
                        }
                        break;                            // its next end lies behind c, or at c under another shift
                    }
                }
                if (!x.running) {
                    if (x.taken >= x.nent) {                   // nothing of it lies behind c
                        x.alive = false; ++n_dropped; again = true;
                        if (x.em) x.em->stop.store(1, std::memory_order_relaxed);
                        break;
                    }
                    return fail(DQ_ERR_HIP, "anchor scan: a record slot was left unfilled");
                }
                // the chain has not got there yet (it started when the followed one did: rare): wait for its entry or its end
                if ((++idle & 63u) != 0) { __builtin_ia32_pause(); continue; }
                const int r = landed(best, false);
                if (r < 0) return r;
                if (gave_up) return 0;
            }
            if (!again) return 0;
        }
    };

    rc = relaunch();
    if (rc != DQ_OK) return rc;
    if (trace) fprintf(stderr, "[dq] anchor scan: first launch out at %.3f ms\n", host_ms());
    for (uint32_t idle = 0; !gave_up;) {
        ScanChain &t = ch[cur];
        if ((t.running || t.taken < t.nent) && t.taken < kAnchorRecs) {
            const unsigned long long v = __atomic_load_n(&t.ring[t.taken], __ATOMIC_ACQUIRE);
            if (v != kAnchorPending) {
                const int64_t cpos = entry_cursor(v);
                const bool silent = (v & kAsSilent) != 0;
                // Has the chain's own emitter been through the entries passed so far, and does it stand where em stands?
                // Then what it writes from here on is what em would write.
                if (want_adopt && !adopting && t.em) {
                    ChainEmitter &e = *t.em;
                    // (an emitter that ran out of room -- its chain walked far beyond its part -- has stopped for good; what it
                    // wrote up to there is as good as any)
                    if (e.seen.load(std::memory_order_acquire) < t.taken) {
                        if (e.failed.load(std::memory_order_acquire)) want_adopt = false;
                    } else {
                        const int64_t nm = e.n_marks.load(std::memory_order_acquire);
                        while (t.mark_at < nm && e.marks[(size_t)t.mark_at].entry < t.taken) ++t.mark_at;
                        const bsdiff::TripleEmitter::Anchor theirs = t.mark_at > 0 ? e.marks[(size_t)t.mark_at - 1].prev : e.first;
                        if (theirs.at == em.prev.at && theirs.in_old == em.prev.in_old) { adopting = true; want_adopt = false; }
                    }
                }
                if (!silent && adopting) {
                    // the chain's emitter has this entry's triple and bytes, or is about to
                    ChainEmitter &e = *t.em;
                    if (e.n_marks.load(std::memory_order_acquire) <= t.mark_at) {
                        if (e.failed.load(std::memory_order_acquire)) { adopting = false; continue; }       // (computed here from now on)
                        if ((++idle & 63u) == 0) std::this_thread::yield(); else __builtin_ia32_pause();
                        continue;
                    }
                    const ChainEmitter::Mark &mk = e.marks[(size_t)t.mark_at];
                    if (mk.entry != t.taken) return fail(DQ_ERR_HIP, "anchor scan: a chain's emitter lost step with its list");
                    const size_t c0 = t.mark_at > 0 ? e.marks[(size_t)t.mark_at - 1].ctrl : 0, d0 = t.mark_at > 0 ? e.marks[(size_t)t.mark_at - 1].diff : 0,
                                 x0 = t.mark_at > 0 ? e.marks[(size_t)t.mark_at - 1].extra : 0;
                    raw.ctrl.insert(raw.ctrl.end(), e.priv.ctrl.data() + c0, e.priv.ctrl.data() + mk.ctrl);
                    raw.diff.insert(raw.diff.end(), e.priv.diff.data() + d0, e.priv.diff.data() + mk.diff);
                    raw.extra.insert(raw.extra.end(), e.priv.extra.data() + x0, e.priv.extra.data() + mk.extra);
                    em.prev = mk.prev;
                    if (em.progress) {
                        em.progress[0].store(raw.diff.size(), std::memory_order_release);
                        em.progress[1].store(raw.extra.size(), std::memory_order_release);
                    }
                    ++t.mark_at;
                    ++n_adopted;
                    t.shift = (int64_t)(uint32_t)v - cpos;
                } else if (!silent) {
                    if (trace) {
                        const auto t0 = std::chrono::steady_clock::now();
                        em.take(cpos, (int64_t)(uint32_t)v);
                        emit_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                    } else {
                        em.take(cpos, (int64_t)(uint32_t)v);
                    }
                    t.shift = (int64_t)(uint32_t)v - cpos;
                }
                idle = 0;
                ++t.taken;
                const int j = try_join(cpos, silent, t.shift);
                if (j < 0) return j;
                continue;
            }
        }
        if (t.running) {
            if ((++idle & 63u) != 0) { if ((idle & 7u) == 0) __builtin_ia32_pause(); continue; }
            const int r = landed(cur, false);
            if (r < 0) return r;
            // (nothing new for thousands of looks: the kernel is inside a long search -- leave the core to the framing
            // and encoder threads of this and other callers for a moment)
            if (r == 0) {
                if ((idle & 0xffffu) == 0) {               // (a launch that died would never say so: ask the stream now and then)
                    const hipError_t q = hipStreamQuery(c.stream);
                    if (q != hipSuccess && q != hipErrorNotReady) return fail(DQ_ERR_HIP, "anchor scan: stream query failed", q);
                    if (q == hipSuccess && landed(cur, false) == 0) return fail(DQ_ERR_HIP, "anchor scan: a launch ended without its result");
                }
                if (idle >= (1u << 14)) std::this_thread::yield();
            }
            continue;
        }
        if (t.taken < t.nent) return fail(DQ_ERR_HIP, "anchor scan: a record slot was left unfilled");
        // the followed chain's launch is over and read to its end
        if (t.st.done) break;
        if (t.nent == 0 && !t.st.mid) return fail(DQ_ERR_HIP, "anchor scan: no progress");
        if (t.st.mid) {                                   // it left a long differing stretch: that iteration alone, with all it can get
            serial_next = true;
        } else if (t.st.extra > 0 && t.st.stop_at == 0) { // a launch that was alone on purpose has ended its iterations
            serial_log2 = std::min(serial_log2 + 1, 12);
        }
        rc = relaunch();
        if (rc != DQ_OK) return rc;
    }
    if (trace) fprintf(stderr, "[dq] anchor scan: end of file at %.3f ms\n", host_ms());
    rc = drain();                                         // (chains that were left behind end by themselves)
    if (rc != DQ_OK) return rc;
    // (a workgroup of a persistent grid did not get onto the device in time -- a device kept full by other work: the
    // caller runs the host loop over windows instead; nothing of this attempt is kept; the next 16 diffs on this device
    // do not try again)
    if (gave_up) {
        if (!t_fault.spin && !env("DQ_SCAN_SPIN_LOG2")) c.scan_skip = 16;     // (not under the tests' own bound)
        if (trace) fprintf(stderr, "[dq] anchor scan: grid barrier timed out\n");
        *retry_on_host = true;
        return DQ_ERR_HIP;                                  // (no fail(): the host loop's DQ_OK must not carry this text)
    }
    unsigned long long searches = ch[cur].st.searches;
    for (const Join &j : joins) {
        if (!j.have_from || !j.have_to) return fail(DQ_ERR_HIP, "anchor scan: a join was left unsettled");
        searches += j.from_v - j.to_v;
    }
    raw.searches += (int64_t)searches;
    t_diff_info[5] = n_launches; t_diff_info[6] = n_joins; t_diff_info[7] = n_dropped; t_diff_info[8] = n_adopted;
    if (em.progress) framer->complete();
    if (trace)
        fprintf(stderr, "[dq] anchor scan: %lld chain launches, %lld joins, %lld chains dropped in %.3f ms; emitter (steps 2 and 3 on the host, beside the kernels): %.2f ms "
                "on this thread, %lld triples taken from the chains' own emitters\n", (long long)n_launches, (long long)n_joins, (long long)n_dropped, host_ms(), emit_ms,
                (long long)n_adopted);
    if (trace) fprintf(stderr, "[dq] emitter: extensions %.2f ms, diff bytes %.2f ms, extra bytes and triple %.2f ms\n", emit_phase_ms[0], emit_phase_ms[1], emit_phase_ms[2]);
    return DQ_OK;
}

// Diff.Create's data path up to the raw streams for one new file: upload it, run the scan loop over windows of answers
int diff_index_scan(const DiffIndex &ix, const uint8_t *nw, int64_t m, bsdiff::RawStreams &raw, PatchFramer *framer = nullptr)
{
    if (m < 0 || (m > 0 && !nw)) return fail(DQ_ERR_BAD_ARGS, "bad arguments");
    if (m > 0x7fffffffLL) return fail(DQ_ERR_TOO_LARGE, "the BSDIFF40 path takes files below 2 GiB (int indices, as the reference)");
    for (int64_t &x : t_diff_info) x = 0;
    if (m == 0) return DQ_OK;
    const int dev = ix.dev;
    HIP_TRY(hipSetDevice(dev));
    DeviceCtx &c = ctx0(dev);
    const bool trace = env("DQ_TRACE") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    auto stamp = [&](const char *what) {
        if (trace) fprintf(stderr, "[dq] bsdiff %-14s at %8.3f ms (%.3f ms of the clock)\n", what,
                           std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(),
                           std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count() -
                               1e3 * (double)(long long)(std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() / 100.0) * 100.0);
    };
    const size_t b_new = align_up((size_t)m + 16);
    // (+ the mailbox of the window kernel; + control block, answers and anchor list of the device's scan)
    int rc = grow_cached(&c.diff_dev, &c.diff_dev_bytes, b_new + 256 + kAnchorScratch, "hipMalloc(bsdiff buffers)");
    if (rc != DQ_OK) return rc;
    if (!c.diff_pinned) {
        hipError_t e = dq_host_malloc((void **)&c.diff_pinned, kDiffPinnedBytes, hipHostMallocCoherent);   // (windows + the packed answers the loop polls)
        if (e != hipSuccess) return fail(DQ_ERR_OOM, "hipHostMalloc(search windows)", e);
        std::memset(c.diff_pinned, 0, kDiffPinnedBytes);   // (no chain's "over" word may read as a launch number to come)
    }
    char *pinned = c.diff_pinned;
    const size_t b_win = align_up((size_t)(SearchWindows::kMaxWindow + 2) * 4);
    static_assert(kDiffWindowBytes >= 2 * ((size_t)(SearchWindows::kMaxWindow + 2) * 4 + 256) +
                  (size_t)(SearchWindows::kWaveWindow + 2 * (SearchWindows::kSecond + 1)) * 8 + 256, "pinned window area");
    char *d_new = c.diff_dev;
    stamp("buffers");
    HIP_TRY(hipMemcpy(d_new, nw, (size_t)m, hipMemcpyHostToDevice));
    stamp("new on device");
    // the anchor search of the scan loop on the device (default), or the host loop over windows of device answers
    // (its answer words have 31 bits for a length, all ones standing for "not exact": files of 2^31 - 1 bytes take the host loop)
    const bool device_scan = (env("DQ_SCAN_DEVICE") ? atoi(env("DQ_SCAN_DEVICE")) != 0 : true) && ix.n < 0x7fffffffLL && m < 0x7fffffffLL;
    if (device_scan) {
        bool retry_on_host = false;
        rc = scan_on_device(ix, c, d_new, d_new + b_new + 256, pinned + kDiffWindowBytes, nw, m, raw, &retry_on_host, framer);
        stamp("scan (device)");
        if (trace)
            fprintf(stderr, "[dq] device scan: %lld searches, %lld windows, %lld stop points, %zu triples%s\n", (long long)raw.searches,
                    (long long)raw.windows, (long long)raw.exact, raw.ctrl.size() / 24, retry_on_host ? " -- given up, host loop instead" : "");
        if (!retry_on_host) {
            t_diff_info[0] = raw.searches; t_diff_info[1] = raw.windows; t_diff_info[2] = raw.exact;
            return rc;
        }
        t_diff_info[3] += 1;                               // (not silently: dq_last_diff_info says the host loop took this file)
        if (framer) framer->abandon();                     // (before the streams it reads go away)
        raw = bsdiff::RawStreams{};
    }
    SearchWindows win{ix.d_old, ix.d_sa, d_new, ix.n, m, dev};
    win.d_ptab = ix.d_tab;
    win.pk = ix.pk;
    win.h_pos = reinterpret_cast<int32_t *>(pinned);
    win.h_len = reinterpret_cast<int32_t *>(pinned + b_win);
    win.h_packed = env("DQ_NO_POLL") ? nullptr : reinterpret_cast<uint64_t *>(pinned + 2 * b_win);
    win.d_mail = d_new + b_new;
    HIP_TRY(hipMemset(win.d_mail, 0, 16));
    win.no_second = env("DQ_NO_SECOND_STAGE") != nullptr;
    if (const char *v = env("DQ_WIN_MIN")) win.min_window = std::min<int64_t>(std::max(16, atoi(v)), SearchWindows::kWaveWindow);
    if (const char *v = env("DQ_WIN_SECOND")) win.second = std::min<int64_t>(std::max(16, atoi(v)), SearchWindows::kSecond);
    win.next_size = win.min_window;
    if (const char *v = env("DQ_WALK_ON")) win.walk_on = atoi(v) != 0;
    win.no_resume = env("DQ_NO_RESUME") != nullptr;
    rc = bsdiff::scan_loop(ix.old, ix.n, nw, m, win, raw);
    raw.windows = win.windows;
    raw.exact = win.exact;
    stamp("scan loop");
    if (trace)
        fprintf(stderr, "[dq] scan loop: %lld searches, %lld windows (%lld of them answered ahead by the second stage), %lld exact repeats\n",
                (long long)raw.searches, (long long)win.windows, (long long)win.predicted, (long long)win.exact);
    // (the loop polled the kernels' own completion counts: drain the stream before the buffers are reused)
    const hipError_t drained = hipStreamSynchronize(c.stream);
    if (rc == DQ_OK && drained != hipSuccess) return fail(DQ_ERR_HIP, "scan loop: stream did not drain", drained);
    t_diff_info[0] = raw.searches; t_diff_info[1] = raw.windows; t_diff_info[2] = raw.exact;
    return rc;
}

// Diff.Create's data path up to the raw streams: sort old on the device, keep the SA there, run the scan loop
int bsdiff_raw(const uint8_t *old, int64_t n, const uint8_t *nw, int64_t m, int32_t device, bsdiff::RawStreams &raw,
               PatchFramer *framer = nullptr)
{
    if (n < 0 || m < 0) return fail(DQ_ERR_BAD_ARGS, "negative length");
    if ((n > 0 && !old) || (m > 0 && !nw)) return fail(DQ_ERR_BAD_ARGS, "null buffer");
    if (n > 0x7fffffffLL || m > 0x7fffffffLL) return fail(DQ_ERR_TOO_LARGE, "the BSDIFF40 path takes files below 2 GiB (int indices, as the reference)");
    int dev = 0;
    int rc = resolve_device(device, &dev);
    if (rc != DQ_OK) return rc;
    if (m == 0) return DQ_OK;
    std::lock_guard<std::mutex> one_diff(ctx0(dev).diff_mu);
    DiffIndex ix;
    rc = diff_index_build(old, n, dev, nullptr, nullptr, /*cached=*/true, &ix);
    if (rc != DQ_OK) return rc;
    return diff_index_scan(ix, nw, m, raw, framer);
}

// one bzip2 stream; the Burrows-Wheeler transform of each block through the suffix sorter (blocks of a long stream
// are encoded on several threads: the sorter is called concurrently, each call leasing its own device context)
int bz2_stream(const std::vector<uint8_t> &src, std::vector<uint8_t> &out, int dev)
{
    BlockSorter sorter;
    sorter.dev = dev;
    const int rc = bz2::bz2_compress(src.data(), src.size(), out, sorter.fn());
    if (rc == -2) { t_err = sorter.err; return sorter.rc.load(); }
    if (rc != 0) return fail(DQ_ERR_HIP, "bzip2 block transform failed");
    return DQ_OK;
}

// header + the three streams (Diff.cs:54-70 / :196-252).  The streams are framed side by side on three host threads:
// their run-length / MTF / Huffman work overlaps, the block sorts take turns on the device.
int frame_patch(const bsdiff::RawStreams &raw, int64_t m, int dev, std::vector<uint8_t> &patch, PatchFramer *framer = nullptr)
{
    const bool trace = env("DQ_TRACE") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    const bool followed = framer && framer->ready();       // diff and extra were framed while they grew: their last blocks are left
    std::vector<uint8_t> z[3];
    const std::vector<uint8_t> *src[3] = {&raw.ctrl, &raw.diff, &raw.extra};
    int rcs[3] = {DQ_OK, DQ_OK, DQ_OK};
    std::string errs[3];
    auto work = [&](int k) {
        try {
            rcs[k] = followed && k > 0 ? framer->finish(k - 1, z[k]) : bz2_stream(*src[k], z[k], dev);
            if (rcs[k] != DQ_OK) errs[k] = t_err;
            if (trace && followed && k > 0)
                fprintf(stderr, "[dq] bsdiff stream %d: %.2f ms of run-length pre-pass and CRC behind the scan\n", k, framer->busy_ms[k - 1]);
            if (trace) fprintf(stderr, "[dq] bsdiff stream %d framed   at %8.3f ms (%zu -> %zu bytes%s)\n", k,
                               std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(),
                               src[k]->size(), z[k].size(), followed && k > 0 ? ", behind the scan" : "");
        } catch (const std::exception &e) {
            rcs[k] = DQ_ERR_OOM;
            errs[k] = std::string("bsdiff: ") + e.what();
        }
    };
    {
        JoinAll threads;
        // (streams of a few KB are not worth a thread)
        const bool parallel = raw.ctrl.size() + raw.diff.size() + raw.extra.size() >= (1u << 16) && !env("DQ_BZ2_SERIAL");
        for (int k = 1; k < 3; ++k) {
            if (!parallel) { work(k); continue; }
            try { threads.v.emplace_back(work, k); } catch (const std::exception &) { work(k); }
        }
        work(0);
    }
    for (int k = 0; k < 3; ++k)
        if (rcs[k] != DQ_OK) { t_err = errs[k]; return rcs[k]; }
    patch.assign((size_t)bsdiff::kHeaderSize, 0);                                  // Diff.cs:54-70 / :247-252
    bsdiff::write_packed_long(&patch[0], bsdiff::kSignature);
    bsdiff::write_packed_long(&patch[8], (int64_t)z[0].size());
    bsdiff::write_packed_long(&patch[16], (int64_t)z[1].size());
    bsdiff::write_packed_long(&patch[24], m);
    patch.reserve(patch.size() + z[0].size() + z[1].size() + z[2].size());
    for (int k = 0; k < 3; ++k) patch.insert(patch.end(), z[k].begin(), z[k].end());
    return DQ_OK;
}
}  // namespace

int bsdiff_create_host(const uint8_t *old, int64_t n, const uint8_t *nw, int64_t m, int32_t device, std::vector<uint8_t> &patch)
{
    int dev = 0;
    int rc = resolve_device(device, &dev);
    if (rc != DQ_OK) return rc;
    bsdiff::RawStreams raw;
    PatchFramer framer(dev);                              // (after raw: it reads the streams until it is gone)
    const bool follow = !env("DQ_FRAME_AFTER");
    rc = bsdiff_raw(old, n, nw, m, device, raw, follow ? &framer : nullptr);
    if (rc != DQ_OK) return rc;
    return frame_patch(raw, m, dev, patch, &framer);
}

// Patch.Apply (Patch.cs:52-168): host only (dq_bspatch.h)
int bspatch_apply_host(const uint8_t *old, int64_t n, const uint8_t *patch, int64_t plen, uint8_t *out, int64_t cap, int64_t *out_len)
{
    if (n < 0 || plen < 0 || cap < 0 || (n > 0 && !old) || !patch) return fail(DQ_ERR_BAD_ARGS, "bad arguments");
    const int rc = bsdiff::apply_patch(old, n, patch, plen, out, cap, out_len);
    if (rc == bsdiff::kPatchSmallBuffer) return fail(DQ_ERR_BAD_ARGS, "output buffer too small");
    if (rc != bsdiff::kPatchOk) return fail(DQ_ERR_BAD_ARGS, "Corrupt patch");
    return DQ_OK;
}

int match_search_dev_i32(const void *d_old, int64_t n, const void *d_sa, const void *d_new, int64_t m, const int64_t *d_scans,
                         int64_t scan0, int64_t count, int64_t cap, void *d_pos, void *d_len, int32_t device, void *stream)
{
    return match_search_dev<int32_t>(d_old, n, d_sa, d_new, m, d_scans, scan0, count, cap, d_pos, d_len, device, stream);
}
int match_search_dev_i64(const void *d_old, int64_t n, const void *d_sa, const void *d_new, int64_t m, const int64_t *d_scans,
                         int64_t scan0, int64_t count, int64_t cap, void *d_pos, void *d_len, int32_t device, void *stream)
{
    return match_search_dev<int64_t>(d_old, n, d_sa, d_new, m, d_scans, scan0, count, cap, d_pos, d_len, device, stream);
}
int match_search_host_i32(const uint8_t *old, int64_t n, const int32_t *sa, const uint8_t *nw, int64_t m, const int64_t *scans,
                          int64_t scan0, int64_t count, int64_t cap, int32_t *pos, int32_t *len, int32_t device)
{
    return match_search_host<int32_t>(old, n, sa, nw, m, scans, scan0, count, cap, pos, len, device);
}
int match_search_host_i64(const uint8_t *old, int64_t n, const int64_t *sa, const uint8_t *nw, int64_t m, const int64_t *scans,
                          int64_t scan0, int64_t count, int64_t cap, int64_t *pos, int64_t *len, int32_t device)
{
    return match_search_host<int64_t>(old, n, sa, nw, m, scans, scan0, count, cap, pos, len, device);
}

// the raw streams of Diff.Create (dq_bsdiff_scan_i32: what the tests compare with the oracle's restated loop)
int bsdiff_scan_raw(const uint8_t *old, int64_t n, const uint8_t *nw, int64_t m, int32_t device, std::vector<int64_t> &ctrl,
                    std::vector<uint8_t> &diff, std::vector<uint8_t> &extra, int64_t stats[3])
{
    bsdiff::RawStreams raw;
    const int rc = bsdiff_raw(old, n, nw, m, device, raw);
    if (rc != DQ_OK) return rc;
    ctrl.resize(raw.ctrl.size() / 8);
    for (size_t i = 0; i < ctrl.size(); ++i) ctrl[i] = bsdiff::read_packed_long(&raw.ctrl[i * 8]);
    diff.swap(raw.diff); extra.swap(raw.extra);
    stats[0] = raw.searches; stats[1] = raw.windows; stats[2] = raw.exact;
    return DQ_OK;
}

int diff_index_new(const uint8_t *old, int64_t n, int32_t device, const void *d_old, const void *d_sa, void **index_out)
{
    DiffIndex *ix = new DiffIndex();
    const int rc = diff_index_build(old, n, device, d_old, d_sa, /*cached=*/false, ix);
    if (rc != DQ_OK) { diff_index_drop(ix); delete ix; return rc; }
    *index_out = ix;
    return DQ_OK;
}

int diff_index_clone(const void *index, int32_t device, void **index_out)
{
    const DiffIndex *src = static_cast<const DiffIndex *>(index);
    DiffIndex *ix = new DiffIndex();
    const int rc = diff_index_copy(*src, device, ix);
    if (rc != DQ_OK) { diff_index_drop(ix); delete ix; return rc; }
    *index_out = ix;
    return DQ_OK;
}

int diff_index_buffers(const void *index, const void **d_old, const void **d_sa, int64_t *n)
{
    const DiffIndex *ix = static_cast<const DiffIndex *>(index);
    if (d_old) *d_old = ix->d_old;
    if (d_sa) *d_sa = ix->d_sa;
    if (n) *n = ix->n;
    return DQ_OK;
}

int diff_index_diff(const void *index, const uint8_t *nw, int64_t m, std::vector<uint8_t> &patch)
{
    const DiffIndex *ix = static_cast<const DiffIndex *>(index);
    bsdiff::RawStreams raw;
    PatchFramer framer(ix->dev);
    {
        std::lock_guard<std::mutex> one_diff(ctx0(ix->dev).diff_mu);      // scan loops take turns on a device
        const int rc = diff_index_scan(*ix, nw, m, raw, env("DQ_FRAME_AFTER") ? nullptr : &framer);
        if (rc != DQ_OK) return rc;
    }
    return frame_patch(raw, m, ix->dev, patch, &framer);                            // (framing overlaps the next caller's scan loop)
}

void diff_index_delete(void *index)
{
    DiffIndex *ix = static_cast<DiffIndex *>(index);
    diff_index_drop(ix);
    delete ix;
}

}  // namespace dq
