// dq_sufsort_hip.hip -- host runtime + C ABI of libdq_sufsort_hip.so.
//
// Suffix-array construction for byte text on one MI355X (gfx950), prefix doubling on ranks:
//   round 0   byte histogram of the text -> key width kb (3..8 bytes); kb stable LSD digit passes
//             (radix_rank_kernel) over packed words (key << ib | suffix) or (key, suffix) pairs,
//             the first pass building its keys from the text, the last one emitting the SA and
//             (packed) the tie bits; group heads / device-wide scan -> ranks (seg_fused_kernel)
//   few ties  groups of <= 8 sorted by direct text comparison, key extension from the text
//   round r   for the suffixes still tied: key2 = rank of the suffix h bytes further on,
//             sort by (rank, key2), rebucket, h *= 2 (only the tied suffixes are touched;
//             groups of <= 8..32 members are finished in one pass per round), until no group
//             has more than one member.
// DESIGN.md section 2 has the whole map.  The result is the unique suffix array, hence
// bit-identical to the reference's
// LibDivSufSort.Sort() (LibDivSufSort.cs:12-29; order = LibDivSufSortTests.cs:43-59).
//
// This file contains no CPU sorting path: if HIP is unusable the entry points fail.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/dq_sufsort.h"
#include "dq_alpha_code.h"
#include "dq_onesweep.h"
#include "dq_radix.h"
#include "dq_sa_kernels.h"
#include "dq_seg_fused.h"
#include "dq_small.h"
#include "dq_small_groups.h"
#include "dq_mid_groups.h"
#include "dq_runs.h"
#include "dq_ties.h"
#include "dq_isa_pairs.h"
#include "dq_bucket_sort.h"
#include "dq_match_search.h"
#include "dq_pair_chains.h"
#include "dq_bz2.h"
#include "dq_bsdiff.h"
#include "dq_bspatch.h"

namespace {

using namespace dq;

// ------------------------------------------------------------------ errors
thread_local std::string t_err;
thread_local int64_t t_info[3] = {0, 0, 0};

int fail(int code, const char *what, hipError_t e = hipSuccess)
{
    char buf[512];
    if (e != hipSuccess)
        snprintf(buf, sizeof buf, "%s: %s (%d)", what, hipGetErrorString(e), (int)e);
    else
        snprintf(buf, sizeof buf, "%s", what);
    t_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                   \
    do {                                                                                \
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
};
thread_local EnvCache t_env;

const char *env(const char *name)
{
    EnvCache &c = t_env;
    for (int i = 0; i < c.count; ++i)
        if (c.e[i].name == name || strcmp(c.e[i].name, name) == 0) return c.e[i].set ? c.e[i].val.c_str() : nullptr;
    const char *v = getenv(name);
    if (c.depth == 0 || c.count == EnvCache::kMax) return v;          // outside an entry point: nothing is kept
    EnvCache::Entry &x = c.e[c.count++];
    x.name = name; x.set = v != nullptr; x.val = v ? v : "";
    return x.set ? x.val.c_str() : nullptr;
}

struct EnvScope {
    EnvScope() { if (t_env.depth++ == 0) t_env.count = 0; }
    ~EnvScope() { --t_env.depth; }
};

// ------------------------------------------------------------------ profiling
struct KernelStat { int64_t launches = 0; double ms = 0; int64_t elems = 0; int64_t bytes = 0; };
std::mutex g_prof_mu;
KernelStat g_prof[DQ_K_COUNT];
std::atomic<int> g_prof_on{0};

const char *const kKernelNames[DQ_K_COUNT] = {
    "text_hist_kernel", "radix_hist_kernel", "radix_rank_kernel", "seg_fused_kernel",
    "tie_seam_kernel", "tie_collect_kernel", "small_group_finish_kernel", "small_group_round_kernel",
    "isa_update_kernel", "isa_from_pairs_kernel", "key2_from_pairs_kernel", "gather_key2_kernel",
    "gather_text_key_kernel", "isa_from_sa_kernel", "small_sufsort_kernel", "bucket_sort_kernel",
    "match_search_kernel", "pair_chain_kernels", "mid_group_round_kernel", "runlen_kernels"};

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
DeviceState g_dev[kMaxDevices];
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

int init_ctx(DeviceCtx &c, int dev)
{
    HIP_TRY(hipSetDevice(dev));
    if (c.dev == dev) return DQ_OK;
    // c.dev is published only once every resource exists: a failure half way (e.g. pinned memory
    // exhausted) frees what was made and leaves the context unbuilt, so the next call retries
    hipError_t e = hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipHostMalloc((void **)&c.pinned, 8192, hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc((void **)&c.pinned_io, kSmallIoBytes, hipHostMallocDefault);
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
void drop_pending(DeviceCtx &c, hipStream_t st)
{
    (void)hipStreamSynchronize(st);
    for (ProfRec &r : c.pending) {
        if (r.a) c.pool.push_back(r.a);
        if (r.b) c.pool.push_back(r.b);
    }
    c.pending.clear();
}

int ensure_ws(DeviceCtx &c, size_t bytes)
{
    if (c.ws_bytes >= bytes) return DQ_OK;
    if (c.ws) { (void)hipFree(c.ws); c.ws = nullptr; c.ws_bytes = 0; }
    hipError_t e = hipMalloc((void **)&c.ws, bytes);
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

int flush_profile(DeviceCtx &c)
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

constexpr int kSgChain = 8;       // small-group rounds chained without a host round trip (4 -> 8: see DESIGN section 5)
constexpr int64_t kSgShortList = 1 << 20;     // below this many tied suffixes a round is launch-bound

// ------------------------------------------------------------------ workspace carving
inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// radix_rank_kernel tile geometry by list length (RankCfg below): status rows a sort of m entries may need
// (2048-key tiles measured: 64 KiB 16.7 -> 11 us per pass, 256 KiB ~18 -> ~14; break-even at 2^20 entries, where 512 tiles
// make the look-back chain as long as 85 big tiles are slow)
constexpr int64_t kSmallTileMaxM = 1ll << 20;
inline bool small_tiles(int64_t m) { return m <= kSmallTileMaxM; }
inline size_t status_tiles(size_t m)
{
    const size_t small = (m < (size_t)kSmallTileMaxM ? m : (size_t)kSmallTileMaxM) / 2048;
    return std::max(m / 8192, small) + 2;
}

template <typename IdxT>
struct Workspace {
    uint8_t *text;
    uint64_t *K0, *K1;
    IdxT *Va, *Vb, *ISA, *SAbuf;
    int64_t *bkt_bounds;        // tile bounds of the bucketed round 0 (dq_bucket_sort.h)
    int64_t *totals;            // [0] active count, [1] sticky look-back timeout flag
    SmallGroupCounters *sg_ctr; // one per chained small-group round
    uint32_t *hist_partial;     // [kHistBlocks][8][256]
    uint16_t *codetab;          // [256] codewords of the coded round 0 (dq_alpha_code.h)
    uint32_t *pc_tiles;         // per-tile counts / prefix sums of the pair-chain phase (dq_pair_chains.h)
    uint32_t *RL;               // run lengths of the text (dq_runs.h; int32 indices only)
    uint32_t *run_lead, *run_carry;   // per 4096-byte chunk
    uint8_t *run_link;
    int64_t *digit_offset;      // [8][256]
    int64_t *bytehist;          // [256]
    char *ctl_status;           // per digit pass: OnesweepCtl (256 B) + the tiles' status words
    size_t ctl_status_bytes;
    size_t ctl_status_stride;   // bytes per pass (set by prepare_status)
    char *seg_status;           // SegCtl (256 B) followed by 3 x ntiles status words
    size_t seg_status_bytes;
    size_t bytes;
};

template <typename IdxT>
Workspace<IdxT> carve(char *base, int64_t n, bool with_sa)
{
    Workspace<IdxT> w{};
    size_t off = 0;
    auto take = [&](size_t b) { char *p = base ? base + off : nullptr; off += align_up(b); return p; };
    const size_t un = (size_t)n;
    w.text = (uint8_t *)take(un + 64);
    // (+2: the lists of a small-group round start their L region on an even entry, see sg_half())
    w.K0 = (uint64_t *)take((un + 2) * 8);
    w.K1 = (uint64_t *)take((un + 2) * 8);
    w.Va = (IdxT *)take((un + 2) * sizeof(IdxT));
    w.Vb = (IdxT *)take((un + 2) * sizeof(IdxT));
    w.ISA = (IdxT *)take(un * sizeof(IdxT));
    w.SAbuf = with_sa ? (IdxT *)take(un * sizeof(IdxT)) : nullptr;
    w.bkt_bounds = (int64_t *)take((un / 4096 + 4) * 8);
    w.totals = (int64_t *)take(64);
    w.sg_ctr = (SmallGroupCounters *)take(kSgChain * sizeof(SmallGroupCounters));
    w.hist_partial = (uint32_t *)take((size_t)kHistBlocks * kMaxPasses * kRadixSize * 4);
    w.digit_offset = (int64_t *)take((size_t)kMaxPasses * kRadixSize * 8);
    w.codetab = (uint16_t *)take(512);
    w.pc_tiles = (uint32_t *)take((un / 2048 + 4) * 8);
    if (sizeof(IdxT) == 4) {
        const size_t nchunks = un / kRunChunk + 2;
        w.RL = (uint32_t *)take(un * 4);
        w.run_lead = (uint32_t *)take(nchunks * 4);
        w.run_carry = (uint32_t *)take(nchunks * 4);
        w.run_link = (uint8_t *)take(nchunks);
    }
    w.bytehist = (int64_t *)take((size_t)(kRadixSize + 16) * 8);       // + the 8 k-gram sample counters + the long-run flag
    // smallest tile is 8192 keys (2048 for lists of up to kSmallTileMaxM entries, see RankCfg); 8-byte status words
    // once a list reaches 2^30 entries
    w.ctl_status_bytes = (size_t)kMaxPasses * align_up(256 + status_tiles(un) * kRadixSize * (un >= (1ull << 30) ? 8 : 4));
    w.ctl_status = take(w.ctl_status_bytes);
    w.seg_status_bytes = 256 + 3 * (un / kSegFusedTile + 2) * 8;
    w.seg_status = take(w.seg_status_bytes);
    w.bytes = off;
    return w;
}

inline int bit_length(uint64_t x) { return x == 0 ? 1 : 64 - __builtin_clzll(x); }

// ------------------------------------------------------------------ onesweep driver
// Tile geometry of radix_rank_kernel per (index type, pass kind), from the kbench sweep
// (tools/kbench, 64 Mi keys, random digits): 512 threads; packed-word passes 24 keys/thread
// (12288-key tiles, ~48-key runs per digit), LDS match tables; pair passes 20 keys/thread,
// ballot match; the tile is staged through LDS in 2 position ranges (half the LDS footprint).
//
// Lists of up to kSmallTileMaxM entries are launch-bound, and what a pass costs there is the LIFE of one tile (load,
// ranking, exchange, look-back, stores: ~17-22 us for the big tiles whatever their number -- 64 KiB ... 1 MiB of
// text spend half their sort in these passes): they take 2048-key tiles (256 threads x 8), several per CU at once.
template <typename IdxT, int kMode, bool kSmall = false> struct RankCfg {
    static constexpr bool kWords = (kMode == kTextPacked || kMode == kKeys || kMode == kKeysLast || kMode == kKeysLastTies);
    // (a 1024-thread tile for the tie-recording last pass, whose runs are 4-byte SA entries, measured +18 %)
    static constexpr int kThreads = kSmall ? 256 : 512;
    static constexpr int kItems = kSmall ? 8 : kWords ? 24 : (sizeof(IdxT) == 4 ? 20 : 16);
    static constexpr int kMinWaves = 2;
    static constexpr int kRounds = 2;
    // LDS match tables beat 8 ballots on near-uniform digits (words: -6%), but equal digits in a wave are
    // same-address LDS atomics: pair passes run on text-like (skewed) data and keep the ballots
    static constexpr bool kLdsMatch = kWords;
    // the first pass of a sort has no earlier order to keep: atomic cursors instead of the look-back
    static constexpr bool kAtomicBase = (kMode == kTextPacked || kMode == kText);
};

// Zero the look-back state (ticket + status words) of ALL digit passes of one sort with a single
// memset, so the passes run back to back.
template <typename IdxT>
int prepare_status(Launcher &L, Workspace<IdxT> &w, int64_t m, int passes, int from = 0)
{
    const size_t word = m < (1ll << 30) ? 4 : 8;
    const size_t stride = align_up(256 + status_tiles((size_t)m) * kRadixSize * word);
    if ((size_t)passes * stride > w.ctl_status_bytes) return fail(DQ_ERR_HIP, "status buffer too small");
    w.ctl_status_stride = stride;
    if (passes > from) HIP_TRY(hipMemsetAsync(w.ctl_status + (size_t)from * stride, 0, (size_t)(passes - from) * stride, L.st));
    return DQ_OK;
}

template <typename IdxT, typename StatusT, int kMode, bool kCoded = false, bool kSmall = false>
int launch_rank_pass(Launcher &L, Workspace<IdxT> &w, const uint64_t *kin, const IdxT *vin,
                     uint64_t *kout, IdxT *vout, int64_t m, int pass, int kb, int ib,
                     uint32_t *ebits = nullptr, uint64_t *seam_tab = nullptr, int shift_override = -1,
                     int keybits = 0)
{
    using Cfg = RankCfg<IdxT, kMode, kSmall>;
    constexpr int kItems = Cfg::kItems;
    constexpr int kThreads = Cfg::kThreads;
    constexpr int kTileN = kThreads * kItems;
    const int64_t ntiles = (m + kTileN - 1) / kTileN;
    const int64_t wb = (int64_t)sizeof(IdxT);
    // the status area of every pass of this sort was zeroed by prepare_status()
    char *area = w.ctl_status + (size_t)pass * w.ctl_status_stride;
    OnesweepCtl *ctl = reinterpret_cast<OnesweepCtl *>(area);
    StatusT *status = reinterpret_cast<StatusT *>(area + 256);
    if (256 + (size_t)ntiles * kRadixSize * sizeof(StatusT) > w.ctl_status_stride)
        return fail(DQ_ERR_HIP, "status buffer too small");
    // algorithmic bytes per element: what the pass must read + write
    const int64_t alg = kMode == kPairs ? 2 * (8 + wb) : kMode == kText ? 1 + 8 + wb
                      : kMode == kTextPacked ? 1 + 8 : kMode == kKeys ? 16 : kMode == kKeysLastTies ? 8 + wb : 16 + wb;
    // the tie-recording pass also writes 1 bit per element and 2 words per (tile, digit)
    const int64_t alg_extra = kMode == kKeysLastTies ? m / 8 + ntiles * kRadixSize * 16 : 0;
    LAUNCH(L, DQ_K_RADIX_RANK, m, m * alg + alg_extra,
           hipLaunchKernelGGL((radix_rank_kernel<IdxT, StatusT, kItems, kMode, Cfg::kMinWaves, kThreads,
                                                 false, Cfg::kLdsMatch, Cfg::kRounds, Cfg::kAtomicBase, kCoded>),
                              dim3((unsigned)ntiles), dim3(kThreads), 0, L.st, kin, vin, kout, vout, m,
                              shift_override >= 0 ? shift_override : pass * kRadixBits + ib,
                              keybits > 0 ? keybits : 8 * kb, ib,
                              (const int64_t *)(w.digit_offset + pass * kRadixSize), status, ctl, w.totals + 1,
                              ebits, seam_tab, (const uint16_t *)w.codetab));
    return DQ_OK;
}

template <typename IdxT, int kMode, bool kCoded = false>
int rank_pass(Launcher &L, Workspace<IdxT> &w, const uint64_t *kin, const IdxT *vin, uint64_t *kout,
              IdxT *vout, int64_t m, int pass, int kb, int ib = 0, uint32_t *ebits = nullptr,
              uint64_t *seam_tab = nullptr, int shift_override = -1, int keybits = 0)
{
    if (small_tiles(m))
        return launch_rank_pass<IdxT, uint32_t, kMode, kCoded, true>(L, w, kin, vin, kout, vout, m, pass, kb, ib, ebits,
                                                                     seam_tab, shift_override, keybits);
    if (m < (1ll << 30))
        return launch_rank_pass<IdxT, uint32_t, kMode, kCoded>(L, w, kin, vin, kout, vout, m, pass, kb, ib, ebits,
                                                               seam_tab, shift_override, keybits);
    return launch_rank_pass<IdxT, uint64_t, kMode, kCoded>(L, w, kin, vin, kout, vout, m, pass, kb, ib, ebits, seam_tab,
                                                           shift_override, keybits);
}

template <int kPasses>
void launch_hist(hipStream_t st, int blocks, const uint64_t *keys, int64_t m, uint32_t *partial)
{
    hipLaunchKernelGGL(radix_hist_kernel<kPasses>, dim3(blocks), dim3(kHistThreads), 0, st, keys, m, partial);
}

// generic pairs: all digit histograms in one read, then one radix_rank_kernel per digit
template <typename IdxT>
int onesweep_sort_pairs(Launcher &L, Workspace<IdxT> &w, uint64_t *K[2], IdxT *V[2], int64_t m,
                        int total_bits, int &cur)
{
    const int passes = (total_bits + kRadixBits - 1) / kRadixBits;
    const int blocks = (int)std::min<int64_t>(kHistBlocks, ((m >> 1) + kHistThreads - 1) / kHistThreads + 1);
    int rc = L.begin(DQ_K_RADIX_HIST, m, m * 8);
    if (rc != DQ_OK) return rc;
    switch (passes) {
        case 1: launch_hist<1>(L.st, blocks, K[cur], m, w.hist_partial); break;
        case 2: launch_hist<2>(L.st, blocks, K[cur], m, w.hist_partial); break;
        case 3: launch_hist<3>(L.st, blocks, K[cur], m, w.hist_partial); break;
        case 4: launch_hist<4>(L.st, blocks, K[cur], m, w.hist_partial); break;
        case 5: launch_hist<5>(L.st, blocks, K[cur], m, w.hist_partial); break;
        case 6: launch_hist<6>(L.st, blocks, K[cur], m, w.hist_partial); break;
        case 7: launch_hist<7>(L.st, blocks, K[cur], m, w.hist_partial); break;
        default: launch_hist<8>(L.st, blocks, K[cur], m, w.hist_partial); break;
    }
    hipLaunchKernelGGL(radix_hist_scan_kernel, dim3(passes), dim3(kHistScanThreads), 0, L.st,
                       (const uint32_t *)w.hist_partial, blocks, w.digit_offset);
    HIP_TRY(hipGetLastError());
    rc = L.end();
    if (rc != DQ_OK) return rc;
    rc = prepare_status<IdxT>(L, w, m, passes);
    if (rc != DQ_OK) return rc;
    for (int p = 0; p < passes; ++p) {
        rc = rank_pass<IdxT, kPairs>(L, w, K[cur], V[cur], K[cur ^ 1], V[cur ^ 1], m, p, 8);
        if (rc != DQ_OK) return rc;
        cur ^= 1;
    }
    return DQ_OK;
}

// Number of leading text bytes worth sorting in round 0: enough bits, under an order-0 model
// of the text, to make ties among n suffixes rare (~n/1000); text-like inputs get all 8.
// If (almost) that many key bytes fit into one 64-bit word next to the suffix index
// (ib = bits of n-1), round 0 sorts PACKED words (key << ib | suffix): 16 B per element per
// pass instead of 24 and no value array; the few extra ties go to the sparse finishing path.
// pair chains (dq_pair_chains.h): tried when a doubling round left > 60% of its list tied, at most this often per sort
constexpr int kPairChainTries = 3;
// Lists shorter than this keep doubling: since the LDS class finishes a round for nearly every group in one cheap pass
// (a round of a 1 MiB text: ~20 us + its rank updates), a chain phase (~20 launches) costs more than the rounds it
// saves -- 64 KiB ... 16 MiB of text are 8-35 % faster without (1 MiB: 1.43 -> 0.94 ms), 64 MiB tar-like and 256 MiB
// enwik-style (lists of 2e7 ... 4e7 entries) 9-10 % slower.  DQ_PAIR_CHAINS=1/2 forces the phases on lists of any length.
constexpr int64_t kPairChainMinM = 1 << 23;

// coded round 0 (dq_alpha_code.h): from this size on, and only if a byte costs at most this many bits on average
constexpr int64_t kCodedMinN = 8ll << 20;
constexpr double kCodedMaxAvgLen = 5.8;       // >= 11 characters per key (a 205-symbol Python source tree: 6.17, no gain)

void choose_key_bytes(const int64_t *bytehist, const int64_t *kgram_coll, int64_t n, int *kb_out, bool *packed_out)
{
    double h0 = 0;
    for (int b = 0; b < 256; ++b) {
        if (bytehist[b] > 0) {
            const double p = (double)bytehist[b] / (double)n;
            h0 -= p * std::log2(p);
        }
    }
    const double need = std::log2((double)std::max<int64_t>(n, 2)) + 10.0;
    int kb = 8;
    if (h0 >= 0.25) kb = std::min(8, std::max(3, (int)std::ceil(need / h0)));
    const int ib = bit_length((uint64_t)(n - 1));
    const int fit = (64 - ib) / 8;
    // packed if the bytes that fit still leave at most ~1/8 of the suffixes tied
    bool packed = fit >= 2 && (kb <= fit || (double)fit * h0 >= std::log2((double)std::max<int64_t>(n, 2)) + 3.0);
    // Veto from the k-gram sample: an order-0 model cannot see repetition.  With S sampled suffixes and C
    // adjacent sorted pairs agreeing on L bytes, a suffix expects about n * 2C / S^2 twins under an L-byte
    // key.  Repetitive (text-like) data takes the 8-byte pair path, which is built for many ties.
    // (It takes 1 sample in 16 with a twin among the samples: a few repeated regions in otherwise random data
    // are what the packed sort and its sparse finishing are good at.)
    if (kgram_coll) {
        const int L = packed ? std::min(kb, fit) : kb;
        const int64_t C = kgram_coll[std::max(L, 1) - 1];
        const double twins = (double)n * 2.0 * (double)C / ((double)kKgramSamples * (double)kKgramSamples);
        if (C >= kKgramSamples / 16 && twins > 0.25) { packed = false; kb = 8; }
    }
    if (const char *v = env("DQ_PACKED")) packed = atoi(v) != 0 && fit >= 2;
    if (packed) kb = std::min(kb, fit);
    *kb_out = kb;
    *packed_out = packed;
}

// round 0, step 1: byte histogram of the text -> key width kb -> per-digit offsets
template <typename IdxT>
int onesweep_sort_text_prepare(Launcher &L, DeviceCtx &c, Workspace<IdxT> &w, int64_t n, int *kb_out,
                               bool *packed_out, bool *coded_out)
{
    *coded_out = false;
    const int blocks = (int)std::min<int64_t>(kHistBlocks, ((n >> 4) + kBlock - 1) / kBlock + 1);
    HIP_TRY(hipMemsetAsync(w.bytehist, 0, (256 + 10) * 8, L.st));
    // (+1 workgroup: the k-gram sample, whose 8 counters sit right behind the byte histogram: one readback)
    LAUNCH(L, DQ_K_TEXT_HIST, n, n,
           hipLaunchKernelGGL(text_hist_kernel, dim3(blocks + 1), dim3(kBlock), 0, L.st,
                              (const uint8_t *)w.text, n, reinterpret_cast<unsigned long long *>(w.bytehist),
                              reinterpret_cast<unsigned long long *>(w.bytehist + 256)));
    int kb = 8;
    bool packed = false;
    HIP_TRY(hipMemcpyAsync(c.pinned, w.bytehist, (256 + 10) * 8, hipMemcpyDeviceToHost, L.st));
    HIP_TRY(hipEventRecord(c.readback, L.st));
    // While the host waits for the histogram and picks the key width, the device zeroes what the passes
    // need whatever that choice is: the look-back state of the first 3 passes (all the bucketed round 0 runs;
    // 33 MB per pass at 256 MiB) and the tie bits; the other passes' state once kb is known.
    constexpr int kEarlyPasses = 3;
    int rc = prepare_status<IdxT>(L, w, n, kEarlyPasses);
    if (rc != DQ_OK) return rc;
    if (n >= (1 << 16)) HIP_TRY(hipMemsetAsync(w.Vb, 0, (size_t)((n + 63) / 64 + 1) * 8, L.st));
    HIP_TRY(hipEventSynchronize(c.readback));
    choose_key_bytes(c.pinned, n >= kKgramSamples * 8 ? c.pinned + 256 : nullptr, n, &kb, &packed);
    if (const char *force = env("DQ_KEY_BYTES")) {
        kb = std::min(8, std::max(1, atoi(force)));
        const int fit = (64 - bit_length((uint64_t)(n - 1))) / 8;
        if (kb > fit || kb < 2) packed = false;
    }
    *kb_out = kb;
    *packed_out = packed;
    rc = prepare_status<IdxT>(L, w, n, kb, kEarlyPasses);
    if (rc != DQ_OK) return rc;
    // Text-like input on the 8-byte pair path: the 64 key bits hold the codewords of an alphabetic prefix code
    // instead of 8 raw bytes (dq_alpha_code.h) when that makes the key reach at least ~10 characters on average.
    // The keys' digits are then no longer text bytes: their histograms take one more read of the text.
    bool coded = !packed && kb == 8 && n >= kCodedMinN;
    if (coded) {
        // the code is built on the host while the device waits (0.2 ms for 73 symbols, 1-2 ms for 200+): only where
        // it can pay -- the expected codeword length is at least the order-0 entropy, and texts with more than
        // 128 symbols must be large enough to hide the construction
        int sigma = 0;
        double h0 = 0;
        for (int b = 0; b < 256; ++b) {
            if (c.pinned[b] > 0) { ++sigma; const double p = (double)c.pinned[b] / (double)n; h0 -= p * std::log2(p); }
        }
        coded = h0 <= kCodedMaxAvgLen - 0.25 && (sigma <= 128 || n >= 2 * kCodedMinN);
    }
    if (const char *v = env("DQ_CODED")) coded = atoi(v) != 0 && !packed && kb == 8 && n >= 64;
    if (coded) {
        AlphaCode code;
        coded = build_alpha_code(c.pinned, &code) && (code.avg_len <= kCodedMaxAvgLen || env("DQ_CODED"));
        if (coded) {
            uint16_t *stage = reinterpret_cast<uint16_t *>(c.pinned + 512);          // the upload half of the pinned area
            memcpy(stage, code.tab, sizeof(code.tab));
            HIP_TRY(hipMemcpyAsync(w.codetab, stage, sizeof(code.tab), hipMemcpyHostToDevice, L.st));
            const int hblocks = (int)std::min<int64_t>(kHistBlocks, ((n >> 2) + kHistThreads - 1) / kHistThreads + 1);
            int rc2 = L.begin(DQ_K_RADIX_HIST, n, n);
            if (rc2 != DQ_OK) return rc2;
            hipLaunchKernelGGL(text_coded_hist_kernel, dim3(hblocks), dim3(kHistThreads), 0, L.st,
                               reinterpret_cast<const uint32_t *>(w.text), n, (const uint16_t *)w.codetab, w.hist_partial);
            hipLaunchKernelGGL(radix_hist_scan_kernel, dim3(kMaxPasses), dim3(kHistScanThreads), 0, L.st,
                               (const uint32_t *)w.hist_partial, hblocks, w.digit_offset);
            HIP_TRY(hipGetLastError());
            rc2 = L.end();
            if (rc2 != DQ_OK) return rc2;
            if (env("DQ_TRACE")) fprintf(stderr, "[dq] coded round 0: %d symbols, %.2f bits per byte\n", code.sigma, code.avg_len);
            *coded_out = true;
            return DQ_OK;
        }
    }
    hipLaunchKernelGGL(text_digit_offsets_kernel, dim3(kb), dim3(kBlock), 0, L.st,
                       (const int64_t *)w.bytehist, (const uint8_t *)w.text, n, kb, w.digit_offset);
    HIP_TRY(hipGetLastError());
    return DQ_OK;
}

// round 0, step 2: kb digit passes; pass 0 builds its keys straight from the text and writes
// buffer 1, pass p writes buffer (p+1)&1.  Packed: words only, the last pass also emits the SA.
template <typename IdxT>
int onesweep_sort_text_passes(Launcher &L, Workspace<IdxT> &w, int64_t n, uint64_t *K[2], IdxT *V[2],
                              int kb, bool packed, IdxT *d_sa, int &cur, uint32_t *ebits = nullptr,
                              uint64_t *seam_tab = nullptr, bool coded = false)
{
    const uint64_t *text64 = reinterpret_cast<const uint64_t *>(w.text);
    int rc = DQ_OK;                 // look-back state zeroed by onesweep_sort_text_prepare()
    if (packed) {
        const int ib = bit_length((uint64_t)(n - 1));
        rc = rank_pass<IdxT, kTextPacked>(L, w, text64, (const IdxT *)nullptr, K[1], (IdxT *)nullptr, n, 0, kb, ib);
        if (rc != DQ_OK) return rc;
        cur = 1;
        for (int p = 1; p < kb; ++p) {
            if (p == kb - 1 && ebits)
                rc = rank_pass<IdxT, kKeysLastTies>(L, w, K[cur], (const IdxT *)nullptr, (uint64_t *)nullptr, d_sa, n, p,
                                                    kb, ib, ebits, seam_tab);
            else if (p == kb - 1)
                rc = rank_pass<IdxT, kKeysLast>(L, w, K[cur], (const IdxT *)nullptr, K[cur ^ 1], d_sa, n, p, kb, ib);
            else
                rc = rank_pass<IdxT, kKeys>(L, w, K[cur], (const IdxT *)nullptr, K[cur ^ 1], (IdxT *)nullptr, n, p, kb, ib);
            if (rc != DQ_OK) return rc;
            cur ^= 1;
        }
        return DQ_OK;
    }
    rc = coded ? rank_pass<IdxT, kText, true>(L, w, text64, (const IdxT *)nullptr, K[1], V[1], n, 0, kb)
               : rank_pass<IdxT, kText>(L, w, text64, (const IdxT *)nullptr, K[1], V[1], n, 0, kb);
    if (rc != DQ_OK) return rc;
    cur = 1;
    for (int p = 1; p < kb; ++p) {
        rc = rank_pass<IdxT, kPairs>(L, w, K[cur], V[cur], K[cur ^ 1], V[cur ^ 1], n, p, kb);
        if (rc != DQ_OK) return rc;
        cur ^= 1;
    }
    return DQ_OK;
}

// After a packed sort whose last pass ran in kKeysLastTies mode: decide the cross-tile pairs, then
// turn the tie bits into the list of tied suffixes.  *overflow: a run of equal keys too long for
// the per-thread walk was met and the caller must take the general rebucket pass instead.
template <typename IdxT>
int collect_ties(Launcher &L, DeviceCtx &c, Workspace<IdxT> &w, int64_t n, int kb, int ib, uint32_t *ebits,
                 const uint64_t *seam_tab, const IdxT *d_sa, uint64_t *act_rank, IdxT *act_suf, int64_t *count,
                 bool *overflow, int64_t fin_cap, uint64_t *fin_rank, IdxT *fin_suf, int64_t *fin_left,
                 bool seams = true, int64_t h_fin = -1)
{
    using Cfg = RankCfg<IdxT, kKeysLastTies>;
    using CfgS = RankCfg<IdxT, kKeysLastTies, true>;
    const int64_t tile_keys = small_tiles(n) ? CfgS::kThreads * CfgS::kItems : Cfg::kThreads * Cfg::kItems;
    const int64_t ntiles = (n + tile_keys - 1) / tile_keys;
    const int pass = kb - 1;
    char *area = w.ctl_status + (size_t)pass * w.ctl_status_stride;
    const int64_t *dofs = w.digit_offset + pass * kRadixSize;
    const int64_t nwords = (n + 63) / 64;
    TieCounters *ctr = reinterpret_cast<TieCounters *>(w.totals + 6);
    const int64_t wb = (int64_t)sizeof(IdxT);
    // (without seams the producer -- bucket_sort_kernel -- has already used ctr->overflow: zeroed by the caller)
    if (seams) HIP_TRY(hipMemsetAsync(ctr, 0, sizeof(TieCounters), L.st));
    const unsigned sg = (unsigned)((ntiles * kRadixSize + kBlock - 1) / kBlock);
    if (!seams) {
    } else if (n < (1ll << 30)) {
        LAUNCH(L, DQ_K_TIE_SEAM, ntiles * kRadixSize, ntiles * kRadixSize * (16 + (int64_t)sizeof(uint64_t)),
               hipLaunchKernelGGL(tie_seam_kernel<uint32_t>, dim3(sg), dim3(kBlock), 0, L.st, seam_tab, ntiles, ib, dofs,
                                  reinterpret_cast<const uint32_t *>(area + 256), ebits));
    } else {
        LAUNCH(L, DQ_K_TIE_SEAM, ntiles * kRadixSize, ntiles * kRadixSize * (16 + (int64_t)sizeof(uint64_t)),
               hipLaunchKernelGGL(tie_seam_kernel<uint64_t>, dim3(sg), dim3(kBlock), 0, L.st, seam_tab, ntiles, ib, dofs,
                                  reinterpret_cast<const uint64_t *>(area + 256), ebits));
    }
    LAUNCH(L, DQ_K_TIE_COLLECT, n, n / 8,
           hipLaunchKernelGGL(tie_collect_kernel<IdxT>, dim3((unsigned)((nwords + kTieThreads - 1) / kTieThreads)),
                              dim3(kTieThreads), 0, L.st, reinterpret_cast<const uint64_t *>(ebits), nwords, n, d_sa,
                              act_rank, act_suf, ctr));
    // Few ties are expected here, so the direct-comparison finisher is launched right away on the list
    // whose length is still on the device (capacity fin_cap), saving a host round trip; its result is
    // used only if the list fits and the sparse path is taken.
    unsigned long long *left_over = reinterpret_cast<unsigned long long *>(w.totals + 3);     // zero since run()
    if (fin_cap > 0) {
        LAUNCH(L, DQ_K_SMALL_FINISH, fin_cap, 0,
               hipLaunchKernelGGL((small_group_finish_kernel<IdxT, 8, 32>),
                                  dim3((unsigned)std::min<int64_t>((fin_cap + kFinishThreads - 1) / kFinishThreads, 256 * 16)),
                                  dim3(kFinishThreads), 0, L.st, (const uint64_t *)act_rank, (const IdxT *)act_suf, (const uint8_t *)w.text,
                                  fin_cap, n, h_fin >= 0 ? h_fin : (int64_t)kb, const_cast<IdxT *>(d_sa), fin_rank, fin_suf, left_over,
                                  (const unsigned long long *)&ctr->count));
    }
    HIP_TRY(hipMemcpyAsync(c.pinned, w.totals, 64, hipMemcpyDeviceToHost, L.st));     // [1] sticky flag, [3] leftovers, [6..7] counters
    HIP_TRY(hipStreamSynchronize(L.st));
    if (c.pinned[1] != 0) return fail(DQ_ERR_HIP, "radix look-back timed out (device spin bound hit)");
    *count = c.pinned[6];
    *overflow = c.pinned[7] != 0;
    if (*overflow && env("DQ_TRACE")) fprintf(stderr, "[dq] tie / bucket overflow flags: %lld\n", (long long)c.pinned[7]);
    *fin_left = c.pinned[3];
    (void)wb;
    return DQ_OK;
}

// Rebucket a list sorted by (composite) key: group heads, device-wide scan, SA / ISA
// scatter, compaction of the still-tied suffixes into (act_rank, act_suf); *active_out = their
// number.  Engine 1: one fused single-pass kernel; engine 0: the legacy three kernels.
// kInitial never writes ISA (it is built later, and only on the dense path).
template <typename IdxT, bool kInitial, bool kWriteSA, bool kWriteISA>
int rebucket(Launcher &L, DeviceCtx &c, Workspace<IdxT> &w, const uint64_t *keys, const IdxT *vals,
             int64_t m, int kbits, int kshift, IdxT *SA, uint64_t *act_rank, IdxT *act_suf,
             int64_t *active_out, int rank_from_isa = 0)
{
    const int64_t wb = (int64_t)sizeof(IdxT);
    const int64_t ntiles = (m + kSegFusedTile - 1) / kSegFusedTile;
    const size_t need = 256 + (size_t)3 * ntiles * 8;
    if (need > w.seg_status_bytes) return fail(DQ_ERR_HIP, "seg status buffer too small");
    HIP_TRY(hipMemsetAsync(w.seg_status, 0, need, L.st));
    LAUNCH(L, DQ_K_SEG_FUSED, m, m * (8 + (kWriteSA ? 2 * wb : 0) + (kWriteISA ? wb : 0)),
           hipLaunchKernelGGL((seg_fused_kernel<IdxT, kInitial, kWriteSA, kWriteISA>),
                              dim3((unsigned)ntiles), dim3(kSegThreads), 0, L.st, keys, vals, m, kbits, kshift, SA, w.ISA, act_rank,
                              act_suf, reinterpret_cast<uint64_t *>(w.seg_status + 256), ntiles,
                              reinterpret_cast<SegCtl *>(w.seg_status), w.totals, w.totals + 1, rank_from_isa));
    HIP_TRY(hipMemcpyAsync(c.pinned, w.totals, 16, hipMemcpyDeviceToHost, L.st));
    HIP_TRY(hipStreamSynchronize(L.st));
    *active_out = c.pinned[0];
    if (c.pinned[1] != 0) return fail(DQ_ERR_HIP, "device look-back timed out (spin bound hit)");
    return DQ_OK;
}

// ------------------------------------------------------------------ the suffix sorter
// One sort = one SuffixSorter.  State that survives between phases: the list of still-tied
// suffixes X = (Kr[rcur], Vr[rcur])[0, m) as (group rank, suffix) with the members of a group
// adjacent, and h = bytes already compared.
template <typename IdxT>
struct SuffixSorter {
    DeviceCtx &c;
    hipStream_t st;
    Workspace<IdxT> &w;        // w.text holds the padded text
    int64_t n;
    IdxT *d_sa;                // n entries on the device
    Launcher L;

    static constexpr int64_t wb = (int64_t)sizeof(IdxT);
    uint64_t *Kr[2] = {nullptr, nullptr};
    IdxT *Vr[2] = {nullptr, nullptr};
    int rcur = 0;
    int64_t m = 0, h = 0;
    int rbits = 0;
    // the list already holds composite keys (rank << kbits | key2) for the next doubling round
    bool keys_ready = false;
    // the list came out of the suffix-binned words in TEXT order (build_isa_binned): the members of a group are not
    // adjacent until a radix round has sorted it, so no small-group / LDS-class round and no pair chains before that
    bool list_ungrouped = false;
    // the last small-group round sent nothing to the radix list: every group has <= small_cap members
    bool only_small_groups = false;
    int small_cap = kSgMaxG;
    // the finisher already ran (speculatively, right after the tie bits were collected)
    bool fin_done = false;
    int64_t fin_cap = 0, fin_left = 0;
    // round 0 was bucketed: however many suffixes are tied, they are tied shallowly (random-like text)
    bool shallow_ties = false;
    // runs of one byte (dq_runs.h): text_hist_kernel saw a run of >= 64 equal bytes; the doubling rounds then order
    // the suffixes inside runs by the run's own structure (w.RL) instead of log2(run length) rounds
    bool runs_wanted = false, runs_on = false;
    int run_order = 0;                  // 1 while the run-order round is being launched
    const uint32_t *rl() const { return runs_on ? w.RL : nullptr; }
    // the first round's list carries its ranks as 32-bit values here (build_isa_binned), not in Kr[rcur]
    const uint32_t *first_rank32 = nullptr;
    // largest group the LDS class finishes: 0 = off; DQ_MID_GROUPS = 0 | 256 | 512 | 1024 forces it.  The walk over a
    // group costs its members ~group size each, the radix passes cost launches: 512 on long lists (256 MiB of
    // enwik-style text: 31.2 ms, 32.3 with 1024, 33.9 without the class), 1024 on the launch-bound short ones
    // (16 MiB: 4.1 - 4.3 ms against 5.2; 64 KiB: 0.56 against 0.76).
    static int mid_group_cap(int64_t list_len)
    {
        if (const char *v = env("DQ_MID_GROUPS")) {
            const int g = atoi(v);
            return g >= 1024 ? 1024 : g >= 512 ? 512 : g >= 256 ? 256 : 0;
        }
        return list_len >= kSgShortList ? 512 : 1024;
    }

    SuffixSorter(DeviceCtx &c_, hipStream_t st_, Workspace<IdxT> &w_, int64_t n_, IdxT *sa_)
        : c(c_), st(st_), w(w_), n(n_), d_sa(sa_), L{c_, st_, g_prof_on.load()} {}

    static unsigned grid_for(int64_t items)
    {
        return (unsigned)std::min<int64_t>((items + kBlock - 1) / kBlock, 256 * 16);
    }

    int sort_pairs(uint64_t *K[2], IdxT *V[2], int64_t cnt, int bits, int &cur)
    {
        return onesweep_sort_pairs<IdxT>(L, w, K, V, cnt, bits, cur);
    }

    // ISA[SA[p]] = p for everybody, then the tied suffixes get their group rank
    int build_isa(const uint64_t *rank, const IdxT *suf, int64_t cnt)
    {
        LAUNCH(L, DQ_K_ISA_FROM_SA, n, n * 3 * wb,
               hipLaunchKernelGGL(isa_from_sa_kernel<IdxT>, dim3(grid_for(n)), dim3(kBlock), 0, st,
                                  (const IdxT *)d_sa, w.ISA, n);
               hipLaunchKernelGGL(isa_scatter_kernel<IdxT>, dim3(grid_for(cnt)), dim3(kBlock), 0, st, rank, suf,
                                  w.ISA, cnt));
        return DQ_OK;
    }

    // ---- dense inputs: first ISA + first key2 gather through suffix-binned words (dq_isa_pairs.h).
    //      keys = the sorted round-0 keys (buffer P1), P0 = the other key buffer (free).  On return the
    //      tied list is (P1, Va) and m its length.
    bool uses_small_round(int64_t mm) const
    {
        return !env("DQ_NO_SMALL") && mm * 2 <= n && n < (1ll << 32);
    }

    int build_isa_binned(uint64_t *keys, uint64_t *P0, int kb, int kshift0)
    {
        const int ib = bit_length((uint64_t)(n - 1));
        const int64_t ntiles = (n + kSegFusedTile - 1) / kSegFusedTile;
        const size_t need = 256 + (size_t)3 * ntiles * 8;
        if (need > w.seg_status_bytes) return fail(DQ_ERR_HIP, "seg status buffer too small");
        HIP_TRY(hipMemsetAsync(w.seg_status, 0, need, st));
        // The tied suffixes are also listed group by group (32-bit ranks in the idle Vb, suffixes in Va): if they
        // are at most n/2, the first doubling round is a small-group round on that list and only the groups of
        // more than 8 go through the radix passes.
        // (32-bit ranks: every 64-bit buffer is busy until the words have been binned.  They go to the run-length buffer
        // when that is idle -- the first round's kernel reads them there -- else to Vb, to be widened into a key buffer)
        const bool rank32_direct = sizeof(IdxT) == 4 && !runs_wanted && mid_group_cap(n) > 0 && !env("DQ_WIDEN_RANKS");
        uint32_t *list_rank = (uses_small_round(0) && !env("DQ_NO_FIRST_SMALL"))
                                  ? (rank32_direct ? w.RL : reinterpret_cast<uint32_t *>(w.Vb)) : nullptr;
        LAUNCH(L, DQ_K_SEG_FUSED, n, n * (8 + wb + 8),
               hipLaunchKernelGGL((seg_fused_kernel<IdxT, true, false, false, true>), dim3((unsigned)ntiles),
                                  dim3(kSegThreads), 0, st, (const uint64_t *)keys, (const IdxT *)d_sa, n, ib, kshift0,
                                  d_sa, w.ISA, P0, w.Va, reinterpret_cast<uint64_t *>(w.seg_status + 256), ntiles,
                                  reinterpret_cast<SegCtl *>(w.seg_status), w.totals, w.totals + 1, 0, list_rank));
        // digit offsets of the two binning passes in closed form: every suffix 0..n-1 occurs once
        const int sh[2] = {ib - 16, ib - 8};
        HIP_TRY(hipMemcpyAsync(c.pinned, w.totals, 16, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        m = c.pinned[0];
        if (c.pinned[1] != 0) return fail(DQ_ERR_HIP, "device look-back timed out (spin bound hit)");
        for (int p = 0; p < 2; ++p) {
            const int64_t unit = 1ll << sh[p];                       // suffixes per digit value inside one cycle
            const int64_t full = n >> (sh[p] + 8), rem = n & ((unit << 8) - 1);
            int64_t acc = 0;
            for (int d = 0; d < 256; ++d) {
                c.pinned[p * 256 + d] = acc;
                acc += full * unit + std::min<int64_t>(std::max<int64_t>(rem - d * unit, 0), unit);
            }
        }
        HIP_TRY(hipMemcpyAsync(w.digit_offset, c.pinned, 2 * 256 * 8, hipMemcpyHostToDevice, st));
        int rc = prepare_status<IdxT>(L, w, n, 2);
        if (rc != DQ_OK) return rc;
        rc = rank_pass<IdxT, kKeys>(L, w, P0, (const IdxT *)nullptr, keys, (IdxT *)nullptr, n, 0, kb, ib, nullptr, nullptr, sh[0]);
        if (rc != DQ_OK) return rc;
        rc = rank_pass<IdxT, kKeys>(L, w, keys, (const IdxT *)nullptr, P0, (IdxT *)nullptr, n, 1, kb, ib, nullptr, nullptr, sh[1]);
        if (rc != DQ_OK) return rc;
        if (ib - 16 <= 12) {
            LAUNCH(L, DQ_K_ISA_FROM_PAIRS, n, n * (8 + wb),
                   hipLaunchKernelGGL((isa_from_pairs_kernel<IdxT, 4096>), dim3((unsigned)((n + 4095) / 4096)),
                                      dim3(kPairThreads), 0, st, (const uint64_t *)P0, n, ib, w.ISA));
        } else {
            LAUNCH(L, DQ_K_ISA_FROM_PAIRS, n, n * (8 + wb),
                   hipLaunchKernelGGL((isa_from_pairs_kernel<IdxT, 32768>), dim3((unsigned)((n + 32767) / 32768)),
                                      dim3(kPairThreads), 0, st, (const uint64_t *)P0, n, ib, w.ISA));
        }
        if (list_rank && m == 0) return DQ_OK;
        if (list_rank && uses_small_round(m) && rank32_direct) {
            first_rank32 = list_rank;
            return DQ_OK;
        }
        if (list_rank && uses_small_round(m)) {
            // (the sorted keys are gone -- their buffer was the output of the first binning pass and is free now)
            LAUNCH(L, DQ_K_KEY2_FROM_PAIRS, m, m * 12,
                   hipLaunchKernelGGL(widen_ranks_kernel, dim3(grid_for(m)), dim3(kBlock), 0, st,
                                      (const uint32_t *)list_rank, m, keys));
            return DQ_OK;
        }
        // Otherwise the list is taken from the words: it comes out in suffix order, not with the members of a
        // group adjacent, so the first doubling round takes the radix path (which sorts it); key2 is gathered here.
        const bool with_key2 = !runs_wanted;               // (runs: the first round's keys are not ISA[s + h], see run())
        const int kbits = bit_length((uint64_t)(n - 1) + (uint64_t)kb);
        unsigned long long *cnt = reinterpret_cast<unsigned long long *>(w.totals + 3);
        HIP_TRY(hipMemsetAsync(cnt, 0, 8, st));
        const int64_t per = (int64_t)kPairThreads * kPairItems;
        LAUNCH(L, DQ_K_KEY2_FROM_PAIRS, n, n * 8 + m * (wb + 8 + wb),
               hipLaunchKernelGGL(key2_from_pairs_kernel<IdxT>, dim3((unsigned)((n + per - 1) / per)), dim3(kPairThreads),
                                  0, st, (const uint64_t *)P0, n, ib, (const IdxT *)w.ISA, (int64_t)kb, kbits, with_key2,
                                  keys, w.Va, cnt));
        keys_ready = with_key2;
        list_ungrouped = true;
        return DQ_OK;
    }

    // ---- bucketed round 0 (see dq_bucket_sort.h).  *done = false: the path does not apply, or it met a
    //      bucket / bin it does not take (the state the plain passes expect has then been restored).
    int round0_bucketed(uint64_t *K[2], int kb, bool packed, bool coded, bool *done)
    {
        *done = false;
        if (coded) return DQ_OK;                          // (the digit offsets on the device are those of the coded keys)
        const int ib = bit_length((uint64_t)(n - 1));
        if (env("DQ_NO_BUCKET") || env("DQ_NO_FUSED_TIES") || env("DQ_SPARSE") || env("DQ_KEY_BYTES")) return DQ_OK;
        const bool forced = env("DQ_BUCKET") != nullptr;
        if (ib > 31 || n < (1 << 16)) return DQ_OK;                           // a suffix must fit 31 bits next to the tie flag
        // a run of >= 64 equal bytes somewhere (zero padding of real binaries; text_hist_kernel saw it): more equal
        // keys than a bin takes -- the pass would only raise its flag and be repeated by the plain passes
        if (!forced && c.pinned[256 + 8] != 0) return DQ_OK;
        // order-0 model of the text (c.pinned still holds the byte histogram): entropy, most frequent byte
        int64_t cmax = 0;
        double h0 = 0;
        for (int b = 0; b < 256; ++b) {
            cmax = std::max(cmax, c.pinned[b]);
            if (c.pinned[b] > 0) { const double p = (double)c.pinned[b] / (double)n; h0 -= p * std::log2(p); }
        }
        const int keybits = std::min(64 - ib, 36);
        if (!packed) {
            // Words were not chosen because too many suffixes would stay tied for the tie-bit path of the plain
            // passes (2 GiB of random bytes: 33 key bits leave 1/4 of them tied).  Those ties are shallow, which the
            // direct-comparison finisher takes; the key must still separate most suffixes, and the k-gram sample
            // must not have seen repetition (it then set kb = 8).
            const double tied = (double)n * std::exp2(-(double)keybits * h0 / 8.0);
            if (!forced && (kb >= 8 || tied > 0.3)) return DQ_OK;
        }
        // longest bucket expected when the words are grouped by their first 2 (3) bytes; tiles are cut for it
        const double pm = (double)cmax / (double)n;
        int bbytes = 2;
        double est = (double)n * pm * pm;
        double need = est + 6.0 * std::sqrt(est) + 64.0;
        // a tile must not span more than 64 two-byte buckets (its keys, relative to its first bucket, take 26
        // bits + 6 arrival bits): buckets of >= 192 words on average, i.e. texts of >= 12 MiB
        const bool force3 = forced && atoi(env("DQ_BUCKET")) == 3;          // (tests: 3-byte buckets on mid-size inputs)
        if (need > 5120 || (!forced && n < (12 << 20)) || force3) {
            if (keybits - 24 >= 8 && (forced || n >= (12 << 20))) {
                bbytes = 3;
                est *= pm;
                need = est + 6.0 * std::sqrt(est) + 64.0;
            }
            if (need > 5120 || (bbytes == 2 && !forced)) {
                if (!forced) return DQ_OK;
                need = 5120;
            }
        }
        const int64_t X = std::min<int64_t>(((int64_t)need + 255) / 256 * 256, 5120);
        const int64_t C = kBktCap - X;
        const int lowbits = keybits - 8 * bbytes;
        const int64_t ntiles = (n + C - 1) / C;
        uint32_t *ebits = reinterpret_cast<uint32_t *>(w.Vb);                  // zeroed by onesweep_sort_text_prepare
        TieCounters *ctr = reinterpret_cast<TieCounters *>(w.totals + 6);     // zero since run()
        BucketFlags *flags = reinterpret_cast<BucketFlags *>(&ctr->overflow);
        const uint64_t *text64 = reinterpret_cast<const uint64_t *>(w.text);
        // digit p of the bucket of suffix i is T[i + bbytes - 1 - p]
        hipLaunchKernelGGL(text_digit_offsets_kernel, dim3(bbytes), dim3(kBlock), 0, st,
                           (const int64_t *)w.bytehist, (const uint8_t *)w.text, n, bbytes, w.digit_offset);
        HIP_TRY(hipGetLastError());
        int rc = rank_pass<IdxT, kTextPacked>(L, w, text64, (const IdxT *)nullptr, K[1], (IdxT *)nullptr, n, 0, kb, ib,
                                             nullptr, nullptr, ib + lowbits, keybits);
        if (rc != DQ_OK) return rc;
        for (int p = 1; p < bbytes; ++p) {                   // pass p reads buffer p & 1 and writes the other
            rc = rank_pass<IdxT, kKeys>(L, w, K[p & 1], (const IdxT *)nullptr, K[(p & 1) ^ 1], (IdxT *)nullptr, n, p, kb, ib,
                                        nullptr, nullptr, ib + lowbits + 8 * p, keybits);
            if (rc != DQ_OK) return rc;
        }
        uint64_t *Ks = K[bbytes & 1], *Kfree = K[(bbytes & 1) ^ 1];           // sorted words / the other buffer
        LAUNCH(L, DQ_K_BUCKET_SORT, ntiles, ntiles * 16 * 8,
               hipLaunchKernelGGL(bucket_bounds_kernel, dim3((unsigned)((ntiles + 1 + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                                  st, (const uint64_t *)Ks, n, ib + lowbits, C, X, ntiles, w.bkt_bounds, flags));
        if (c.ncu <= 0) {
            int v = 0;
            c.ncu = hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, c.dev) == hipSuccess && v > 0 ? v : 256;
        }
        LAUNCH(L, DQ_K_BUCKET_SORT, n, n * (8 + wb) + n / 8,                  // persistent: one workgroup per CU
               hipLaunchKernelGGL(bucket_sort_kernel<IdxT>, dim3((unsigned)std::min<int64_t>(ntiles, c.ncu)),
                                  dim3(kBktThreads), 0, st, (const uint64_t *)Ks, ib, lowbits,
                                  (const int64_t *)w.bkt_bounds, ntiles, d_sa, ebits, flags));
        bool overflow = false;
        fin_cap = n / 8;
        const int64_t hb = keybits / 8;                  // whole bytes the members of a tie group share
        rc = collect_ties<IdxT>(L, c, w, n, kb, ib, ebits, nullptr, (const IdxT *)d_sa, Kfree, w.Va, &m, &overflow,
                                fin_cap, Ks, w.Vb, &fin_left, /*seams=*/false, hb);
        if (rc != DQ_OK) return rc;
        if (overflow) {
            // a bucket or a bin this path does not take (or a run of equal keys too long for the tie walk):
            // back to the plain digit passes, with the state they expect
            if (env("DQ_TRACE")) fprintf(stderr, "[dq] bucketed round 0 gave up (n=%lld): plain digit passes\n", (long long)n);
            rc = prepare_status<IdxT>(L, w, n, kMaxPasses);
            if (rc != DQ_OK) return rc;
            HIP_TRY(hipMemsetAsync(w.Vb, 0, (size_t)((n + 63) / 64 + 1) * 8, st));
            HIP_TRY(hipMemsetAsync(w.totals, 0, 64, st));
            hipLaunchKernelGGL(text_digit_offsets_kernel, dim3(kb), dim3(kBlock), 0, st,
                               (const int64_t *)w.bytehist, (const uint8_t *)w.text, n, kb, w.digit_offset);
            HIP_TRY(hipGetLastError());
            m = 0; fin_cap = 0; fin_left = 0;
            return DQ_OK;
        }
        fin_done = m <= fin_cap;
        Kr[0] = Kfree; Kr[1] = Ks;
        Vr[0] = w.Va; Vr[1] = w.Vb;
        rcur = 0;
        h = hb;
        rbits = ib;
        shallow_ties = true;                             // ties of random-like text: the finisher, not the ISA
        *done = true;
        return DQ_OK;
    }

    // ---- round 0: leading kb bytes of every suffix as a key (or packed word), full radix ranking,
    //      then the first rebucket: X = members of groups of size > 1.  *dense_built tells whether
    //      the rebucket pass already wrote the inverse suffix array.
    int round0(bool *dense_built)
    {
        uint64_t *K[2] = {w.K0, w.K1};
        IdxT *V[2];
        int cur = 0, kb = 8, rc;
        bool packed = false, coded = false;
        // pass p writes buffer (p+1)&1, so the last pass (kb-1) writes buffer kb&1: that one
        // must be the caller's SA, which is why the key width is chosen first
        rc = onesweep_sort_text_prepare<IdxT>(L, c, w, n, &kb, &packed, &coded);
        if (rc != DQ_OK) return rc;
        // (c.pinned still holds the byte histogram, the k-gram sample and the long-run flag of text_hist_kernel)
        // (run lengths + the run-order round cost about one doubling round: worth it where a good part of the text lies
        // in runs -- padded images, sparse files; measured on the image's shared libraries, whose long tie tails are
        // code repeated for several targets, not runs: 5-20 % slower with it.  1/16 of the text in 16-byte chunks of one value)
        runs_wanted = sizeof(IdxT) == 4 && c.pinned[256 + 8] != 0 && n >= (1 << 16) && c.pinned[256 + 9] * 16 * 16 >= n;
        if (const char *v = env("DQ_RUNS")) runs_wanted = sizeof(IdxT) == 4 && atoi(v) != 0;
        if (const char *v = env("DQ_MID_GROUPS")) runs_wanted = runs_wanted && atoi(v) >= 256;   // (the LDS class carries the run offsets)
        V[kb & 1] = d_sa;
        V[(kb & 1) ^ 1] = w.Va;
        // Random-like input (packed words = few ties expected) of a size whose 2-byte buckets fit a workgroup's
        // LDS: two digit passes on the top 16 key bits, then every bucket is finished in LDS (dq_bucket_sort.h).
        {
            bool done = false;
            rc = round0_bucketed(K, kb, packed, coded, &done);
            if (rc != DQ_OK) return rc;
            if (done) { *dense_built = false; return DQ_OK; }
        }
        // Packed words were chosen because few ties are expected: the last pass then records the tie
        // structure itself (1 bit per suffix + 2 words per tile and digit, in the idle Vb buffer)
        // instead of writing the sorted words for a rebucket pass to read back.
        const bool fused_ties = packed && kb >= 2 && n >= (1 << 16) && !env("DQ_NO_FUSED_TIES") &&
                                !env("DQ_SPARSE");
        if (fused_ties) {
            const int ib = bit_length((uint64_t)(n - 1));
            const int64_t nwords = (n + 63) / 64;
            uint32_t *ebits = reinterpret_cast<uint32_t *>(w.Vb);
            uint64_t *seam_tab = reinterpret_cast<uint64_t *>(reinterpret_cast<char *>(w.Vb) +
                                                              align_up((size_t)(nwords + 1) * 8));
            // (the tie bits were zeroed by onesweep_sort_text_prepare while the key width was chosen)
            rc = onesweep_sort_text_passes<IdxT>(L, w, n, K, V, kb, packed, d_sa, cur, ebits, seam_tab);
            if (rc != DQ_OK) return rc;
            // cur names the buffer the last pass would have written: it is free, the pass's input
            // K[cur ^ 1] stays intact for the fallback
            bool overflow = false;
            fin_cap = n / 8;
            rc = collect_ties<IdxT>(L, c, w, n, kb, ib, ebits, seam_tab, (const IdxT *)d_sa, K[cur], w.Va, &m,
                                    &overflow, fin_cap, K[cur ^ 1], w.Vb, &fin_left);
            if (rc != DQ_OK) return rc;
            fin_done = !overflow && m <= fin_cap;
            if (!overflow) {
                *dense_built = false;
                Kr[0] = K[cur]; Kr[1] = K[cur ^ 1];
                Vr[0] = w.Va; Vr[1] = w.Vb;
                rcur = 0;
                h = kb;
                rbits = ib;
                return DQ_OK;
            }
            // a long run of equal keys: redo the last pass with the sorted words as output and take
            // the general rebucket pass below
            HIP_TRY(hipMemsetAsync(w.ctl_status + (size_t)(kb - 1) * w.ctl_status_stride, 0, w.ctl_status_stride, st));
            rc = rank_pass<IdxT, kKeysLast>(L, w, K[cur ^ 1], (const IdxT *)nullptr, K[cur], d_sa, n, kb - 1, kb, ib);
            if (rc != DQ_OK) return rc;
        } else {
            rc = onesweep_sort_text_passes<IdxT>(L, w, n, K, V, kb, packed, d_sa, cur, nullptr, nullptr, coded);
            if (rc != DQ_OK) return rc;
        }
        // sorted keys (or packed words) are in K[cur], suffixes in d_sa
        const int kshift0 = packed ? bit_length((uint64_t)(n - 1)) : 0;
        uint64_t *act_rank = K[cur ^ 1];

        // Few ties (random-like input): they are finished by direct comparison / key extension from
        // the text, without the n random writes of a full inverse suffix array.  Many ties: the ISA
        // is needed for doubling.  4096 sampled adjacent pairs predict which, so that the dense case
        // writes the ISA in the rebucket pass itself.  (Inputs whose order-0 entropy already promised
        // few ties -- packed words or a short key -- skip the sample and its host round trip.)
        bool predict_dense = false;
        if (n >= (1 << 16) && !packed && kb == 8) {
            constexpr int kSamples = 4096;
            HIP_TRY(hipMemsetAsync(w.totals + 2, 0, 8, st));
            hipLaunchKernelGGL(sample_ties_kernel, dim3(kSamples / kBlock), dim3(kBlock), 0, st,
                               (const uint64_t *)K[cur], n, kshift0, kSamples, w.totals + 2);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(c.pinned, w.totals + 2, 8, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            // a pair ties with probability ~ (tied fraction) * (1 - 1/group size); 1/12 ~ tied fraction 1/6
            predict_dense = c.pinned[0] * 12 > kSamples;
        }
        if (const char *v = env("DQ_SPARSE")) predict_dense = atoi(v) == 0;
        // (the suffix-binned build pays once the inverse suffix array outgrows the last-level cache: 4n > 128 MiB.  Below
        // that the plain scatter is ahead -- 64 KiB ... 16 MiB of text: 1-6 %.  DQ_BINNED_ISA=1: from 64 KiB on, for the tests)
        const bool binned_pays = env("DQ_BINNED_ISA") ? atoi(env("DQ_BINNED_ISA")) != 0 : n > (32ll << 20);
        const bool binned = predict_dense && n >= (1 << 16) && binned_pays &&
                            2 * bit_length((uint64_t)(n - 1)) <= 63 && !env("DQ_NO_BINNED_ISA");
        if (binned) {
            rc = build_isa_binned(K[cur], K[cur ^ 1], kb, kshift0);
            if (rc != DQ_OK) return rc;
            *dense_built = true;
            Kr[0] = K[cur]; Kr[1] = K[cur ^ 1];
            Vr[0] = w.Va; Vr[1] = w.Vb;
            rcur = 0;
            h = kb;
            rbits = bit_length((uint64_t)(n - 1));
            return DQ_OK;
        }
        if (predict_dense)
            rc = rebucket<IdxT, true, false, true>(L, c, w, K[cur], (const IdxT *)d_sa, n, 0, kshift0, d_sa, act_rank,
                                                   w.Va, &m);
        else
            rc = rebucket<IdxT, true, false, false>(L, c, w, K[cur], (const IdxT *)d_sa, n, 0, kshift0, d_sa, act_rank,
                                                    w.Va, &m);
        if (rc != DQ_OK) return rc;
        *dense_built = predict_dense;
        // ping-pong buffers of the tied list: (act_rank buffer, Va) <-> (other key buffer, Vb)
        Kr[0] = act_rank; Kr[1] = K[cur];
        Vr[0] = w.Va; Vr[1] = w.Vb;
        rcur = 0;
        h = kb;                          // bytes already compared: the round-0 key width
        rbits = bit_length((uint64_t)(n - 1));
        return DQ_OK;
    }

    // ---- sparse finishing: direct comparison of tiny groups, then up to 3 rounds of key extension
    //      from the text; whatever is still tied afterwards (long repeats) goes to doubling.
    int finish_sparse()
    {
        int rc;
        // tiny groups with a short remaining common prefix; the leftovers come back as a new list
        t_info[0] += 1;
        t_info[2] += m;
        int64_t m2 = fin_left;
        if (!fin_done) {
            unsigned long long *left_over = reinterpret_cast<unsigned long long *>(w.totals + 3);
            HIP_TRY(hipMemsetAsync(left_over, 0, 8, st));
            LAUNCH(L, DQ_K_SMALL_FINISH, m, m * (8 + wb + 16 + wb),
                   hipLaunchKernelGGL((small_group_finish_kernel<IdxT, 8, 32>),
                                      dim3((unsigned)((m + kFinishThreads - 1) / kFinishThreads)), dim3(kFinishThreads),
                                      0, st, (const uint64_t *)Kr[rcur], (const IdxT *)Vr[rcur],
                                      (const uint8_t *)w.text, m, n, h, d_sa, Kr[rcur ^ 1], Vr[rcur ^ 1], left_over));
            HIP_TRY(hipMemcpyAsync(c.pinned, left_over, 8, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            m2 = c.pinned[0];
        }
        if (m2 > 0) rcur ^= 1;
        m = m2;

        const int ebytes = std::max(1, std::min(4, (64 - rbits - 3) / 8));
        const int kbits = 8 * ebytes + 3;
        for (int r = 0; r < 3 && m > 0; ++r) {
            t_info[0] += 1;
            t_info[2] += m;
            LAUNCH(L, DQ_K_GATHER_TEXT_KEY, m, m * (8 + wb + ebytes + 8),
                   hipLaunchKernelGGL(gather_text_key_kernel<IdxT>, dim3(grid_for(m)), dim3(kBlock), 0, st, Kr[rcur],
                                      (const IdxT *)Vr[rcur], (const uint8_t *)w.text, m, n, h, ebytes));
            rc = sort_pairs(Kr, Vr, m, kbits + rbits, rcur);
            if (rc != DQ_OK) return rc;
            rc = rebucket<IdxT, false, true, false>(L, c, w, Kr[rcur], (const IdxT *)Vr[rcur], m, kbits, 0, d_sa,
                                                    Kr[rcur ^ 1], Vr[rcur ^ 1], &m2);
            if (rc != DQ_OK) return rc;
            rcur ^= 1;
            m = m2;
            h += ebytes;
        }
        // long repeats after all: materialise the ranks for the doubling rounds
        return m > 0 ? build_isa(Kr[rcur], Vr[rcur], m) : DQ_OK;
    }

    // ---- one doubling round, everything through the radix path
    int doubling_round_radix(int kbits, int rshift = 0)
    {
        if (keys_ready) {
            keys_ready = false;          // build_isa_binned() gathered key2 while the ranks were still local
        } else {
            LAUNCH(L, DQ_K_GATHER_KEY2, m, m * (8 + wb + wb + 8),
                   hipLaunchKernelGGL(gather_key2_kernel<IdxT>, dim3(grid_for(m)), dim3(kBlock), 0, st, Kr[rcur],
                                      (const IdxT *)Vr[rcur], (const IdxT *)w.ISA, m, n, h, kbits, rshift, rl(),
                                      (const uint8_t *)w.text, run_order));
        }
        int rc = sort_pairs(Kr, Vr, m, kbits + rbits - rshift, rcur);
        if (rc != DQ_OK) return rc;
        list_ungrouped = false;
        int64_t m2 = 0;
        rc = rebucket<IdxT, false, true, true>(L, c, w, Kr[rcur], (const IdxT *)Vr[rcur], m, kbits, 0, d_sa,
                                               Kr[rcur ^ 1], Vr[rcur ^ 1], &m2, rshift);
        if (rc != DQ_OK) return rc;
        rcur ^= 1;
        m = m2;
        return DQ_OK;
    }

    // ---- one doubling round with the groups of <= 8 finished in a single pass (dq_small_groups.h)
    //      and only the larger groups through the radix path.  Needs m <= n/2: every buffer has room
    //      for n entries, X sits in the first half of (A, As), and the other buffer pair receives
    //      T (next list, from 0), L (large groups, from n/2) and U (rank updates, downward from n).
    // L region of a small-group round: first even entry at or after n/2, so that the 16-byte key loads of
    // the radix histogram over it are aligned; U then grows downward from n + 2 (the buffers have the slack)
    int64_t sg_half() const { return (n / 2 + 1) & ~(int64_t)1; }
    int64_t sg_top() const { return n + 2; }

    int doubling_round_small(int kbits)
    {
        uint64_t *A = Kr[rcur], *B = Kr[rcur ^ 1];
        IdxT *As = Vr[rcur], *Bs = Vr[rcur ^ 1];
        const int64_t half = sg_half(), top = sg_top();
        SmallGroupCounters *ctr = reinterpret_cast<SmallGroupCounters *>(w.totals + 4);
        HIP_TRY(hipMemsetAsync(ctr, 0, sizeof(SmallGroupCounters), st));
        const bool cap32 = m < kSgShortList;           // (cap 32 on long lists measured: radix -2.4 ms, this kernel +2.8 ms)
        // The groups of up to mid_g members are finished inside LDS by mid_group_round_kernel (dq_mid_groups.h); only
        // longer ones take the radix passes.  DQ_MID_GROUPS=0: the two-class scheme of before (groups of <= 8, or
        // <= 32 on short lists, in small_group_round_kernel; everything else through the radix passes).
        const int mid_g = mid_group_cap(m);
        const bool use_mid = mid_g > 0;
        if (use_mid) {
            int rc = launch_mid_round(mid_g, m, A, As, B, Bs, h, kbits, ctr, nullptr, m * (8 + wb + wb + wb + 8 + wb));
            if (rc != DQ_OK) return rc;
            first_rank32 = nullptr;
        } else if (cap32) {
            constexpr int kTile = sg_tile<kSgMaxGShort>();
            LAUNCH(L, DQ_K_SMALL_ROUND, m, m * (8 + wb + wb + wb + 8 + wb),
                   hipLaunchKernelGGL((small_group_round_kernel<IdxT, kSgMaxGShort>), dim3((unsigned)((m + kTile - 1) / kTile)),
                                      dim3(kSgThreads), 0, st, (const uint64_t *)A, (const IdxT *)As,
                                      (const IdxT *)w.ISA, m, n, h, kbits, d_sa, B, Bs, B + half, Bs + half, B + top,
                                      Bs + top, ctr, (const SmallGroupCounters *)nullptr));
        } else {
            constexpr int kTile = sg_tile<kSgMaxG>();
            LAUNCH(L, DQ_K_SMALL_ROUND, m, m * (8 + wb + wb + wb + 8 + wb),
                   hipLaunchKernelGGL((small_group_round_kernel<IdxT, kSgMaxG>), dim3((unsigned)((m + kTile - 1) / kTile)),
                                      dim3(kSgThreads), 0, st, (const uint64_t *)A, (const IdxT *)As,
                                      (const IdxT *)w.ISA, m, n, h, kbits, d_sa, B, Bs, B + half, Bs + half, B + top,
                                      Bs + top, ctr, (const SmallGroupCounters *)nullptr));
        }
        HIP_TRY(hipMemcpyAsync(c.pinned, ctr, sizeof(SmallGroupCounters), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        const int64_t m1 = c.pinned[0] & 0xffffffffll, mU = (int64_t)((uint64_t)c.pinned[0] >> 32), mL = c.pinned[1];
        if (env("DQ_TRACE"))
            fprintf(stderr, "[dq] %s round h=%lld m=%lld -> tied %lld, to radix %lld, moved %lld\n",
                    use_mid ? "mid-group" : "small", (long long)h,
                    (long long)m, (long long)m1, (long long)mL, (long long)mU);
        if (mU > 0) {
            LAUNCH(L, DQ_K_ISA_UPDATE, mU, mU * (8 + wb + wb),
                   hipLaunchKernelGGL(isa_update_kernel<IdxT>, dim3(grid_for(mU)), dim3(kBlock), 0, st,
                                      (const uint64_t *)(B + top), (const IdxT *)(Bs + top), mU, w.ISA));
        }
        int64_t mLs = 0;
        if (mL > 0) {
            // radix ping-pong partner: the unused second half of X's own buffers
            uint64_t *Kx[2] = {B + half, A + half};
            IdxT *Vx[2] = {Bs + half, As + half};
            int xcur = 0;
            int rc = onesweep_sort_pairs<IdxT>(L, w, Kx, Vx, mL, kbits + rbits, xcur);
            if (rc != DQ_OK) return rc;
            rc = rebucket<IdxT, false, true, true>(L, c, w, Kx[xcur], (const IdxT *)Vx[xcur], mL, kbits, 0, d_sa,
                                                   B + m1, Bs + m1, &mLs);
            if (rc != DQ_OK) return rc;
        }
        rcur ^= 1;
        m = m1 + mLs;
        // groups only ever split: once nothing went to the radix list, every group fits this round's cap
        if (mL == 0) small_cap = use_mid ? mid_g : cap32 ? kSgMaxGShort : kSgMaxG;
        only_small_groups = mL == 0;
        return DQ_OK;
    }

    // ---- kSgChain small-group rounds back to back, lengths handed over on the device.  Only valid once
    //      every group has <= small_cap members (nothing goes to the radix list any more); the grids are sized for
    //      the current m, which is an upper bound for all later rounds.
    int doubling_rounds_small_chain()
    {
        return small_cap > kSgMaxGShort ? small_chain<0>() : small_cap == kSgMaxGShort ? small_chain<kSgMaxGShort>() : small_chain<kSgMaxG>();
    }

    // one round of mid_group_round_kernel<kG> on the list (A, As)[0, mm); prev: the previous chained round's counters
    int launch_mid_round(int g, int64_t mm, const uint64_t *A, const IdxT *As, uint64_t *B, IdxT *Bs, int64_t hh, int kbits,
                         SmallGroupCounters *ctr, const SmallGroupCounters *prev, int64_t alg_bytes)
    {
        const int64_t half = sg_half(), top = sg_top();
        const int64_t tile = g == 256 ? mg_tile<256>() : g == 512 ? mg_tile<512>() : mg_tile<1024>();
        const dim3 grid((unsigned)((mm + tile - 1) / tile));
        auto go = [&](auto kern) -> int {
            LAUNCH(L, DQ_K_MID_ROUND, mm, alg_bytes,
                   hipLaunchKernelGGL(kern, grid, dim3(kMgThreads), 0, st, A, As, (const IdxT *)w.ISA, mm, n, hh, kbits, d_sa, B, Bs,
                                      B + half, Bs + half, B + top, Bs + top, ctr, prev, rl(), (const uint8_t *)w.text, run_order,
                                      first_rank32));
            return DQ_OK;
        };
        return g == 256 ? go(mid_group_round_kernel<IdxT, 256>) : g == 512 ? go(mid_group_round_kernel<IdxT, 512>)
                                                                            : go(mid_group_round_kernel<IdxT, 1024>);
    }

    template <int kCap>
    int small_chain()
    {
        const int64_t half = sg_half(), top = sg_top();
        const int64_t m_in = m;
        HIP_TRY(hipMemsetAsync(w.sg_ctr, 0, kSgChain * sizeof(SmallGroupCounters), st));
        int64_t hr = h;
        for (int r = 0; r < kSgChain; ++r) {
            uint64_t *A = Kr[rcur], *B = Kr[rcur ^ 1];
            IdxT *As = Vr[rcur], *Bs = Vr[rcur ^ 1];
            const int kbits = std::min(bit_length((uint64_t)(n - 1) + (uint64_t)hr), 64 - rbits);   // (no radix keys are made)
            if constexpr (kCap == 0) {                                       // (every group has <= small_cap members here)
                const int rc = launch_mid_round(small_cap, m_in, A, As, B, Bs, hr, kbits, w.sg_ctr + r,
                                                r == 0 ? (const SmallGroupCounters *)nullptr : w.sg_ctr + r - 1, 0);
                if (rc != DQ_OK) return rc;
            } else {
                constexpr int kTile = sg_tile<kCap>();                       // (every group has <= kCap members here)
                LAUNCH(L, DQ_K_SMALL_ROUND, m_in, 0,
                       hipLaunchKernelGGL((small_group_round_kernel<IdxT, kCap>), dim3((unsigned)((m_in + kTile - 1) / kTile)),
                                          dim3(kSgThreads), 0, st, (const uint64_t *)A, (const IdxT *)As,
                                          (const IdxT *)w.ISA, m_in, n, hr, kbits, d_sa, B, Bs, B + half, Bs + half, B + top,
                                          Bs + top, w.sg_ctr + r, r == 0 ? (const SmallGroupCounters *)nullptr : w.sg_ctr + r - 1));
            }
            LAUNCH(L, DQ_K_ISA_UPDATE, m_in, 0,
                   hipLaunchKernelGGL(isa_update_kernel<IdxT>, dim3(grid_for(m_in)), dim3(kBlock), 0, st,
                                      (const uint64_t *)(B + top), (const IdxT *)(Bs + top), (int64_t)0, w.ISA,
                                      (const SmallGroupCounters *)(w.sg_ctr + r)));
            rcur ^= 1;
            hr *= 2;
        }
        HIP_TRY(hipMemcpyAsync(c.pinned, w.sg_ctr, kSgChain * sizeof(SmallGroupCounters), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        int64_t cur_m = m_in;
        for (int r = 0; r < kSgChain; ++r) {
            if (c.pinned[2 * r + 1] != 0) return fail(DQ_ERR_HIP, "chained small-group round met a large group");
            if (cur_m > 0) { t_info[0] += 1; t_info[2] += cur_m; }
            cur_m = c.pinned[2 * r] & 0xffffffffll;
        }
        m = cur_m;
        h = hr;
        return DQ_OK;
    }

    // ---- small tie groups inside long repeats, decided chain by chain (dq_pair_chains.h).  Needs the ISA and room for
    //      the records behind the list (always there for m <= n/2; for longer lists if the count pass says so).
    //      h is not advanced: the groups that stay behind (>= 3 members, pairs blocked by them) go on doubling.
    // *outcome: 0 = given up after the count (most of the list sits in larger groups: their chains would end
    // blocked), 1 = ran, 2 = ran and finished at least half of the list.
    int pair_chain_phase(int *outcome, bool forced)
    {
        *outcome = 1;
        uint64_t *A = Kr[rcur], *B = Kr[rcur ^ 1];
        IdxT *As = Vr[rcur], *Bs = Vr[rcur ^ 1];
        const int ib = bit_length((uint64_t)(n - 1));
        const int64_t ntiles = (m + kPcTile - 1) / kPcTile;
        const size_t scratch = (size_t)kHistBlocks * kMaxPasses * kRadixSize * 4;          // w.hist_partial
        // Long lists: pairs only (records sorted by x alone, 4 digit passes; groups of 3 and 4 keep doubling, which
        // is cheap per round there).  Short lists are launch-bound: every round saved counts, so groups up to 4 -- or
        // up to 3 when the list is longer than n/3: the records (1.5 per entry for groups of 4, at most 1 for
        // pairs and triples) must fit behind `half`.
        int maxg = m >= kSgShortList ? 2 : (m * 3 <= n ? kPcMaxG : (m * 2 <= n ? 3 : 2));
        // (experiment knob: groups of 3 / 4 as their pairs on long lists too, where the records fit)
        if (const char *v = env("DQ_PAIR_MAXG_LONG")) {
            const int g = atoi(v);
            if (m >= kSgShortList && g >= 3) maxg = (m * 3 <= n && g >= 4) ? kPcMaxG : (m * 2 <= n ? 3 : 2);
        }
        if (const char *v = env("DQ_PAIR_MAXG")) maxg = std::min(maxg >= 3 ? maxg : 2, std::max(2, atoi(v)));
        // record = d << xbits | x.  Pairs only: x padded to whole digits, so that the digit passes over x see nothing of d
        const int xbits = maxg == 2 ? (ib + 7) / 8 * 8 : ib;
        uint32_t *tile_cnt = w.pc_tiles;
        PairCounters *ctr = reinterpret_cast<PairCounters *>(w.totals + 4);
        const int64_t m_in = m;
        int rc = L.begin(DQ_K_PAIR_CHAINS, m, m * 2 * (8 + wb));
        if (rc != DQ_OK) return rc;
        hipLaunchKernelGGL((pair_split_kernel<IdxT, false>), dim3((unsigned)ntiles), dim3(kPcThreads), 0, st,
                           (const uint64_t *)A, (const IdxT *)As, m, xbits, maxg, tile_cnt, (uint64_t *)nullptr, (IdxT *)nullptr,
                           (uint64_t *)nullptr, (IdxT *)nullptr);
        hipLaunchKernelGGL(pair_scan_kernel, dim3(1), dim3(kPcScanThreads), 0, st, tile_cnt, ntiles, ctr);
        HIP_TRY(hipGetLastError());
        rc = L.end();
        if (rc != DQ_OK) return rc;
        HIP_TRY(hipMemcpyAsync(c.pinned, ctr, sizeof(PairCounters), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        const int64_t cnt = c.pinned[0], copied = c.pinned[1];
        // the records (and the ping-pong partner of their sort) go behind the list in both buffer pairs: from n/2 as in
        // a small-group round, or from the end of a longer list if they still fit
        const int64_t half = std::max(sg_half(), (m + 1) & ~(int64_t)1);
        if (cnt == 0 || half + cnt > n || (!forced && copied * 5 > m * 3)) {
            if (env("DQ_TRACE"))
                fprintf(stderr, "[dq] pair chains h=%lld m=%lld: given up, %lld entries in groups > %d\n", (long long)h, (long long)m,
                        (long long)copied, maxg);
            *outcome = 0;
            return DQ_OK;
        }
        t_info[0] += 1;
        t_info[2] += m_in;
        LAUNCH(L, DQ_K_PAIR_CHAINS, m, m * (8 + wb) + cnt * (8 + wb) + copied * (8 + wb),
               hipLaunchKernelGGL((pair_split_kernel<IdxT, true>), dim3((unsigned)ntiles), dim3(kPcThreads), 0, st,
                                  (const uint64_t *)A, (const IdxT *)As, m, xbits, maxg, tile_cnt, B + half, Bs + half, B, Bs));
        const int64_t rtiles = (cnt + kPcTile - 1) / kPcTile;
        // (n close to 2^32 with ~2^31 records: the per-tile scratch would not fit -- leave the list as it is)
        if (2 * align_up((size_t)rtiles) * 4 > scratch) { *outcome = 0; return DQ_OK; }
        uint64_t *Kx[2] = {B + half, A + half};
        IdxT *Vx[2] = {Bs + half, As + half};
        int xcur = 0;
        rc = onesweep_sort_pairs<IdxT>(L, w, Kx, Vx, cnt, maxg == 2 ? xbits : 2 * ib, xcur);
        const uint64_t sort_mask = maxg == 2 ? (1ull << xbits) - 1 : ~0ull;
        if (rc != DQ_OK) return rc;
        // per record: next chain end (4 B) + status (1 B) + answer by ordinal (1 B) in the idle key buffer, far links
        // (4 B) in the idle value buffer
        uint32_t *nt = reinterpret_cast<uint32_t *>(Kx[xcur ^ 1]);
        uint8_t *tstat = reinterpret_cast<uint8_t *>(nt + cnt);
        uint8_t *answer = tstat + cnt;
        uint32_t *far = reinterpret_cast<uint32_t *>(Vx[xcur ^ 1]);
        uint32_t *tile_head = w.hist_partial;
        uint32_t *carry = tile_head + align_up((size_t)rtiles);
        rc = L.begin(DQ_K_PAIR_CHAINS, cnt, cnt * (2 * (8 + wb) + 4 * wb));
        if (rc != DQ_OK) return rc;
        const unsigned rgrid = (unsigned)((cnt + kPcThreads - 1) / kPcThreads);
        hipLaunchKernelGGL(pair_link_kernel<IdxT>, dim3((unsigned)rtiles), dim3(kPcThreads), 0, st,
                           (const uint64_t *)Kx[xcur], cnt, xbits, sort_mask, (const IdxT *)w.ISA, n, h, nt, tstat, far, tile_head);
        hipLaunchKernelGGL(pair_carry_kernel, dim3(1), dim3(kPcScanThreads), 0, st, (const uint32_t *)tile_head, rtiles, carry);
        for (int r = 0; r < kPcResolveRounds; ++r)
            hipLaunchKernelGGL(pair_resolve_kernel, dim3(rgrid), dim3(kPcThreads), 0, st, (const uint32_t *)nt,
                               (const uint32_t *)carry, cnt, tstat, far);
        if (maxg == 2) {
            hipLaunchKernelGGL(pair_emit_kernel<IdxT>, dim3(rgrid), dim3(kPcThreads), 0, st, (const uint64_t *)Kx[xcur],
                               (const IdxT *)Vx[xcur], cnt, xbits, (const uint32_t *)nt, (const uint32_t *)carry,
                               (const uint8_t *)tstat, d_sa, w.ISA, B, Bs, ctr);
        } else {
            hipLaunchKernelGGL(pair_answer_kernel<IdxT>, dim3(rgrid), dim3(kPcThreads), 0, st, (const IdxT *)Vx[xcur], cnt,
                               (const uint32_t *)nt, (const uint32_t *)carry, (const uint8_t *)tstat, answer);
            hipLaunchKernelGGL(pair_finish_kernel<IdxT>, dim3((unsigned)ntiles), dim3(kPcThreads), 0, st, (const uint64_t *)A,
                               (const IdxT *)As, m_in, maxg, (const uint32_t *)tile_cnt, (const uint8_t *)answer, d_sa, w.ISA,
                               B, Bs, ctr);
        }
        HIP_TRY(hipGetLastError());
        rc = L.end();
        if (rc != DQ_OK) return rc;
        HIP_TRY(hipMemcpyAsync(c.pinned, ctr, sizeof(PairCounters), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        rcur ^= 1;
        m = c.pinned[1];
        if ((m_in - m) * 2 >= m_in) *outcome = 2;         // at least half of the list was finished
        if (env("DQ_TRACE"))
            fprintf(stderr, "[dq] pair chains h=%lld m=%lld (groups <= %d): %lld pair records, %lld entries in larger groups, %lld entries left\n",
                    (long long)h, (long long)m_in, maxg, (long long)cnt, (long long)copied, (long long)m);
        return DQ_OK;
    }

    // RL[i] = number of equal bytes the text has from position i on (dq_runs.h): chunk pass, carry across chunks, final pass
    int compute_run_lengths()
    {
        if constexpr (sizeof(IdxT) != 4) {
            return fail(DQ_ERR_HIP, "run lengths: int32 indices only");
        } else {
            const int64_t nchunks = (n + kRunChunk - 1) / kRunChunk;
            LAUNCH(L, DQ_K_RUNS, n, 2 * n + 4 * n,
                   hipLaunchKernelGGL(runlen_chunk_kernel<false>, dim3((unsigned)nchunks), dim3(kRunThreads), 0, st,
                                      (const uint8_t *)w.text, n, w.run_lead, w.run_link, (const uint32_t *)nullptr, (uint32_t *)nullptr);
                   hipLaunchKernelGGL(runlen_carry_kernel, dim3(1), dim3(kRunScanThreads), 0, st, (const uint32_t *)w.run_lead,
                                      (const uint8_t *)w.run_link, nchunks, w.run_carry);
                   hipLaunchKernelGGL(runlen_chunk_kernel<true>, dim3((unsigned)nchunks), dim3(kRunThreads), 0, st,
                                      (const uint8_t *)w.text, n, (uint32_t *)nullptr, (uint8_t *)nullptr,
                                      (const uint32_t *)w.run_carry, w.RL));
            return DQ_OK;
        }
    }

    int run()
    {
        t_info[0] = t_info[1] = t_info[2] = 0;
        HIP_TRY(hipMemsetAsync(w.totals, 0, 64, st));
        bool dense_built = false;
        int rc = round0(&dense_built);
        if (rc != DQ_OK) return rc;
        t_info[1] = m;
        if (m == 0) return flush_profile(c);

        bool sparse = m * 6 <= n || shallow_ties;
        if (const char *v = env("DQ_SPARSE")) sparse = atoi(v) != 0;
        // the ISA exists and the list is keyed / laid out for a radix round -- or carries its ranks as 32-bit values
        // for the first LDS-class round (first_rank32: nothing else may read Kr[rcur] as ranks before that round)
        if (keys_ready || list_ungrouped || first_rank32) sparse = false;
        if (sparse) rc = finish_sparse();
        else if (!dense_built) rc = build_isa(Kr[rcur], Vr[rcur], m);
        if (rc != DQ_OK) return rc;

        // Runs of one byte (dq_runs.h): run lengths of the text, then ONE round that orders the members of every
        // group inside a run by the run's own structure; from then on they gather the rank behind their run.
        if (runs_wanted && m > 0 && 32 + rbits <= 64 && ((uses_small_round(m) && !list_ungrouped) ? mid_group_cap(m) > 0 : true)) {
            rc = compute_run_lengths();
            if (rc != DQ_OK) return rc;
            runs_on = true;
            run_order = 1;
            t_info[0] += 1;
            t_info[2] += m;
            if (env("DQ_TRACE")) fprintf(stderr, "[dq] run-order round at h=%lld on %lld tied suffixes\n", (long long)h, (long long)m);
            rc = (uses_small_round(m) && !list_ungrouped) ? doubling_round_small(32) : doubling_round_radix(32, 0);
            run_order = 0;
            if (rc != DQ_OK) return rc;
        }

        int64_t m_before = 0;             // list length before the last round (0: no round yet)
        int pair_tries = 0, pair_aborts = 0;
        bool pair_paid = true;            // the last pair-chain phase finished at least half of its list
        int64_t pair_h = 0;               // h of the last phase
        int64_t abort_h = 0, abort_m = 0; // h and list length when a phase last gave up after its count
        while (m > 0) {
            // Small tie groups inside long repeats are decided chain by chain (dq_pair_chains.h): tried once after the
            // first doubling round; again after a round that left most of its list tied if the phase before paid
            // off, or -- if it did not -- once h has grown 16-fold (chain ends step over larger groups h characters
            // at a time).  A phase gives up after its count pass when most of the list sits in larger groups, and
            // is tried again once the list has halved or h has grown 16-fold.
            const bool stagnant = m_before > 0 && m * 5 > m_before * 3;
            const char *pc = env("DQ_PAIR_CHAINS");
            const int64_t pair_chain_min = env("DQ_PAIR_CHAINS_MIN") ? std::max(1, atoi(env("DQ_PAIR_CHAINS_MIN"))) : kPairChainMinM;
            const bool after_abort = abort_h == 0 || m * 2 <= abort_m || h >= 16 * abort_h;
            const bool want = pc ? atoi(pc) != 0 && (m_before > 0 || atoi(pc) > 1)
                                 : m_before > 0 && m >= pair_chain_min && after_abort &&
                                   (pair_tries == 0 || (pair_paid ? stagnant : h >= 16 * pair_h));
            const int max_tries = env("DQ_PAIR_TRIES") ? atoi(env("DQ_PAIR_TRIES")) : kPairChainTries;
            if (want && pair_tries < max_tries && pair_aborts < 2 * kPairChainTries && !env("DQ_NO_SMALL") && n < (1ll << 32) &&
                m < n && !keys_ready && !list_ungrouped && !first_rank32) {
                int outcome = 0;
                m_before = 0;
                const int64_t m_try = m;
                rc = pair_chain_phase(&outcome, pc != nullptr);
                if (rc != DQ_OK) return rc;
                if (outcome == 0) { ++pair_aborts; abort_h = h; abort_m = m_try; }
                else { ++pair_tries; pair_paid = outcome == 2; pair_h = h; abort_h = 0; }
                continue;
            }
            m_before = m;
            if (only_small_groups && uses_small_round(m) && !keys_ready && !list_ungrouped && !env("DQ_NO_CHAIN")) {
                rc = doubling_rounds_small_chain();           // several rounds, one host round trip; updates h
                if (rc != DQ_OK) return rc;
                continue;
            }
            t_info[0] += 1;
            t_info[2] += m;
            const int kbits = bit_length((uint64_t)(n - 1) + (uint64_t)h);
            // (rank << kbits | key2) must fit 64 bits.  For 2^31 < n <= 2^32 a repeat longer than 2^32 - n bytes
            // needs 33 + 32: the key then carries rank >> 1 (unique per group: tied groups have >= 2 members)
            // and the rebucket pass reads the true rank from the ISA.  check_args() keeps n <= 2^32.
            // (DQ_FORCE_RSHIFT: the tests take this path on small inputs)
            const int rshift = (kbits + rbits > 64 || env("DQ_FORCE_RSHIFT")) ? 1 : 0;
            if (kbits + rbits - rshift > 64) return fail(DQ_ERR_TOO_LARGE, "composite key exceeds 64 bits");
            if (rshift && keys_ready) {
                // the list came keyed from build_isa_binned() (rank << kbits | key2, unshifted): take the group
                // ranks back out of the keys and let the round gather its own, shifted ones
                LAUNCH(L, DQ_K_GATHER_KEY2, m, m * 16,
                       hipLaunchKernelGGL(keys_to_ranks_kernel, dim3(grid_for(m)), dim3(kBlock), 0, st, Kr[rcur], m, kbits));
                keys_ready = false;
            }
            rc = (uses_small_round(m) && !keys_ready && !rshift && !list_ungrouped) ? doubling_round_small(kbits)
                                                                 : doubling_round_radix(kbits, rshift);
            if (rc != DQ_OK) return rc;
            h *= 2;
        }
        HIP_TRY(hipStreamSynchronize(st));
        return flush_profile(c);
    }
};

template <typename IdxT>
int sufsort_device(DeviceCtx &c, hipStream_t st, Workspace<IdxT> &w, int64_t n, IdxT *d_sa)
{
    SuffixSorter<IdxT> sorter(c, st, w, n, d_sa);
    return sorter.run();
}

// ------------------------------------------------------------------ short texts: one launch
// Largest n the single-workgroup sorter takes (DQ_SMALL_N=0 sends everything down the
// device-wide pipeline; the tests use that to keep the pipeline covered on the fixtures).
int64_t small_limit()
{
    if (const char *v = env("DQ_SMALL_N")) return std::min<int64_t>(std::max(0, atoi(v)), kSmallMaxN);
    return kSmallMaxN;
}

template <typename IdxT>
int sufsort_small(DeviceCtx &c, hipStream_t st, const uint8_t *text, int64_t n, IdxT *sa)
{
    Launcher L{c, st, g_prof_on.load()};
    t_info[0] = t_info[1] = t_info[2] = 0;
    LAUNCH(L, DQ_K_SMALL_SORT, n, n * (1 + (int64_t)sizeof(IdxT)),
           hipLaunchKernelGGL(small_sufsort_kernel<IdxT>, dim3(1), dim3(kSmallThreads), 0, st, text, (int)n, sa));
    HIP_TRY(hipStreamSynchronize(st));
    return flush_profile(c);
}

int resolve_device(int32_t device, int *out)
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

template <typename IdxT>
int check_args(const void *text, int64_t n, const void *sa)
{
    if (n < 0) return fail(DQ_ERR_BAD_ARGS, "negative length");
    if (n > 0 && (!text || !sa)) return fail(DQ_ERR_BAD_ARGS, "null buffer");
    if (sizeof(IdxT) == 4 && n > 0x7fffffffLL)
        return fail(DQ_ERR_TOO_LARGE, "n exceeds 2^31-1; use the i64 entry point");
    // a doubling round sorts (rank, key2) as ONE 64-bit word: 32 + 32 bits at most (see run())
    if (n > (1ll << 32))
        return fail(DQ_ERR_TOO_LARGE, "n exceeds 2^32: the 64-bit entry points take texts of up to 4 GiB");
    return DQ_OK;
}

// host buffers in / out  (ISuffixSort.Sort(text, suffixes))
template <typename IdxT>
int sufsort_host(const uint8_t *text, int64_t n, IdxT *sa, int32_t device)
{
    int rc = check_args<IdxT>(text, n, sa);
    if (rc != DQ_OK) return rc;
    int dev = 0;
    rc = resolve_device(device, &dev);
    if (rc != DQ_OK) return rc;
    // DivSufSort.cs:22-38: the reference special-cases n = 0, 1, 2
    if (n == 0) return DQ_OK;
    if (n == 1) { sa[0] = 0; return DQ_OK; }
    if (n == 2) {
        const bool lt = text[0] < text[1];
        sa[0] = lt ? 0 : 1; sa[1] = lt ? 1 : 0;
        return DQ_OK;
    }
    SlotLease lease(dev, n);
    DeviceCtx &c = *lease.c;
    rc = init_ctx(c, dev);
    if (rc != DQ_OK) return rc;
    if (n <= small_limit()) {
        // the kernel reads the text from, and writes the SA to, pinned host memory: one launch, no copies
        IdxT *io_sa = reinterpret_cast<IdxT *>(c.pinned_io + kSmallTextArea);
        memcpy(c.pinned_io, text, (size_t)n);
        rc = sufsort_small<IdxT>(c, c.stream, c.pinned_io, n, io_sa);
        if (rc != DQ_OK) { drop_pending(c, c.stream); return rc; }
        memcpy(sa, io_sa, (size_t)n * sizeof(IdxT));
        return DQ_OK;
    }
    Workspace<IdxT> w = carve<IdxT>(nullptr, n, true);
    rc = ensure_ws(c, w.bytes);
    if (rc != DQ_OK) return rc;
    w = carve<IdxT>(c.ws, n, true);
    hipStream_t st = c.stream;
    // The caller's buffers are ordinary pageable memory; the runtime's staged copies already run
    // at PCIe rate here (page-locking them per call with hipHostRegister measured no gain).
    HIP_TRY(hipMemcpyAsync(w.text, text, (size_t)n, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemsetAsync(w.text + n, 0, 64, st));
    rc = sufsort_device<IdxT>(c, st, w, n, w.SAbuf);
    if (rc != DQ_OK) { drop_pending(c, st); return rc; }
    HIP_TRY(hipMemcpyAsync(sa, w.SAbuf, (size_t)n * sizeof(IdxT), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return DQ_OK;
}

// device buffers in / out
template <typename IdxT>
int sufsort_dev(const void *d_text, int64_t n, void *d_sa, int32_t device, void *stream)
{
    int rc = check_args<IdxT>(d_text, n, d_sa);
    if (rc != DQ_OK) return rc;
    int dev = 0;
    rc = resolve_device(device, &dev);
    if (rc != DQ_OK) return rc;
    if (n == 0) return DQ_OK;
    SlotLease lease(dev, n);
    DeviceCtx &c = *lease.c;
    rc = init_ctx(c, dev);
    if (rc != DQ_OK) return rc;
    if (n <= small_limit()) {
        hipStream_t sst = stream ? (hipStream_t)stream : c.stream;
        rc = sufsort_small<IdxT>(c, sst, (const uint8_t *)d_text, n, (IdxT *)d_sa);
        if (rc != DQ_OK) drop_pending(c, sst);
        return rc;
    }
    Workspace<IdxT> w = carve<IdxT>(nullptr, n, false);
    rc = ensure_ws(c, w.bytes);
    if (rc != DQ_OK) return rc;
    w = carve<IdxT>(c.ws, n, false);
    hipStream_t st = stream ? (hipStream_t)stream : c.stream;
    HIP_TRY(hipMemcpyAsync(w.text, d_text, (size_t)n, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemsetAsync(w.text + n, 0, 64, st));
    rc = sufsort_device<IdxT>(c, st, w, n, (IdxT *)d_sa);
    if (rc != DQ_OK) { drop_pending(c, st); return rc; }
    HIP_TRY(hipStreamSynchronize(st));
    return DQ_OK;
}

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
        HIP_TRY(hipMalloc(&tmp_tab.p, (size_t)total * sizeof(IdxT)));
        hipLaunchKernelGGL(prefix_bounds_kernel<IdxT>, dim3((unsigned)((total + kBlock - 1) / kBlock)), dim3(kBlock), 0, st,
                           (const uint8_t *)d_old, n, (const IdxT *)d_sa, pk, (IdxT *)tmp_tab.p);
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
    hipError_t e = hipMalloc((void **)&base, b_old + b_sa + b_new + b_sc + 2 * b_out);
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

struct JoinAll {                        // joins whatever was started, also when leaving by exception
    std::vector<std::thread> v;
    ~JoinAll() { for (std::thread &t : v) if (t.joinable()) t.join(); }
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

constexpr size_t kDiffPinnedBytes = 2 * ((size_t)(65536 + 2) * 4 + 256) + (size_t)(2048 + 2 * (1024 + 1)) * 8 + 256;

size_t diff_tab_bytes(int64_t n, int *pk_out)
{
    // prefix table of the match search: 3 bytes (64 MiB of entries) for old files from 4 MiB, 2 bytes from 64 KiB
    const int pk = n >= (4 << 20) ? 3 : n >= (1 << 16) ? 2 : 0;
    *pk_out = pk;
    return pk ? align_up(((size_t)1 << (8 * pk)) * 4 + 16) : 0;
}

int grow_cached(char **buf, size_t *have, size_t want, const char *what)
{
    if (*have >= want) return DQ_OK;
    if (*buf) { (void)hipFree(*buf); *buf = nullptr; *have = 0; }
    hipError_t e = hipMalloc((void **)buf, want);
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
            hipError_t e = hipMalloc((void **)&ix->own, total);
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
        hipLaunchKernelGGL(prefix_bounds_kernel<int32_t>, dim3((unsigned)((total_e + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                           nullptr, (const uint8_t *)ix->d_old, n, (const int32_t *)ix->d_sa, pk, (int32_t *)tab);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipDeviceSynchronize());
        ix->d_tab = tab;
    }
    return DQ_OK;
}

void diff_index_drop(DiffIndex *ix)
{
    if (ix->own && !ix->own_cached) { (void)hipSetDevice(ix->dev); (void)hipFree(ix->own); }
    ix->own = nullptr;
}

// Diff.Create's data path up to the raw streams for one new file: upload it, run the scan loop over windows of answers
int diff_index_scan(const DiffIndex &ix, const uint8_t *nw, int64_t m, bsdiff::RawStreams &raw)
{
    if (m < 0 || (m > 0 && !nw)) return fail(DQ_ERR_BAD_ARGS, "bad arguments");
    if (m > 0x7fffffffLL) return fail(DQ_ERR_TOO_LARGE, "the BSDIFF40 path takes files below 2 GiB (int indices, as the reference)");
    if (m == 0) return DQ_OK;
    const int dev = ix.dev;
    HIP_TRY(hipSetDevice(dev));
    DeviceCtx &c = ctx0(dev);
    const bool trace = env("DQ_TRACE") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    auto stamp = [&](const char *what) {
        if (trace) fprintf(stderr, "[dq] bsdiff %-14s at %8.3f ms\n", what,
                           std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
    };
    const size_t b_new = align_up((size_t)m + 16);
    int rc = grow_cached(&c.diff_dev, &c.diff_dev_bytes, b_new + 256, "hipMalloc(bsdiff buffers)");      // (+ the mailbox of the window kernel)
    if (rc != DQ_OK) return rc;
    if (!c.diff_pinned) {
        hipError_t e = hipHostMalloc((void **)&c.diff_pinned, kDiffPinnedBytes, hipHostMallocCoherent);   // (windows + the packed answers the loop polls)
        if (e != hipSuccess) return fail(DQ_ERR_OOM, "hipHostMalloc(search windows)", e);
    }
    char *pinned = c.diff_pinned;
    const size_t b_win = align_up((size_t)(SearchWindows::kMaxWindow + 2) * 4);
    static_assert(kDiffPinnedBytes >= 2 * ((size_t)(SearchWindows::kMaxWindow + 2) * 4 + 256) +
                  (size_t)(SearchWindows::kWaveWindow + 2 * (SearchWindows::kSecond + 1)) * 8 + 256, "pinned window area");
    char *d_new = c.diff_dev;
    stamp("buffers");
    HIP_TRY(hipMemcpy(d_new, nw, (size_t)m, hipMemcpyHostToDevice));
    stamp("new on device");
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
    return rc;
}

// Diff.Create's data path up to the raw streams: sort old on the device, keep the SA there, run the scan loop
int bsdiff_raw(const uint8_t *old, int64_t n, const uint8_t *nw, int64_t m, int32_t device, bsdiff::RawStreams &raw)
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
    return diff_index_scan(ix, nw, m, raw);
}

// one bzip2 stream; the Burrows-Wheeler transform of each block through the suffix sorter (blocks of a long stream
// are encoded on several threads: the sorter is called concurrently, each call leasing its own device context)
int bz2_stream(const std::vector<uint8_t> &src, std::vector<uint8_t> &out, int dev)
{
    std::atomic<int> sort_rc{DQ_OK};
    std::mutex err_mu;
    std::string err;
    const int rc = bz2::bz2_compress(src.data(), src.size(), out,
                                     [&](const uint8_t *t, int64_t n2, int32_t *sa) -> int {
                                         const int r = sufsort_host<int32_t>(t, n2, sa, dev);
                                         if (r == DQ_OK) return 0;
                                         int expect = DQ_OK;
                                         if (sort_rc.compare_exchange_strong(expect, r)) {
                                             std::lock_guard<std::mutex> lk(err_mu);
                                             err = t_err;                      // (thread-local on the worker: carried over)
                                         }
                                         return -2;
                                     });
    if (rc == -2) { t_err = err; return sort_rc.load(); }
    if (rc != 0) return fail(DQ_ERR_HIP, "bzip2 block transform failed");
    return DQ_OK;
}

// header + the three streams (Diff.cs:54-70 / :196-252).  The streams are framed side by side on three host threads:
// their run-length / MTF / Huffman work overlaps, the block sorts take turns on the device.
int frame_patch(const bsdiff::RawStreams &raw, int64_t m, int dev, std::vector<uint8_t> &patch)
{
    std::vector<uint8_t> z[3];
    const std::vector<uint8_t> *src[3] = {&raw.ctrl, &raw.diff, &raw.extra};
    int rcs[3] = {DQ_OK, DQ_OK, DQ_OK};
    std::string errs[3];
    auto work = [&](int k) {
        try {
            rcs[k] = bz2_stream(*src[k], z[k], dev);
            if (rcs[k] != DQ_OK) errs[k] = t_err;
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

int bsdiff_create_host(const uint8_t *old, int64_t n, const uint8_t *nw, int64_t m, int32_t device, std::vector<uint8_t> &patch)
{
    bsdiff::RawStreams raw;
    int rc = bsdiff_raw(old, n, nw, m, device, raw);
    if (rc != DQ_OK) return rc;
    int dev = 0;
    rc = resolve_device(device, &dev);
    if (rc != DQ_OK) return rc;
    return frame_patch(raw, m, dev, patch);
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

// ------------------------------------------------------------------ batch: one device's share, pipelined
// Three stages on three host threads and three streams, kBatchSlots device buffers in flight:
//   copy-in   text j -> slot          (pageable host memory: the copy blocks its thread, not the others)
//   sort      slot's text -> slot's SA (device-resident sorter; one sort at a time per device anyway)
//   copy-out  slot's SA -> sas[j]
// so the PCIe transfers of neighbouring inputs overlap the sort.  Inputs that need the
// short-text path or that are larger than the slot size go through the plain host entry point.
constexpr int kBatchSlots = 3;


int batch_on_device(int device, const std::vector<int> &jobs, const uint8_t *const *texts, const int64_t *lens,
                    int32_t *const *sas, std::string *err)
{
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
                ok = ok && hipMalloc((void **)&bc.bslot_text[k], (size_t)cap + 64) == hipSuccess &&
                     hipMalloc((void **)&bc.bslot_sa[k], (size_t)cap * sizeof(int32_t)) == hipSuccess;
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

    auto stage_in = [&]() {
        (void)hipSetDevice(device);
        for (int j : jobs) {
            if (lens[j] <= direct) continue;                                // handled after the pipeline
            const int k = take(freeq, nullptr);
            if (k < 0 || failed.load() != DQ_OK) break;
            slots[k].job = j;
            hipError_t e = hipMemcpyAsync(slots[k].text, texts[j], (size_t)lens[j], hipMemcpyHostToDevice, s_in);
            if (e == hipSuccess) e = hipStreamSynchronize(s_in);
            if (e != hipSuccess) { fail_stage(0, DQ_ERR_HIP, std::string("batch copy-in: ") + hipGetErrorString(e)); break; }
            give(filled, k);
        }
        { std::lock_guard<std::mutex> lk(mu); in_done = true; }
        cv.notify_all();
    };
    auto stage_sort = [&]() {
        (void)hipSetDevice(device);
        for (;;) {
            const int k = take(filled, &in_done);
            if (k < 0 || failed.load() != DQ_OK) break;
            const int j = slots[k].job;
            int rc = sufsort_dev<int32_t>(slots[k].text, lens[j], slots[k].sa, device, s_sort);
            if (rc != DQ_OK) { fail_stage(1, rc, t_err); break; }
            give(sorted, k);
        }
        { std::lock_guard<std::mutex> lk(mu); sort_done = true; }
        cv.notify_all();
    };
    auto stage_out = [&]() {
        (void)hipSetDevice(device);
        for (;;) {
            const int k = take(sorted, &sort_done);
            if (k < 0 || failed.load() != DQ_OK) break;
            const int j = slots[k].job;
            hipError_t e = hipMemcpyAsync(sas[j], slots[k].sa, (size_t)lens[j] * sizeof(int32_t), hipMemcpyDeviceToHost, s_out);
            if (e == hipSuccess) e = hipStreamSynchronize(s_out);
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
    if (failed.load() != DQ_OK) {
        for (const std::string &e : errs) if (!e.empty()) { *err = e; break; }
        return failed.load();
    }
    for (int j : jobs)
        if (lens[j] <= direct) { int rc = plain(j); if (rc != DQ_OK) return rc; }
    return DQ_OK;
}

}  // namespace

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
    {
        JoinAll threads;
        for (int d = 0; d < ndev; ++d) {
            threads.v.emplace_back([&, d]() {
                const int device = devs ? devs[d] : d;
                try {
                    rcs[d] = batch_on_device(device, share[d], texts, lens, sas, &errs[d]);
                } catch (const std::exception &e) {
                    rcs[d] = DQ_ERR_OOM;
                    errs[d] = std::string("batch: ") + e.what();
                }
            });
        }
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
    return match_search_dev<int32_t>(d_old, n, d_sa, d_new, m, d_scans, scan0, count, cap, d_pos, d_len, device, stream);
}

int32_t dq_bsdiff_search_dev_i64(const void *d_old, int64_t n, const void *d_sa, const void *d_new, int64_t m,
                                 const int64_t *d_scans, int64_t scan0, int64_t count, int64_t cap, void *d_pos,
                                 void *d_len, int32_t device, void *stream)
{
    EnvScope flags;
    return match_search_dev<int64_t>(d_old, n, d_sa, d_new, m, d_scans, scan0, count, cap, d_pos, d_len, device, stream);
}

int32_t dq_bsdiff_search_i32(const uint8_t *old_data, int64_t n, const int32_t *sa, const uint8_t *new_data, int64_t m,
                             const int64_t *scans, int64_t scan0, int64_t count, int64_t cap, int32_t *pos, int32_t *len,
                             int32_t device)
{
    EnvScope flags;
    return match_search_host<int32_t>(old_data, n, sa, new_data, m, scans, scan0, count, cap, pos, len, device);
}

int32_t dq_bsdiff_search_i64(const uint8_t *old_data, int64_t n, const int64_t *sa, const uint8_t *new_data, int64_t m,
                             const int64_t *scans, int64_t scan0, int64_t count, int64_t cap, int64_t *pos, int64_t *len,
                             int32_t device)
{
    EnvScope flags;
    return match_search_host<int64_t>(old_data, n, sa, new_data, m, scans, scan0, count, cap, pos, len, device);
}

int32_t dq_bsdiff_scan_i32(const uint8_t *old_data, int64_t n, const uint8_t *new_data, int64_t m, int64_t *ctrl,
                           int64_t ctrl_cap, int64_t *nctrl, uint8_t *diff, int64_t *ndiff, uint8_t *extra, int64_t *nextra,
                           int64_t *stats, int32_t device)
{
    EnvScope flags;
    try {
        bsdiff::RawStreams raw;
        const int rc = bsdiff_raw(old_data, n, new_data, m, device, raw);
        if (rc != DQ_OK) return rc;
        const int64_t triples = (int64_t)raw.ctrl.size() / 24;
        if (triples > ctrl_cap) return fail(DQ_ERR_BAD_ARGS, "control buffer too small");
        for (int64_t i = 0; i < 3 * triples; ++i) ctrl[i] = bsdiff::read_packed_long(&raw.ctrl[(size_t)i * 8]);
        if (!raw.diff.empty()) memcpy(diff, raw.diff.data(), raw.diff.size());
        if (!raw.extra.empty()) memcpy(extra, raw.extra.data(), raw.extra.size());
        *nctrl = triples; *ndiff = (int64_t)raw.diff.size(); *nextra = (int64_t)raw.extra.size();
        if (stats) { stats[0] = raw.searches; stats[1] = raw.windows; stats[2] = raw.exact; }
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
        DiffIndex *ix = new DiffIndex();
        const int rc = diff_index_build(old_data, n, device, d_old, d_sa, /*cached=*/false, ix);
        if (rc != DQ_OK) { diff_index_drop(ix); delete ix; return rc; }
        *index_out = ix;
        return DQ_OK;
    } catch (const std::bad_alloc &) {
        return fail(DQ_ERR_OOM, "bsdiff: host allocation failed");
    } catch (const std::exception &e) {
        return fail(DQ_ERR_HIP, e.what());
    }
}

int32_t dq_bsdiff_index_buffers(const void *index, const void **d_old, const void **d_sa, int64_t *n)
{
    if (!index) return fail(DQ_ERR_BAD_ARGS, "null index");
    const DiffIndex *ix = static_cast<const DiffIndex *>(index);
    if (d_old) *d_old = ix->d_old;
    if (d_sa) *d_sa = ix->d_sa;
    if (n) *n = ix->n;
    return DQ_OK;
}

int32_t dq_bsdiff_index_diff(const void *index, const uint8_t *new_data, int64_t m, uint8_t *patch, int64_t cap,
                             int64_t *patch_len)
{
    EnvScope flags;
    if (!index) return fail(DQ_ERR_BAD_ARGS, "null index");
    const DiffIndex *ix = static_cast<const DiffIndex *>(index);
    try {
        std::vector<uint8_t> v;
        {
            bsdiff::RawStreams raw;
            {
                std::lock_guard<std::mutex> one_diff(ctx0(ix->dev).diff_mu);      // scan loops take turns on a device
                const int rc = diff_index_scan(*ix, new_data, m, raw);
                if (rc != DQ_OK) return rc;
            }
            const int rc = frame_patch(raw, m, ix->dev, v);                        // (framing overlaps the next caller's scan loop)
            if (rc != DQ_OK) return rc;
        }
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
    DiffIndex *ix = static_cast<DiffIndex *>(index);
    diff_index_drop(ix);
    delete ix;
}

int64_t dq_bsdiff_patch_bound(int64_t n, int64_t m)
{
    if (n < 0 || m < 0) return -1;
    // three bzip2 streams: 24 bytes of control per triple (at most m + 1 triples), m diff + extra bytes in total;
    // bzip2 never grows its input by more than 1 % + 600 bytes per stream
    const int64_t raw = 24 * (m + 1) + m;
    return bsdiff::kHeaderSize + raw + raw / 100 + 3 * 600 + 64;
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
    if (index_bytes == 4) return (int64_t)carve<int32_t>(nullptr, n, false).bytes;
    if (index_bytes == 8) return (int64_t)carve<int64_t>(nullptr, n, false).bytes;
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
        for (int k = 0; k < kCtxSlots; ++k) any = any || g_dev[d].slot[k].dev >= 0;
        if (!any || hipSetDevice(d) != hipSuccess) continue;
        for (int k = 0; k < 3; ++k) {
            if (c0.bslot_text[k]) (void)hipFree(c0.bslot_text[k]);
            if (c0.bslot_sa[k]) (void)hipFree(c0.bslot_sa[k]);
            c0.bslot_text[k] = nullptr; c0.bslot_sa[k] = nullptr;
        }
        c0.bslot_cap = 0;
        for (hipStream_t *st : {&c0.b_in, &c0.b_sort, &c0.b_out}) { if (*st) (void)hipStreamDestroy(*st); *st = nullptr; }
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

}  // extern "C"
