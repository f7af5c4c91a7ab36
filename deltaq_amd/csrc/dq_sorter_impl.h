// dq_sorter_impl.h -- the suffix sorter: host driver of the HIP kernels, templated on the index type.
// Included by dq_sorter_i32.hip and dq_sorter_i64.hip, which instantiate the entry points of dq_runtime.h.
//
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
#pragma once
#include "dq_runtime.h"
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
#include "dq_pair_chains.h"
#include "dq_tail.h"
#include "dq_split_round0.h"

namespace dq {
namespace {

constexpr int kSgChain = 8;       // small-group rounds chained without a host round trip (4 -> 8: see DESIGN section 5)
constexpr int64_t kSgShortList = 1 << 20;     // below this many tied suffixes a round is launch-bound


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
    uint64_t *X;                // third list buffer (keys / update words) of a doubling round over more than n/2 tied suffixes
    IdxT *Xs;                   // ... and its suffixes
    int64_t *bkt_bounds;        // tile bounds of the bucketed round 0 (dq_bucket_sort.h)
    int64_t *totals;            // [0] active count, [1] sticky look-back timeout flag
    SmallGroupCounters *sg_ctr; // one per chained small-group round
    uint32_t *hist_partial;     // scratch: [8][256] 64-bit digit counters of the histogram kernels in front, pair-chain tables
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
    // round 0 as a sample sort (dq_split_round0.h; int32 indices, texts of >= kSplitMinN bytes): splitter tables, cursors, plans
    uint64_t *sp_top, *sp_sub;
    unsigned long long *sp_cnt_a, *sp_cursor_a, *sp_cursor_b;
    int64_t *sp_off, *sp_out_base, *sp_ovf_src, *sp_ovf_dst;
    uint64_t *sp_low;           // lower key bound of every bucket
    uint8_t *sp_pure;           // bucket holds copies of one key only
    uint32_t *sp_tile_first;
    ScanPart *sp_part;
    SplitCtl *sp_ctl;
    size_t bytes;
};

// smallest text the sample-sort round 0 can take: its 2 Mi sampled keys are sorted in idle key buffers, and pass A's spill (n / 8
// + 1024 entries per top bucket) must fit a quarter of the suffix array
constexpr int64_t kSplitMinN = 5ll << 20;

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
    // (n > 2^32 is refused before anything is allocated; texts of more than n/2 tied suffixes after round 0 -- real
    // binaries -- take their first doubling rounds through the LDS class too, whose three output lists then need a
    // buffer of their own: +12 n / +16 n bytes of a 288 GB device)
    if (un < (1ull << 32)) {            // (exactly 2^32 bytes: no small-group rounds, uses_small_round())
        w.X = (uint64_t *)take((un + 2) * 8);
        w.Xs = (IdxT *)take((un + 2) * sizeof(IdxT));
    }
    w.bkt_bounds = (int64_t *)take((un / 4096 + 4) * 8);
    w.totals = (int64_t *)take(64);
    w.sg_ctr = (SmallGroupCounters *)take((kSgChain + 2) * sizeof(SmallGroupCounters));     // (+ the tail kernel's result, dq_tail.h)
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
    if (sizeof(IdxT) == 4 && n >= kSplitMinN) {             // ~4.7 MB of tables
        w.sp_top = (uint64_t *)take((size_t)kSplitTop * 8);
        w.sp_sub = (uint64_t *)take((size_t)kSplitBuckets * 8);
        w.sp_cnt_a = (unsigned long long *)take((size_t)kSplitTop * 8);
        w.sp_cursor_a = (unsigned long long *)take((size_t)kSplitTop * 8);
        w.sp_cursor_b = (unsigned long long *)take((size_t)kSplitBuckets * 8);
        w.sp_off = (int64_t *)take((size_t)(kSplitTop + 1) * 8);
        w.sp_out_base = (int64_t *)take((size_t)(kSplitBuckets + 1) * 8);
        w.sp_ovf_src = (int64_t *)take((size_t)kSplitBuckets * 8);
        w.sp_ovf_dst = (int64_t *)take((size_t)kSplitBuckets * 8);
        w.sp_low = (uint64_t *)take((size_t)kSplitBuckets * 8);
        w.sp_pure = (uint8_t *)take((size_t)kSplitBuckets);
        w.sp_tile_first = (uint32_t *)take((size_t)(kSplitTop + 1) * 4);
        w.sp_part = (ScanPart *)take((size_t)kScanBlocks * sizeof(ScanPart));
        w.sp_ctl = (SplitCtl *)take(sizeof(SplitCtl));
    }
    w.bytes = off;
    return w;
}


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
    static constexpr bool kExtra = (kMode == kTextPackedExt || kMode == kKeysExt);          // words + one more key byte each
    static constexpr bool kWords = (kMode == kTextPacked || kMode == kKeys || kMode == kKeysLast || kMode == kKeysLastTies || kExtra);
    // (a 1024-thread tile for the tie-recording last pass, whose runs are 4-byte SA entries, measured +18 %)
    static constexpr int kThreads = kSmall ? 256 : 512;
    static constexpr int kItems = kSmall ? 8 : kExtra ? 20 : kWords ? 24 : (sizeof(IdxT) == 4 ? 20 : 16);
    static constexpr int kMinWaves = 2;
    static constexpr int kRounds = 2;
    // LDS match tables beat 8 ballots on near-uniform digits (words: -6%), but equal digits in a wave are
    // same-address LDS atomics: pair passes run on text-like (skewed) data and keep the ballots
    static constexpr bool kLdsMatch = kWords;
    // the first pass of a sort has no earlier order to keep: atomic cursors instead of the look-back
    static constexpr bool kAtomicBase = (kMode == kTextPacked || kMode == kText || kMode == kTextPackedExt);
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

// XCD-aware tile order of the first digit pass of a sort (radix_rank_kernel, kAtomicBase): tiles per XCD and group.
// DQ_XCD_GROUP = 0 (blockIdx order) | 1 .. 64.
// what the look-back spins of this call's launches give up at (dq_device_utils.h: a kernel argument)
inline uint32_t spin_bound() { return t_fault.spin ? 0u : kSpinLimit; }

inline int xcd_tile_group()
{
    if (const char *v = env("DQ_XCD_GROUP")) return std::max(0, std::min(64, atoi(v)));
    return 8;
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
                              ebits, seam_tab, (const uint16_t *)w.codetab, xcd_tile_group(), spin_bound()));
    return DQ_OK;
}

// a digit pass over packed words that travel with one more byte of key each (kTextPackedExt: made from the text; kKeysExt)
template <typename IdxT, int kMode>
int rank_pass_ext(Launcher &L, Workspace<IdxT> &w, const uint64_t *kin, const uint8_t *ein, uint64_t *kout, uint8_t *eout,
                  int64_t m, int pass, int ib, int shift, int keybits)
{
    static_assert(kMode == kTextPackedExt || kMode == kKeysExt, "extra-byte modes");
    using Cfg = RankCfg<IdxT, kMode>;
    constexpr int kTileN = Cfg::kThreads * Cfg::kItems;
    const int64_t ntiles = (m + kTileN - 1) / kTileN;
    char *area = w.ctl_status + (size_t)pass * w.ctl_status_stride;
    OnesweepCtl *ctl = reinterpret_cast<OnesweepCtl *>(area);
    auto go = [&](auto status_tag) -> int {
        using StatusT = decltype(status_tag);
        StatusT *status = reinterpret_cast<StatusT *>(area + 256);
        if (256 + (size_t)ntiles * kRadixSize * sizeof(StatusT) > w.ctl_status_stride)
            return fail(DQ_ERR_HIP, "status buffer too small");
        LAUNCH(L, DQ_K_RADIX_RANK, m, m * (kMode == kTextPackedExt ? 1 + 8 + 1 : 18),
               hipLaunchKernelGGL((radix_rank_kernel<IdxT, StatusT, Cfg::kItems, kMode, Cfg::kMinWaves, Cfg::kThreads, false,
                                                     Cfg::kLdsMatch, Cfg::kRounds, Cfg::kAtomicBase, false, uint8_t>),
                                  dim3((unsigned)ntiles), dim3(Cfg::kThreads), 0, L.st, kin, ein, kout, eout, m, shift, keybits, ib,
                                  (const int64_t *)(w.digit_offset + pass * kRadixSize), status, ctl, w.totals + 1,
                                  (uint32_t *)nullptr, (uint64_t *)nullptr, (const uint16_t *)w.codetab, xcd_tile_group(), spin_bound()));
        return DQ_OK;
    };
    return m < (1ll << 30) ? go(uint32_t{}) : go(uint64_t{});
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
void launch_hist(hipStream_t st, int blocks, const uint64_t *keys, int64_t m, uint32_t *acc_area, int shift0 = 0)
{
    // (the counters the workgroups add into: the first 16 KB of the histogram scratch area, zeroed here)
    (void)hipMemsetAsync(acc_area, 0, (size_t)kMaxPasses * kRadixSize * 8, st);
    hipLaunchKernelGGL(radix_hist_kernel<kPasses>, dim3(blocks), dim3(kHistThreads), 0, st, keys, m,
                       reinterpret_cast<unsigned long long *>(acc_area), shift0);
}

// generic pairs: all digit histograms in one read, then one radix_rank_kernel per digit
template <typename IdxT>
int onesweep_sort_pairs(Launcher &L, Workspace<IdxT> &w, uint64_t *K[2], IdxT *V[2], int64_t m,
                        int total_bits, int &cur, int shift0 = 0 /* the sort field starts at this bit */)
{
    const int passes = (total_bits + kRadixBits - 1) / kRadixBits;
    const int blocks = (int)std::min<int64_t>(kHistBlocks, ((m >> 1) + kHistThreads - 1) / kHistThreads + 1);
    int rc = L.begin(DQ_K_RADIX_HIST, m, m * 8);
    if (rc != DQ_OK) return rc;
    switch (passes) {
        case 1: launch_hist<1>(L.st, blocks, K[cur], m, w.hist_partial, shift0); break;
        case 2: launch_hist<2>(L.st, blocks, K[cur], m, w.hist_partial, shift0); break;
        case 3: launch_hist<3>(L.st, blocks, K[cur], m, w.hist_partial, shift0); break;
        case 4: launch_hist<4>(L.st, blocks, K[cur], m, w.hist_partial, shift0); break;
        case 5: launch_hist<5>(L.st, blocks, K[cur], m, w.hist_partial, shift0); break;
        case 6: launch_hist<6>(L.st, blocks, K[cur], m, w.hist_partial, shift0); break;
        case 7: launch_hist<7>(L.st, blocks, K[cur], m, w.hist_partial, shift0); break;
        default: launch_hist<8>(L.st, blocks, K[cur], m, w.hist_partial, shift0); break;
    }
    hipLaunchKernelGGL(radix_hist_scan_kernel, dim3(passes), dim3(kHistScanThreads), 0, L.st,
                       (const unsigned long long *)w.hist_partial, w.digit_offset);
    HIP_TRY(hipGetLastError());
    rc = L.end();
    if (rc != DQ_OK) return rc;
    rc = prepare_status<IdxT>(L, w, m, passes);
    if (rc != DQ_OK) return rc;
    for (int p = 0; p < passes; ++p) {
        rc = rank_pass<IdxT, kPairs>(L, w, K[cur], V[cur], K[cur ^ 1], V[cur ^ 1], m, p, 8, 0, nullptr, nullptr,
                                     shift0 > 0 ? shift0 + p * kRadixBits : -1);
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
int launch_coded_hist(Launcher &L, Workspace<IdxT> &w, int64_t n)
{
    const int hblocks = (int)std::min<int64_t>(kHistBlocks, ((n >> 2) + kHistThreads - 1) / kHistThreads + 1);
    int rc2 = L.begin(DQ_K_RADIX_HIST, n, n);
    if (rc2 != DQ_OK) return rc2;
    HIP_TRY(hipMemsetAsync(w.hist_partial, 0, (size_t)kMaxPasses * kRadixSize * 8, L.st));
    hipLaunchKernelGGL(text_coded_hist_kernel, dim3(hblocks), dim3(kHistThreads), 0, L.st,
                       reinterpret_cast<const uint32_t *>(w.text), n, (const uint16_t *)w.codetab,
                       reinterpret_cast<unsigned long long *>(w.hist_partial));
    hipLaunchKernelGGL(radix_hist_scan_kernel, dim3(kMaxPasses), dim3(kHistScanThreads), 0, L.st,
                       (const unsigned long long *)w.hist_partial, w.digit_offset);
    HIP_TRY(hipGetLastError());
    return L.end();
}

// Round 0 as a sample sort (dq_split_round0.h) instead of eight digit passes: the 8-byte pair path -- coded keys (text-like
// input) or raw ones (real binaries) --, int32 indices, from 64 MiB on.  Measured, sample sort against digit passes in one
// process: enwik-style text 64 MiB 7.29 / 7.86 ms, 96 MiB 9.87 / 11.11, 128 MiB 12.42 / 14.12, 256 MiB 24.9 / 29.3 (32 MiB, an
// earlier build: 5.54 / 4.60 -- its fixed costs, a 2 Mi-key sample sorted and 262 144 workgroups of the finish kernel, want
// a long text); first 128 MiB of libtorch_cpu.so 19.9 / 21.75 (17 M copies of heavy keys placed unsorted; while they
// took the sorted overflow route: 22.7).  Up to the size whose mean bucket is half the finish kernel's capacity (256 MiB).
// DQ_SPLIT = 0 | 1 | 2 overrides (1: from kSplitMinN on; 2: also past what the sample says about heavy keys -- for the tests).
template <typename IdxT>
bool split_round0_wanted(int64_t n, bool packed, int kb, bool coded)
{
    if (sizeof(IdxT) != 4 || packed || kb != 8 || n < kSplitMinN || n > (int64_t)kSplitBuckets * (kFinCap / 2)) return false;
    if (const char *v = env("DQ_SPLIT")) return atoi(v) != 0;
    if (env("DQ_KEY_BYTES") || env("DQ_NO_BUCKET")) return false;     // (forced plain paths of the tests stay what they were)
    (void)coded;
    return n >= (64ll << 20);
}

template <typename IdxT>
int onesweep_sort_text_prepare(Launcher &L, DeviceCtx &c, Workspace<IdxT> &w, int64_t n, int *kb_out,
                               bool *packed_out, bool *coded_out, const uint8_t *text_src = nullptr, bool *hist_deferred = nullptr)
{
    *coded_out = false;
    // (256-thread workgroups: the pass is a chain of 16-byte loads and LDS adds, bound by how many are in flight.  512 / 1024 /
    // 2048 / 4096 workgroups at 256 MiB: 165 / 131 / 139 / 143 us -- more waves hide more latency until the 256 global adds
    // each workgroup ends with pile up.  DQ_TEXT_HIST_BLOCKS tries other grids.)
    const int64_t hist_cap = env("DQ_TEXT_HIST_BLOCKS") ? std::max(1, std::min(8192, atoi(env("DQ_TEXT_HIST_BLOCKS")))) : 1024;
    const int blocks = (int)std::min<int64_t>(hist_cap, ((n >> 4) + kBlock - 1) / kBlock + 1);
    HIP_TRY(hipMemsetAsync(w.bytehist, 0, (256 + 10) * 8, L.st));
    // (+1 workgroup: the k-gram sample, whose 8 counters sit right behind the byte histogram: one readback)
    LAUNCH(L, DQ_K_TEXT_HIST, n, n,
           // (text_src: the caller's device buffer, not copied yet -- this pass reads it and fills w.text, see the kernel)
           hipLaunchKernelGGL(text_hist_kernel, dim3(blocks + 1), dim3(kBlock), 0, L.st,
                              text_src ? text_src : (const uint8_t *)w.text, n, reinterpret_cast<unsigned long long *>(w.bytehist),
                              reinterpret_cast<unsigned long long *>(w.bytehist + 256), text_src ? w.text : (uint8_t *)nullptr));
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
            // (the eight digit histograms of the coded keys -- one more read of the text -- only if the digit passes
            // will run: the sample-sort round 0 does not need them and launches them itself should it give up)
            if (hist_deferred && split_round0_wanted<IdxT>(n, packed, kb, true)) {
                *hist_deferred = true;
            } else {
                const int rc2 = launch_coded_hist<IdxT>(L, w, n);
                if (rc2 != DQ_OK) return rc2;
            }
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
    // byte model of the speculative finisher, now that the list length is known: list entry in, one 64-byte
    // sector of text per tied suffix, SA entry out
    if (fin_cap > 0 && L.active && !c.pending.empty() && c.pending.back().cat == DQ_K_SMALL_FINISH) {
        const int64_t cnt = std::min<int64_t>(*count, fin_cap);
        c.pending.back().elems = cnt;
        c.pending.back().bytes = cnt * (8 + wb + 64 + wb);
    }
    return DQ_OK;
}

// Rebucket a list sorted by (composite) key: group heads, device-wide scan, SA / ISA
// scatter, compaction of the still-tied suffixes into (act_rank, act_suf); *active_out = their
// number.  Engine 1: one fused single-pass kernel; engine 0: the legacy three kernels.
// kInitial never writes ISA (it is built later, and only on the dense path).
template <typename IdxT, bool kInitial, bool kWriteSA, bool kWriteISA>
int rebucket(Launcher &L, DeviceCtx &c, Workspace<IdxT> &w, const uint64_t *keys, const IdxT *vals,
             int64_t m, int kbits, int kshift, IdxT *SA, uint64_t *act_rank, IdxT *act_suf,
             int64_t *active_out, int rank_from_isa = 0, int rank_lo = 0)
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
                              reinterpret_cast<SegCtl *>(w.seg_status), w.totals, w.totals + 1, rank_from_isa,
                              (uint32_t *)nullptr, rank_lo, spin_bound()));
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
    // ... or only once the rounds show them: text_hist_kernel saw a long run somewhere (long_run_seen), and the members
    // of the large groups (last_large: what the last round sent to the radix list, prev_large the round before) stop
    // getting fewer -- a run of L bytes keeps ~L - h of its suffixes in one group for log2(L / h) rounds.  The run
    // lengths and the run-order round are then paid at that point, on the list as it is by then (libtorch_cpu.so:
    // one 5.5 MB run of 'X' among 128 MiB kept 16 rounds of 8 radix passes over ~6 M entries alive).
    bool long_run_seen = false, late_runs_possible = false, runs_late_tried = false;
    int64_t last_large = -1, prev_large = -1;
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

    // digit offsets of the two binning passes over the top 16 bits of the suffix, in closed form (every suffix
    // 0..n-1 occurs once): staged in the pinned area, uploaded to w.digit_offset[0..1]
    int upload_suffix_bin_offsets(int ib)
    {
        const int sh[2] = {ib - 16, ib - 8};
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
        return DQ_OK;
    }

    // ISA[SA[p]] = p for everybody, then the tied suffixes get their group rank.  (rank, suf) = (Kr[rcur], Vr[rcur])[0, cnt).
    // Large texts: the n random 4-byte writes of the plain scatter (256 MiB: 9.9 ms) are replaced by the suffix-binned
    // build of dq_isa_pairs.h -- words (p << ib | SA[p]), two word passes over the top 16 bits of the suffix, LDS
    // images written coalesced (~3 ms) -- with the tied list's ranks parked in the idle index buffer meanwhile.
    int build_isa(const uint64_t *rank, const IdxT *suf, int64_t cnt)
    {
        const int ib = bit_length((uint64_t)(n - 1));
        const bool pays = env("DQ_BINNED_ISA") ? atoi(env("DQ_BINNED_ISA")) != 0 : n > (32ll << 20);
        const bool fits = (size_t)cnt * 8 <= (size_t)(n + 2) * sizeof(IdxT) && rank == Kr[rcur] && suf == Vr[rcur];
        if (pays && fits && n >= (1 << 16) && 2 * ib <= 63 && !env("DQ_NO_BINNED_ISA")) {
            uint64_t *P0 = Kr[rcur ^ 1], *P1 = Kr[rcur];
            uint64_t *stash = reinterpret_cast<uint64_t *>(Vr[rcur ^ 1]);
            if (cnt > 0) HIP_TRY(hipMemcpyAsync(stash, P1, (size_t)cnt * 8, hipMemcpyDeviceToDevice, st));
            LAUNCH(L, DQ_K_ISA_FROM_SA, n, n * (wb + 8),
                   hipLaunchKernelGGL(sa_words_kernel<IdxT>, dim3(grid_for(n)), dim3(kBlock), 0, st, (const IdxT *)d_sa, n, ib, P0));
            int rc = upload_suffix_bin_offsets(ib);
            if (rc != DQ_OK) return rc;
            rc = prepare_status<IdxT>(L, w, n, 2);
            if (rc != DQ_OK) return rc;
            rc = rank_pass<IdxT, kKeys>(L, w, P0, (const IdxT *)nullptr, P1, (IdxT *)nullptr, n, 0, 8, ib, nullptr, nullptr, ib - 16);
            if (rc != DQ_OK) return rc;
            rc = rank_pass<IdxT, kKeys>(L, w, P1, (const IdxT *)nullptr, P0, (IdxT *)nullptr, n, 1, 8, ib, nullptr, nullptr, ib - 8);
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
            if (cnt > 0) HIP_TRY(hipMemcpyAsync(P1, stash, (size_t)cnt * 8, hipMemcpyDeviceToDevice, st));
            LAUNCH(L, DQ_K_ISA_FROM_SA, cnt, cnt * (8 + 2 * wb),
                   hipLaunchKernelGGL(isa_scatter_kernel<IdxT>, dim3(grid_for(cnt)), dim3(kBlock), 0, st, rank, suf, w.ISA, cnt));
            return DQ_OK;
        }
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
    // (wide: a list of more than n/2 entries -- the output lists of its round do not fit beside each other in the
    // partner buffers, see round_layout(); DQ_NO_WIDE_SMALL=1: such lists take the radix path as before round 5)
    bool wide_list(int64_t mm) const { return mm * 2 > n; }
    bool uses_small_round(int64_t mm) const
    {
        return !env("DQ_NO_SMALL") && n < (1ll << 32) && (!wide_list(mm) || !env("DQ_NO_WIDE_SMALL"));
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
                                  reinterpret_cast<SegCtl *>(w.seg_status), w.totals, w.totals + 1, 0, list_rank, 0, spin_bound()));
        // digit offsets of the two binning passes in closed form: every suffix 0..n-1 occurs once
        const int sh[2] = {ib - 16, ib - 8};
        HIP_TRY(hipMemcpyAsync(c.pinned, w.totals, 16, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        m = c.pinned[0];
        if (c.pinned[1] != 0) return fail(DQ_ERR_HIP, "device look-back timed out (spin bound hit)");
        int rc = upload_suffix_bin_offsets(ib);
        if (rc != DQ_OK) return rc;
        rc = prepare_status<IdxT>(L, w, n, 2);
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
        int keybits = std::min(64 - ib, 36);
        if (const char *v = env("DQ_BUCKET_KEYBITS")) keybits = std::max(17, std::min(keybits, atoi(v)));      // (tests: few key bits on small inputs)
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
        // One more byte of key beside every word (kTextPackedExt / kKeysExt passes, bucket_sort_kernel<kExt>) where the
        // word's own key bits would leave more than a few per cent of the suffixes tied: 2 GiB of random bytes have 33
        // bits beside the 31-bit suffix -- 22 % tied, 19 ms of direct comparisons behind one 64-byte sector each --
        // and 41 with the byte (0.1 %).  The bytes live in the idle index buffer Va (two arrays of n).
        // DQ_BUCKET_EXT = 0 | 1 overrides (tests: small inputs).
        bool ext = !coded && keybits + 8 <= 56 && lowbits + 8 <= 18 &&
                   (double)n * std::exp2(-(double)keybits * h0 / 8.0) > 0.02 && (size_t)2 * align_up((size_t)n) <= (size_t)(n + 2) * sizeof(IdxT);
        if (const char *v = env("DQ_BUCKET_EXT")) ext = atoi(v) != 0 && keybits + 8 <= 56 && (size_t)2 * align_up((size_t)n) <= (size_t)(n + 2) * sizeof(IdxT);
        uint8_t *E[2] = {reinterpret_cast<uint8_t *>(w.Va), reinterpret_cast<uint8_t *>(w.Va) + align_up((size_t)n)};
        const int64_t ntiles = (n + C - 1) / C;
        uint32_t *ebits = reinterpret_cast<uint32_t *>(w.Vb);                  // zeroed by onesweep_sort_text_prepare
        TieCounters *ctr = reinterpret_cast<TieCounters *>(w.totals + 6);     // zero since run()
        BucketFlags *flags = reinterpret_cast<BucketFlags *>(&ctr->overflow);
        const uint64_t *text64 = reinterpret_cast<const uint64_t *>(w.text);
        // digit p of the bucket of suffix i is T[i + bbytes - 1 - p]
        hipLaunchKernelGGL(text_digit_offsets_kernel, dim3(bbytes), dim3(kBlock), 0, st,
                           (const int64_t *)w.bytehist, (const uint8_t *)w.text, n, bbytes, w.digit_offset);
        HIP_TRY(hipGetLastError());
        int rc = ext ? rank_pass_ext<IdxT, kTextPackedExt>(L, w, text64, (const uint8_t *)nullptr, K[1], E[1], n, 0, ib, ib + lowbits, keybits)
                     : rank_pass<IdxT, kTextPacked>(L, w, text64, (const IdxT *)nullptr, K[1], (IdxT *)nullptr, n, 0, kb, ib,
                                                    nullptr, nullptr, ib + lowbits, keybits);
        if (rc != DQ_OK) return rc;
        for (int p = 1; p < bbytes; ++p) {                   // pass p reads buffer p & 1 and writes the other
            rc = ext ? rank_pass_ext<IdxT, kKeysExt>(L, w, K[p & 1], E[p & 1], K[(p & 1) ^ 1], E[(p & 1) ^ 1], n, p, ib, ib + lowbits + 8 * p, keybits)
                     : rank_pass<IdxT, kKeys>(L, w, K[p & 1], (const IdxT *)nullptr, K[(p & 1) ^ 1], (IdxT *)nullptr, n, p, kb, ib,
                                              nullptr, nullptr, ib + lowbits + 8 * p, keybits);
            if (rc != DQ_OK) return rc;
        }
        uint64_t *Ks = K[bbytes & 1], *Kfree = K[(bbytes & 1) ^ 1];           // sorted words / the other buffer
        const uint8_t *Es = E[bbytes & 1];
        LAUNCH(L, DQ_K_BUCKET_SORT, ntiles, ntiles * 16 * 8,
               hipLaunchKernelGGL(bucket_bounds_kernel, dim3((unsigned)((ntiles + 1 + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                                  st, (const uint64_t *)Ks, n, ib + lowbits, C, X, ntiles, w.bkt_bounds, flags));
        if (c.ncu <= 0) {
            int v = 0;
            c.ncu = hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, c.dev) == hipSuccess && v > 0 ? v : 256;
        }
        if (ext) {
            LAUNCH(L, DQ_K_BUCKET_SORT, n, n * (9 + wb) + n / 8,              // persistent: one workgroup per CU
                   hipLaunchKernelGGL((bucket_sort_kernel<IdxT, true>), dim3((unsigned)std::min<int64_t>(ntiles, c.ncu)),
                                      dim3(kBktThreads), 0, st, (const uint64_t *)Ks, ib, lowbits,
                                      (const int64_t *)w.bkt_bounds, ntiles, d_sa, ebits, flags, Es));
        } else {
            LAUNCH(L, DQ_K_BUCKET_SORT, n, n * (8 + wb) + n / 8,              // persistent: one workgroup per CU
                   hipLaunchKernelGGL((bucket_sort_kernel<IdxT, false>), dim3((unsigned)std::min<int64_t>(ntiles, c.ncu)),
                                      dim3(kBktThreads), 0, st, (const uint64_t *)Ks, ib, lowbits,
                                      (const int64_t *)w.bkt_bounds, ntiles, d_sa, ebits, flags, (const uint8_t *)nullptr));
        }
        if (ext && env("DQ_TRACE")) fprintf(stderr, "[dq] bucketed round 0 with %d + 8 key bits per suffix (n=%lld)\n", keybits, (long long)n);
        bool overflow = false;
        fin_cap = n / 8;
        const int64_t hb = (keybits + (ext ? 8 : 0)) / 8;    // whole bytes the members of a tie group share
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

    // ---- round 0 as a sample sort (dq_split_round0.h): on return with *done the 64-bit keys lie sorted in K[1] and the
    //      suffixes in d_sa, as after the eight digit passes (which would have ended in K[0]).  *done = false: the
    //      overflow list ran full (a text made of a few heavy keys) -- nothing the digit passes need has been touched
    //      but the look-back state, which the caller zeroes again.
    int round0_split(uint64_t *K[2], bool coded, bool *done)
    {
        *done = false;
        if constexpr (sizeof(IdxT) != 4) {
            return DQ_OK;
        } else {
            if (!w.sp_top || !w.X || !w.RL) return DQ_OK;
            const int64_t cap = std::min<int64_t>(kFinCap, (n + 2) / (kSplitBuckets / 2));       // slot entries per bucket: twice the mean
            if (cap < 2) return DQ_OK;
            const uint32_t *t32 = reinterpret_cast<const uint32_t *>(w.text);
            const uint64_t *text64 = reinterpret_cast<const uint64_t *>(w.text);
            const uint16_t *ctab = (const uint16_t *)w.codetab;
            // idle buffers: the sample and its sort, then the bucket slots -- keys in K[0] (first half of the buckets) and X,
            // suffixes in Vb and Xs; pass A's pairs in (K[1], Va); the overflow arena -- n / 2 entries: the list of the
            // oversize buckets from its start, the pure list from its end -- in the inverse suffix array's and the run
            // lengths' memory; the overflow list's sort ping-pongs with the slot buffers, dead by then
            uint64_t *Ks[2] = {K[0], w.X};
            IdxT *Vs[2] = {w.Vb, w.Xs};
            const int64_t ovf_cap = (n / 2) & ~(int64_t)1;
            uint64_t *ovf_k[2] = {reinterpret_cast<uint64_t *>(w.ISA), K[0]};
            IdxT *ovf_v[2] = {reinterpret_cast<IdxT *>(w.RL), w.Vb};
            // (DQ_TRACE=2: the stream is drained after every phase and the phase named -- tests/manual/t_split_small.py)
            const bool dbg = env("DQ_TRACE") && atoi(env("DQ_TRACE")) >= 2;
            auto phase = [&](const char *what) -> int {
                if (!dbg) return DQ_OK;
                HIP_TRY(hipStreamSynchronize(st));
                fprintf(stderr, "[dq] split round 0: %s done\n", what);
                return DQ_OK;
            };
            const unsigned sgrid = (unsigned)((kSplitSample + kBlock - 1) / kBlock);
            if (coded) {
                LAUNCH(L, DQ_K_SPLIT_AUX, kSplitSample, kSplitSample * (20 + 8),
                       hipLaunchKernelGGL(sample_keys_kernel<true>, dim3(sgrid), dim3(kBlock), 0, st, t32, n, ctab, kSplitSample, Ks[0]));
            } else {
                LAUNCH(L, DQ_K_SPLIT_AUX, kSplitSample, kSplitSample * (12 + 8),
                       hipLaunchKernelGGL(sample_keys_kernel<false>, dim3(sgrid), dim3(kBlock), 0, st, t32, n, ctab, kSplitSample, Ks[0]));
            }
            int rc = phase("sample");
            if (rc != DQ_OK) return rc;
            int scur = 0;
            rc = onesweep_sort_pairs<IdxT>(L, w, Ks, Vs, kSplitSample, 64, scur);
            if (rc != DQ_OK) return rc;
            if ((rc = phase("sample sort")) != DQ_OK) return rc;
            // texts made of a few heavy keys (runs, short periods, tiny alphabets) would only fill the overflow list: the sorted
            // sample tells before anything is moved (one small kernel and a host round trip)
            HIP_TRY(hipMemsetAsync(w.sp_ctl, 0, sizeof(SplitCtl), st));
            hipLaunchKernelGGL(sample_heavy_kernel, dim3(sgrid), dim3(kBlock), 0, st, (const uint64_t *)Ks[scur], &w.sp_ctl->ovf_count);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(c.pinned, w.sp_ctl, sizeof(SplitCtl), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            const int64_t heavy = c.pinned[0];
            // (heavy keys have buckets of their own and are placed unsorted; but their copies beyond a slot wait in the
            // same arena of n / 2 entries as the oversize buckets: a text that is mostly heavy keys does not fit it)
            if (env("DQ_TRACE"))
                fprintf(stderr, "[dq] sample-sort round 0: %.1f %% of the sampled keys are copies of keys too heavy for a bucket%s\n",
                        100.0 * (double)heavy / (double)kSplitSample, heavy * 5 > 2 * kSplitSample ? " -- the digit passes instead" : "");
            if (heavy * 5 > 2 * kSplitSample && !(env("DQ_SPLIT") && atoi(env("DQ_SPLIT")) >= 2)) return DQ_OK;      // (DQ_SPLIT=2: the tests go on regardless)
            HIP_TRY(hipMemsetAsync(w.sp_cursor_b, 0, (size_t)kSplitBuckets * 8, st));
            HIP_TRY(hipMemsetAsync(w.sp_ctl, 0, sizeof(SplitCtl), st));
            // pass A's output: a virtual array of ~1.13 n entries -- the first n_main in (K[1], Va), the rest spilled into the
            // suffix array's memory (keys from its start, suffixes from its middle: 0.13 n x 12 bytes of its 4 n)
            const int64_t n_main = n & ~(int64_t)1;
            uint64_t *spill_k = reinterpret_cast<uint64_t *>(d_sa);
            IdxT *spill_v = d_sa + (n / 2 + 1);
            if (n / 8 + 1025 * (int64_t)kSplitTop + 1024 > n / 4) return DQ_OK;                 // (the spill -- sum of the regions' room minus n -- must fit n / 4 entries: keys below the middle of the array, suffixes above)
            LAUNCH(L, DQ_K_SPLIT_AUX, kSplitBuckets, (int64_t)kSplitBuckets * 16,
                   hipLaunchKernelGGL(make_splitters_kernel, dim3(kSplitBuckets / kBlock), dim3(kBlock), 0, st, (const uint64_t *)Ks[scur], w.sp_top, w.sp_sub,
                                      w.sp_low, w.sp_pure);
                   hipLaunchKernelGGL(split_estimate_kernel, dim3(1), dim3(kSplitTop), 0, st, (const uint64_t *)Ks[scur], (const uint64_t *)w.sp_top, n, w.sp_off,
                                      w.sp_cursor_a));
            if ((rc = phase("splitters, region estimates")) != DQ_OK) return rc;
            const unsigned grid_a = (unsigned)((n + kSplitTileA - 1) / kSplitTileA);
            if (coded) {
                LAUNCH(L, DQ_K_SPLIT_PASS, n, n * (1 + 8 + wb),
                       hipLaunchKernelGGL((split_pass_kernel<IdxT, true, true>), dim3(grid_a), dim3(kSplitThreads), 0, st, text64, (const IdxT *)nullptr,
                                          (const uint64_t *)nullptr, (const IdxT *)nullptr, (int64_t)0, n, (const uint64_t *)w.sp_top, w.sp_cursor_a,
                                          (const int64_t *)w.sp_off, (const unsigned long long *)nullptr, (const uint32_t *)w.sp_tile_first, K[1], w.Va, spill_k, spill_v,
                                          n_main, (uint64_t *)nullptr, (IdxT *)nullptr, (int64_t)0, w.sp_ctl, ctab));
            } else {
                LAUNCH(L, DQ_K_SPLIT_PASS, n, n * (1 + 8 + wb),
                       hipLaunchKernelGGL((split_pass_kernel<IdxT, true, false>), dim3(grid_a), dim3(kSplitThreads), 0, st, text64, (const IdxT *)nullptr,
                                          (const uint64_t *)nullptr, (const IdxT *)nullptr, (int64_t)0, n, (const uint64_t *)w.sp_top, w.sp_cursor_a,
                                          (const int64_t *)w.sp_off, (const unsigned long long *)nullptr, (const uint32_t *)w.sp_tile_first, K[1], w.Va, spill_k, spill_v,
                                          n_main, (uint64_t *)nullptr, (IdxT *)nullptr, (int64_t)0, w.sp_ctl, ctab));
            }
            if ((rc = phase("pass A")) != DQ_OK) return rc;
            // (pass B's grid is an upper bound -- every top bucket may end in a ragged tile; the workgroups beyond the plan's count leave at once)
            const unsigned grid_b = (unsigned)(n / kSplitTileB + kSplitTop);
            LAUNCH(L, DQ_K_SPLIT_PASS, n, n * 2 * (8 + wb),
                   hipLaunchKernelGGL(split_plan_kernel, dim3(1), dim3(kSplitTop), 0, st, (const unsigned long long *)w.sp_cursor_a, (const int64_t *)w.sp_off, w.sp_cnt_a,
                                      w.sp_tile_first, w.sp_ctl);
                   hipLaunchKernelGGL((split_pass_kernel<IdxT, false, false>), dim3(grid_b), dim3(kSplitThreads), 0, st, (const uint64_t *)K[1], (const IdxT *)w.Va,
                                      (const uint64_t *)spill_k, (const IdxT *)spill_v, n_main, n, (const uint64_t *)w.sp_sub, w.sp_cursor_b, (const int64_t *)w.sp_off,
                                      (const unsigned long long *)w.sp_cnt_a, (const uint32_t *)w.sp_tile_first, Ks[0], Vs[0], Ks[1], Vs[1],
                                      cap, ovf_k[0], ovf_v[0], ovf_cap, w.sp_ctl, ctab, (const uint8_t *)w.sp_pure));
            if ((rc = phase("pass B")) != DQ_OK) return rc;
            LAUNCH(L, DQ_K_SPLIT_AUX, kSplitBuckets, (int64_t)kSplitBuckets * 36,
                   hipLaunchKernelGGL(bucket_sum_kernel, dim3(kScanBlocks), dim3(kScanThreads), 0, st, (const unsigned long long *)w.sp_cursor_b, cap,
                                      (const uint8_t *)w.sp_pure, w.sp_part);
                   hipLaunchKernelGGL(bucket_scan_kernel, dim3(kScanBlocks), dim3(kScanThreads), 0, st, (const unsigned long long *)w.sp_cursor_b, cap,
                                      (const uint8_t *)w.sp_pure, (const ScanPart *)w.sp_part, w.sp_out_base, w.sp_ovf_src, w.sp_ovf_dst, w.sp_ctl));
            if ((rc = phase("bucket scan")) != DQ_OK) return rc;
            // two geometries by bucket size (dq_split_round0.h: what a CU gets through is set by how many buckets it holds at
            // once): <= 1024 entries with 16 KB of LDS, eight workgroups per CU; the others with 31 KB, five.  The last launch
            // also moves the oversize buckets to the overflow list.
            const bool two = cap > kFinSmallCap && !env("DQ_SPLIT_ONE_CLASS");
            LAUNCH(L, DQ_K_SPLIT_FINISH, n, n * 2 * (8 + wb),
                   if (two || cap <= kFinSmallCap)
                       hipLaunchKernelGGL((bucket_finish_kernel<IdxT, 256, 4>), dim3(kSplitBuckets), dim3(256), 0, st, (const uint64_t *)Ks[0], (const IdxT *)Vs[0],
                                          (const uint64_t *)Ks[1], (const IdxT *)Vs[1], cap, (int64_t)0, (int64_t)kFinSmallCap, !two,
                                          (const unsigned long long *)w.sp_cursor_b, (const int64_t *)w.sp_out_base, K[1], d_sa, ovf_k[0], ovf_v[0], ovf_cap, w.sp_ctl, (const uint8_t *)w.sp_pure);
                   if (cap > kFinSmallCap)
                       hipLaunchKernelGGL((bucket_finish_kernel<IdxT, 256, 8>), dim3(kSplitBuckets), dim3(256), 0, st, (const uint64_t *)Ks[0], (const IdxT *)Vs[0],
                                          (const uint64_t *)Ks[1], (const IdxT *)Vs[1], cap, (int64_t)(two ? kFinSmallCap : 0), (int64_t)kFinCap, true,
                                          (const unsigned long long *)w.sp_cursor_b, (const int64_t *)w.sp_out_base, K[1], d_sa, ovf_k[0], ovf_v[0], ovf_cap, w.sp_ctl, (const uint8_t *)w.sp_pure));
            HIP_TRY(hipMemcpyAsync(c.pinned, w.sp_ctl, sizeof(SplitCtl), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            const int64_t ovf = c.pinned[0], ovf_buckets = c.pinned[1], npure = c.pinned[4];
            const bool abandon = c.pinned[3] != 0 || ovf + npure > ovf_cap;            // (the two lists share one arena, from either end)
            if (env("DQ_TRACE"))
                fprintf(stderr, "[dq] sample-sort round 0 (%s keys, %d buckets of <= %lld): %lld suffixes in %lld oversize buckets, %lld copies of heavy keys placed unsorted%s\n",
                        coded ? "coded" : "raw", kSplitBuckets, (long long)cap, (long long)ovf, (long long)ovf_buckets, (long long)npure,
                        abandon ? " -- overflow lists full, given up" : "");
            if (abandon) return DQ_OK;
            if (npure > 0) {
                LAUNCH(L, DQ_K_SPLIT_AUX, npure, npure * 2 * (8 + wb),
                       hipLaunchKernelGGL(pure_place_kernel<IdxT>, dim3((unsigned)((npure + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, npure, ovf_cap,
                                          (const uint64_t *)ovf_k[0], (const IdxT *)ovf_v[0], (const int64_t *)w.sp_out_base, (const uint64_t *)w.sp_low, K[1], d_sa));
            }
            if (ovf > 0) {
                int xcur = 0;
                rc = onesweep_sort_pairs<IdxT>(L, w, ovf_k, ovf_v, ovf, 64, xcur);
                if (rc != DQ_OK) return rc;
                LAUNCH(L, DQ_K_SPLIT_AUX, ovf, ovf * 2 * (8 + wb),
                       hipLaunchKernelGGL(overflow_place_kernel<IdxT>, dim3((unsigned)((ovf + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, ovf, ovf_buckets,
                                          (const int64_t *)w.sp_ovf_src, (const int64_t *)w.sp_ovf_dst,
                                          (const uint64_t *)ovf_k[xcur], (const IdxT *)ovf_v[xcur], K[1], d_sa));
            }
            *done = true;
            return DQ_OK;
        }
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
        bool hist_deferred = false;
        rc = onesweep_sort_text_prepare<IdxT>(L, c, w, n, &kb, &packed, &coded, text_src, &hist_deferred);
        if (rc != DQ_OK) return rc;
        // (c.pinned still holds the byte histogram, the k-gram sample and the long-run flag of text_hist_kernel)
        // (run lengths + the run-order round cost about one doubling round: worth it where a good part of the text lies
        // in runs -- padded images, sparse files; measured on the image's shared libraries, whose long tie tails are
        // code repeated for several targets, not runs: 5-20 % slower with it.  1/16 of the text in 16-byte chunks of one value)
        runs_wanted = sizeof(IdxT) == 4 && c.pinned[256 + 8] != 0 && n >= (1 << 16) && c.pinned[256 + 9] * 16 * 16 >= n;
        long_run_seen = sizeof(IdxT) == 4 && c.pinned[256 + 8] != 0 && n >= (1 << 16);
        // (the late rounds also take stretches that repeat with a period > 1, which the histogram pass does not see:
        // large groups that stop shrinking are what calls them)
        late_runs_possible = sizeof(IdxT) == 4 && n >= (1 << 16);
        if (period_hint > 0 && sizeof(IdxT) == 4 && n >= (1 << 16)) runs_wanted = true;      // (the caller has seen the stretches)
        if (const char *v = env("DQ_RUNS")) { runs_wanted = sizeof(IdxT) == 4 && atoi(v) != 0; late_runs_possible = late_runs_possible && atoi(v) != 0; }
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
            bool split_done = false;
            // (a text that has a good part of itself in runs -- runs_wanted: padded images, sparse files -- is a text of heavy
            // keys: the sorted sample would only say so, 0.5 ms later)
            const bool split_wanted = split_round0_wanted<IdxT>(n, packed, kb, coded);
            if (split_wanted && (!runs_wanted || env("DQ_SPLIT"))) {
                rc = round0_split(K, coded, &split_done);
                if (rc != DQ_OK) return rc;
            }
            if (split_wanted && !split_done) {
                // not taken after all, or given up: the digit passes, with the state they expect -- their digit offsets (the
                // coded keys' histograms were left out for the sample sort's sake; the sorts of the sample and of the overflow
                // list have used the table since) and look-back state
                if (coded) {
                    rc = launch_coded_hist<IdxT>(L, w, n);
                    if (rc != DQ_OK) return rc;
                } else {
                    hipLaunchKernelGGL(text_digit_offsets_kernel, dim3(kb), dim3(kBlock), 0, st,
                                       (const int64_t *)w.bytehist, (const uint8_t *)w.text, n, kb, w.digit_offset);
                    HIP_TRY(hipGetLastError());
                }
                rc = prepare_status<IdxT>(L, w, n, kb);
                if (rc != DQ_OK) return rc;
            }
            if (split_done) {
                cur = 1;
            } else {
                rc = onesweep_sort_text_passes<IdxT>(L, w, n, K, V, kb, packed, d_sa, cur, nullptr, nullptr, coded);
                if (rc != DQ_OK) return rc;
            }
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
        // (below 8 MiB the sample's host round trip costs more than a wrong guess: 8-byte pair keys were chosen because the
        // text repeats itself, so "many ties" is the guess, and the inverse suffix array of a short text is cheap either way)
        if (n >= (1 << 16) && n < (8 << 20) && !packed && kb == 8 && !env("DQ_SAMPLE_TIES")) {
            predict_dense = true;
        } else if (n >= (1 << 16) && !packed && kb == 8) {
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

    // Where a round over the list (A, As)[0, mm) puts what it produces.  T (still tied, next list) always grows from the
    // start of the partner pair (B, Bs).  A list of at most n/2 entries leaves room there for L (members of large
    // groups, from n/2) and U (rank updates, downward from n + 2); the radix sort of L ping-pongs with the idle second
    // half of A.  A longer list (real binaries after raw 8-byte keys: 2/3 of the suffixes tied) would run T into L, so L
    // and U go to the third buffer (w.X, w.Xs): an entry goes to L or is flagged for T / U, never both, so L (upward) and
    // U (downward) share it; L's sort ping-pongs with A itself, which is dead once the round's updates are applied, and
    // so never ends in B, where its survivors are appended behind T.
    struct RoundLayout {
        uint64_t *t_rank; IdxT *t_suf;
        uint64_t *l_key; IdxT *l_suf;
        uint64_t *u_end; IdxT *u_suf_end;
        uint64_t *l_partner; IdxT *l_partner_suf;
    };
    RoundLayout round_layout(int64_t mm, uint64_t *A, IdxT *As, uint64_t *B, IdxT *Bs) const
    {
        const int64_t half = sg_half(), top = sg_top();
        if (!wide_list(mm)) return {B, Bs, B + half, Bs + half, B + top, Bs + top, A + half, As + half};
        return {B, Bs, w.X, w.Xs, w.X + top, w.Xs + top, A, As};
    }

    // update entries of the LDS-class rounds as single words (rank << ib | suffix) where two indices fit one
    int upd_ib() const
    {
        const int ib = bit_length((uint64_t)(n - 1));
        return (2 * ib <= 64 && !env("DQ_NO_UPD_WORDS")) ? ib : 0;
    }

    // ISA[s] = new rank for the mU entries a round left in U (stored downward from B + top / Bs + top).  A long list of
    // update words is first binned by the top 8 bits of the suffix with one word pass of the radix sorter (into the
    // buffer of the round's input list, which is dead by now), so that the 4-byte writes of the moment fall into one
    // 1/256 of the array (isa_update_words_kernel).  Measured on 256 MiB of enwik-style text, first doubling round,
    // 75 M updates: 2.75 ms as random writes; one pass 0.35 + 1.7 ms; two passes (16 bits, DQ_UPD_BIN=2) 0.75 + 1.07 ms
    // -- the passes eat most of what the writes gain, and lists of a few million entries gain nothing.
    static constexpr int64_t kUpdBinMin = 1ll << 24;
    int apply_rank_updates(uint64_t *A, IdxT *As, uint64_t *u_end, IdxT *u_suf_end, int64_t mU, int u_ib)
    {
        // (the entries lie downward from u_end / u_suf_end: B + top of the round's partner pair, or the third buffer's)
        int passes = env("DQ_UPD_BIN") ? std::max(0, std::min(2, atoi(env("DQ_UPD_BIN")))) : 1;
        const int64_t min_len = env("DQ_UPD_BIN_MIN") ? std::max(1, atoi(env("DQ_UPD_BIN_MIN"))) : kUpdBinMin;
        // (the second pass writes into the dead suffix buffer of the round's input list, as 64-bit words)
        const bool second_fits = (size_t)(mU + 1) * 8 <= (size_t)(n + 2) * sizeof(IdxT);
        // Dense updates (at least 1/4 of the array moves: the first round of a text-like input or a binary) are binned by
        // the top 16 bits of the suffix and applied span by span inside LDS (isa_update_window_kernel): 256 MiB of text,
        // 75 M updates: 0.4 + 1.45 ms -> 0.7 + 0.55 ms (the sort 29.87 -> 29.44 ms).  The window kernel moves the whole
        // array once whatever the number of updates, and the second pass costs what it costs: at 1/6 of the array
        // (second round of libtorch_cpu.so, 20.7 M of 134 M) the one-pass form is ahead again.  DQ_UPD_WINDOW = 0 | 1 overrides.
        bool window = u_ib >= 16 && 2 * u_ib <= 63 && second_fits && mU >= min_len && mU * 4 >= n && !env("DQ_UPD_BIN");
        if (const char *v = env("DQ_UPD_WINDOW")) window = atoi(v) != 0 && u_ib >= 16 && 2 * u_ib <= 63 && second_fits;
        if (window) passes = 2;
        if (u_ib < 16 || mU < min_len || (passes == 2 && !second_fits)) passes = window ? 2 : 0;
        if (passes == 0) {
            LAUNCH(L, DQ_K_ISA_UPDATE, mU, mU * (u_ib ? 8 + wb : 8 + wb + wb),
                   hipLaunchKernelGGL(isa_update_kernel<IdxT>, dim3(grid_for(mU)), dim3(kBlock), 0, st,
                                      (const uint64_t *)u_end, (const IdxT *)u_suf_end, mU, w.ISA,
                                      (const SmallGroupCounters *)nullptr, u_ib));
            return DQ_OK;
        }
        // the word list starts on a 16-byte boundary (key loads of the histogram kernel): one filler word in
        // front of it if need be -- all ones, which no real word is (bit 63 of rank << ib | suffix is clear): the update
        // kernels skip exactly that word (not "suffix field >= n": for n = 2^ib the filler's field reads n - 1)
        uint64_t *U = u_end - mU;
        int64_t cnt = mU;
        if (reinterpret_cast<uintptr_t>(U) & 8) {
            --U; ++cnt;
            HIP_TRY(hipMemsetAsync(U, 0xff, 8, st));
        }
        const int sh0 = passes == 2 ? u_ib - 16 : u_ib - 8;
        const int blocks = (int)std::min<int64_t>(kHistBlocks, ((cnt >> 1) + kHistThreads - 1) / kHistThreads + 1);
        int rc = L.begin(DQ_K_RADIX_HIST, cnt, cnt * 8);
        if (rc != DQ_OK) return rc;
        if (passes == 2) launch_hist<2>(st, blocks, U, cnt, w.hist_partial, sh0);
        else launch_hist<1>(st, blocks, U, cnt, w.hist_partial, sh0);
        hipLaunchKernelGGL(radix_hist_scan_kernel, dim3(passes), dim3(kHistScanThreads), 0, st,
                           (const unsigned long long *)w.hist_partial, w.digit_offset);
        HIP_TRY(hipGetLastError());
        rc = L.end();
        if (rc != DQ_OK) return rc;
        rc = prepare_status<IdxT>(L, w, cnt, passes);
        if (rc != DQ_OK) return rc;
        uint64_t *W1 = A, *W2 = reinterpret_cast<uint64_t *>(As);
        rc = rank_pass<IdxT, kKeys>(L, w, U, (const IdxT *)nullptr, W1, (IdxT *)nullptr, cnt, 0, 8, u_ib, nullptr, nullptr, sh0);
        if (rc != DQ_OK) return rc;
        const uint64_t *binned = W1;
        if (passes == 2) {
            rc = rank_pass<IdxT, kKeys>(L, w, W1, (const IdxT *)nullptr, W2, (IdxT *)nullptr, cnt, 1, 8, u_ib, nullptr, nullptr, sh0 + 8);
            if (rc != DQ_OK) return rc;
            binned = W2;
        }
        if (window) {
            // spans of 4096 suffixes (32768 where a bin of the 16-bit binning is wider than that), as isa_from_pairs_kernel
            const int span_log2 = u_ib - 16 <= 12 ? 12 : 15;
            const int64_t nspans = (n + ((int64_t)1 << span_log2) - 1) >> span_log2;
            if ((size_t)(nspans + 1) * 8 > (size_t)((size_t)n / 4096 + 4) * 8) return fail(DQ_ERR_HIP, "window bounds do not fit");
            LAUNCH(L, DQ_K_ISA_UPDATE, cnt, cnt * 8 + 2 * n * wb,
                   hipLaunchKernelGGL(window_bounds_kernel, dim3((unsigned)((nspans + 1 + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, binned, cnt,
                                      u_ib, span_log2, nspans, w.bkt_bounds);
                   if (span_log2 == 12)
                       hipLaunchKernelGGL((isa_update_window_kernel<IdxT, 4096>), dim3((unsigned)nspans), dim3(kPairThreads), 0, st, binned,
                                          (const int64_t *)w.bkt_bounds, n, u_ib, w.ISA);
                   else
                       hipLaunchKernelGGL((isa_update_window_kernel<IdxT, 32768>), dim3((unsigned)nspans), dim3(kPairThreads), 0, st, binned,
                                          (const int64_t *)w.bkt_bounds, n, u_ib, w.ISA));
            return DQ_OK;
        }
        LAUNCH(L, DQ_K_ISA_UPDATE, cnt, cnt * (8 + wb),
               hipLaunchKernelGGL(isa_update_words_kernel<IdxT>, dim3(grid_for(cnt)), dim3(kBlock), 0, st, binned, cnt, u_ib, n, w.ISA));
        return DQ_OK;
    }

    int doubling_round_small(int kbits)
    {
        uint64_t *A = Kr[rcur], *B = Kr[rcur ^ 1];
        IdxT *As = Vr[rcur], *Bs = Vr[rcur ^ 1];
        const RoundLayout lay = round_layout(m, A, As, B, Bs);
        SmallGroupCounters *ctr = reinterpret_cast<SmallGroupCounters *>(w.totals + 4);
        HIP_TRY(hipMemsetAsync(ctr, 0, sizeof(SmallGroupCounters), st));
        const bool cap32 = m < kSgShortList;           // (cap 32 on long lists measured: radix -2.4 ms, this kernel +2.8 ms)
        // The groups of up to mid_g members are finished inside LDS by mid_group_round_kernel (dq_mid_groups.h); only
        // longer ones take the radix passes.  DQ_MID_GROUPS=0: the two-class scheme of before (groups of <= 8, or
        // <= 32 on short lists, in small_group_round_kernel; everything else through the radix passes).
        const int mid_g = mid_group_cap(m);
        const bool use_mid = mid_g > 0;
        // the radix list's composite keys with rank >> log2(mid_g) as the rank field (dq_mid_groups.h): 57 -> 48 bits for
        // 256 MiB of text, 8 -> 6 digit passes per large-group sort.  DQ_NO_L_SHIFT=1: the full rank, as before round 5.
        const int l_shift = (use_mid && !env("DQ_NO_L_SHIFT")) ? (mid_g >= 1024 ? 10 : mid_g >= 512 ? 9 : 8) : 0;
        if (use_mid) {
            int rc = launch_mid_round(mid_g, m, A, As, lay, h, kbits, ctr, nullptr, m * (8 + wb + wb + wb + 8 + wb), l_shift);
            if (rc != DQ_OK) return rc;
            first_rank32 = nullptr;
        } else if (cap32) {
            constexpr int kTile = sg_tile<kSgMaxGShort>();
            LAUNCH(L, DQ_K_SMALL_ROUND, m, m * (8 + wb + wb + wb + 8 + wb),
                   hipLaunchKernelGGL((small_group_round_kernel<IdxT, kSgMaxGShort>), dim3((unsigned)((m + kTile - 1) / kTile)),
                                      dim3(kSgThreads), 0, st, (const uint64_t *)A, (const IdxT *)As,
                                      (const IdxT *)w.ISA, m, n, h, kbits, d_sa, lay.t_rank, lay.t_suf, lay.l_key, lay.l_suf, lay.u_end,
                                      lay.u_suf_end, ctr, (const SmallGroupCounters *)nullptr));
        } else {
            constexpr int kTile = sg_tile<kSgMaxG>();
            LAUNCH(L, DQ_K_SMALL_ROUND, m, m * (8 + wb + wb + wb + 8 + wb),
                   hipLaunchKernelGGL((small_group_round_kernel<IdxT, kSgMaxG>), dim3((unsigned)((m + kTile - 1) / kTile)),
                                      dim3(kSgThreads), 0, st, (const uint64_t *)A, (const IdxT *)As,
                                      (const IdxT *)w.ISA, m, n, h, kbits, d_sa, lay.t_rank, lay.t_suf, lay.l_key, lay.l_suf, lay.u_end,
                                      lay.u_suf_end, ctr, (const SmallGroupCounters *)nullptr));
        }
        HIP_TRY(hipMemcpyAsync(c.pinned, ctr, sizeof(SmallGroupCounters), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        const int64_t m1 = c.pinned[0] & 0xffffffffll, mU = (int64_t)((uint64_t)c.pinned[0] >> 32), mL = c.pinned[1];
        if (env("DQ_TRACE"))
            fprintf(stderr, "[dq] %s round h=%lld m=%lld%s -> tied %lld, to radix %lld, moved %lld\n",
                    use_mid ? "mid-group" : "small", (long long)h,
                    (long long)m, wide_list(m) ? " (wide)" : "", (long long)m1, (long long)mL, (long long)mU);
        if (mU > 0) {
            const int rc = apply_rank_updates(A, As, lay.u_end, lay.u_suf_end, mU, use_mid ? upd_ib() : 0);
            if (rc != DQ_OK) return rc;
        }
        int64_t mLs = 0;
        if (mL > 0) {
            // radix ping-pong partner: the unused second half of X's own buffers (a wide list: all of them, see round_layout)
            uint64_t *Kx[2] = {lay.l_key, lay.l_partner};
            IdxT *Vx[2] = {lay.l_suf, lay.l_partner_suf};
            int xcur = 0;
            int rc = onesweep_sort_pairs<IdxT>(L, w, Kx, Vx, mL, std::max(1, kbits + rbits - l_shift), xcur, l_shift);
            if (rc != DQ_OK) return rc;
            rc = rebucket<IdxT, false, true, true>(L, c, w, Kx[xcur], (const IdxT *)Vx[xcur], mL, kbits, l_shift, d_sa,
                                                   B + m1, Bs + m1, &mLs, 0, l_shift);
            if (rc != DQ_OK) return rc;
        }
        rcur ^= 1;
        m = m1 + mLs;
        // groups only ever split: once nothing went to the radix list, every group fits this round's cap
        if (mL == 0) small_cap = use_mid ? mid_g : cap32 ? kSgMaxGShort : kSgMaxG;
        only_small_groups = mL == 0;
        prev_large = last_large;
        last_large = mL;
        return DQ_OK;
    }

    // ---- kSgChain small-group rounds back to back, lengths handed over on the device.  Only valid once
    //      every group has <= small_cap members (nothing goes to the radix list any more); the grids are sized for
    //      the current m, which is an upper bound for all later rounds.
    // (a list within reach of the tail kernel -- dq_tail.h: the rest of the sort in one launch once <= tail_max suffixes
    // are tied -- runs two rounds per host round trip instead of eight, so that the hand-over is not slept through)
    int chain_len = kSgChain;
    bool tail_behind_chain = false;
    int spec_misses = 0;                 // chains that had the tail kernel behind them so far
    int doubling_rounds_small_chain()
    {
        return small_cap > kSgMaxGShort ? small_chain<0>() : small_cap == kSgMaxGShort ? small_chain<kSgMaxGShort>() : small_chain<kSgMaxG>();
    }

    // one round of mid_group_round_kernel<kG> on the list (A, As)[0, mm); prev: the previous chained round's counters
    // steps: elements of the key tuple (dq_mid_groups.h, kSteps): 1, or 3 on the chained rounds of short lists
    static constexpr int kChainSteps = 3;
    int launch_mid_round(int g, int64_t mm, const uint64_t *A, const IdxT *As, const RoundLayout &lay, int64_t hh, int kbits,
                         SmallGroupCounters *ctr, const SmallGroupCounters *prev, int64_t alg_bytes, int l_shift = 0, int steps = 1)
    {
        const int64_t tile = g == 256 ? mg_tile<256>() : g == 512 ? mg_tile<512>() : mg_tile<1024>();
        const dim3 grid((unsigned)((mm + tile - 1) / tile));
        auto go = [&](auto kern) -> int {
            LAUNCH(L, DQ_K_MID_ROUND, mm, alg_bytes,
                   hipLaunchKernelGGL(kern, grid, dim3(kMgThreads), 0, st, A, As, (const IdxT *)w.ISA, mm, n, hh, kbits, d_sa, lay.t_rank, lay.t_suf,
                                      lay.l_key, lay.l_suf, lay.u_end, lay.u_suf_end, ctr, prev, rl(), (const uint8_t *)w.text, run_order,
                                      first_rank32, upd_ib(), l_shift));
            return DQ_OK;
        };
        if constexpr (sizeof(IdxT) == 4) {               // (64-bit key tuples do not fit the LDS: int64 sorts keep one step)
            if (steps == kChainSteps)
                return g == 256 ? go(mid_group_round_kernel<IdxT, 256, kChainSteps>) : g == 512 ? go(mid_group_round_kernel<IdxT, 512, kChainSteps>)
                                                                                                 : go(mid_group_round_kernel<IdxT, 1024, kChainSteps>);
        }
        return g == 256 ? go(mid_group_round_kernel<IdxT, 256>) : g == 512 ? go(mid_group_round_kernel<IdxT, 512>)
                                                                            : go(mid_group_round_kernel<IdxT, 1024>);
    }

    template <int kCap>
    int small_chain()
    {
        const int64_t m_in = m;
        HIP_TRY(hipMemsetAsync(w.sg_ctr, 0, (kSgChain + 2) * sizeof(SmallGroupCounters), st));
        int64_t hr = h;
        // Short lists (launch-bound rounds) compare (kChainSteps + 1) h bytes a round instead of 2 h: the key of a member is
        // the tuple of the ranks h, 2h, 3h bytes further on.  Not with run lengths in force (a member inside a run takes
        // ONE rank, behind its run) and not on long lists, whose rounds are bound by the sectors their gathers move.
        // DQ_CHAIN_STEPS = 1 | 3 overrides.
        int steps = (kCap == 0 && sizeof(IdxT) == 4 && m_in < kSgShortList && !runs_on) ? kChainSteps : 1;
        if (const char *v = env("DQ_CHAIN_STEPS")) steps = (atoi(v) >= kChainSteps && kCap == 0 && sizeof(IdxT) == 4 && !runs_on) ? kChainSteps : 1;
        for (int r = 0; r < chain_len; ++r) {
            uint64_t *A = Kr[rcur], *B = Kr[rcur ^ 1];
            IdxT *As = Vr[rcur], *Bs = Vr[rcur ^ 1];
            // (every round of the chain is laid out for the list length the chain starts with: an upper bound of the others')
            const RoundLayout lay = round_layout(m_in, A, As, B, Bs);
            const int kbits = std::min(bit_length((uint64_t)(n - 1) + (uint64_t)hr), 64 - rbits);   // (no radix keys are made)
            if constexpr (kCap == 0) {                                       // (every group has <= small_cap members here)
                const int rc = launch_mid_round(small_cap, m_in, A, As, lay, hr, kbits, w.sg_ctr + r,
                                                r == 0 ? (const SmallGroupCounters *)nullptr : w.sg_ctr + r - 1, 0, 0, steps);
                if (rc != DQ_OK) return rc;
            } else {
                constexpr int kTile = sg_tile<kCap>();                       // (every group has <= kCap members here)
                LAUNCH(L, DQ_K_SMALL_ROUND, m_in, 0,
                       hipLaunchKernelGGL((small_group_round_kernel<IdxT, kCap>), dim3((unsigned)((m_in + kTile - 1) / kTile)),
                                          dim3(kSgThreads), 0, st, (const uint64_t *)A, (const IdxT *)As,
                                          (const IdxT *)w.ISA, m_in, n, hr, kbits, d_sa, lay.t_rank, lay.t_suf, lay.l_key, lay.l_suf, lay.u_end,
                                          lay.u_suf_end, w.sg_ctr + r, r == 0 ? (const SmallGroupCounters *)nullptr : w.sg_ctr + r - 1));
            }
            LAUNCH(L, DQ_K_ISA_UPDATE, m_in, 0,
                   hipLaunchKernelGGL(isa_update_kernel<IdxT>, dim3(grid_for(m_in)), dim3(kBlock), 0, st,
                                      (const uint64_t *)lay.u_end, (const IdxT *)lay.u_suf_end, (int64_t)0, w.ISA,
                                      (const SmallGroupCounters *)(w.sg_ctr + r), kCap == 0 ? upd_ib() : 0));
            rcur ^= 1;
            hr *= (kCap == 0 ? steps + 1 : 2);
        }
        // A list within reach of the tail kernel: it is launched right behind the chain, on the list and at the depth the
        // chain leaves, and reads the list's length on the device -- if that is <= kTailMax the sort ends in this same
        // host round trip, otherwise the kernel does nothing (no host round trip spent on finding out).
        TailResult *res = reinterpret_cast<TailResult *>(w.sg_ctr + kSgChain);
        const bool spec_tail = tail_behind_chain;
        if (spec_tail) {
            LAUNCH(L, DQ_K_SMALL_ROUND, m_in, 0,
                   launch_tail((const uint64_t *)Kr[rcur], (const IdxT *)Vr[rcur], (int)0, hr, res,
                               (const unsigned long long *)&w.sg_ctr[chain_len - 1].tied_moved));
        }
        HIP_TRY(hipMemcpyAsync(c.pinned, w.sg_ctr, (kSgChain + 2) * sizeof(SmallGroupCounters), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        int64_t cur_m = m_in;
        for (int r = 0; r < chain_len; ++r) {
            if (c.pinned[2 * r + 1] != 0) return fail(DQ_ERR_HIP, "chained small-group round met a large group");
            if (cur_m > 0) { t_info[0] += 1; t_info[2] += cur_m; }
            cur_m = c.pinned[2 * r] & 0xffffffffll;
        }
        m = cur_m;
        h = hr;
        if (spec_tail && cur_m > 0) {
            const int64_t rounds = c.pinned[2 * kSgChain], entries = c.pinned[2 * kSgChain + 1], left = c.pinned[2 * kSgChain + 2];
            if (cur_m <= kTailMax) {                       // the kernel took the list
                if (env("DQ_TRACE"))
                    fprintf(stderr, "[dq] tail behind a chain of %d: %lld tied suffixes from h=%lld on, %lld rounds (%lld list entries in all)\n",
                            chain_len, (long long)cur_m, (long long)hr, (long long)rounds, (long long)entries);
                if (left != 0) return fail(DQ_ERR_HIP, "tail rounds did not finish (round bound hit)");
                t_info[0] += rounds;
                t_info[2] += entries;
                m = 0;
            }
        }
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

    // (the three-step form of the tail kernel where its 32-bit keys hold rank + h and no run lengths are in force)
    void launch_tail(const uint64_t *rank, const IdxT *suf, int mm, int64_t hh, TailResult *res, const unsigned long long *m_dev)
    {
        const bool three = sizeof(IdxT) == 4 && !runs_on && !(env("DQ_CHAIN_STEPS") && atoi(env("DQ_CHAIN_STEPS")) < kChainSteps);
        if constexpr (sizeof(IdxT) == 4) {
            if (three) {
                hipLaunchKernelGGL((tail_rounds_kernel<IdxT, uint32_t, kChainSteps>), dim3(1), dim3(kTailThreads), 0, st, rank, suf, mm, n, hh,
                                   w.ISA, d_sa, (const uint32_t *)nullptr, res, m_dev);
                return;
            }
        }
        (void)three;
        hipLaunchKernelGGL((tail_rounds_kernel<IdxT, uint64_t, 1>), dim3(1), dim3(kTailThreads), 0, st, rank, suf, mm, n, hh, w.ISA, d_sa,
                           rl(), res, m_dev);
    }

    // ---- the last rounds in one launch (dq_tail.h): at most kTailMax tied suffixes, one workgroup, the list in LDS
    int tail_rounds()
    {
        TailResult *res = reinterpret_cast<TailResult *>(w.sg_ctr + kSgChain);
        static_assert(sizeof(TailResult) <= 2 * sizeof(SmallGroupCounters), "the result sits behind the chain counters");
        HIP_TRY(hipMemsetAsync(res, 0, sizeof(TailResult), st));
        LAUNCH(L, DQ_K_SMALL_ROUND, m, m * (8 + wb + wb + 64),
               launch_tail((const uint64_t *)Kr[rcur], (const IdxT *)Vr[rcur], (int)m, h, res, (const unsigned long long *)nullptr));
        HIP_TRY(hipMemcpyAsync(c.pinned, res, sizeof(TailResult), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        const int64_t rounds = c.pinned[0], entries = c.pinned[1], left = c.pinned[2];
        if (env("DQ_TRACE"))
            fprintf(stderr, "[dq] tail: %lld tied suffixes from h=%lld on, %lld rounds in one launch (%lld list entries in all)\n",
                    (long long)m, (long long)h, (long long)rounds, (long long)entries);
        if (left != 0) return fail(DQ_ERR_HIP, "tail rounds did not finish (round bound hit)");
        t_info[0] += rounds;
        t_info[2] += entries;
        m = 0;
        return DQ_OK;
    }

    // RL[i] = number of equal bytes the text has from position i on (dq_runs.h): chunk pass, carry across chunks, final pass
    int compute_run_lengths(int period = 1)
    {
        if constexpr (sizeof(IdxT) != 4) {
            return fail(DQ_ERR_HIP, "run lengths: int32 indices only");
        } else {
            const int64_t nchunks = (n + kRunChunk - 1) / kRunChunk;
            LAUNCH(L, DQ_K_RUNS, n, 2 * n + 4 * n,
                   hipLaunchKernelGGL(runlen_chunk_kernel<false>, dim3((unsigned)nchunks), dim3(kRunThreads), 0, st,
                                      (const uint8_t *)w.text, n, w.run_lead, w.run_link, (const uint32_t *)nullptr, (uint32_t *)nullptr, period);
                   hipLaunchKernelGGL(runlen_carry_kernel, dim3(1), dim3(kRunScanThreads), 0, st, (const uint32_t *)w.run_lead,
                                      (const uint8_t *)w.run_link, nchunks, w.run_carry);
                   hipLaunchKernelGGL(runlen_chunk_kernel<true>, dim3((unsigned)nchunks), dim3(kRunThreads), 0, st,
                                      (const uint8_t *)w.text, n, (uint32_t *)nullptr, (uint8_t *)nullptr,
                                      (const uint32_t *)w.run_carry, w.RL, period));
            return DQ_OK;
        }
    }

    // ---- doubled text (dq_small_groups.h, twin_mark_kernel): the tie groups that are a pair (i, i + half) are written
    //      down and leave the list; *done: nothing is left.
    int64_t twin_half = 0;
    const uint8_t *text_src = nullptr;   // the caller's device-resident text when w.text is still to be filled from it
    int period_hint = 0;                // (SortHints::run_period)
    int twin_pairs_step(bool *done)
    {
        *done = false;
        static_assert(kTwTile == 2048, "w.pc_tiles holds two counters per 2048 list entries");
        const int64_t ntiles = (m + kTwTile - 1) / kTwTile;
        uint32_t *tile_cnt = w.pc_tiles;
        PairCounters *ctr = reinterpret_cast<PairCounters *>(w.totals + 4);
        int rc = L.begin(DQ_K_SMALL_ROUND, m, m * (8 + wb));
        if (rc != DQ_OK) return rc;
        hipLaunchKernelGGL(twin_mark_kernel<IdxT>, dim3((unsigned)ntiles), dim3(kTwThreads), 0, st, (const uint64_t *)Kr[rcur],
                           (const uint32_t *)first_rank32, (const IdxT *)Vr[rcur], m, twin_half, d_sa, w.ISA, tile_cnt);
        hipLaunchKernelGGL(pair_scan_kernel, dim3(1), dim3(kPcScanThreads), 0, st, tile_cnt, ntiles, ctr);
        HIP_TRY(hipGetLastError());
        rc = L.end();
        if (rc != DQ_OK) return rc;
        HIP_TRY(hipMemcpyAsync(c.pinned, ctr, sizeof(PairCounters), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        const int64_t groups = c.pinned[0], kept = c.pinned[1];
        if (env("DQ_TRACE"))
            fprintf(stderr, "[dq] doubled text at h=%lld: %lld of %lld tied suffixes in pairs (i, i + n/2), %lld other groups\n", (long long)h,
                    (long long)(m - kept), (long long)m, (long long)groups);
        if (kept == 0) { *done = true; return DQ_OK; }
        // (a look that takes nothing costs a launch and a host round trip per round; it is not given up after idle
        // looks -- the pairs only become "all that is left of their group" as the rounds split the groups around them)
        if (kept == m) return DQ_OK;
        LAUNCH(L, DQ_K_SMALL_ROUND, m, m * (8 + wb) + kept * (8 + wb),
               hipLaunchKernelGGL(twin_compact_kernel<IdxT>, dim3((unsigned)ntiles), dim3(kTwThreads), 0, st, (const uint64_t *)Kr[rcur],
                                  (const uint32_t *)first_rank32, (const IdxT *)Vr[rcur], m, twin_half, (const uint32_t *)tile_cnt,
                                  Kr[rcur ^ 1], Vr[rcur ^ 1]));
        rcur ^= 1;
        first_rank32 = nullptr;                            // (the list carries 64-bit ranks now)
        m = kept;
        return DQ_OK;
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
            // (a period other than 1 -- the caller's hint, or DQ_RUN_PERIOD in the tests -- must not exceed the depth the
            // groups are tied to: the rules of dq_runs.h hold for P <= h)
            int period = period_hint > 0 ? period_hint : 1;
            if (const char *v = env("DQ_RUN_PERIOD")) period = std::max(1, atoi(v));
            if (period > h || period > 64) period = 1;
            rc = compute_run_lengths(period);
            if (rc != DQ_OK) return rc;
            runs_on = true;
            run_order = period;
            t_info[0] += 1;
            t_info[2] += m;
            if (env("DQ_TRACE"))
                fprintf(stderr, "[dq] run-order round (period %d) at h=%lld on %lld tied suffixes\n", period, (long long)h, (long long)m);
            rc = (uses_small_round(m) && !list_ungrouped) ? doubling_round_small(32) : doubling_round_radix(32, 0);
            run_order = 0;
            if (rc != DQ_OK) return rc;
        }

        int64_t m_before = 0;             // list length before the last round (0: no round yet)
        int pair_tries = 0, pair_aborts = 0;
        bool pair_paid = true;            // the last pair-chain phase finished at least half of its list
        int64_t pair_h = 0;               // h of the last phase
        int64_t abort_h = 0, abort_m = 0; // h and list length when a phase last gave up after its count
        // (DQ_TAIL_MAX = 0 ... 4096: the list length from which the rest of the sort is one launch; 0 = never)
        const int64_t tail_max = env("DQ_TAIL_MAX") ? std::max(0, std::min(kTailMax, atoi(env("DQ_TAIL_MAX")))) : kTailMax;
        while (m > 0) {
            if (m <= tail_max && n < (1ll << 32) && !keys_ready && !list_ungrouped && !first_rank32 && !run_order) {
                rc = tail_rounds();
                if (rc != DQ_OK) return rc;
                break;
            }
            if (twin_half > 0 && !keys_ready && !list_ungrouped && !run_order) {
                bool done = false;
                rc = twin_pairs_step(&done);
                if (rc != DQ_OK) return rc;
                if (done) break;
            }
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
            // runs seen late (see long_run_seen): the large groups have stopped shrinking -- run lengths now, one
            // run-order round at the current depth on the current list, the rank behind the run from then on
            const int64_t late_min = env("DQ_LATE_RUNS_MIN") ? std::max(1, atoi(env("DQ_LATE_RUNS_MIN"))) : (1 << 15);   // (tests: small inputs)
            // (the run lengths are a sweep over the whole text, what they save is a few passes over the large groups:
            // librocsparse.so, 256 MiB, 0.32 M members of large groups -- 2.2 ms of run lengths for nothing)
            const int64_t late_share = env("DQ_LATE_RUNS_SHARE") ? std::max(1, atoi(env("DQ_LATE_RUNS_SHARE")))
                                       : env("DQ_LATE_RUNS_MIN") ? (int64_t)1 << 30 : 64;      // (the tests' knob lifts this bar too)
            // (stagnation: the large groups kept at least late_ratio / 8 of their members over the last round.  Measured with
            // 5 / 8 and 4 / 8, which call the round one doubling earlier on libtorch_cpu.so: 25.3 / 25.5 ms against 24.8)
            const int64_t late_ratio = env("DQ_LATE_RUNS_RATIO") ? std::max(1, std::min(8, atoi(env("DQ_LATE_RUNS_RATIO")))) : 7;
            if (late_runs_possible && !runs_on && !runs_late_tried && !run_order && last_large >= late_min && prev_large > 0 &&
                last_large * late_share >= n && last_large * 8 >= prev_large * late_ratio && h >= 32 && 32 + rbits <= 64 && uses_small_round(m) && !keys_ready &&
                !list_ungrouped && !first_rank32 && mid_group_cap(m) > 0 && !env("DQ_NO_LATE_RUNS")) {
                runs_late_tried = true;
                // the rules hold for stretches that repeat with any period P <= h (tests/test_models_cpu.py has the
                // model): P = 64 (or the largest power of two <= h) takes runs of one byte and tables of 2-, 4-, ...
                // 64-byte entries alike
                int period = 1;
                while (period * 2 <= 64 && period * 2 <= h) period *= 2;
                if (const char *v = env("DQ_RUN_PERIOD")) period = std::max(1, std::min<int>((int)std::min<int64_t>(h, 1 << 20), atoi(v)));
                rc = compute_run_lengths(period);
                if (rc != DQ_OK) return rc;
                runs_on = true;
                run_order = period;                        // (the kernels take the period from here)
                t_info[0] += 1;
                t_info[2] += m;
                if (env("DQ_TRACE"))
                    fprintf(stderr, "[dq] late run-order round (period %d) at h=%lld on %lld tied suffixes (%lld in large groups, %lld the round before)\n",
                            period, (long long)h, (long long)m, (long long)last_large, (long long)prev_large);
                rc = doubling_round_small(32);
                run_order = 0;
                if (rc != DQ_OK) return rc;
                m_before = 0;
                continue;
            }
            m_before = m;
            // (doubled text: a look at the pairs after every round while the list is long)
            if (only_small_groups && uses_small_round(m) && !keys_ready && !list_ungrouped && (twin_half == 0 || m < (1 << 16)) &&
                !env("DQ_NO_CHAIN")) {
                tail_behind_chain = tail_max >= kTailMax && m <= 4 * tail_max && n < (1ll << 32);      // (the kernel's own bound is kTailMax)
                // (two rounds, then four, then eight per host round trip: a list that hovers just above the tail
                // kernel's reach -- a long repeat among a few thousand suffixes -- must not pay a round trip every two rounds)
                chain_len = tail_behind_chain ? std::min(kSgChain, 2 << std::min(spec_misses, 2)) : kSgChain;
                if (tail_behind_chain) ++spec_misses;
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
            // (a list that still carries its ranks as 32-bit values for the first LDS-class round -- first_rank32 -- has
            // nothing in Kr[rcur] for the radix path to read: the test flag is ignored for that round; kbits + rbits > 64
            // cannot coincide with it, n < 2^32 there)
            const int rshift = (kbits + rbits > 64 || (env("DQ_FORCE_RSHIFT") && !first_rank32)) ? 1 : 0;
            if (rshift && first_rank32) return fail(DQ_ERR_HIP, "rank-shift round on a list with 32-bit ranks");
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
int sufsort_device(DeviceCtx &c, hipStream_t st, Workspace<IdxT> &w, int64_t n, IdxT *d_sa, SortHints hints = SortHints(),
                   const uint8_t *text_src = nullptr)
{
    SuffixSorter<IdxT> sorter(c, st, w, n, d_sa);
    // (DQ_ASSUME_DOUBLED: the tests vouch for their inputs through the public entry points)
    if ((hints.doubled || env("DQ_ASSUME_DOUBLED")) && n % 2 == 0 && !env("DQ_NO_TWINS")) sorter.twin_half = n / 2;
    if (hints.run_period > 0 && !env("DQ_NO_PERIOD_HINT")) sorter.period_hint = hints.run_period;
    sorter.text_src = text_src;
    return sorter.run();
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

}  // namespace

// host buffers in / out  (ISuffixSort.Sort(text, suffixes))
template <typename IdxT>
int sufsort_host(const uint8_t *text, int64_t n, IdxT *sa, int32_t device, SortHints hints)
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
    rc = sufsort_device<IdxT>(c, st, w, n, w.SAbuf, hints);
    if (rc != DQ_OK) { drop_pending(c, st); return rc; }
    // (one checked step: a failure must not leave a copy into the caller's array in flight behind the return)
    auto copy_out = [&]() -> hipError_t {
        const hipError_t e = hipMemcpyAsync(sa, w.SAbuf, (size_t)n * sizeof(IdxT), hipMemcpyDeviceToHost, st);
        const hipError_t e2 = hipStreamSynchronize(st);
        return e != hipSuccess ? e : e2;
    };
    HIP_TRY(copy_out());
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
    // The library works on a padded, 16-byte aligned copy of the text.  The copy is made by the pass that reads the
    // text first anyway (text_hist_kernel) when the caller's buffer is 16-byte aligned; DQ_TEXT_COPY=1: by a copy in front.
    const bool fused_copy = (reinterpret_cast<uintptr_t>(d_text) & 15) == 0 && !env("DQ_TEXT_COPY");
    if (!fused_copy) HIP_TRY(hipMemcpyAsync(w.text, d_text, (size_t)n, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemsetAsync(w.text + n, 0, 64, st));
    rc = sufsort_device<IdxT>(c, st, w, n, (IdxT *)d_sa, SortHints(), fused_copy ? (const uint8_t *)d_text : nullptr);
    if (rc != DQ_OK) { drop_pending(c, st); return rc; }
    HIP_TRY(hipStreamSynchronize(st));
    return DQ_OK;
}

template <typename IdxT> int64_t sufsort_workspace_bytes(int64_t n) { return (int64_t)carve<IdxT>(nullptr, n, false).bytes; }

}  // namespace dq
