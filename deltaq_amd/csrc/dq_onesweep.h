// dq_onesweep.h -- single-read LSD radix ranking ("onesweep" shape) for gfx950.
//
// One digit pass = ONE kernel that reads every (key, suffix) pair once and writes it once:
//   radix_hist_kernel       all digit histograms of the keys in a single read (per-workgroup
//                           partials, reduced + scanned by radix_hist_scan_kernel)
//   radix_rank_kernel       per tile: ticket -> load -> wave64 ballot multi-split rank ->
//                           publish the tile's 256 digit counts -> decoupled look-back over
//                           the predecessor tiles' status words -> stage through LDS -> write
//
// Inter-workgroup protocol (MI355X: 8 XCDs, L2s not coherent with each other): the ONLY thing a
// tile hands to later tiles is one self-describing status word per digit -- {2-bit state,
// count/prefix} written by ONE agent-scope relaxed atomic store and read by agent-scope relaxed
// atomic loads (both lower to sc1 accesses that bypass the non-coherent levels).  The word is
// its own flag ("the data IS the flag"), so no fence, no release/acquire pair and no ordering
// between different words is needed.  Status words are zeroed by a memset node before every
// sort; tiles are handed out by an atomic ticket, so a tile's predecessors are always already
// running (no dependence on dispatch order or XCD placement); every spin is bounded and
// reports through an error word instead of hanging.
#pragma once
#include "dq_coded_keys.h"
#include "dq_radix.h"

namespace dq {

constexpr int kHistBlocks = 512;
constexpr int kHistThreads = 1024;   // 16 waves share one set of sub-histograms: 512 workgroups fill the 256 CUs
constexpr int kMaxPasses = 8;


// ---------------------------------------------------------------------------------
// radix_hist_kernel: partial[g][p][d] = #keys in workgroup g's share whose digit p is d,
// for all kPasses digit places in ONE read of the keys.  LDS atomics into 4 interleaved
// sub-histograms (hist[p][d][lane & 3]: equal digits from different lanes hit 4 banks);
// a wave whose 64 keys share a digit (constant high digits of composite keys) adds once.
// ---------------------------------------------------------------------------------
template <int kPasses>
__global__ __launch_bounds__(kHistThreads) void radix_hist_kernel(const uint64_t *__restrict__ keys,
                                                            int64_t m, unsigned long long *__restrict__ acc /*[kMaxPasses][256], zeroed*/,
                                                            int shift0 = 0 /* digit p sits at bit shift0 + 8 p */)
{
    __shared__ uint32_t hist[kPasses][kRadixSize * 4];     // kPasses * 4 KiB
    const int tid = threadIdx.x;
    const int lane = lane_id();
    const int sub = tid & 3;
    for (int i = tid; i < kPasses * kRadixSize * 4; i += kHistThreads) (&hist[0][0])[i] = 0;
    __syncthreads();

    const int64_t pairs = m >> 1;
    const ulonglong2 *k2 = reinterpret_cast<const ulonglong2 *>(keys);
    // i0 is wave-uniform, so a wave enters/leaves the loop as a whole and the wave-uniform
    // shortcut below always sees 64 lanes; out-of-range lanes contribute nothing
    for (int64_t i0 = (int64_t)blockIdx.x * kHistThreads + (tid & ~(kWave - 1)); i0 < pairs;
         i0 += (int64_t)gridDim.x * kHistThreads) {
        const int64_t i = i0 + lane;
        const bool ok = i < pairs;
        const bool full = (i0 + kWave) <= pairs;
        ulonglong2 v;
        v.x = 0; v.y = 0;
        if (ok) v = k2[i];
#pragma unroll
        for (int p = 0; p < kPasses; ++p) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint32_t d = digit_of(h ? v.y : v.x, shift0 + p * kRadixBits);
                const uint32_t d0 = __builtin_amdgcn_readfirstlane(d);
                if (full && __all(d == d0)) {
                    if (lane == 0) atomicAdd(&hist[p][d0 << 2], (uint32_t)kWave);
                } else if (ok) {
                    atomicAdd(&hist[p][(d << 2) | sub], 1u);
                }
            }
        }
    }
    if ((m & 1) && blockIdx.x == 0 && tid == 0) {
        const uint64_t k = keys[m - 1];
        for (int p = 0; p < kPasses; ++p) atomicAdd(&hist[p][digit_of(k, shift0 + p * kRadixBits) << 2], 1u);
    }
    __syncthreads();
    // (round 5: one global add per workgroup and counter -- 512 adds per address, ~6 us of them spread over the kernel's
    // life -- instead of per-workgroup partial histograms that a second kernel summed in 38 us for 512 workgroups)
    for (int i = tid; i < kPasses * kRadixSize; i += kHistThreads) {
        const uint32_t *h4 = &(&hist[0][0])[i * 4];
        const uint32_t c = h4[0] + h4[1] + h4[2] + h4[3];
        if (c) atomicAdd(&acc[i], (unsigned long long)c);
    }
}

// Same 8 histograms for the CODED round-0 keys (dq_coded_keys.h), which exist nowhere in memory yet: every lane
// builds the keys of 4 consecutive suffixes from 20 text bytes, as the first digit pass will.
static __global__ __launch_bounds__(kHistThreads) void text_coded_hist_kernel(const uint32_t *__restrict__ t32, int64_t n,
                                                                       const uint16_t *__restrict__ codetab,
                                                                       unsigned long long *__restrict__ acc /*[kMaxPasses][256], zeroed*/)
{
    __shared__ uint32_t hist[kMaxPasses][kRadixSize * 4];
    __shared__ uint16_t ctab[256];
    const int tid = threadIdx.x;
    const int sub = tid & 3;
    for (int i = tid; i < kMaxPasses * kRadixSize * 4; i += kHistThreads) (&hist[0][0])[i] = 0;
    if (tid < 256) ctab[tid] = codetab[tid];
    __syncthreads();
    const int64_t quads = (n + 3) >> 2;
    for (int64_t q = (int64_t)blockIdx.x * kHistThreads + tid; q < quads; q += (int64_t)gridDim.x * kHistThreads) {
        uint32_t tw[5];
#pragma unroll
        for (int c = 0; c < 5; ++c) tw[c] = t32[q + c];
        uint64_t key[4];
        coded_keys4(tw, ctab, key);
        const int cnt = (n - q * 4) < 4 ? (int)(n - q * 4) : 4;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c < cnt) {
#pragma unroll
                for (int p = 0; p < kMaxPasses; ++p)
                    atomicAdd(&hist[p][(digit_of(key[c], p * kRadixBits) << 2) | sub], 1u);
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < kMaxPasses * kRadixSize; i += kHistThreads) {
        const uint32_t *h4 = &(&hist[0][0])[i * 4];
        const uint32_t c = h4[0] + h4[1] + h4[2] + h4[3];
        if (c) atomicAdd(&acc[i], (unsigned long long)c);
    }
}

// digit_offset[p][d] = number of keys whose digit p is < d   (one workgroup of 256 threads per digit place)
constexpr int kHistScanThreads = kRadixSize;
static __global__ __launch_bounds__(kHistScanThreads) void radix_hist_scan_kernel(const unsigned long long *__restrict__ acc,
                                                                           int64_t *__restrict__ digit_offset)
{
    __shared__ int64_t tmp[kHistScanThreads / kWave];
    const int p = blockIdx.x, d = threadIdx.x;
    int64_t total;
    digit_offset[p * kRadixSize + d] = block_excl_sum((int64_t)acc[p * kRadixSize + d], tmp, &total);
}

// ---------------------------------------------------------------------------------
// status words
// ---------------------------------------------------------------------------------
template <typename StatusT> struct StatusBits;
template <> struct StatusBits<uint32_t> {
    static constexpr uint32_t kAgg = 1u << 30, kPrefix = 2u << 30, kMask = (1u << 30) - 1;
};
template <> struct StatusBits<uint64_t> {
    static constexpr uint64_t kAgg = 1ull << 62, kPrefix = 2ull << 62, kMask = (1ull << 62) - 1;
};

template <typename StatusT>
__device__ __forceinline__ void status_store(StatusT *p, StatusT v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename StatusT>
__device__ __forceinline__ StatusT status_load(StatusT *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct OnesweepCtl {
    uint32_t ticket;        // next tile to hand out
    uint32_t error;         // set when a look-back spin timed out
};

// ---------------------------------------------------------------------------------
// radix_rank_kernel (onesweep): THE dominant kernel of the pipeline
//   kItems   keys per thread per tile (tile = 256 * kItems)
//   kMode    kPairs : (key, suffix) pairs are loaded            (every pass but the first)
//            kText  : FIRST pass of round 0: keys are built on the fly from the text
//                     (key = first `keybits` bits at the suffix, big-endian, zero padded past
//                     the end) and the suffix index is synthesised.  Lanes own 4 consecutive
//                     suffixes (3 dwords of text each); the pass need not be stable with
//                     respect to the text order because members of a tie group are re-ranked
//                     by the doubling rounds anyway.
// ---------------------------------------------------------------------------------
// kPairs      (key, suffix) pairs in two arrays
// kText       round 0, first pass: keys built from the text, suffix index synthesised
// kTextPacked same, but the suffix index is packed into the low `ib` bits of the key word
//             (word = key << ib | suffix) and no value array exists
// kKeys       packed words only
// kKeysLast   packed words; the last pass also emits SA[o] = word & mask
// kKeysLastTies  last pass of a packed sort that is expected to leave few ties: emits ONLY the SA
//             (no sorted words) and, from the sorted tile it already holds, the tie structure:
//             ebits bit o = 1 iff the keys at output positions o-1 and o are equal, for every
//             pair inside one digit run of the tile (atomic OR of the few set bits), plus the
//             first and last word of each of the tile's digit runs in seam_tab so that
//             tie_seam_kernel (dq_ties.h) can decide the pairs that straddle two tiles
// kTextPackedExt / kKeysExt: packed words as above PLUS one more byte of key per element in a value array (ValT =
//             uint8_t): at n >= 2^30 a 64-bit word has room for 33 key bits beside the suffix, which leaves 22 % of
//             2^31 random suffixes tied; 8 more bits, carried through the digit passes and used by the bucket pass as
//             low key bits, leave 0.1 %
enum OnesweepMode { kPairs = 0, kText = 1, kTextPacked = 2, kKeys = 3, kKeysLast = 4, kKeysLastTies = 5, kTextPackedExt = 6, kKeysExt = 7 };
constexpr uint64_t kSeamEmpty = ~0ull;          // seam_tab marker: the tile has no key with this digit

// (tools/kbench/kbench.hip defines DQ_PHASE before it includes this header to stamp a tile's phases; the library does not)
#ifndef DQ_PHASE
#define DQ_PHASE(i) do { } while (0)
#endif

template <typename IdxT, typename StatusT, int kItems, int kMode, int kMinWaves, int kThreads = kBlock,
          bool kEarlyVals = false, bool kLdsMatch = true, int kExchRounds = 1, bool kAtomicBase = false,
          bool kCoded = false, typename ValT = IdxT>
__global__ __launch_bounds__(kThreads, kMinWaves) void radix_rank_kernel(
    const uint64_t *__restrict__ kin, const ValT *__restrict__ vin,
    uint64_t *__restrict__ kout, ValT *__restrict__ vout, int64_t m, int shift, int keybits, int ib,
    const int64_t *__restrict__ digit_offset /*[256] for this pass*/,
    StatusT *__restrict__ status /*[ntiles][256]*/, OnesweepCtl *__restrict__ ctl,
    int64_t *__restrict__ sticky_error, uint32_t *__restrict__ ebits = nullptr,
    uint64_t *__restrict__ seam_tab /*[ntiles][256][2]*/ = nullptr,
    const uint16_t *__restrict__ codetab /*[256], kCoded*/ = nullptr,
    int xcd_group = 0 /* kAtomicBase: tiles per XCD and group of the XCD-aware tile order, 0 = blockIdx order */,
    uint32_t spin_limit = kSpinLimit /* empty polls of the look-back before it gives up (DQ_FAULT=spin: 0) */)
{
    constexpr bool kFromText = (kMode == kText || kMode == kTextPacked || kMode == kTextPackedExt);
    constexpr bool kPackedText = (kMode == kTextPacked || kMode == kTextPackedExt);
    static_assert(!kCoded || kMode == kText, "coded keys: first pass of the pair sort only");
    // arrival-order ranking (below) only where digits are near-uniform -- packed words are chosen for random-like
    // text; a skewed digit would put hundreds of same-address LDS atomics of a tile in a row
    constexpr bool kAtomicRank = kAtomicBase && kPackedText;
    constexpr bool kTies = (kMode == kKeysLastTies);
    constexpr bool kHasVals = (kMode == kPairs || kMode == kText || kMode == kTextPackedExt || kMode == kKeysExt);
    static_assert((kMode != kTextPackedExt && kMode != kKeysExt) || sizeof(ValT) == 1, "the extra key bits travel as bytes");
    static_assert(!kFromText || (kItems % 4) == 0, "text mode packs 4 suffixes per lane");
    static_assert(kThreads % kRadixSize == 0, "threads 0..255 own one digit each");
    constexpr int kWavesB = kThreads / kWave;
    constexpr int kTileN = kThreads * kItems;
    constexpr int kWaveN = kWave * kItems;
    using SB = StatusBits<StatusT>;

    // the exchange buffer holds 1/kExchRounds of the tile: the tile is staged through LDS in
    // kExchRounds position ranges, trading barriers for LDS footprint (= occupancy)
    static_assert(kItems % kExchRounds == 0, "exchange rounds split the items evenly");
    constexpr int kExchN = kTileN / kExchRounds;
    __shared__ __attribute__((aligned(16))) uint64_t exch[kExchN];
    __shared__ uint32_t whist[kWavesB][kRadixSize];
    __shared__ uint32_t tile_base[kRadixSize];
    __shared__ IdxT gofs[kRadixSize];
    __shared__ uint32_t wtmp[kRadixSize / kWave];
    __shared__ uint32_t s_tile;
    __shared__ uint16_t ctab[kCoded ? 256 : 1];          // byte -> codeword << 4 | length (dq_alpha_code.h)

    const int tid = threadIdx.x;
    const int w = tid >> 6;
    const int lane = lane_id();
    if (kCoded && tid < 256) ctab[tid] = codetab[tid];

    // kAtomicBase (first pass of a sort only: nothing to be stable against): a tile reserves its place in every
    // digit's output region with ONE returning atomic add per digit on a table of 256 cursors (the first 256
    // 64-bit words of `status`, zeroed like the status words) instead of the decoupled look-back -- no
    // dependence on other tiles at all, so no ticket either.  Tiles land in whatever order the adds arrive.
    if (kAtomicBase) {
        // No tile depends on another here, so the tile a workgroup takes is a pure placement choice: workgroup b runs
        // on XCD b % 8 (observed, MI355X_MICROARCH.md; nothing breaks if not), and with blockIdx order the 8 tiles
        // whose digit runs are neighbours in every output region are written through 8 different L2s -- partial lines
        // all round.  Inside every group of 8 G workgroups XCD x takes tiles x G ... x G + G - 1 instead: the runs of G
        // consecutive tiles meet in one L2 and leave it as whole lines (tools/kbench/scatter.hip: 64 Mi words in
        // 384-byte runs to 256 regions 184 -> 155 us).
        if (tid == 0) {
            uint32_t t = blockIdx.x;
            if (xcd_group > 0) {
                const uint32_t grp = 8u * (uint32_t)xcd_group, g = t / grp, r = t % grp;
                if ((g + 1) * grp <= gridDim.x) t = g * grp + (r & 7u) * (uint32_t)xcd_group + (r >> 3);
            }
            s_tile = t;
        }
    } else if (tid == 0) {
        // Tiles are handed out by an atomic ticket, so a tile's predecessors are always running.
        // (Measured and dropped: XCD-aware orders under the look-back.  Round 1: ticket -> tile permuted inside groups of
        // 64 (-4 % pairs, +2 % text pass).  Round 5: per-XCD tile queues keyed by HW_REG_XCC_ID, G = 4 / 8 / 16 tiles per
        // die and group -- the write pattern that gains 16 % in tools/kbench/scatter.hip -- made every look-back pass
        // SLOWER: 256 MiB of text 30.35 -> 31.4 ms, libtorch_cpu.so 23.3 -> 24.1 ms; a tile waits longer for
        // predecessors that another die's queue hands out later than a single ticket would.  Only the first pass of a
        // sort, which has no look-back, takes the XCD-aware order -- above.)
        const uint32_t t = atomicAdd(&ctl->ticket, 1u);
        s_tile = t;
    }
    for (int i = tid; i < kWavesB * kRadixSize; i += kThreads) (&whist[0][0])[i] = 0;
    if (kLdsMatch && !kAtomicRank) {
        // per-wave digit -> lane-mask tables live in the (not yet used) exchange buffer
        static_assert(kExchN >= kWavesB * kRadixSize, "exchange buffer must hold the match tables");
        for (int i = tid; i < kWavesB * kRadixSize; i += kThreads) exch[i] = 0;
    }
    __syncthreads();
    DQ_PHASE(0);
    const int64_t tile = s_tile;
    const int64_t base = tile * kTileN;
    const int valid = (m - base) < kTileN ? (int)(m - base) : kTileN;
    const int wbase = w * kWaveN + lane;

    // element index (inside the tile) of this lane's item k
    auto elem = [&](int k) -> int {
        return kFromText ? ((k >> 2) * kThreads + tid) * 4 + (k & 3) : wbase + k * kWave;
    };

    uint64_t key[kItems];
    ValT val[kItems];
    // (kTextPackedExt: the extra key bytes of a lane's 4 consecutive suffixes wait packed in one register -- one
    // register each cost the text pass its second workgroup per CU: 142 VGPRs)
    uint32_t vpack[kMode == kTextPackedExt ? kItems / 4 : 1];
    auto val_of = [&](int k) -> ValT {
        if (kMode == kTextPackedExt) return (ValT)((vpack[k >> 2] >> (8 * (k & 3))) & 0xffu);
        return val[k];
    };
    if (kFromText) {
        const uint32_t *t32 = reinterpret_cast<const uint32_t *>(kin);
        const int kshift = 64 - keybits;                      // key = leading `keybits` bits of the suffix
#pragma unroll
        for (int j = 0; j < kItems / 4; ++j) {
            const int e0 = (j * kThreads + tid) * 4;           // first of this lane's 4 suffixes
            if (kCoded) {
                // keys = the first 64 bits of the codewords of T[i], T[i+1], ... (dq_coded_keys.h); the text is
                // followed by 64 zero bytes, so the 20 bytes of the last lanes are there
                if (e0 < valid) {
                    const int64_t qd = (base + e0) >> 2;
                    uint32_t tw[5];
#pragma unroll
                    for (int c = 0; c < 5; ++c) tw[c] = t32[qd + c];
                    coded_keys4(tw, ctab, &key[4 * j]);
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) key[4 * j + c] = ~0ull;
                }
            } else if (e0 < valid) {
                const int64_t qd = (base + e0) >> 2;
                const uint32_t w0 = t32[qd], w1 = t32[qd + 1], w2 = t32[qd + 2];
                const uint64_t x = __builtin_bswap64((uint64_t)w0 | ((uint64_t)w1 << 32));
                const uint64_t y = (uint64_t)__builtin_bswap32(w2) << 32;
                const uint64_t x4[4] = {x, (x << 8) | (y >> 56), (x << 16) | (y >> 48), (x << 24) | (y >> 40)};
                if (kMode == kTextPackedExt) vpack[j] = 0;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    key[4 * j + c] = x4[c] >> kshift;
                    // (the 8 key bits behind the word's: kshift >= 8 on this path -- the word leaves them no room)
                    if (kMode == kTextPackedExt) vpack[j] |= (uint32_t)((x4[c] >> (kshift - 8)) & 0xffu) << (8 * c);
                }
                if (kPackedText) {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        key[4 * j + c] = (key[4 * j + c] << ib) | (uint64_t)(base + e0 + c);
                }
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) key[4 * j + c] = ~0ull;
                if (kMode == kTextPackedExt) vpack[j] = 0;
            }
        }
    } else if (valid == kTileN) {
#pragma unroll
        for (int k = 0; k < kItems; ++k) key[k] = kin[base + wbase + k * kWave];
    } else {
#pragma unroll
        for (int k = 0; k < kItems; ++k) {
            const int e = wbase + k * kWave;
            key[k] = e < valid ? kin[base + e] : ~0ull;
        }
    }

    // values (suffix indices): synthesised, or loaded -- early (more bytes in flight, more
    // registers) or after the ranking loop
    auto load_vals = [&]() {
        if (!kHasVals || kMode == kTextPackedExt) {
            // packed words carry their suffix index (and the extra key byte was made with the key)
        } else if (kMode == kText) {
#pragma unroll
            for (int k = 0; k < kItems; ++k) val[k] = (ValT)(base + elem(k));
        } else if (valid == kTileN) {
#pragma unroll
            for (int k = 0; k < kItems; ++k) val[k] = vin[base + wbase + k * kWave];
        } else {
#pragma unroll
            for (int k = 0; k < kItems; ++k) {
                const int e = wbase + k * kWave;
                val[k] = e < valid ? vin[base + e] : (ValT)0;
            }
        }
    };
    if (kEarlyVals) load_vals();
    if (kMode == kPairs || kMode == kKeysExt) { asm volatile("" :: "v"(key[kItems - 1])); }
    DQ_PHASE(1);

    // ---- rank inside the wave.  peers(d) = lanes of this wave holding digit d in this round:
    //      either 8 ballots (VALU heavy) or, default, through LDS: every lane ORs its lane bit
    //      into a per-wave table entry mtab[d] (ds_or_b64), reads the entry back (LDS executes a
    //      wave's instructions in order, so the read sees all 64 lanes' bits) and the group's
    //      first lane clears it again.  rank = running per-wave digit count + peers below me. ----
    uint32_t pos[kItems];
    uint32_t *myhist = whist[w];
    unsigned long long *mtab = reinterpret_cast<unsigned long long *>(exch) + w * kRadixSize;
    const unsigned long long lanebit = 1ull << lane;
    auto peers_of = [&](uint32_t d) -> uint64_t {
        if (kLdsMatch) {
            __hip_atomic_fetch_or(&mtab[d], lanebit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            return __hip_atomic_load(&mtab[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        return match_digit8(d);
    };
    (void)lanebit;
    if (kAtomicRank) {
        // The first pass of a sort keeps no earlier order, so the rank of a key inside (tile, digit) may be its
        // arrival number: ONE returning LDS add per key on a digit histogram shared by the whole workgroup
        // (row 0 of whist), all 24 of a lane in flight together -- instead of the match table, its read-back and
        // the per-wave counter, three dependent LDS round trips per key.
#pragma unroll
        for (int k = 0; k < kItems; ++k) {
            const uint32_t d = digit_of(key[k], shift);
            pos[k] = atomicAdd(&whist[0][d], elem(k) < valid ? 1u : 0u);
        }
    } else if (valid == kTileN) {
#pragma unroll
        for (int k = 0; k < kItems; ++k) {
            const uint32_t d = digit_of(key[k], shift);
            // a digit shared by the whole wave (constant high digits of composite keys, runs of
            // equal bytes) would be a 64-way same-address LDS atomic: skip the table for it
            const bool uni = kLdsMatch && __all(d == (uint32_t)__builtin_amdgcn_readfirstlane(d));
            const uint64_t peers = uni ? ~0ull : peers_of(d);
            const uint32_t before = myhist[d];                 // same value for every peer
            const int r = mask_rank_lt(peers);
            if (r == 0) {
                myhist[d] = before + (uint32_t)__popcll(peers);
                if (kLdsMatch && !uni) mtab[d] = 0;
            }
            pos[k] = before + (uint32_t)r;
        }
    } else {
        // ragged last tile: out-of-range items take no part in the ranking
#pragma unroll
        for (int k = 0; k < kItems; ++k) {
            const bool ok = elem(k) < valid;
            const uint32_t d = digit_of(key[k], shift);
            pos[k] = 0;
            if (kLdsMatch) {
                if (ok) {
                    const uint64_t peers = peers_of(d);
                    const uint32_t before = myhist[d];
                    const int r = mask_rank_lt(peers);
                    if (r == 0) { myhist[d] = before + (uint32_t)__popcll(peers); mtab[d] = 0; }
                    pos[k] = before + (uint32_t)r;
                }
            } else {
                const uint64_t peers = match_digit8(d) & __ballot(ok);
                if (ok) {
                    const uint32_t before = myhist[d];
                    const int r = mask_rank_lt(peers);
                    if (r == 0) myhist[d] = before + (uint32_t)__popcll(peers);
                    pos[k] = before + (uint32_t)r;
                }
            }
        }
    }
    if (kLdsMatch && !kAtomicRank) __syncthreads();          // the match tables alias the exchange buffer
    DQ_PHASE(2);
    if (!kEarlyVals) load_vals();
    __syncthreads();
    DQ_PHASE(3);

    // ---- digit totals of the tile (thread d < 256 owns digit d): publish the aggregate and
    //      start fetching the predecessors' status words; the look-back is RESOLVED LATE, after
    //      the LDS exchanges, so its cross-XCD round trips and any straggling predecessor
    //      overlap this tile's own exchange work instead of stalling the workgroup ----
    constexpr int kLookWin = 8;            // status words fetched at publish time (16 / 32 measured: no gain)
    uint32_t tot = 0, incl = 0;
    StatusT sw[kLookWin];
    unsigned long long abase = 0;
    StatusT *mine = status + tile * kRadixSize + (tid & (kRadixSize - 1));
    if (tid < kRadixSize) {
#pragma unroll
        for (int i = 0; i < kWavesB; ++i) {
            const uint32_t c = whist[i][tid];
            whist[i][tid] = tot;
            tot += c;
        }
        if (kAtomicBase) {
            // (the returned value is not needed before the LDS exchanges are done: the round trip overlaps them)
            abase = atomicAdd(reinterpret_cast<unsigned long long *>(status) + tid, (unsigned long long)tot);
        } else {
        if (tile == 0) status_store<StatusT>(mine, SB::kPrefix | (StatusT)tot);
        else status_store<StatusT>(mine, SB::kAgg | (StatusT)tot);
#pragma unroll
        for (int j = 0; j < kLookWin; ++j)
            sw[j] = (tile - 1 - j >= 0) ? status_load<StatusT>(status + (tile - 1 - j) * kRadixSize + tid)
                                        : SB::kPrefix;
        }
        incl = wave_incl_sum(tot);                             // tile-local scan over digits
        if (lane == kWave - 1) wtmp[w] = incl;
    }
    __syncthreads();
    uint32_t excl_tile = 0;
    if (tid < kRadixSize) {
        uint32_t off = 0;
#pragma unroll
        for (int i = 0; i < kRadixSize / kWave; ++i) if (i < w) off += wtmp[i];
        excl_tile = off + incl - tot;
        tile_base[tid] = excl_tile;
    }
    __syncthreads();
    DQ_PHASE(4);

    // ---- stage keys, then values, in digit order through LDS; keep the sorted tile in registers ----
#pragma unroll
    for (int k = 0; k < kItems; ++k) {
        const uint32_t d = digit_of(key[k], shift);
        // (kAtomicRank: pos is already the place inside the tile's digit run; the per-wave offsets are not used)
        pos[k] += kAtomicRank ? tile_base[d] : tile_base[d] + myhist[d];
    }
    if (valid != kTileN) {
#pragma unroll
        for (int k = 0; k < kItems; ++k)
            if (elem(k) >= valid) pos[k] = 0xffffffffu;                      // never staged
    }
    uint64_t skey[kItems];
#pragma unroll
    for (int r = 0; r < kExchRounds; ++r) {
        if (r > 0) __syncthreads();
#pragma unroll
        for (int k = 0; k < kItems; ++k) {
            const uint32_t p = pos[k] - (uint32_t)(r * kExchN);
            if (p < (uint32_t)kExchN) exch[p] = key[k];
        }
        __syncthreads();
#pragma unroll
        for (int k = r * (kItems / kExchRounds); k < (r + 1) * (kItems / kExchRounds); ++k)
            skey[k] = exch[k * kThreads + tid - r * kExchN];
    }
    ValT sval[kItems];
    if (kHasVals) {
        constexpr int kValN = kExchN * (int)(sizeof(uint64_t) / sizeof(ValT));    // values per round
        constexpr int kValRounds = kTileN / kValN > 0 ? kTileN / kValN : 1;
        constexpr int kValCap = kTileN / kValRounds;
        ValT *exv = reinterpret_cast<ValT *>(exch);
#pragma unroll
        for (int r = 0; r < kValRounds; ++r) {
            __syncthreads();
#pragma unroll
            for (int k = 0; k < kItems; ++k) {
                const uint32_t p = pos[k] - (uint32_t)(r * kValCap);
                if (p < (uint32_t)kValCap) exv[p] = val_of(k);
            }
            __syncthreads();
#pragma unroll
            for (int k = r * (kItems / kValRounds); k < (r + 1) * (kItems / kValRounds); ++k)
                sval[k] = exv[k * kThreads + tid - r * kValCap];
        }
    }
    DQ_PHASE(5);

    // ---- resolve the look-back: sum aggregates until an inclusive prefix shows up ----
    if (tid < kRadixSize) {
        StatusT excl = 0;
        if (kAtomicBase) {
            excl = (StatusT)abase;
        } else if (tile > 0) {
            int64_t t = tile - 1;
            uint32_t spins = 0;
            bool done = false;
            // the window prefetched at publish time first ...
            {
                int used = 0;
#pragma unroll
                for (int j = 0; j < kLookWin; ++j) {
                    if (!done && used == j) {
                        if (sw[j] & SB::kPrefix) { excl += sw[j] & SB::kMask; done = true; }
                        else if (sw[j] & SB::kAgg) { excl += sw[j] & SB::kMask; used = j + 1; }
                    }
                }
                t -= used;
            }
            // ... then kLookWin2 status words per round trip until an inclusive prefix shows up
            constexpr int kLookWin2 = 8;
            while (!done) {
                StatusT s2[kLookWin2];
#pragma unroll
                for (int j = 0; j < kLookWin2; ++j)
                    s2[j] = (t - j >= 0) ? status_load<StatusT>(status + (t - j) * kRadixSize + tid) : SB::kPrefix;
                int used = 0;
#pragma unroll
                for (int j = 0; j < kLookWin2; ++j) {
                    if (!done && used == j) {
                        if (s2[j] & SB::kPrefix) { excl += s2[j] & SB::kMask; done = true; }
                        else if (s2[j] & SB::kAgg) { excl += s2[j] & SB::kMask; used = j + 1; }
                    }
                }
                t -= used;
                if (!done && used == 0) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > spin_limit) { atomicExch(&ctl->error, 1u); *sticky_error = 1; break; }
                }
            }
            status_store<StatusT>(mine, SB::kPrefix | (StatusT)(excl + tot));
        }
        gofs[tid] = (IdxT)(digit_offset[tid] + (int64_t)excl) - (IdxT)excl_tile;
    }
    __syncthreads();
    DQ_PHASE(6);

    if (kTies) {
        // ---- tie structure of the sorted tile.  q = k*kThreads + tid is the position in the sorted
        //      tile; the predecessor q-1 sits in lane-1 (DPP), in the previous wave's lane 63 or, for
        //      tid 0, in the last lane of item k-1: those come through LDS (the exchange buffer is idle) ----
        uint64_t *edge = exch;                                   // [kItems][kWavesB]
        uint64_t *run_first = exch + kItems * kWavesB;           // [256] first / last word of each digit run,
        uint64_t *run_last = run_first + kRadixSize;             // staged here and written out coalesced
        static_assert(kExchN >= kItems * kWavesB + 2 * kRadixSize, "exchange buffer holds the seam staging");
        if (lane == kWave - 1) {
#pragma unroll
            for (int k = 0; k < kItems; ++k) edge[k * kWavesB + w] = skey[k];
        }
        if (tid < kRadixSize) { run_first[tid] = kSeamEmpty; run_last[tid] = kSeamEmpty; }
        __syncthreads();
        const uint64_t smask = (1ull << ib) - 1;
        // lane 0's predecessors for all items, fetched with ONE LDS read (lane k holds item k's) and
        // handed over by readlane inside the loop: no LDS round trip per item
        static_assert(kItems <= kWave, "one lane per item");
        uint64_t eprev = 0;
        if (lane < kItems) {
            if (w > 0) eprev = edge[lane * kWavesB + w - 1];
            else if (lane > 0) eprev = edge[(lane - 1) * kWavesB + kWavesB - 1];
        }
#pragma unroll
        for (int k = 0; k < kItems; ++k) {
            const int q = k * kThreads + tid;
            const uint32_t plo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)skey[k], 0x138, 0xf, 0xf, false);
            const uint32_t phi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(skey[k] >> 32), 0x138, 0xf, 0xf, false);
            uint64_t prev = ((uint64_t)phi << 32) | plo;
            const uint64_t e0 = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(eprev >> 32), k) << 32) |
                                (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)eprev, k);
            if (lane == 0) prev = e0;
            if (q < valid) {
                const uint32_t d = digit_of(skey[k], shift);
                const IdxT o = gofs[d] + (IdxT)q;
                const bool hasprev = q > 0;
                // equal keys imply equal digits, so a tie never crosses a run boundary
                if (hasprev && ((skey[k] ^ prev) >> ib) == 0)
                    atomicOr(&ebits[(uint64_t)o >> 5], 1u << ((uint32_t)o & 31u));
                const uint32_t dp = digit_of(prev, shift);
                if (!hasprev || dp != d) {
                    run_first[d] = skey[k];
                    if (hasprev) run_last[dp] = prev;
                }
                if (q == valid - 1) run_last[d] = skey[k];
                vout[o] = (ValT)(skey[k] & smask);
            }
        }
        __syncthreads();
        if (tid < kRadixSize) {
            seam_tab[(tile * kRadixSize + tid) * 2] = run_first[tid];
            seam_tab[(tile * kRadixSize + tid) * 2 + 1] = run_last[tid];
        }
    } else {
    // ---- coalesced run writes ----
#pragma unroll
    for (int k = 0; k < kItems; ++k) {
        const int q = k * kThreads + tid;
        if (q < valid) {
            const IdxT o = gofs[digit_of(skey[k], shift)] + (IdxT)q;
            kout[o] = skey[k];
            if (kHasVals) vout[o] = sval[k];
            if (kMode == kKeysLast) vout[o] = (ValT)(skey[k] & ((1ull << ib) - 1));
        }
    }
    }
    DQ_PHASE(7);
}

// ---------------------------------------------------------------------------------
// Round-0 digit histograms straight from the text: digit p of suffix i is the byte
// T[i + kb-1-p] (zero past the end), so every digit place shares ONE byte histogram of the
// text, corrected for the first / last kb-1 positions.
// ---------------------------------------------------------------------------------
//
// Workgroup 0 of the grid does something else (kgram_coll != nullptr): the byte histogram says nothing
// about repetition (text has ~4.5 bits of order-0 entropy per byte and ~2 bits of real entropy), so it
// samples kKgramSamples evenly spaced suffixes and counts, for every prefix length L = 1..8, the samples
// whose first L bytes were already seen in an earlier sample (one hashed bit set per length in LDS; a
// ~1.6 % full table gives ~16 false hits, far below the threshold the host applies).  With C such samples
// out of S, a suffix expects about n * 2C / S^2 twins under an L-byte key.  kgram_coll[L-1] = C for L.
// It runs beside the histogram workgroups, so the sample costs no time of its own.
constexpr int kKgramSamples = 1024;
constexpr int kKgramBits = 32768;                    // bits per length in the seen-set

__device__ __forceinline__ void sample_kgrams(const uint8_t *__restrict__ text, int64_t n,
                                              unsigned long long *__restrict__ coll, uint32_t *seen /*[8][kKgramBits/32]*/,
                                              uint32_t *cnt)
{
    constexpr int kPer = kKgramSamples / kBlock;
    const int t = threadIdx.x;
    for (int i = t; i < 8 * kKgramBits / 32; i += kBlock) seen[i] = 0;
    if (t < 8) cnt[t] = 0;
    uint64_t key[kPer];
#pragma unroll
    for (int q = 0; q < kPer; ++q) {
        const int64_t p = (int64_t)((__int128)(q * kBlock + t) * n / kKgramSamples);
        uint64_t k = 0;
#pragma unroll
        for (int b = 0; b < 8; ++b) k = (k << 8) | (p + b < n ? (uint64_t)text[p + b] : 0ull);
        key[q] = k;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < kPer; ++q) {
#pragma unroll
        for (int L = 1; L <= 8; ++L) {
            uint64_t x = (key[q] >> (64 - 8 * L)) + 0x9E3779B97F4A7C15ull * (uint64_t)L;
            x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
            x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
            const uint32_t bit = (uint32_t)(x >> 40) & (kKgramBits - 1);
            const uint32_t old = atomicOr(&seen[(L - 1) * (kKgramBits / 32) + (bit >> 5)], 1u << (bit & 31));
            if (old & (1u << (bit & 31))) atomicAdd(&cnt[L - 1], 1u);
        }
    }
    __syncthreads();
    if (t < 8) coll[t] = (unsigned long long)cnt[t];
}

// copy_out (round 5): the caller's device-resident text is read HERE for the first time -- the histogram pass also writes
// the library's padded copy of it (16 bytes stored per 16 bytes loaded, behind LDS atomics that bound the kernel anyway)
// instead of a device-to-device copy in front of the sort: 98 us of the 3.73 ms of a 256 MiB sort.
static __global__ __launch_bounds__(kBlock) void text_hist_kernel(const uint8_t *__restrict__ text, int64_t n,
                                                           unsigned long long *__restrict__ bytehist /*[256], zeroed*/,
                                                           unsigned long long *__restrict__ kgram_coll = nullptr,
                                                           uint8_t *__restrict__ copy_out = nullptr /* 16-byte aligned, or none */)
{
    // 4 interleaved sub-histograms (hist[d][lane & 3]) spread equal bytes over 4 banks
    __shared__ uint32_t hist[kRadixSize * 4];
    if (kgram_coll && blockIdx.x == 0) {
        __shared__ uint32_t s_seen[8 * kKgramBits / 32];
        sample_kgrams(text, n, kgram_coll, s_seen, hist);
        return;
    }
    const int tid = threadIdx.x;
    const int sub = tid & 3;
    const int64_t nblocks = (int64_t)gridDim.x - (kgram_coll ? 1 : 0);          // workgroups that build the histogram
    const int64_t hblock = (int64_t)blockIdx.x - (kgram_coll ? 1 : 0);
    for (int i = tid; i < kRadixSize * 4; i += kBlock) hist[i] = 0;
    __syncthreads();
    const uint4 *t16 = reinterpret_cast<const uint4 *>(text);      // text is 16-byte aligned
    const int64_t chunks = n >> 4;
    bool long_run = false;
    unsigned long long flat_chunks = 0;                  // 16-byte chunks made of one byte value (wave-uniform count)
    uint4 *o16 = reinterpret_cast<uint4 *>(copy_out);
    for (int64_t i = hblock * kBlock + tid; i < chunks; i += nblocks * kBlock) {
        const uint4 v = t16[i];
        if (copy_out) o16[i] = v;
        const uint32_t wds[4] = {v.x, v.y, v.z, v.w};
        // a run of >= 64 equal bytes shows as 4 consecutive lanes whose 16 bytes are all one value (the lanes of a
        // wave hold consecutive chunks): more equal round-0 keys than one bin of the bucket pass takes
        const bool flat = v.x == v.y && v.y == v.z && v.z == v.w && v.x == (v.x & 0xffu) * 0x01010101u;
        const uint64_t fb = __ballot(flat);
        long_run |= (fb & (fb >> 1) & (fb >> 2) & (fb >> 3)) != 0;
        flat_chunks += (unsigned long long)__popcll(fb);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int b = 0; b < 4; ++b) atomicAdd(&hist[(((wds[j] >> (8 * b)) & 0xff) << 2) | sub], 1u);
        }
    }
    if (hblock == 0) {
        for (int64_t i = (chunks << 4) + tid; i < n; i += kBlock) {
            const uint8_t b = text[i];
            if (copy_out) copy_out[i] = b;
            atomicAdd(&hist[((uint32_t)b << 2) | sub], 1u);
        }
    }
    __syncthreads();
    const uint32_t c = hist[tid * 4] + hist[tid * 4 + 1] + hist[tid * 4 + 2] + hist[tid * 4 + 3];
    if (c) atomicAdd(&bytehist[tid], (unsigned long long)c);
    if (kgram_coll && long_run && lane_id() == 0) atomicOr(&kgram_coll[8], 1ull);      // (wave-uniform flag)
    // kgram_coll[9]: how much of the text lies in runs -- dq_runs.h pays where that is a good part of it
    if (kgram_coll && flat_chunks && lane_id() == 0) atomicAdd(&kgram_coll[9], flat_chunks);
}

// digit_offset[p][d] for p < kb (one workgroup per digit place p)
static __global__ __launch_bounds__(kBlock) void text_digit_offsets_kernel(const int64_t *__restrict__ bytehist,
                                                                    const uint8_t *__restrict__ text,
                                                                    int64_t n, int kb,
                                                                    int64_t *__restrict__ digit_offset)
{
    __shared__ int64_t tmp[kWavesPerBlock];
    const int p = blockIdx.x;
    const int d = threadIdx.x;
    const int64_t off = kb - 1 - p;                       // digit p of suffix i is T[i + off]
    const int64_t lead = off < n ? off : n;               // text positions < off are never a digit p
    int64_t c = bytehist[d];
    for (int64_t j = 0; j < lead; ++j) c -= (text[j] == d);
    if (d == 0) c += lead;                                // ... and `lead` suffixes read the zero pad
    int64_t total;
    const int64_t excl = block_excl_sum(c, tmp, &total);
    digit_offset[p * kRadixSize + d] = excl;
}

}  // namespace dq
