// dq_split_round0.h -- round 0 of a text-like input as a SAMPLE SORT instead of eight stable digit passes (round 6).
//
// What it replaces: the 8-byte pair path of round 0 (dq_sorter_impl.h: one pass that builds 64-bit keys from the text
// + seven stable pair passes of 24 B per suffix, 11.5 of the 28.8 ms of a 256 MiB text sort).  Fixed key bits cannot
// be used MSD-first on skewed text (after the top 16 coded bits 73 % of the suffixes of the text config sit in buckets
// too long for LDS); ORDER-PRESERVING SPLITTERS drawn from a sorted sample of the keys can: ranked
// by splitter index the suffixes fall into near-equal buckets whatever the alphabet (tools/exp/splitter_buckets.cpp, the
// go / no-go measurement with 65 536 buckets: 97.9 % of the suffixes of the 256 MiB text config and 91.1 % of 128 MiB of
// libtorch_cpu.so in buckets of <= 12 288; what is left are keys that occur thousands of times, which no splitter separates).
//
//   sample_keys_kernel      kSplitSample = 2 Mi keys at scattered text positions (the same coded / raw 64-bit keys the digit
//                           passes build); sorted with the ordinary pair sorter (~0.2 ms)
//   sample_heavy_kernel     share of the sample inside runs of equal keys too long for a bucket: a text made of a few
//                           heavy keys is left to the digit passes before anything is moved
//   make_splitters_kernel   every 8th sorted sample key is a splitter: top[kSplitTop - 1] (every 4096th) and, per top
//                           bucket t, sub[t][kSplitSub - 1].  bucket(key) = (t, s), t = #{top <= key}, s = #{sub[t] <= key}:
//                           monotone in the key, so bucket order is key order
//   split_estimate_kernel   pass A's output regions from the sample alone: k_t sampled keys in top bucket t => n k_t / S
//                           suffixes +- 1.6 %; room for 1/8 more (no counting pass over the text: it cost 1.08 ms)
//   split_pass_kernel<A>    text -> (key, suffix) pairs grouped by t.  A tile ranks its keys by arrival (LDS atomics: nothing
//                           to be stable against), reserves its place in every region with one returning global add per
//                           digit, stages the tile through LDS in digit order and writes runs -- the shape of
//                           radix_rank_kernel's first pass with "digit = rank among the splitters"; the digits of a tile
//                           are near-uniform BY CONSTRUCTION, whatever the text
//   split_plan_kernel       what every region received; tiles of pass B per top bucket
//   split_pass_kernel<B>    pairs of one top bucket (tiles never straddle two: their sub-splitter table is 4 KB of LDS) ->
//                           the bucket's SLOT: kSplitBuckets slots of `cap` entries (twice the mean bucket) in idle
//                           buffers, filled through one cursor per bucket; what does not fit goes to the overflow list
//   bucket_sum / _scan      final position of every bucket (exclusive scan of the cursors), the oversize buckets
//   bucket_finish_kernel    one workgroup per bucket: the bucket is sorted inside LDS by its full 64-bit keys -- a
//                           sample sort again (see the kernel) -- and leaves as sorted keys + suffixes.  An oversize
//                           bucket is moved to the overflow list instead.
//   (host)                  the overflow list -- ALL members of the oversize buckets, a few per cent of the text -- is sorted by
//                           the ordinary pair sorter; sorted by key it is sorted by bucket, so
//   overflow_place_kernel   copies every entry of it to its place in its bucket's final stretch.
//
// Measured (1 x MI355X, 256 MiB of enwik-style text, profiles/r06/): pass A 1.69 ms, pass B 1.83, finish 3.48, sample +
// its sort + overflow sort + placement ~1.0 -- 8.0 ms against 11.6 for the histograms + eight digit passes; the sort
// 28.6 -> 25.5 ms; 128 MiB 14.1 -> 12.8.
//
// The result is what the digit passes leave: keys sorted in one buffer, suffixes in the suffix array, equal keys in
// arbitrary order -- the rebucket pass and everything behind it run unchanged.  Per suffix: 1 + 12 (pass A) + 12 + 12
// (pass B) + 12 + 12 (finish) = 61 B against 1 (histograms) + 21 + 7 x 24 = 190 B.
// Reference context: the phase replaced is still LibDivSufSort.Sort (LibDivSufSort.cs:12-29; its B* substring sort,
// SsSort.cs:934-1269, is what dominates the reference on text); the suffix array is unchanged by any of this.
#pragma once
#include "dq_onesweep.h"

namespace dq {

constexpr int kSplitTop = 512;                                   // top buckets = regions of pass A
constexpr int kSplitSub = 512;                                   // parts of a top bucket = regions of pass B
constexpr int kSplitBuckets = kSplitTop * kSplitSub;             // 262 144: a 256 MiB text has 1024 suffixes per bucket
// sampled keys per bucket: 8 -> bucket sizes spread like Gamma(8) around the mean (sigma 35 %): one bucket in ~250 grows past
// twice the mean, its slot, and takes the overflow route -- cheaper than sorting a sample twice as long (16: 0.25 ms more)
constexpr int kSplitOversample = 8;
constexpr int64_t kSplitSample = (int64_t)kSplitBuckets * kSplitOversample;      // 2 Mi sampled keys
constexpr int kSplitThreads = 512;
constexpr int kSplitItemsA = 20, kSplitItemsB = 16;              // keys per thread: pass A (text -> pairs), pass B (pairs -> slots: 16 keep it inside 128 registers)
constexpr int kSplitTileA = kSplitThreads * kSplitItemsA;        // 10 240 keys per tile
constexpr int kSplitTileB = kSplitThreads * kSplitItemsB;        // 8 192
constexpr int kFinCap = 2048;                                    // the longest bucket the finish kernel sorts
constexpr int kFinSmallCap = 1024;                               // ... and what its small geometry takes

struct SplitCtl {
    unsigned long long ovf_count;       // entries of the overflow list (before that: heavy sampled keys, sample_heavy_kernel)
    unsigned long long ovf_buckets;     // oversize buckets (bucket_scan_kernel)
    unsigned long long tiles_b;         // tiles of pass B (split_plan_kernel)
    unsigned long long abandon;         // the overflow list ran full: the caller takes the digit passes instead
    unsigned long long pure_count;      // entries of the pure list (copies of heavy keys that did not fit their bucket's slot)
};

// the 64-bit round-0 key of ONE suffix p, as the digit passes build it (coded: dq_coded_keys.h; raw: 8 bytes big-endian)
template <bool kCoded>
__device__ __forceinline__ uint64_t split_key_at(const uint32_t *__restrict__ t32, int64_t p, const uint16_t *ctab)
{
    const int64_t q = p >> 2;
    const int c = (int)(p & 3);
    if (kCoded) {
        uint32_t tw[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) tw[i] = t32[q + i];
        uint64_t key[4];
        coded_keys4(tw, ctab, key);
        return c == 0 ? key[0] : c == 1 ? key[1] : c == 2 ? key[2] : key[3];
    }
    const uint32_t w0 = t32[q], w1 = t32[q + 1], w2 = t32[q + 2];
    const uint64_t x = __builtin_bswap64((uint64_t)w0 | ((uint64_t)w1 << 32));
    const uint64_t y = (uint64_t)__builtin_bswap32(w2) << 32;
    return c == 0 ? x : c == 1 ? ((x << 8) | (y >> 56)) : c == 2 ? ((x << 16) | (y >> 48)) : ((x << 24) | (y >> 40));
}

// the keys of a lane's 4 consecutive suffixes (q = index of their first dword)
template <bool kCoded>
__device__ __forceinline__ void split_keys4(const uint32_t *__restrict__ t32, int64_t q, const uint16_t *ctab, uint64_t key[4])
{
    if (kCoded) {
        uint32_t tw[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) tw[i] = t32[q + i];
        coded_keys4(tw, ctab, key);
    } else {
        const uint32_t w0 = t32[q], w1 = t32[q + 1], w2 = t32[q + 2];
        const uint64_t x = __builtin_bswap64((uint64_t)w0 | ((uint64_t)w1 << 32));
        const uint64_t y = (uint64_t)__builtin_bswap32(w2) << 32;
        key[0] = x; key[1] = (x << 8) | (y >> 56); key[2] = (x << 16) | (y >> 48); key[3] = (x << 24) | (y >> 40);
    }
}

// #{j < kN - 1 : tab[j] <= key}: a branch-free binary search over kN - 1 sorted splitters (kN a power of two)
template <int kN>
__device__ __forceinline__ uint32_t split_rank(const uint64_t *tab, uint64_t key)
{
    uint32_t d = 0;
#pragma unroll
    for (int step = kN / 2; step >= 1; step >>= 1) d += (tab[d + step - 1] <= key) ? (uint32_t)step : 0u;
    return d;
}

// sample j sits at text position floor(frac(j * golden ratio) * n): scattered, reproducible, every position equally likely
__device__ __forceinline__ int64_t split_sample_pos(int64_t j, int64_t n)
{
    return (int64_t)(((unsigned __int128)((uint64_t)j * 0x9E3779B97F4A7C15ull) * (unsigned __int128)(uint64_t)n) >> 64);
}

template <bool kCoded>
__global__ __launch_bounds__(kBlock) void sample_keys_kernel(const uint32_t *__restrict__ t32, int64_t n, const uint16_t *__restrict__ codetab,
                                                          int64_t count, uint64_t *__restrict__ out)
{
    __shared__ uint16_t ctab[256];
    if (kCoded) ctab[threadIdx.x] = codetab[threadIdx.x];
    __syncthreads();
    const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j < count) out[j] = split_key_at<kCoded>(t32, split_sample_pos(j, n), ctab);
}

// How much of the text would end on the overflow list?  A key whose copies fill more than one slot (cap entries; a bucket
// holds cap / 2 on average = kSplitOversample sampled keys) shows in the sorted sample as a run of more than
// 2 kSplitOversample equal keys: *heavy = sampled keys inside such runs.  (One thread per sampled key steps to the ends of
// its run -- runs are short unless the text is made of a few keys, and then the count is all that matters: the walk
// stops at 4 kSplitOversample.)
static __global__ __launch_bounds__(kBlock) void sample_heavy_kernel(const uint64_t *__restrict__ sorted, unsigned long long *__restrict__ heavy)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    bool h = false;
    if (i < kSplitSample) {
        const uint64_t me = sorted[i];
        int run = 1;
        for (int64_t j = i - 1; j >= 0 && run <= 4 * kSplitOversample && sorted[j] == me; --j) ++run;
        for (int64_t j = i + 1; j < kSplitSample && run <= 4 * kSplitOversample && sorted[j] == me; ++j) ++run;
        h = run > 2 * kSplitOversample;
    }
    const uint64_t bal = __ballot(h);
    if (lane_id() == 0 && bal) atomicAdd(heavy, (unsigned long long)__popcll(bal));
}

// The bucket boundaries: U[b] = sorted sample key (b + 1) * kSplitOversample for b < kSplitBuckets - 1 -- top[t] is U[t * kSplitSub
// + kSplitSub - 1], sub[t][s] is U[t * kSplitSub + s]; the last entry of every table is never read by split_rank (all ones).
// HEAVY KEYS GET A BUCKET OF THEIR OWN: a key K that occurs more often than two mean buckets hold shows up as two or more
// consecutive boundaries; the second becomes K + 1 (keys are integers, the sequence stays non-decreasing), so that bucket
// [K, K + 1) holds exactly the copies of K.  Such a bucket is PURE (pure[b] = 1, low[b] = K): its entries need no order
// (round 0 need not be stable), however many there are -- pass B sends what does not fit its slot to a list of (bucket,
// arrival number, suffix) triples that is placed WITHOUT being sorted, and the finish kernel copies its slot as it is.
// (Before: 7 % of the suffixes of the 256 MiB text went through the overflow list's eight digit passes; most were copies of
// a few thousand heavy keys.)
static __global__ __launch_bounds__(kBlock) void make_splitters_kernel(const uint64_t *__restrict__ sorted, uint64_t *__restrict__ top,
                                                               uint64_t *__restrict__ sub, uint64_t *__restrict__ low, uint8_t *__restrict__ pure)
{
    constexpr int64_t per_sub = kSplitSample / kSplitBuckets;
    const int64_t b = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (b >= kSplitBuckets) return;
    auto U = [&](int64_t i) -> uint64_t { return i < 0 ? 0ull : i >= kSplitBuckets - 1 ? ~0ull : sorted[(i + 1) * per_sub]; };
    auto fixed = [&](int64_t i) -> uint64_t {               // U'[i]
        const uint64_t u = U(i);
        return (i >= 1 && i < kSplitBuckets - 1 && u == U(i - 1) && u != ~0ull) ? u + 1 : u;
    };
    const uint64_t up = fixed(b), lo = b == 0 ? 0ull : fixed(b - 1);
    sub[b] = up;
    if (b % kSplitSub == kSplitSub - 1) top[b / kSplitSub] = up;
    low[b] = lo;
    pure[b] = (b >= 1 && b < kSplitBuckets - 1 && lo != ~0ull && up == lo + 1) ? 1 : 0;
}

// Pass A's output regions WITHOUT counting the text first (an exact histogram of the top buckets is one more read of the text
// with a 9-step search per key: 1.08 ms of the 256 MiB text sort): the sorted sample already says how many suffixes a
// top bucket holds -- k_t sampled keys fall into bucket t (two binary searches per bucket; 4096 of them unless a heavy key
// swallows splitters), so n k_t / S suffixes, to within 1 / sqrt(k_t) = 1.6 % (one sigma).  Region t gets room for 1/8 more
// than that + 1024 entries (eight sigma); the regions lie one after the other in a VIRTUAL array of ~1.13 n entries
// whose first `n_main` entries are pass A's output buffers and whose rest spills into the idle suffix array.  Pass A fills
// a region through its cursor; should one run full all the same, the sort is left to the digit passes (`abandon`).
//   start[t] = first virtual entry of region t, start[kSplitTop] = end of the last; cursor_a[t] = start[t]
static __global__ __launch_bounds__(kSplitTop) void split_estimate_kernel(const uint64_t *__restrict__ sorted, const uint64_t *__restrict__ top, int64_t n,
                                                                    int64_t *__restrict__ start, unsigned long long *__restrict__ cursor_a)
{
    __shared__ int64_t wsum[kSplitTop / kWave];
    const int t = threadIdx.x, lane = lane_id(), w = t >> 6;
    auto lower = [&](uint64_t x) -> int64_t {               // first sampled key >= x
        int64_t lo = 0, hi = kSplitSample;
        while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (sorted[mid] < x) lo = mid + 1; else hi = mid; }
        return lo;
    };
    // bucket t holds the keys with top[t - 1] <= key < top[t]  (split_rank counts the splitters <= key)
    const int64_t lo = t == 0 ? 0 : lower(top[t - 1]);
    const int64_t hi = t == kSplitTop - 1 ? kSplitSample : lower(top[t]);
    const int64_t k = hi - lo;
    const int64_t est = (int64_t)(((unsigned __int128)(uint64_t)k * (uint64_t)n + kSplitSample - 1) / kSplitSample);
    const int64_t room = est + est / 8 + 1024;
    const int64_t incl = wave_incl_sum(room);
    if (lane == kWave - 1) wsum[w] = incl;
    __syncthreads();
    int64_t b = 0;
    for (int i = 0; i < w; ++i) b += wsum[i];
    start[t] = b + incl - room;
    cursor_a[t] = (unsigned long long)(b + incl - room);
    if (t == kSplitTop - 1) start[kSplitTop] = b + incl;
}

// After pass A: cnt[t] = entries region t received (its cursor minus its start), tile_first[t] = first tile of pass B in top
// bucket t (tile_first[kSplitTop] = all of them).  One workgroup of kSplitTop threads.
static __global__ __launch_bounds__(kSplitTop) void split_plan_kernel(const unsigned long long *__restrict__ cursor_a, const int64_t *__restrict__ start,
                                                                unsigned long long *__restrict__ cnt, uint32_t *__restrict__ tile_first,
                                                                SplitCtl *__restrict__ ctl)
{
    __shared__ int64_t wtl[kSplitTop / kWave];
    const int t = threadIdx.x, lane = lane_id(), w = t >> 6;
    const int64_t c = (int64_t)cursor_a[t] - start[t];
    cnt[t] = (unsigned long long)c;
    const int64_t tiles = (c + kSplitTileB - 1) / kSplitTileB;
    const int64_t it = wave_incl_sum(tiles);
    if (lane == kWave - 1) wtl[w] = it;
    __syncthreads();
    int64_t bt = 0;
    for (int i = 0; i < w; ++i) bt += wtl[i];
    tile_first[t] = (uint32_t)(bt + it - tiles);
    if (t == kSplitTop - 1) {
        tile_first[kSplitTop] = (uint32_t)(bt + it);
        ctl->tiles_b = (unsigned long long)(bt + it);
    }
}

// Pass A (kFromText): kin = the text, one tile = kSplitTileA consecutive suffixes; digit = top bucket; output position = a
// VIRTUAL entry from cursor[digit] (preset to the region starts, split_estimate_kernel): entries below `cap` (= n_main for
// this pass) lie in (kout0, vout0), the others in (kout1, vout1); an entry at or past its region's end raises `abandon`.
// Pass B: (kin, vin) / (kin1, vin1) = pass A's output, main and spill part (n_main = the boundary); workgroup b takes tile
// b - tile_first[t] of top bucket t, whose entries are region t's first cnt_a[t]; digit = part s of t
// (table = sub + t * kSplitSub); bucket = t * kSplitSub + s; the arrival number q of an entry in its bucket comes from
// cursor[bucket]; q < cap: slot entry q of the bucket -- the slots of the first half of the buckets lie in (kout0, vout0),
// the others in (kout1, vout1) --, else the overflow list.
template <typename IdxT, bool kFromText, bool kCoded>
__global__ __launch_bounds__(kSplitThreads, 4) void split_pass_kernel(
    const uint64_t *__restrict__ kin, const IdxT *__restrict__ vin, const uint64_t *__restrict__ kin1, const IdxT *__restrict__ vin1, int64_t n_main,
    int64_t n, const uint64_t *__restrict__ table, unsigned long long *__restrict__ cursor, const int64_t *__restrict__ off /* region starts */,
    const unsigned long long *__restrict__ cnt_a /* B: entries per region */, const uint32_t *__restrict__ tile_first,
    uint64_t *__restrict__ kout0, IdxT *__restrict__ vout0, uint64_t *__restrict__ kout1, IdxT *__restrict__ vout1, int64_t cap,
    uint64_t *__restrict__ ovf_key, IdxT *__restrict__ ovf_idx, int64_t ovf_cap, SplitCtl *__restrict__ ctl,
    const uint16_t *__restrict__ codetab, const uint8_t *__restrict__ pure = nullptr /* B: bucket holds copies of one key only */)
{
    constexpr int kDigits = kFromText ? kSplitTop : kSplitSub;
    constexpr int kSplitItems = kFromText ? kSplitItemsA : kSplitItemsB;
    constexpr int kSplitTile = kSplitThreads * kSplitItems;
    constexpr int kWavesB = kSplitThreads / kWave;
    constexpr int kExchRounds = 2;
    constexpr int kExchN = kSplitTile / kExchRounds;
    static_assert(kSplitItems % 4 == 0 && kSplitItems % kExchRounds == 0, "tile geometry");
    static_assert(kDigits <= kSplitThreads, "one thread per digit");
    __shared__ __attribute__((aligned(16))) uint64_t exch[kExchN];
    __shared__ uint64_t tab[kDigits];
    __shared__ uint32_t cnt[kDigits];
    __shared__ uint32_t tile_base[kDigits];
    __shared__ long long gofs[kDigits];                     // global position (A) / arrival number in the bucket (B) of a run's first entry, minus its place in the tile
    __shared__ long long oofs[kFromText ? 1 : kDigits];     // B: place on the overflow list of a run's first entry that does not fit, minus its place in the tile
    __shared__ uint16_t dig_of[kSplitTile];                 // digit of every position of the sorted tile
    __shared__ uint32_t wtmp[kWavesB];
    __shared__ unsigned long long s_obase, s_pbase;
    __shared__ uint16_t ctab[kCoded ? 256 : 1];
    __shared__ int s_t;

    const int tid = threadIdx.x, lane = lane_id(), w = tid >> 6;
    int64_t base = 0;
    int valid = 0;
    int t_b = 0;
    if (kFromText) {
        base = (int64_t)blockIdx.x * kSplitTile;
        valid = (n - base) < kSplitTile ? (int)(n - base) : kSplitTile;
        for (int i = tid; i < kDigits; i += kSplitThreads) { tab[i] = table[i]; cnt[i] = 0; }
        if (kCoded && tid < 256) ctab[tid] = codetab[tid];
    } else {
        if (blockIdx.x >= tile_first[kSplitTop]) return;    // (the grid is an upper bound)
        if (tid == 0) {
            int lo = 0, hi = kSplitTop;                     // last t with tile_first[t] <= blockIdx.x: the one that has this tile
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (tile_first[mid] <= blockIdx.x) lo = mid; else hi = mid; }
            s_t = lo;
        }
        __syncthreads();
        t_b = s_t;
        base = off[t_b] + (int64_t)(blockIdx.x - tile_first[t_b]) * kSplitTile;
        const int64_t left = off[t_b] + (int64_t)cnt_a[t_b] - base;
        valid = left < kSplitTile ? (int)left : kSplitTile;
        for (int i = tid; i < kDigits; i += kSplitThreads) { tab[i] = table[(int64_t)t_b * kSplitSub + i]; cnt[i] = 0; }
    }
    __syncthreads();

    // element index (inside the tile) of this lane's item k
    const int wbase = w * (kWave * kSplitItems) + lane;
    auto elem = [&](int k) -> int { return kFromText ? ((k >> 2) * kSplitThreads + tid) * 4 + (k & 3) : wbase + k * kWave; };

    uint64_t key[kSplitItems];
    IdxT val[kSplitItems];
    if (kFromText) {
        const uint32_t *t32 = reinterpret_cast<const uint32_t *>(kin);
#pragma unroll
        for (int j = 0; j < kSplitItems / 4; ++j) {
            const int e0 = (j * kSplitThreads + tid) * 4;
            if (e0 < valid) split_keys4<kCoded>(t32, (base + e0) >> 2, ctab, &key[4 * j]);
            else { key[4 * j] = key[4 * j + 1] = key[4 * j + 2] = key[4 * j + 3] = ~0ull; }
#pragma unroll
            for (int c = 0; c < 4; ++c) val[4 * j + c] = (IdxT)(base + e0 + c);
        }
    } else {
#pragma unroll
        for (int k = 0; k < kSplitItems; ++k) {
            const int e = wbase + k * kWave;
            const int64_t v = base + (e < valid ? e : valid - 1);       // (clamped, not predicated: the loads stay in flight together)
            key[k] = v < n_main ? kin[v] : kin1[v - n_main];
            val[k] = v < n_main ? vin[v] : vin1[v - n_main];
        }
    }

    // ---- digit = rank among the splitters; place inside the tile's digit run = arrival number (one returning LDS add).
    //      A wave whose 64 keys share a digit -- heavy keys come in runs: padding in text order, a heavy bucket's tiles in
    //      pass B -- would be 64 adds on one LDS address: it adds once and its lanes take consecutive numbers. ----
    uint32_t pd[kSplitItems];                                // digit << 16 | arrival number, then the place in the sorted tile
#pragma unroll
    for (int k = 0; k < kSplitItems; ++k) {
        const uint32_t d = split_rank<kDigits>(tab, key[k]);
        const bool ok = elem(k) < valid;
        const uint32_t d0 = (uint32_t)__builtin_amdgcn_readfirstlane(d);
        uint32_t a = 0;
        if (__all(ok && d == d0)) {
            uint32_t a0 = 0;
            if (lane == 0) a0 = atomicAdd(&cnt[d0], (uint32_t)kWave);
            a = (uint32_t)__builtin_amdgcn_readfirstlane(a0) + (uint32_t)lane;
        } else if (ok) {
            a = atomicAdd(&cnt[d], 1u);
        }
        pd[k] = ok ? ((d << 16) | a) : 0xffffffffu;
    }
    __syncthreads();

    // ---- digit totals: reserve the tile's place in every output region / bucket (the returned value is not needed
    //      before the LDS exchanges are done: the round trip overlaps them), tile-local scan ----
    unsigned long long abase = 0;
    uint32_t tot = 0, incl = 0, excl = 0;
    if (tid < kDigits) {
        tot = cnt[tid];
        if (tot) abase = atomicAdd(&cursor[kFromText ? tid : t_b * kSplitSub + tid], (unsigned long long)tot);
    }
    incl = wave_incl_sum(tot);
    if (lane == kWave - 1) wtmp[w] = incl;
    __syncthreads();
    if (tid < kDigits) {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < kWavesB; ++i) if (i < w) o += wtmp[i];
        excl = o + incl - tot;
        tile_base[tid] = excl;
    }
    __syncthreads();

    // ---- stage keys, then values, in digit order through LDS (two position ranges: half the footprint) ----
#pragma unroll
    for (int k = 0; k < kSplitItems; ++k) {
        if (pd[k] != 0xffffffffu) {
            const uint32_t d = pd[k] >> 16;
            const uint32_t p = tile_base[d] + (pd[k] & 0xffffu);
            dig_of[p] = (uint16_t)d;
            pd[k] = p;
        }
    }
    uint64_t skey[kSplitItems];
#pragma unroll
    for (int r = 0; r < kExchRounds; ++r) {
        if (r > 0) __syncthreads();
#pragma unroll
        for (int k = 0; k < kSplitItems; ++k) {
            const uint32_t p = pd[k] - (uint32_t)(r * kExchN);
            if (p < (uint32_t)kExchN) exch[p] = key[k];
        }
        __syncthreads();
#pragma unroll
        for (int k = r * (kSplitItems / kExchRounds); k < (r + 1) * (kSplitItems / kExchRounds); ++k)
            skey[k] = exch[k * kSplitThreads + tid - r * kExchN];
    }
    IdxT sval[kSplitItems];
    {
        constexpr int kValN = kExchN * (int)(sizeof(uint64_t) / sizeof(IdxT));
        constexpr int kValRounds = kSplitTile / kValN > 0 ? kSplitTile / kValN : 1;
        constexpr int kValCap = kSplitTile / kValRounds;
        IdxT *exv = reinterpret_cast<IdxT *>(exch);
#pragma unroll
        for (int r = 0; r < kValRounds; ++r) {
            __syncthreads();
#pragma unroll
            for (int k = 0; k < kSplitItems; ++k) {
                const uint32_t p = pd[k] - (uint32_t)(r * kValCap);
                if (p < (uint32_t)kValCap) exv[p] = val[k];
            }
            __syncthreads();
#pragma unroll
            for (int k = r * (kSplitItems / kValRounds); k < (r + 1) * (kSplitItems / kValRounds); ++k)
                sval[k] = exv[k * kSplitThreads + tid - r * kValCap];
        }
    }

    // ---- now the reserved places.  B: the entries of a digit run that do not fit their slot (arrival number >= cap) are
    //      its tail; the tile takes ONE stretch of the overflow list for all of them (one global add per tile that has
    //      any -- a heavy bucket sends whole tiles there: one add per wave and store instruction was 300 000 adds on one
    //      address for libtorch_cpu.so) ----
    if (tid < kDigits) {
        gofs[tid] = (long long)abase - (long long)excl;
        if (kFromText && tot && (long long)(abase + tot) > off[tid + 1]) {
            // the region is full (the estimate was off by more than its eight sigma of room): nothing of this run is written,
            // the caller takes the digit passes instead
            ctl->abandon = 1;
            gofs[tid] = -(1ll << 60);
        }
    }
    if (!kFromText) {
        // (a PURE bucket's overflow -- copies of one heavy key -- goes to the pure list instead: triples (bucket, arrival
        // number, suffix), growing downward from the end of the same arena, placed later without a sort; oofs of such a
        // digit is negative: -(1 + place of its first overflowing entry, counted from the arena's end) - ...)
        uint32_t fit = 0, over_n = 0, pure_n = 0;
        bool is_pure = false;
        if (tid < kDigits) {
            const long long room = cap - (long long)abase;
            fit = room <= 0 ? 0u : room >= (long long)tot ? tot : (uint32_t)room;
            over_n = tot - fit;
            is_pure = over_n != 0 && pure[(int64_t)t_b * kSplitSub + tid] != 0;
            if (is_pure) { pure_n = over_n; over_n = 0; }
        }
        // two counts in one scan: the pure ones in the upper half of the word (a tile holds < 2^16 entries)
        const uint32_t both = over_n | (pure_n << 16);
        const uint32_t oincl = wave_incl_sum(both);
        if (lane == kWave - 1) wtmp[w] = oincl;
        __syncthreads();
        uint32_t o = 0, total = 0;
#pragma unroll
        for (int i = 0; i < kWavesB; ++i) { if (i < w) o += wtmp[i]; total += wtmp[i]; }
        if (total) {                                         // (the same for every thread)
            if (tid == 0) {
                s_obase = (total & 0xffffu) ? atomicAdd(&ctl->ovf_count, (unsigned long long)(total & 0xffffu)) : 0ull;
                s_pbase = (total >> 16) ? atomicAdd(&ctl->pure_count, (unsigned long long)(total >> 16)) : 0ull;
            }
            __syncthreads();
            if (tid < kDigits) {
                const uint32_t ex = o + oincl - both;       // exclusive counts of the digits in front: low half normal, high half pure
                if (is_pure) oofs[tid] = -(1ll << 62) + (long long)s_pbase + (long long)(ex >> 16) - (long long)fit - (long long)excl;
                else oofs[tid] = (long long)s_obase + (long long)(ex & 0xffffu) - (long long)fit - (long long)excl;
            }
        }
    }
    __syncthreads();

    // ---- out: position p of the sorted tile belongs to digit dig_of[p]; consecutive lanes write consecutive entries of a run ----
#pragma unroll
    for (int k = 0; k < kSplitItems; ++k) {
        const int p = k * kSplitThreads + tid;
        const bool ok = p < valid;
        const uint32_t d = ok ? dig_of[p] : 0u;
        const long long q = gofs[d] + p;                     // A: position in the output; B: arrival number in the bucket
        if (kFromText) {
            if (ok && q >= 0) {                              // (q < 0: a run whose region ran full, see above)
                if (q < cap) { kout0[q] = skey[k]; vout0[q] = sval[k]; }
                else { kout1[q - cap] = skey[k]; vout1[q - cap] = sval[k]; }
            }
        } else if (ok) {
            if (q < cap) {
                const int64_t b = (int64_t)t_b * kSplitSub + d;
                if (b < kSplitBuckets / 2) { kout0[b * cap + q] = skey[k]; vout0[b * cap + q] = sval[k]; }
                else { kout1[(b - kSplitBuckets / 2) * cap + q] = skey[k]; vout1[(b - kSplitBuckets / 2) * cap + q] = sval[k]; }
            } else {
                const long long o = oofs[d] + p;
                if (o < 0) {                                 // a pure bucket's: (bucket, arrival number), downward from the arena's end
                    const long long j = o + (1ll << 62);
                    if (j < ovf_cap) {
                        ovf_key[ovf_cap - 1 - j] = ((uint64_t)((int64_t)t_b * kSplitSub + d) << 32) | (uint64_t)q;
                        ovf_idx[ovf_cap - 1 - j] = sval[k];
                    } else ctl->abandon = 1;
                } else if (o < ovf_cap) { ovf_key[o] = skey[k]; ovf_idx[o] = sval[k]; }
                else ctl->abandon = 1;
            }
        }
    }
}

// out_base[b] = final position of bucket b (exclusive scan of the bucket sizes = the cursors of pass B), out_base[NB] = n;
// for the k-th oversize bucket: ovf_src[k] = position of its first entry in the SORTED overflow list (exclusive scan of the
// sizes of the oversize buckets), ovf_dst[k] = its final position.  Two launches of kScanBlocks workgroups, every
// load coalesced: sums per 1024 buckets, then every workgroup adds up the sums in front of it and scans its own 1024.
constexpr int kScanThreads = 1024;
constexpr int kScanBlocks = kSplitBuckets / kScanThreads;
struct ScanPart { long long sum, osum, ocnt; };

static __global__ __launch_bounds__(kScanThreads) void bucket_sum_kernel(const unsigned long long *__restrict__ cursor, int64_t cap,
                                                                   const uint8_t *__restrict__ pure, ScanPart *__restrict__ part)
{
    __shared__ long long ws[3][kScanThreads / kWave];
    const int tid = threadIdx.x, lane = lane_id(), w = tid >> 6;
    const long long c = (long long)cursor[(int64_t)blockIdx.x * kScanThreads + tid];
    const bool over = c > cap && !pure[(int64_t)blockIdx.x * kScanThreads + tid];      // (a pure bucket's overflow is not on the sorted list)
    const long long a = wave_sum(c), o = wave_sum(over ? c : 0ll), k = wave_sum(over ? 1ll : 0ll);
    if (lane == 0) { ws[0][w] = a; ws[1][w] = o; ws[2][w] = k; }
    __syncthreads();
    if (tid == 0) {
        ScanPart p{0, 0, 0};
        for (int i = 0; i < kScanThreads / kWave; ++i) { p.sum += ws[0][i]; p.osum += ws[1][i]; p.ocnt += ws[2][i]; }
        part[blockIdx.x] = p;
    }
}

static __global__ __launch_bounds__(kScanThreads) void bucket_scan_kernel(const unsigned long long *__restrict__ cursor, int64_t cap,
                                                                    const uint8_t *__restrict__ pure, const ScanPart *__restrict__ part, int64_t *__restrict__ out_base,
                                                                    int64_t *__restrict__ ovf_src, int64_t *__restrict__ ovf_dst,
                                                                    SplitCtl *__restrict__ ctl)
{
    __shared__ long long ws[3][kScanThreads / kWave];
    __shared__ long long s_pre[3];
    const int tid = threadIdx.x, lane = lane_id(), w = tid >> 6;
    if (w == 0) {                                            // what lies in front of this workgroup's 1024 buckets
        long long a = 0, o = 0, k = 0;
        for (int i = lane; i < (int)blockIdx.x; i += kWave) { a += part[i].sum; o += part[i].osum; k += part[i].ocnt; }
        a = wave_sum(a); o = wave_sum(o); k = wave_sum(k);
        if (lane == 0) { s_pre[0] = a; s_pre[1] = o; s_pre[2] = k; }
    }
    const int64_t b = (int64_t)blockIdx.x * kScanThreads + tid;
    const long long c = (long long)cursor[b];
    const bool over = c > cap && !pure[b];
    const long long ia = wave_incl_sum(c), io = wave_incl_sum(over ? c : 0ll), ik = wave_incl_sum(over ? 1ll : 0ll);
    if (lane == kWave - 1) { ws[0][w] = ia; ws[1][w] = io; ws[2][w] = ik; }
    __syncthreads();
    long long a = s_pre[0], o = s_pre[1], k = s_pre[2];
    for (int i = 0; i < w; ++i) { a += ws[0][i]; o += ws[1][i]; k += ws[2][i]; }
    out_base[b] = a + ia - c;
    if (over) { ovf_src[k + ik - 1] = o + io - c; ovf_dst[k + ik - 1] = a + ia - c; }
    if (b == kSplitBuckets - 1) { out_base[kSplitBuckets] = a + ia; ctl->ovf_buckets = (unsigned long long)(k + ik); }
}

// One workgroup per bucket: sort its (key, suffix) entries by key inside LDS, write them to the bucket's final position.
//   1. kThreads / 2 of the bucket's own keys, ranked by counting, are its local splitters: part of a key = 2 #{splitters <
//      key} + (key is a splitter value) -- rank-based, so the parts are balanced whatever the keys look like, and a key that
//      occurs often is a splitter value with near certainty: its copies form a part of their own, which needs no order
//      (round 0 need not be stable)
//   2. count, scan, scatter (key, suffix, part) into part order
//   3. every POSITION of the part-ordered bucket (a wave = 64 consecutive positions = a few parts: the loop below has one
//      trip count for most of the wave and reads the same LDS words in every lane -- broadcasts) counts the smaller keys
//      of its part and the equal ones in front of it: its final place; the entry leaves from there, the 64 lanes writing
//      inside one window of a few hundred bytes.
// What a CU gets through is set by its LDS (~47 LDS instructions per entry: the searches' random 8-byte reads, the scatter,
// the broadcast reads of the ranking and of the walk) and by how many buckets it holds at once: 256 threads x 8 entries,
// 31 KB of LDS, five workgroups per CU; item slots beyond the bucket's size are skipped as a whole (a 256 MiB text has 1024
// entries per bucket: four of the eight).  A <512, 8> geometry for 4096-entry buckets (62 KB: two per CU) took twice as
// long per entry -- hence 2^18 buckets of <= 2048 rather than 2^17 of <= 4096.  lo < size <= hi selects the buckets of a
// launch (one launch today); the launch with `oversize` set also moves the buckets that are longer than their slot to
// the overflow list.
template <typename IdxT, int kThreads, int kItems, int kSample = kThreads / 2>
__global__ __launch_bounds__(kThreads, kThreads * kItems <= 1024 ? 8 : kThreads == 256 ? 5 : 4) void bucket_finish_kernel(
    const uint64_t *__restrict__ kslot0, const IdxT *__restrict__ vslot0, const uint64_t *__restrict__ kslot1, const IdxT *__restrict__ vslot1,
    int64_t cap, int64_t lo, int64_t hi, bool oversize, const unsigned long long *__restrict__ cursor, const int64_t *__restrict__ out_base,
    uint64_t *__restrict__ kout, IdxT *__restrict__ sa, uint64_t *__restrict__ ovf_key, IdxT *__restrict__ ovf_idx, int64_t ovf_cap,
    SplitCtl *__restrict__ ctl, const uint8_t *__restrict__ pure)
{
    constexpr int kCap = kThreads * kItems;
    // local splitters.  Per entry of a 1024-entry bucket: log2(kSample) + 2 random LDS reads for its search, kSample^2 / 1024
    // broadcast reads for ranking the sample, ~1024 / kSample steps of the walk over its part (a broadcast read + ~8 VALU
    // each).  Measured on the 256 MiB text: 128 splitters 3.48 ms, 64 splitters 4.17 -- the walk costs more than its reads
    constexpr int kParts = 2 * kSample + 1;
    static_assert(2 * kSample <= kThreads && kParts <= kThreads + 1, "two threads rank a splitter, one thread owns a part (the last one two)");
    __shared__ uint64_t skey[kCap];
    __shared__ uint32_t sidx[kCap];
    __shared__ uint16_t spart[kCap];
    __shared__ uint64_t smp[kSample], srt[kSample];
    __shared__ uint32_t pcnt[kParts + 1];
    __shared__ uint32_t wtmp[kThreads / kWave];
    __shared__ unsigned long long s_o0;
    const int tid = threadIdx.x, lane = lane_id(), w = tid >> 6;
    const int64_t b = blockIdx.x;
    const uint64_t *ks = b < kSplitBuckets / 2 ? kslot0 + b * cap : kslot1 + (b - kSplitBuckets / 2) * cap;
    const IdxT *vs = b < kSplitBuckets / 2 ? vslot0 + b * cap : vslot1 + (b - kSplitBuckets / 2) * cap;
    // (the size first: a bucket of the other launch's class costs one load, not its slot)
    const int64_t c64 = (int64_t)cursor[b];
    const int64_t ob = out_base[b];
    if (pure[b]) {
        // copies of one key: nothing to sort -- the slot's entries (what did not fit is on the pure list) go out as they
        // are, by the launch that also takes the oversize buckets
        if (!oversize) return;
        const int64_t cnt = c64 < cap ? c64 : cap;
        for (int64_t i = tid; i < cnt; i += kThreads) { kout[ob + i] = ks[i]; sa[ob + i] = vs[i]; }
        return;
    }
    if (c64 <= cap ? (c64 <= lo || c64 > hi) : !oversize) return;       // (empty, or the other launch's)
    uint64_t key[kItems];
    uint32_t idx[kItems];
    const int last = (int)(cap < kCap ? cap : kCap) - 1;
#pragma unroll
    for (int k = 0; k < kItems; ++k) {
        const int e = k * kThreads + tid;
        const int ec = e < last ? e : last;                  // (inside the slot whatever the size: the loads do not wait for it)
        key[k] = ks[ec];
        idx[k] = (uint32_t)vs[ec];
    }
    if (c64 > cap) {
        // oversize: the entries that did fit the slot join the rest of the bucket on the overflow list
        if (tid == 0) s_o0 = atomicAdd(&ctl->ovf_count, (unsigned long long)cap);
        __syncthreads();
        const long long o0 = (long long)s_o0;
        if (o0 + cap > ovf_cap) { if (tid == 0) ctl->abandon = 1; return; }
        for (int64_t i = tid; i < cap; i += kThreads) { ovf_key[o0 + i] = ks[i]; ovf_idx[o0 + i] = vs[i]; }
        return;
    }
    const int c = (int)c64;                                  // <= kCap
    const int kmax = (c + kThreads - 1) / kThreads;          // item slots in use (the same for every thread: the others are skipped as a whole)
    for (int i = tid; i <= kParts; i += kThreads) pcnt[i] = 0;
    // ---- local splitters: entry floor(i c / kSample) of the slot for i < kSample, taken from the registers that hold it ----
#pragma unroll
    for (int k = 0; k < kItems; ++k) {
        const int e = k * kThreads + tid;
        if (k < kmax && e < c) {
            for (int i = (int)(((int64_t)e * kSample + c - 1) / c); i < kSample && (int)(((int64_t)i * c) / kSample) == e; ++i) smp[i] = key[k];
        }
    }
    __syncthreads();
    if (tid < 2 * kSample) {   // ranked by counting, two threads per splitter (half of the others each); whole waves take part
        const int i = tid >> 1, half = tid & 1;
        const uint64_t me = smp[i];
        uint32_t r = 0;
#pragma unroll 8
        for (int j = half * (kSample / 2); j < (half + 1) * (kSample / 2); ++j) {
            const uint64_t o = smp[j];
            r += ((o < me) || (o == me && j < i)) ? 1u : 0u;
        }
        r += (uint32_t)__shfl_xor((int)r, 1, kWave);
        if (half == 0) srt[r] = me;
    }
    __syncthreads();
    // ---- part of every key; its arrival number in the part ----
    uint32_t part[kItems];                                   // part << 16 | arrival number
#pragma unroll
    for (int k = 0; k < kItems; ++k) {
        part[k] = 0xffffffffu;
        if (k >= kmax) continue;
        uint32_t lb = 0;                                     // #{srt < key}
#pragma unroll
        for (int step = kSample / 2; step >= 1; step >>= 1) lb += (srt[lb + step - 1] < key[k]) ? (uint32_t)step : 0u;
        lb += (srt[lb] < key[k]) ? 1u : 0u;                  // (kSample entries, not kSample - 1: one more step)
        const uint32_t eq = (lb < (uint32_t)kSample && srt[lb < (uint32_t)kSample ? lb : 0] == key[k]) ? 1u : 0u;
        const uint32_t p = 2 * lb + eq;
        const bool ok = k * kThreads + tid < c;
        const uint32_t a = ok ? atomicAdd(&pcnt[p], 1u) : 0u;
        part[k] = ok ? ((p << 16) | a) : 0xffffffffu;
    }
    __syncthreads();
    // ---- exclusive scan of the part sizes (thread t owns part t; the last thread also part kThreads if there is one) ----
    {
        const uint32_t mine = tid < kParts ? pcnt[tid] : 0u;
        const uint32_t v = mine + ((kParts > kThreads && tid == kThreads - 1) ? pcnt[kThreads] : 0u);
        const uint32_t incl = wave_incl_sum(v);
        if (lane == kWave - 1) wtmp[w] = incl;
        __syncthreads();
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < kThreads / kWave; ++i) if (i < w) o += wtmp[i];
        const uint32_t excl = o + incl - v;
        if (tid < kParts) pcnt[tid] = excl;                  // (every count was read before the barrier above)
        if (kParts > kThreads && tid == kThreads - 1) pcnt[kThreads] = excl + mine;
        if (tid == 0) pcnt[kParts] = (uint32_t)c;
    }
    __syncthreads();
    // ---- into part order ----
#pragma unroll
    for (int k = 0; k < kItems; ++k) {
        if (k < kmax && part[k] != 0xffffffffu) {
            const uint32_t p = part[k] >> 16;
            const uint32_t slot = pcnt[p] + (part[k] & 0xffffu);
            skey[slot] = key[k];
            sidx[slot] = idx[k];
            spart[slot] = (uint16_t)p;
        }
    }
    __syncthreads();
    // ---- final place of every position; out ----
#pragma unroll
    for (int k = 0; k < kItems; ++k) {
        const int pos = k * kThreads + tid;
        if (k < kmax && pos < c) {
            const uint64_t me = skey[pos];
            const uint32_t p = spart[pos];
            uint32_t r = (uint32_t)pos;                       // a splitter-valued key stays where it is
            if (!(p & 1u)) {
                const uint32_t s0 = pcnt[p], s1 = pcnt[p + 1];
                r = s0;
                for (uint32_t j = s0; j < s1; j += 4) {        // (four loads in flight; the clamped ones count nothing)
#pragma unroll
                    for (uint32_t i = 0; i < 4; ++i) {
                        const uint32_t jj = j + i;
                        const uint64_t o = skey[jj < s1 ? jj : s1 - 1];
                        r += (jj < s1 && ((o < me) || (o == me && jj < (uint32_t)pos))) ? 1u : 0u;
                    }
                }
            }
            kout[ob + r] = me;
            sa[ob + r] = (IdxT)sidx[pos];
        }
    }
}

// (Measured against this kernel and dropped, round 6: the same finish as BITONIC NETWORKS IN REGISTERS, one wave per bucket of
// <= 1024 entries (16 per lane, distances below 16 inside the lane, 21 shuffle stages across lanes, 96-bit compares so that
// pads sort last), two waves + an LDS exchange for 1025 ... 2048 -- straight-line VALU work instead of LDS round trips between
// barriers, bit-exact on every test: 7.60 ms against 3.46 on the 256 MiB text, 3.21 against 1.91 on 128 MiB.  144 / 168 VGPRs
// leave three waves per SIMD and O(log^2) stages of dependent selects do not beat ~47 LDS instructions per entry.)

// entry j of the sorted overflow list belongs to the last oversize bucket k with ovf_src[k] <= j (binary search: a few
// thousand buckets at most) and goes to ovf_dst[k] + (j - ovf_src[k]).  One thread per ENTRY: one bucket may hold
// millions (a run of 5.5 MB of 'X' in libtorch_cpu.so; a workgroup per bucket took 10 ms over it).
template <typename IdxT>
__global__ __launch_bounds__(kBlock) void overflow_place_kernel(int64_t count, int64_t nbuckets, const int64_t *__restrict__ ovf_src,
                                                             const int64_t *__restrict__ ovf_dst, const uint64_t *__restrict__ okey,
                                                             const IdxT *__restrict__ oidx, uint64_t *__restrict__ kout, IdxT *__restrict__ sa)
{
    const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= count) return;
    int64_t lo = 0, hi = nbuckets;                           // ovf_src[0] = 0 <= j
    while (hi - lo > 1) { const int64_t mid = (lo + hi) >> 1; if (ovf_src[mid] <= j) lo = mid; else hi = mid; }
    const int64_t dst = ovf_dst[lo] + (j - ovf_src[lo]);
    kout[dst] = okey[j];
    sa[dst] = oidx[j];
}

// entry j of the pure list (stored downward from the end of the overflow arena): copy number q of bucket b's one key ->
// position out_base[b] + q, key low[b].  No sort: one thread per entry.
template <typename IdxT>
__global__ __launch_bounds__(kBlock) void pure_place_kernel(int64_t count, int64_t ovf_cap, const uint64_t *__restrict__ pkey,
                                                         const IdxT *__restrict__ pidx, const int64_t *__restrict__ out_base,
                                                         const uint64_t *__restrict__ low, uint64_t *__restrict__ kout, IdxT *__restrict__ sa)
{
    const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= count) return;
    const uint64_t w = pkey[ovf_cap - 1 - j];
    const int64_t b = (int64_t)(w >> 32), q = (int64_t)(w & 0xffffffffull);
    kout[out_base[b] + q] = low[b];
    sa[out_base[b] + q] = pidx[ovf_cap - 1 - j];
}

}  // namespace dq
