// dq_sa_kernels.h -- suffix-array specific kernels around the radix engine:
//   pack_keys        text -> 8-byte big-endian keys (zero padded past the end)
//   seg_reduce/scan/apply   "rank rebucketing": group heads, device-wide
//                    (max, max, sum) scan, SA / ISA scatter, compaction of the
//                    suffixes that are still in groups of size > 1
//   gather_key2      composite (rank, ISA[s + h]) keys for the next doubling round
//
// Order contract (reference: LibDivSufSortTests.cs:43-59 -- unsigned bytes,
// a proper prefix sorts first): suffixes that run off the end of the text are
// resolved by gather_key2's "past the end" rule, never by the zero padding.
#pragma once
#include "dq_device_utils.h"

namespace dq {

constexpr int kSegItems = 8;
constexpr int kSegTile = kBlock * kSegItems;    // 2048 elements per workgroup

// ---------------------------------------------------------------------------------
// pack_keys: key[i] = T[i] T[i+1] ... T[i+7] as a big-endian u64.
// `text` is the library's own padded copy: >= 16 zero bytes follow T[n-1] and the
// base is 16-byte aligned, so 4-byte loads at i, i+4, i+8 are always in bounds.
// Each lane packs 4 consecutive suffixes (one dword of text + 2 dwords of halo).
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void pack_keys_kernel(const uint8_t *__restrict__ text,
                                                           int64_t n, uint64_t *__restrict__ keys)
{
    const uint32_t *t32 = reinterpret_cast<const uint32_t *>(text);
    const int64_t nquads = (n + 3) >> 2;
    for (int64_t qd = (int64_t)blockIdx.x * kBlock + threadIdx.x; qd < nquads;
         qd += (int64_t)gridDim.x * kBlock) {
        const uint32_t w0 = t32[qd], w1 = t32[qd + 1], w2 = t32[qd + 2];
        const uint64_t x = __builtin_bswap64((uint64_t)w0 | ((uint64_t)w1 << 32));   // b0..b7
        const uint64_t y = (uint64_t)__builtin_bswap32(w2) << 32;                    // b8..b11
        uint64_t k[4];
        k[0] = x;
        k[1] = (x << 8) | (y >> 56);
        k[2] = (x << 16) | (y >> 48);
        k[3] = (x << 24) | (y >> 40);
        const int64_t i = qd << 2;
        if (i + 4 <= n) {
            ulonglong2 a, b;
            a.x = k[0]; a.y = k[1]; b.x = k[2]; b.y = k[3];
            *reinterpret_cast<ulonglong2 *>(keys + i) = a;
            *reinterpret_cast<ulonglong2 *>(keys + i + 2) = b;
        } else {
            for (int j = 0; j < 4; ++j)
                if (i + j < n) keys[i + j] = k[j];
        }
    }
}

// ---------------------------------------------------------------------------------
// Rebucketing over a list of m (composite key, suffix) pairs sorted by composite key.
//   kInitial: the list is the whole suffix array sorted by its 8-byte text key; there
//             is one parent group [0, n) with rank 0.
//   else:     composite key = (parent rank << kbits) | key2; parent groups are runs of
//             equal rank, and the parent rank IS the SA position of the group's first
//             member, so member j lands at SA[rank + (j - first j of its group)].
// For element j:  nh(j) = last new-group head <= j,  gh(j) = last parent-group head <= j
//   SA position p = rank + (j - gh),  new rank = rank + (nh - gh).
// A suffix stays active iff its new group has more than one member.
// ---------------------------------------------------------------------------------
template <typename IdxT>
struct SegPartials {
    IdxT *nh;    // per-workgroup: last new head index in the tile, or -1
    IdxT *gh;    // per-workgroup: last parent-group head index in the tile, or -1
    IdxT *cnt;   // per-workgroup: number of still-active elements
};

template <typename IdxT, bool kInitial>
struct SegTileLoader {
    // loads this thread's kSegItems consecutive composite keys plus the neighbours
    // needed for head / activity tests
    uint64_t ck[kSegItems];
    uint64_t prev, next;
    bool has_prev, has_next;
    // kshift > 0: the keys are packed (key << kshift | suffix) words; compare the key part only
    __device__ __forceinline__ void load(const uint64_t *__restrict__ keys, int64_t m, int64_t j0, int kshift)
    {
#pragma unroll
        for (int i = 0; i < kSegItems; ++i) ck[i] = (j0 + i < m) ? keys[j0 + i] >> kshift : 0;
        has_prev = j0 > 0 && j0 <= m;
        prev = has_prev ? keys[j0 - 1] >> kshift : 0;
        has_next = j0 + kSegItems < m;
        next = has_next ? keys[j0 + kSegItems] >> kshift : 0;
    }
};

template <typename IdxT, bool kInitial>
__global__ __launch_bounds__(kBlock) void seg_reduce_kernel(const uint64_t *__restrict__ keys,
                                                            int64_t m, int kbits,
                                                            SegPartials<IdxT> part, int kshift)
{
    __shared__ IdxT tmp[3][kWavesPerBlock];
    const int tid = threadIdx.x;
    const int64_t j0 = (int64_t)blockIdx.x * kSegTile + (int64_t)tid * kSegItems;
    SegTileLoader<IdxT, kInitial> t;
    t.load(keys, m, j0, kshift);

    IdxT last_nh = -1, last_gh = -1, cnt = 0;
    bool head[kSegItems + 1];
#pragma unroll
    for (int i = 0; i < kSegItems; ++i) {
        const int64_t j = j0 + i;
        const uint64_t p = i == 0 ? t.prev : t.ck[i - 1];
        const bool first = (j == 0);
        head[i] = first || t.ck[i] != p;
        const bool ghead = kInitial ? first : (first || (t.ck[i] >> kbits) != (p >> kbits));
        if (j < m) {
            if (head[i]) last_nh = (IdxT)j;
            if (ghead) last_gh = (IdxT)j;
        }
    }
    head[kSegItems] = !t.has_next || t.next != t.ck[kSegItems - 1];
#pragma unroll
    for (int i = 0; i < kSegItems; ++i) {
        const int64_t j = j0 + i;
        const bool next_head = (j + 1 >= m) ? true : head[i + 1];
        if (j < m && !(head[i] && next_head)) ++cnt;
    }
    last_nh = wave_max(last_nh);
    last_gh = wave_max(last_gh);
    cnt = wave_sum(cnt);
    const int w = tid >> 6;
    if (lane_id() == 0) { tmp[0][w] = last_nh; tmp[1][w] = last_gh; tmp[2][w] = cnt; }
    __syncthreads();
    if (tid == 0) {
        IdxT a = tmp[0][0], b = tmp[1][0], c = tmp[2][0];
        for (int i = 1; i < kWavesPerBlock; ++i) {
            a = tmp[0][i] > a ? tmp[0][i] : a;
            b = tmp[1][i] > b ? tmp[1][i] : b;
            c += tmp[2][i];
        }
        part.nh[blockIdx.x] = a;
        part.gh[blockIdx.x] = b;
        part.cnt[blockIdx.x] = c;
    }
}

// Exclusive scan of the per-workgroup partials by ONE workgroup of 1024 threads
// (prefix max, prefix max, prefix sum), in coalesced chunks of 4096 entries with a running
// carry.  totals[0] receives the active count.
template <typename IdxT>
__global__ __launch_bounds__(1024) void seg_scan_kernel(SegPartials<IdxT> part, int64_t nparts,
                                                        int64_t *__restrict__ totals)
{
    constexpr int kPer = 4;
    constexpr int kWaves = 1024 / kWave;
    __shared__ IdxT s_nh[kWaves], s_gh[kWaves], s_cnt[kWaves];
    const int tid = threadIdx.x;
    const int lane = lane_id();
    const int w = tid >> 6;
    IdxT carry_nh = -1, carry_gh = -1, carry_cnt = 0;
    for (int64_t c0 = 0; c0 < nparts; c0 += 1024 * kPer) {
        const int64_t i0 = c0 + (int64_t)tid * kPer;
        IdxT x[kPer], y[kPer], z[kPer];
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            const bool ok = i0 + k < nparts;
            x[k] = ok ? part.nh[i0 + k] : (IdxT)-1;
            y[k] = ok ? part.gh[i0 + k] : (IdxT)-1;
            z[k] = ok ? part.cnt[i0 + k] : (IdxT)0;
        }
        IdxT a = x[0], b = y[0], c = z[0];
#pragma unroll
        for (int k = 1; k < kPer; ++k) { a = x[k] > a ? x[k] : a; b = y[k] > b ? y[k] : b; c += z[k]; }
        // inclusive scans across the 1024 threads
        IdxT ia = wave_incl_max(a), ib = wave_incl_max(b), ic = wave_incl_sum(c);
        if (lane == kWave - 1) { s_nh[w] = ia; s_gh[w] = ib; s_cnt[w] = ic; }
        __syncthreads();
        IdxT pa = carry_nh, pb = carry_gh, pc = carry_cnt;       // prefix of earlier waves + chunks
        IdxT ta = carry_nh, tb = carry_gh, tc = carry_cnt;       // ... including this whole chunk
#pragma unroll
        for (int i = 0; i < kWaves; ++i) {
            const IdxT u = s_nh[i], v = s_gh[i], q = s_cnt[i];
            if (i < w) { pa = u > pa ? u : pa; pb = v > pb ? v : pb; pc += q; }
            ta = u > ta ? u : ta; tb = v > tb ? v : tb; tc += q;
        }
        // exclusive value for this thread = earlier waves/chunks + earlier lanes of this wave
        IdxT ea = __shfl_up(ia, 1, kWave), eb = __shfl_up(ib, 1, kWave), ec = __shfl_up(ic, 1, kWave);
        if (lane == 0) { ea = -1; eb = -1; ec = 0; }
        pa = ea > pa ? ea : pa; pb = eb > pb ? eb : pb; pc += ec;
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            if (i0 + k < nparts) { part.nh[i0 + k] = pa; part.gh[i0 + k] = pb; part.cnt[i0 + k] = pc; }
            pa = x[k] > pa ? x[k] : pa; pb = y[k] > pb ? y[k] : pb; pc += z[k];
        }
        carry_nh = ta; carry_gh = tb; carry_cnt = tc;
        __syncthreads();
    }
    if (tid == 0) totals[0] = (int64_t)carry_cnt;
}

template <typename IdxT, bool kInitial, bool kWriteSA, bool kWriteISA>
__global__ __launch_bounds__(kBlock) void seg_apply_kernel(
    const uint64_t *__restrict__ keys, const IdxT *__restrict__ vals, int64_t m, int kbits,
    SegPartials<IdxT> part, IdxT *__restrict__ SA, IdxT *__restrict__ ISA,
    uint64_t *__restrict__ act_rank, IdxT *__restrict__ act_suf, int kshift)
{
    __shared__ IdxT tmp[kWavesPerBlock];
    const int tid = threadIdx.x;
    const int64_t j0 = (int64_t)blockIdx.x * kSegTile + (int64_t)tid * kSegItems;
    SegTileLoader<IdxT, kInitial> t;
    t.load(keys, m, j0, kshift);
    IdxT suf[kSegItems];
#pragma unroll
    for (int i = 0; i < kSegItems; ++i) suf[i] = (j0 + i < m) ? vals[j0 + i] : (IdxT)0;

    bool head[kSegItems + 1];
    bool ghead[kSegItems];
    IdxT last_nh = -1, last_gh = -1;
#pragma unroll
    for (int i = 0; i < kSegItems; ++i) {
        const int64_t j = j0 + i;
        const uint64_t p = i == 0 ? t.prev : t.ck[i - 1];
        const bool first = (j == 0);
        head[i] = first || t.ck[i] != p;
        ghead[i] = kInitial ? first : (first || (t.ck[i] >> kbits) != (p >> kbits));
        if (j < m) {
            if (head[i]) last_nh = (IdxT)j;
            if (ghead[i]) last_gh = (IdxT)j;
        }
    }
    head[kSegItems] = !t.has_next || t.next != t.ck[kSegItems - 1];
    bool active[kSegItems];
    IdxT cnt = 0;
#pragma unroll
    for (int i = 0; i < kSegItems; ++i) {
        const int64_t j = j0 + i;
        const bool next_head = (j + 1 >= m) ? true : head[i + 1];
        active[i] = j < m && !(head[i] && next_head);
        cnt += active[i] ? 1 : 0;
    }

    // workgroup-wide exclusive scans over the threads: running last-head indices
    // (max) and the running active count (sum)
    IdxT run_nh = block_excl_max(last_nh, tmp);
    __syncthreads();
    IdxT run_gh = block_excl_max(last_gh, tmp);
    __syncthreads();
    IdxT dummy;
    const IdxT cnt_excl = block_excl_sum(cnt, tmp, &dummy);

    const IdxT blk_nh = part.nh[blockIdx.x], blk_gh = part.gh[blockIdx.x];
    run_nh = blk_nh > run_nh ? blk_nh : run_nh;
    run_gh = blk_gh > run_gh ? blk_gh : run_gh;
    int64_t o = (int64_t)part.cnt[blockIdx.x] + (int64_t)cnt_excl;

#pragma unroll
    for (int i = 0; i < kSegItems; ++i) {
        const int64_t j = j0 + i;
        if (j >= m) break;
        if (head[i]) run_nh = (IdxT)j;
        if (ghead[i]) run_gh = (IdxT)j;
        const IdxT rank = kInitial ? (IdxT)0 : (IdxT)(t.ck[i] >> kbits);
        const IdxT p = rank + ((IdxT)j - run_gh);
        const IdxT nr = rank + (run_nh - run_gh);
        if (kWriteSA) SA[p] = suf[i];
        if (kWriteISA) ISA[suf[i]] = nr;
        if (active[i]) {
            act_rank[o] = (uint64_t)nr;
            act_suf[o] = suf[i];
            ++o;
        }
    }
}

// ---------------------------------------------------------------------------------
// gather_key2: composite[j] = (rank_j << kbits) | key2_j in place, where
//   key2 = ISA[s + h] + h          when s + h < n   (rank of the suffix h further on)
//        = n - 1 - s               otherwise        (ran off the end: the SHORTER suffix,
//                                                    i.e. the larger s, sorts first)
// Past-the-end values are < h, in-range values are >= h, so a suffix that is a proper
// prefix of another sorts first, exactly like ReadOnlySpan<byte>.SequenceCompareTo.
// ---------------------------------------------------------------------------------
template <typename IdxT>
__global__ __launch_bounds__(kBlock) void gather_key2_kernel(uint64_t *__restrict__ comp,
                                                             const IdxT *__restrict__ suf,
                                                             const IdxT *__restrict__ ISA,
                                                             int64_t m, int64_t n, int64_t h, int kbits)
{
    for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < m;
         j += (int64_t)gridDim.x * kBlock) {
        const int64_t s = (int64_t)suf[j];
        const int64_t q = s + h;
        const uint64_t k2 = q < n ? (uint64_t)((int64_t)ISA[q] + h) : (uint64_t)(n - 1 - s);
        comp[j] = (comp[j] << kbits) | k2;
    }
}

// ---------------------------------------------------------------------------------
// Sparse finishing (few suffixes still tied, e.g. random-like inputs): instead of building
// the full inverse suffix array, extend the tied suffixes' keys with the next `ebytes` bytes
// of text:  composite = (rank << kbits) | (bytes, zero padded) << 3 | valid_len,
// kbits = 8*ebytes + 3.  (padded bytes, valid length) orders a suffix that ends inside the
// window before one that continues with real zero bytes -- SequenceCompareTo again.
// ---------------------------------------------------------------------------------
template <typename IdxT>
__global__ __launch_bounds__(kBlock) void gather_text_key_kernel(uint64_t *__restrict__ comp,
                                                                 const IdxT *__restrict__ suf,
                                                                 const uint8_t *__restrict__ text,
                                                                 int64_t m, int64_t n, int64_t h, int ebytes)
{
    const int kbits = 8 * ebytes + 3;
    for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < m;
         j += (int64_t)gridDim.x * kBlock) {
        const int64_t q = (int64_t)suf[j] + h;
        int64_t len = n - q;
        len = len < 0 ? 0 : (len > ebytes ? ebytes : len);
        uint64_t bytes = 0;
        for (int b = 0; b < ebytes; ++b) bytes = (bytes << 8) | (b < len ? (uint64_t)text[q + b] : 0ull);
        comp[j] = (comp[j] << kbits) | (bytes << 3) | (uint64_t)len;
    }
}

// ---------------------------------------------------------------------------------
// small_group_finish: most ties left by round 0 on random-like data are groups of 2-3
// suffixes that differ a few bytes further on.  The first member's lane sorts a group of
// <= kMaxG suffixes by direct text comparison from offset h (at most kMaxLen bytes; a suffix
// that ends first sorts first) and writes SA[rank + i].  What it cannot finish -- groups that
// are larger, or still undecided after kMaxLen bytes -- is appended to the list (out_rank,
// out_suf) for the key-extension rounds: one atomic per workgroup that has leftovers, order
// arbitrary (the next step radix-sorts the list anyway).  *out_count must be zero on entry.
// With m_dev the list length is read on the device (launch without a host round trip).
// ---------------------------------------------------------------------------------
constexpr int kFinishThreads = 256;

template <typename IdxT, int kMaxG, int kMaxLen>
__global__ __launch_bounds__(kFinishThreads) void small_group_finish_kernel(
    const uint64_t *__restrict__ rank, const IdxT *__restrict__ suf, const uint8_t *__restrict__ text,
    int64_t m, int64_t n, int64_t h, IdxT *__restrict__ SA, uint64_t *__restrict__ out_rank,
    IdxT *__restrict__ out_suf, unsigned long long *__restrict__ out_count,
    const unsigned long long *__restrict__ m_dev = nullptr)
{
    // speculative launch (the list length is still on the device): m is the capacity the grid was
    // sized for; a longer list makes the host redo the step, so nothing is done for it here
    if (m_dev) {
        const unsigned long long real = m_dev[0];
        if (real > (unsigned long long)m || m_dev[1] != 0) return;      // m_dev[1]: the producer gave up (TieCounters)
        m = (int64_t)real;
    }
    __shared__ uint32_t wave_tot[kFinishThreads / kWave];
    __shared__ unsigned long long s_base;
    const int lane = lane_id();
    const int w = threadIdx.x >> 6;
    // grid-stride over tiles of kFinishThreads entries (the trip count is uniform per workgroup)
    for (int64_t j0 = (int64_t)blockIdx.x * kFinishThreads; j0 < m; j0 += (int64_t)gridDim.x * kFinishThreads) {
    const int64_t j = j0 + threadIdx.x;

    uint32_t emit = 0;                 // entries this lane appends: itself (large group) or its whole undecided group
    uint64_t r = 0;
    int g = 1;
    bool head = false;
    if (j < m) {
        r = rank[j];
        int left = 0, right = 0;
#pragma unroll
        for (int i = 1; i <= kMaxG; ++i)
            if (left == i - 1 && j - i >= 0 && rank[j - i] == r) left = i;
#pragma unroll
        for (int i = 1; i <= kMaxG; ++i)
            if (right == i - 1 && j + i < m && rank[j + i] == r) right = i;
        g = left + right + 1;
        if (g > kMaxG) emit = 1;
        else head = left == 0;
    }
    int64_t s[kMaxG];
    if (head) {
#pragma unroll
        for (int i = 0; i < kMaxG; ++i) s[i] = i < g ? (int64_t)suf[j + i] : 0;
        // -1: a < b, +1: a > b, 0: undecided within kMaxLen bytes
        auto cmp = [&](int64_t a, int64_t b) -> int {
            const int64_t pa = a + h, pb = b + h;
            for (int k = 0; k < kMaxLen; ++k) {
                const bool ea = pa + k >= n, eb = pb + k >= n;
                if (ea || eb) return ea ? (eb ? (a > b ? -1 : 1) : -1) : 1;   // the shorter suffix first
                const int ca = text[pa + k], cb = text[pb + k];
                if (ca != cb) return ca < cb ? -1 : 1;
            }
            return 0;
        };
        // bubble passes with static indices keep s[] in registers.  Nearly every group on random-like
        // data is a pair: it gets its own single comparison instead of walking the 28-site network.
        bool decided = true;
        if (g == 2) {
            const int c = cmp(s[0], s[1]);
            if (c == 0) decided = false;
            if (c > 0) { const int64_t t = s[0]; s[0] = s[1]; s[1] = t; }
        } else
#pragma unroll
        for (int pass = 0; pass < kMaxG - 1; ++pass) {
#pragma unroll
            for (int i = 0; i < kMaxG - 1; ++i) {
                if (i + 1 < g && i < g - 1 - pass) {
                    const int c = cmp(s[i], s[i + 1]);
                    if (c == 0) decided = false;
                    if (c > 0) { const int64_t t = s[i]; s[i] = s[i + 1]; s[i + 1] = t; }
                }
            }
        }
        if (decided) {
#pragma unroll
            for (int i = 0; i < kMaxG; ++i)
                if (i < g) SA[(int64_t)r + i] = (IdxT)s[i];
        } else {
            emit = (uint32_t)g;
        }
    }
    const uint32_t incl = wave_incl_sum(emit);
    if (lane == kWave - 1) wave_tot[w] = incl;
    __syncthreads();
    uint32_t off = incl - emit, tot = 0;
#pragma unroll
    for (int k = 0; k < kFinishThreads / kWave; ++k) {
        const uint32_t c = wave_tot[k];
        if (k < w) off += c;
        tot += c;
    }
    if (threadIdx.x == 0) s_base = tot ? atomicAdd(out_count, (unsigned long long)tot) : 0ull;
    __syncthreads();
    if (emit != 0) {
        const int64_t o = (int64_t)s_base + off;
        if (head) {
            // undecided small group: its members as they stand in the list
            for (int i = 0; i < g; ++i) { out_rank[o + i] = r; out_suf[o + i] = suf[j + i]; }
        } else {
            out_rank[o] = r;
            out_suf[o] = suf[j];
        }
    }
    __syncthreads();                   // wave_tot / s_base are reused by the next tile
    }
}

// Estimate of the tie fraction after round 0: kSamples evenly spaced adjacent pairs of the
// sorted key list; *count = pairs with equal keys.  Decides whether the rebucket pass should
// write the full inverse suffix array right away (dense doubling expected).
__global__ __launch_bounds__(kBlock) void sample_ties_kernel(const uint64_t *__restrict__ keys, int64_t m,
                                                             int kshift, int samples, int64_t *__restrict__ count)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    bool tie = false;
    if (i < samples && m > 1) {
        const int64_t p = (int64_t)((__int128)i * (m - 1) / samples);
        tie = (keys[p] >> kshift) == (keys[p + 1] >> kshift);
    }
    const uint64_t b = __ballot(tie);
    if (lane_id() == 0 && b) atomicAdd((unsigned long long *)count, (unsigned long long)__popcll(b));
}

// ISA[SA[p]] = p : ranks of all singleton groups (switching from sparse to dense doubling)
template <typename IdxT>
__global__ __launch_bounds__(kBlock) void isa_from_sa_kernel(const IdxT *__restrict__ SA,
                                                             IdxT *__restrict__ ISA, int64_t n)
{
    for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < n; p += (int64_t)gridDim.x * kBlock)
        ISA[SA[p]] = (IdxT)p;
}

// ISA[suf[j]] = rank[j] for the still-tied suffixes
template <typename IdxT>
__global__ __launch_bounds__(kBlock) void isa_scatter_kernel(const uint64_t *__restrict__ rank,
                                                             const IdxT *__restrict__ suf,
                                                             IdxT *__restrict__ ISA, int64_t m)
{
    for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < m; j += (int64_t)gridDim.x * kBlock)
        ISA[suf[j]] = (IdxT)rank[j];
}

}  // namespace dq
