/*
 * dq_sufsort.h -- C ABI of libdq_sufsort_hip.so, the MI355X (gfx950) suffix-sorting
 * backend that drops in behind DeltaQ's ISuffixSort plugin interface.
 *
 * Reference interface replaced (paths relative to the jzebedee/deltaq tree):
 *   src/DeltaQ.SuffixSorting.Abstractions/ISuffixSort.cs:18   IMemoryOwner<int> Sort(ReadOnlySpan<byte> text)
 *   src/DeltaQ.SuffixSorting.Abstractions/ISuffixSort.cs:27   void Sort(ReadOnlySpan<byte> text, Span<int> suffixes)
 *   src/DeltaQ.SuffixSorting.LibDivSufSort/LibDivSufSort.cs:12-29  (the default provider both overloads end in
 *   DivSufSort.divsufsort(T, SA), DivSufSort.cs:18-42)
 * The only production caller is Diff.Create (src/DeltaQ.BsDiff/Diff.cs:89-90):
 *   suffixSort.Sort(oldData, I[..^1]).
 *
 * Contract (identical to the reference's): sa receives exactly n entries, a permutation
 * of 0..n-1 in strict lexicographic suffix order over UNSIGNED bytes where a proper
 * prefix sorts first (ReadOnlySpan<byte>.SequenceCompareTo, LibDivSufSortTests.cs:43-59).
 * That array is unique, so the output is bit-identical to LibDivSufSort.Sort().
 * No sentinel slot is written: sa[n] (Diff.cs:78 allocates n+1) is never touched.
 *
 * All entry points are blocking, re-entrant and thread-safe; none retains a caller
 * pointer after returning; none throws or aborts.  There is NO CPU fallback in this
 * library: without a usable HIP device every sort entry point fails with DQ_ERR_NO_DEVICE.
 */
#ifndef DQ_SUFSORT_H
#define DQ_SUFSORT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DQ_ABI_VERSION 1

/* return codes */
#define DQ_OK              0
#define DQ_ERR_BAD_ARGS   (-1)   /* null pointer with n > 0, negative n, bad device / count        */
#define DQ_ERR_OOM        (-2)   /* device (or pinned host) allocation failed                       */
#define DQ_ERR_HIP        (-3)   /* any other HIP runtime error; see dq_last_error()                */
#define DQ_ERR_TOO_LARGE  (-4)   /* n exceeds the index width (i32: n > 2^31-1, ISuffixSort's limit;  */
                                 /* i64: n > 2^32)                                                  */
#define DQ_ERR_NO_DEVICE  (-5)   /* no HIP device visible                                           */

int32_t dq_abi_version(void);
int32_t dq_device_count(void);                 /* 0 when no device / no driver                      */
const char *dq_last_error(void);               /* thread-local, never NULL                          */

/* ---- ISuffixSort.Sort(text, suffixes): host buffers in, host buffers out -----------------------
 * text: n bytes; sa: n entries, written only (may hold garbage, LibDivSufSort.cs:14).
 * n == 0 is a no-op; n == 1 -> {0}; n == 2 -> {0,1} iff text[0] < text[1] else {1,0}
 * (DivSufSort.cs:22-38).  device: HIP device ordinal, or -1 for DQ_HIP_DEVICE / device 0. */
int32_t dq_sufsort_hip_i32(const uint8_t *text, int64_t n, int32_t *sa, int32_t device);
/* Same contract with 64-bit indices, for inputs beyond ISuffixSort's int limit: 2^31 <= n <= 2^32
 * bytes (any n is accepted up to that).  n > 2^32 returns DQ_ERR_TOO_LARGE before any device work:
 * a doubling round sorts (rank, rank of the suffix h bytes on) as one 64-bit word, 32 + 32 bits at
 * most, and the workspace of ~42 n bytes would not fit 288 GB much beyond that anyway. */
int32_t dq_sufsort_hip_i64(const uint8_t *text, int64_t n, int64_t *sa, int32_t device);

/* ---- device-resident variant: text and sa are device pointers on `device` ----------------------
 * d_text: n bytes, any alignment; d_sa: n entries.  Work is enqueued on `stream`
 * (a hipStream_t, NULL = the library's own stream for that device) and the call returns
 * after the stream has drained.  The next consumer (Diff.Create's match search,
 * Diff.cs:100-125) can read d_sa without a D2H copy. */
int32_t dq_sufsort_hip_dev_i32(const void *d_text, int64_t n, void *d_sa, int32_t device, void *stream);
int32_t dq_sufsort_hip_dev_i64(const void *d_text, int64_t n, void *d_sa, int32_t device, void *stream);

/* ---- batch of independent inputs (the many-files bsdiff path) -----------------------------------
 * count inputs, texts[j] of lens[j] bytes -> sas[j] (lens[j] entries).  Inputs are assigned to
 * the ndev devices in devs[] longest-first (LPT); each device runs its share on its own host
 * thread.  devs == NULL means devices 0..ndev-1.  Returns the first failing input's code. */
int32_t dq_sufsort_hip_batch_i32(int32_t count, const uint8_t *const *texts, const int64_t *lens,
                                 int32_t *const *sas, int32_t ndev, const int32_t *devs);

/* ---- Diff.Create's match search on the device-resident suffix array ----------
 * Replaces, for a batch of scan positions, the reference's
 *   Search(I, oldData, newData[scan..], 0, oldData.Length, out pos)          src/DeltaQ.BsDiff/Diff.cs:267-298
 * as called by the scan loop (Diff.cs:106): for every query q the pair (pos[q], len[q]) is exactly what Search
 * returns -- the suffix-array neighbour of the query with the longer common prefix (MatchLength, Diff.cs:248-265),
 * ties to the upper neighbour, the zeroed sentinel slot I[n] = 0 of Diff.cs:78 included.  sa is the n-entry
 * suffix array of old_data (what dq_sufsort_hip_dev_* leaves on the device); the sentinel is implied.
 * Queries: scan = scans[q] when scans != NULL, else scan0 + q; 0 <= scan <= m.
 * cap: 0 = exact for every query.  cap > 0 = a query whose comparison would run more than `cap` bytes past what
 * is already known to match is given up and returns len = -1, pos = 0 (speculative batches inside a long
 * match must not cost O(match length) each: the caller repeats that one position with cap = 0).
 * The _dev_ forms take device pointers on `device` (d_scans too) and enqueue on `stream` (NULL = the library's
 * stream), returning after the stream has drained; the plain forms take host pointers and copy. */
int32_t dq_bsdiff_search_dev_i32(const void *d_old, int64_t n, const void *d_sa, const void *d_new, int64_t m,
                                 const int64_t *d_scans, int64_t scan0, int64_t count, int64_t cap, void *d_pos,
                                 void *d_len, int32_t device, void *stream);
int32_t dq_bsdiff_search_dev_i64(const void *d_old, int64_t n, const void *d_sa, const void *d_new, int64_t m,
                                 const int64_t *d_scans, int64_t scan0, int64_t count, int64_t cap, void *d_pos,
                                 void *d_len, int32_t device, void *stream);
int32_t dq_bsdiff_search_i32(const uint8_t *old_data, int64_t n, const int32_t *sa, const uint8_t *new_data, int64_t m,
                             const int64_t *scans, int64_t scan0, int64_t count, int64_t cap, int32_t *pos, int32_t *len,
                             int32_t device);
int32_t dq_bsdiff_search_i64(const uint8_t *old_data, int64_t n, const int64_t *sa, const uint8_t *new_data, int64_t m,
                             const int64_t *scans, int64_t scan0, int64_t count, int64_t cap, int64_t *pos, int64_t *len,
                             int32_t device);

/* ---- Diff.Create / Patch.Apply natively: the BSDIFF40 container ----------------
 * dq_bsdiff_create   = Diff.Create(oldData, newData, output, suffixSort)         src/DeltaQ.BsDiff/Diff.cs:27-253
 *   old file -> suffix array on the device (stays there) -> match search kernel, asked for windows of scan
 *   positions by the reference's own scan loop (Diff.cs:100-232, kept statement for statement on the host) ->
 *   control triples / diff / extra -> three bzip2 streams (each block's Burrows-Wheeler transform is one more run
 *   of the device sorter) -> "BSDIFF40" header (Constants.cs, SpanExtensions.cs packed longs) + streams.
 *   The raw streams equal the reference loop's byte for byte; the bzip2 framing is a valid encoding of them (any
 *   bzip2 decoder reads it; the reference does not pin SharpZipLib's bytes either).  patch: cap bytes
 *   (dq_bsdiff_patch_bound(n, m) always suffices); *patch_len receives the length.  Files below 2 GiB (int).
 * dq_bsdiff_scan_i32 = the same up to the raw streams: ctrl receives *nctrl (add, copy, seek) triples (capacity
 *   ctrl_cap triples; m + 1 always suffices), diff / extra the raw bytes (capacity m each); stats (optional,
 *   3 entries): Search calls of the loop, windows requested from the device, positions asked again exactly.
 *   (diff and extra are written without a capacity argument: together they never exceed m bytes.)
 * dq_bspatch_apply   = Patch.Apply(input, openPatchStream, output)                src/DeltaQ.BsDiff/Patch.cs:52-168
 *   host code only (no device needed).  out == NULL: only *out_len = size of the new file.  A patch the
 *   reference would reject with "Corrupt patch" returns DQ_ERR_BAD_ARGS with that message in dq_last_error(). */
int32_t dq_bsdiff_create(const uint8_t *old_data, int64_t n, const uint8_t *new_data, int64_t m, uint8_t *patch,
                         int64_t cap, int64_t *patch_len, int32_t device);
int64_t dq_bsdiff_patch_bound(int64_t n, int64_t m);

/* ---- one old file, many new files (the many-files bsdiff path of the batch mode) --------------------------------
 * Diff.Create sorts oldData on every call (Diff.cs:89-90); the suffix array depends on the old file alone.  An index
 * holds (old, suffix array, the match search's prefix table) on one device; any number of new files are diffed
 * against it, each call returning the patch dq_bsdiff_create(old, new) returns.
 *   dq_bsdiff_index_create   d_old == d_sa == NULL: uploads old_data and sorts it (the index owns the buffers).
 *                            Otherwise d_old / d_sa are the caller's device-resident text (n bytes) and suffix array
 *                            (n int32) -- e.g. received by RCCL broadcast from the rank that sorted -- and must stay
 *                            valid until dq_bsdiff_index_free.  old_data (host) is read by every diff (the scan loop
 *                            walks it, Diff.cs:129-191) and must stay valid as long as the index.
 *   dq_bsdiff_index_clone    one more copy of an index on `device` (the same device or another one of the node): text,
 *                            suffix array and prefix table travel device to device -- over xGMI between devices --
 *                            instead of being computed again.  For a host without a collective library in its process
 *                            (the C# shim): clones to the other devices of a node, made from one thread each, use one
 *                            point-to-point link each.  Shares the source's host copy of old_data; freed on its own.
 *   dq_bsdiff_index_buffers  the device pointers (for a broadcast / gather by the caller) and n.
 *   dq_bsdiff_index_diff     = Diff.Create(oldData, newData, ...) without its suffix sort.  Thread-safe: scan loops of
 *                            concurrent callers take turns on the device, their bzip2 framing overlaps.
 *   dq_bsdiff_index_free     releases the index (not the caller's buffers). */
int32_t dq_bsdiff_index_create(const uint8_t *old_data, int64_t n, const void *d_old, const void *d_sa, int32_t device,
                               void **index_out);
int32_t dq_bsdiff_index_clone(const void *index, int32_t device, void **index_out);
int32_t dq_bsdiff_index_buffers(const void *index, const void **d_old, const void **d_sa, int64_t *n);
int32_t dq_bsdiff_index_diff(const void *index, const uint8_t *new_data, int64_t m, uint8_t *patch, int64_t cap,
                             int64_t *patch_len);
void dq_bsdiff_index_free(void *index);
int32_t dq_bsdiff_scan_i32(const uint8_t *old_data, int64_t n, const uint8_t *new_data, int64_t m, int64_t *ctrl,
                           int64_t ctrl_cap, int64_t *nctrl, uint8_t *diff, int64_t *ndiff, uint8_t *extra, int64_t *nextra,
                           int64_t *stats, int32_t device);
int32_t dq_bspatch_apply(const uint8_t *old_data, int64_t n, const uint8_t *patch, int64_t patch_len, uint8_t *out,
                         int64_t cap, int64_t *out_len);

/* Device workspace (bytes) a sort of n bytes with index_bytes (4 or 8) wide indices needs,
 * excluding the caller's text and sa buffers. */
int64_t dq_sufsort_hip_workspace_bytes(int64_t n, int32_t index_bytes);
/* Free every cached device workspace / pinned staging buffer / stream. */
void dq_sufsort_hip_release(void);

/* ---- measurement hooks (bench.py) -----------------------------------------------------------------
 * With profiling on, every kernel launch is bracketed by hipEvents on the launch stream and the
 * elapsed time is accumulated per kernel category when the sort finishes. */
#define DQ_K_TEXT_HIST            0   /* text_hist_kernel (+ text_digit_offsets_kernel): 1 B/text byte            */
#define DQ_K_RADIX_HIST           1   /* radix_hist_kernel + radix_hist_scan_kernel: 8 B/key                        */
#define DQ_K_RADIX_RANK           2   /* radix_rank_kernel, one digit pass; B/element by pass kind (w = index      */
                                      /* bytes): text->words 9, words 16, last words pass 16+w, tie-recording      */
                                      /* last pass 8+w+1/8 (+16 B per tile and digit), pairs 2*(8+w), text->pairs  */
                                      /* 9+w                                                                       */
#define DQ_K_SEG_FUSED            3   /* seg_fused_kernel: rebucket of a sorted list, 8 (+w..3w when writing)       */
#define DQ_K_TIE_SEAM             4   /* tie_seam_kernel: ties across tile seams, 24 B per (tile, digit)            */
#define DQ_K_TIE_COLLECT          5   /* tie_collect_kernel: tie bits -> list of tied suffixes, 1/8 B/suffix        */
#define DQ_K_SMALL_FINISH         6   /* small_group_finish_kernel: groups <= 8 by direct text comparison           */
#define DQ_K_SMALL_ROUND          7   /* small_group_round_kernel: one doubling round for groups <= 8 (32)          */
#define DQ_K_ISA_UPDATE           8   /* isa_update_kernel: deferred rank updates of a small-group round            */
#define DQ_K_ISA_FROM_PAIRS       9   /* isa_from_pairs_kernel: first ISA from suffix-binned words, 8+w             */
#define DQ_K_KEY2_FROM_PAIRS     10   /* key2_from_pairs_kernel: first key2 gather inside the suffix window         */
#define DQ_K_GATHER_KEY2         11   /* gather_key2_kernel: (rank, ISA[s+h]) composite keys, 16+2w                 */
#define DQ_K_GATHER_TEXT_KEY     12   /* gather_text_key_kernel: (rank, next bytes of text) keys                    */
#define DQ_K_ISA_FROM_SA         13   /* isa_from_sa_kernel + isa_scatter_kernel: ISA for the switch to doubling    */
#define DQ_K_SMALL_SORT          14   /* small_sufsort_kernel: a whole short text (n <= 8192) in one workgroup      */
#define DQ_K_BUCKET_SORT         15   /* bucket_sort_kernel (+ bucket_bounds_kernel): buckets finished in LDS, 8+w+1/8  */
#define DQ_K_MATCH_SEARCH        16   /* match_search_kernel: Diff.cs Search for a batch of scan positions              */
#define DQ_K_PAIR_CHAINS         17   /* dq_pair_chains.h: tied pairs inside long repeats decided chain by chain    */
#define DQ_K_MID_ROUND           18   /* dq_mid_groups.h: a doubling round for tie groups of up to 1024 members, inside LDS */
#define DQ_K_RUNS                19   /* dq_runs.h: run lengths of the text (three small kernels)                          */
#define DQ_K_SPLIT_PASS          20   /* dq_split_round0.h: split_pass_kernel, round 0 as a sample sort: text -> pairs by top bucket (1+12), pairs -> bucket slots (12+12) */
#define DQ_K_SPLIT_FINISH        21   /* bucket_finish_kernel: every bucket sorted by its 64-bit keys inside LDS, 12+12           */
#define DQ_K_SPLIT_AUX           22   /* sample, splitter tables, top-bucket histogram (1 B/text byte), plans, scans, overflow placement */
#define DQ_K_COUNT               23

/* 0 off, 1 every kernel, 2 only radix_rank_kernel, 100 + c only category c (cheapest: the timed region) */
int32_t dq_profile_enable(int32_t on);
void    dq_profile_reset(void);
/* launches, summed milliseconds, summed elements processed, summed algorithmic bytes */
int32_t dq_profile_get(int32_t category, int64_t *launches, double *total_ms, int64_t *elements,
                       int64_t *alg_bytes);
const char *dq_profile_kernel_name(int32_t category);
int32_t dq_profile_category_count(void);   /* DQ_K_COUNT of the loaded library */

/* Shape of the last sort on this thread: doubling rounds after the initial 8-byte sort, number
 * of suffixes still in non-singleton groups after the initial sort, and the sum of that count
 * over all rounds. */
int32_t dq_last_sort_info(int64_t *rounds, int64_t *initial_active, int64_t *sum_active);

/* Shape of the last dq_bsdiff_create / dq_bsdiff_scan_i32 / dq_bsdiff_index_diff on this thread, `count` entries (9 are
 * defined, further ones read 0): Search calls the reference's loop makes (Diff.cs:106), windows of scan positions,
 * positions asked again exactly, files the host loop took instead of the device's anchor scan (host_loop_fallbacks: a
 * launch whose persistent grid waited in vain -- a device kept full by other work --, AND each of the 16 diffs after it
 * that skip the device scan on that device, and devices that hold fewer than 8 workgroups of the grid; the patch is
 * the same, the call slower, dq_last_error() is left untouched by it), workgroups of that grid; [5..8] where several
 * grids walked the new file at once: grids launched, grids the followed one was joined to (same place, same shift:
 * their entries are the loop's from there on), grids dropped unjoined, control triples taken over from the grids' own
 * emitter threads. */
int32_t dq_last_diff_info(int64_t *info, int32_t count);

/* Shape of the last dq_sufsort_hip_batch_i32 on this thread, `count` entries (6 are defined, further ones read 0):
 * inputs that went through the three-stage pipelines; microseconds the copy-in, the sort and the copy-out stages were
 * busy, each summed over the device shares (a share whose sort stage is busy all the time waits for the GPU, one whose
 * copy stages are waits for host memory / PCIe: what an 8-GPU run needs to tell the two apart); wall microseconds of
 * the slowest share; device shares whose host threads were bound to their device's NUMA node.  New API like the batch
 * entry itself: the reference has no multi-file call ("Who calls it"). */
int32_t dq_last_batch_info(int64_t *info, int32_t count);

/* NUMA node the device's PCIe function hangs off (/sys/bus/pci/devices/<bdf>/numa_node), -1 where the platform does not
 * say (single-socket hosts, containers without sysfs) or the ordinal is out of range.  The batch pipeline binds the host
 * threads it starts for a device to that node's CPUs (never the caller's thread; DQ_NUMA_BIND=0 turns it off); a host
 * that runs one process per GPU -- bench.py's ranks, deltaq_amd/batch.py -- binds itself with this. */
int32_t dq_device_numa_node(int32_t device);

#ifdef __cplusplus
}
#endif
#endif /* DQ_SUFSORT_H */
