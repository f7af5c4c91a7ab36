// dq_bsdiff.h -- the BSDIFF40 container around the hot path: Diff.Create and Patch.Apply, natively (host code
// driving the device kernels).  SURVEY.md section 8(f) row 3.
//
// Reference: src/DeltaQ.BsDiff/Diff.cs:27-241 (Create: header :54-70, suffix sort :89-90, scan loop :100-232, the
//            three bzip2 streams :85-87 / :235-245, header rewrite :247-252), Patch.cs:52-168 (Apply),
//            SpanExtensions.cs:7-44 (packed longs), Constants.cs:5-12 (layout, "BSDIFF40").
//
//   old file --H2D--> suffix array (the HIP sorter, stays on the device) --> match search kernel, asked for
//   WINDOWS of scan positions by the sequential scan loop on the host --> control triples, diff and extra bytes
//   --> three bzip2 streams (dq_bz2.h; the Burrows-Wheeler transform of every block is one more run of the HIP
//   sorter) --> header + streams.
//
// The scan loop consumes Search only through (pos, len) (Diff.cs:106), so it is kept exactly as the reference
// has it and reads from a window of answers the device filled speculatively for the positions ahead of `scan`:
// in a region where old and new differ the loop advances byte by byte and uses every answer; after a match it
// jumps by `len`, and a jump out of the window simply starts the next window at the new position.  A position
// whose comparison ran into the cap (inside a long match) comes back undecided and is asked again alone, exactly.
// The patch is therefore the one the reference's loop produces, byte for byte in its raw streams
// (tests/test_gpu_bsdiff.py compares them with the oracle's restatement); the bzip2 framing is a valid encoding of
// them, not necessarily SharpZipLib's bytes (which the reference does not pin either).
#pragma once
#include <stdint.h>

#include <cstring>
#include <vector>

namespace dq {
namespace bsdiff {

constexpr int kHeaderSize = 32;                                   // Constants.cs:7
constexpr int64_t kSignature = 0x3034464649445342ll;              // "BSDIFF40", Constants.cs:14

// SpanExtensions.cs:7-29: sign-magnitude, little endian, bit 7 of byte 7 = sign
inline void write_packed_long(uint8_t *p, int64_t y)
{
    uint64_t u;
    uint8_t sign = 0;
    if (y < 0) { u = (uint64_t)(-(y + 1)) + 1; sign = 0x80; } else u = (uint64_t)y;
    for (int i = 0; i < 8; ++i) p[i] = (uint8_t)(u >> (8 * i));
    p[7] = (uint8_t)((p[7] & 0x7f) | sign);
}

// SpanExtensions.cs:31-44
inline int64_t read_packed_long(const uint8_t *p)
{
    int64_t y = p[7] & 0x7f;
    for (int i = 6; i >= 0; --i) y = (y << 8) + p[i];
    return (p[7] & 0x80) ? -y : y;
}

struct RawStreams {
    std::vector<uint8_t> ctrl, diff, extra;                       // ctrl: 24 bytes (three packed longs) per triple
    int64_t searches = 0, windows = 0, exact = 0;
};

// Search provider: int operator()(int64_t scan, int64_t *pos, int64_t *len)  -> 0 or an error code.
// The loop below is Diff.cs:91-232 statement for statement; `I` is never touched by it except through Search.
template <typename SearchFn>
int scan_loop(const uint8_t *old, int64_t n, const uint8_t *nw, int64_t m, SearchFn &&search, RawStreams &out)
{
    int64_t scan = 0, pos = 0, len = 0, lastscan = 0, lastpos = 0, lastoffset = 0;
    uint8_t buf[8];
    while (scan < m) {                                                              // :100
        int64_t oldscore = 0;
        int64_t scsc;
        for (scsc = scan += len; scan < m; scan++) {                                // :104
            const int rc = search(scan, &pos, &len);                                // :106
            if (rc != 0) return rc;
            ++out.searches;
            for (; scsc < scan + len; scsc++)
                if ((scsc + lastoffset < n) && (old[scsc + lastoffset] == nw[scsc])) oldscore++;
            if ((len == oldscore && len != 0) || (len > oldscore + 8)) break;
            if ((scan + lastoffset < n) && (old[scan + lastoffset] == nw[scan])) oldscore--;
        }
        if (len != oldscore || scan == m) {                                         // :127
            int64_t s = 0, sf = 0, lenf = 0;
            for (int64_t i = 0; (lastscan + i < scan) && (lastpos + i < n);) {
                if (old[lastpos + i] == nw[lastscan + i]) s++;
                i++;
                if (s * 2 - i > sf * 2 - lenf) { sf = s; lenf = i; }
            }
            int64_t lenb = 0;
            if (scan < m) {                                                         // :147
                s = 0;
                int64_t sb = 0;
                for (int64_t i = 1; (scan >= lastscan + i) && (pos >= i); i++) {
                    if (old[pos - i] == nw[scan - i]) s++;
                    if (s * 2 - i > sb * 2 - lenb) { sb = s; lenb = i; }
                }
            }
            if (lastscan + lenf > scan - lenb) {                                    // :167
                const int64_t overlap = (lastscan + lenf) - (scan - lenb);
                s = 0;
                int64_t ss = 0, lens = 0;
                for (int64_t i = 0; i < overlap; i++) {
                    if (nw[lastscan + lenf - overlap + i] == old[lastpos + lenf - overlap + i]) s++;
                    if (nw[scan - lenb + i] == old[pos - lenb + i]) s--;
                    if (s > ss) { ss = s; lens = i + 1; }
                }
                lenf += lens - overlap;
                lenb -= lens;
            }
            const size_t d0 = out.diff.size();                                      // :196 diff string
            out.diff.resize(d0 + (size_t)lenf);
            for (int64_t i = 0; i < lenf; i++) out.diff[d0 + (size_t)i] = (uint8_t)(nw[lastscan + i] - old[lastpos + i]);
            const int64_t extra_len = (scan - lenb) - (lastscan + lenf);            // :203 extra string
            if (extra_len > 0) out.extra.insert(out.extra.end(), nw + lastscan + lenf, nw + lastscan + lenf + extra_len);
            const int64_t triple[3] = {lenf, extra_len, (pos - lenb) - (lastpos + lenf)};       // :210-217 ctrl block
            for (int k = 0; k < 3; ++k) {
                write_packed_long(buf, triple[k]);
                out.ctrl.insert(out.ctrl.end(), buf, buf + 8);
            }
            lastscan = scan - lenb;
            lastpos = pos - lenb;
            lastoffset = pos - scan;
        }
    }
    return 0;
}

// Patch.cs:95-168 on the three decoded streams.  Returns 0, or -1 for what the reference reports as "Corrupt patch".
inline int apply_streams(const uint8_t *old, int64_t n, const std::vector<uint8_t> &ctrl, const std::vector<uint8_t> &diff,
                         const std::vector<uint8_t> &extra, int64_t newsize, uint8_t *out)
{
    int64_t outpos = 0, oldpos = 0;
    size_t cpos = 0, dpos = 0, epos = 0;
    while (outpos < newsize) {
        if (cpos + 24 > ctrl.size()) return -1;
        const int64_t add = read_packed_long(&ctrl[cpos]), copy = read_packed_long(&ctrl[cpos + 8]),
                      seek = read_packed_long(&ctrl[cpos + 16]);
        cpos += 24;
        if (add < 0 || copy < 0 || outpos + add > newsize) return -1;                // :131 sanity-check
        if (dpos + (size_t)add > diff.size() || oldpos < 0 || oldpos + add > n) return -1;       // short reads
        for (int64_t i = 0; i < add; i++) out[outpos + i] = (uint8_t)(diff[dpos + (size_t)i] + old[oldpos + i]);
        outpos += add; dpos += (size_t)add; oldpos += add;
        if (outpos + copy > newsize || epos + (size_t)copy > extra.size()) return -1;           // :153
        if (copy > 0) memcpy(out + outpos, &extra[epos], (size_t)copy);
        outpos += copy; epos += (size_t)copy;
        oldpos += seek;                                                              // :165
    }
    return 0;
}

struct Header {
    int64_t ctrl_len = 0, diff_len = 0, new_size = 0;
};

// Patch.cs:52-93.  0, or -1 ("Corrupt patch").
inline int parse_header(const uint8_t *patch, int64_t plen, Header *h)
{
    if (plen < kHeaderSize) return -1;
    if (read_packed_long(patch) != kSignature) return -1;
    h->ctrl_len = read_packed_long(patch + 8);
    h->diff_len = read_packed_long(patch + 16);
    h->new_size = read_packed_long(patch + 24);
    if (h->ctrl_len < 0 || h->diff_len < 0 || h->new_size < 0) return -1;
    if (kHeaderSize + h->ctrl_len + h->diff_len > plen) return -1;
    return 0;
}

}  // namespace bsdiff
}  // namespace dq
