// dq_bspatch.h -- Patch.Apply (src/DeltaQ.BsDiff/Patch.cs:52-168) on a whole patch in memory: header, three bzip2
// streams, add / copy / seek.  Host code only (no device work: the reference's reader is a byte loop over streams).
//
// A patch is untrusted input.  Every length in it is a 63-bit number taken from the file, so
//   * all bounds are compared in a form that cannot overflow (dq_bsdiff.h: parse_header, apply_streams);
//   * the streams are decoded only as far as Patch.Apply could read them: the reference stops at newSize
//     (Patch.cs:115), so at most newSize diff bytes, newSize extra bytes and the control triples that produce them
//     are ever looked at -- a stream that decodes to more is cut there instead of being expanded in full (a bzip2
//     block expands up to ~50x per level and streams concatenate: a few KB could otherwise ask for gigabytes).
// Negative add / copy sizes are rejected here (as bspatch 4.3 does); the reference's loop would step over them.
#pragma once
#include "dq_bsdiff.h"
#include "dq_bz2.h"

namespace dq {
namespace bsdiff {

enum { kPatchOk = 0, kPatchCorrupt = -1, kPatchSmallBuffer = -2 };

// out == nullptr: size query (*out_len = the new file's size).
inline int apply_patch(const uint8_t *old, int64_t n, const uint8_t *patch, int64_t plen, uint8_t *out, int64_t cap,
                       int64_t *out_len)
{
    Header h;
    if (parse_header(patch, plen, &h) != 0) return kPatchCorrupt;
    // The header's newSize is what a caller allocates BEFORE any stream has been looked at (the size query below), so a
    // claim no patch of this length could honour is rejected here instead of becoming an out-of-memory error in the
    // wrapper: every byte of the new file is decoded from the diff or the extra stream, and bzip2 cannot expand by
    // more than ~10^6 (a 900 kB block of run-length pairs = 46 MB of one byte, in ~45 bytes); 2^21 leaves a margin.
    if (h.new_size > ((int64_t)1 << 21) * (plen - kHeaderSize - h.ctrl_len + 64)) return kPatchCorrupt;
    if (out_len) *out_len = h.new_size;
    if (!out) return kPatchOk;
    if (cap < h.new_size) return kPatchSmallBuffer;
    std::vector<uint8_t> ctrl, diff, extra;
    const uint8_t *pc = patch + kHeaderSize, *pd = pc + h.ctrl_len, *pe = pd + h.diff_len;
    const size_t elen = (size_t)(plen - kHeaderSize - h.ctrl_len - h.diff_len);
    // Triples the reader may look at.  Diff.Create's scan position advances with every triple, so its patches hold at
    // most newSize + 1 of them -- but a triple may produce no byte at all (the first one routinely is (0, 0, seek)),
    // and Patch.Apply accepts a third-party patch with any number of those.  The cut-off is therefore generous rather
    // than exact: two triples per output byte + 4096; a control stream that decodes to more than that before the
    // new file is complete is treated as corrupt (the one documented divergence from Patch.cs:115-160, which would
    // keep reading: it is what keeps a few KB of bzip2 from asking for gigabytes of control data).
    const size_t ctrl_max = h.new_size < (int64_t)1 << 55 ? 24 * (2 * (size_t)h.new_size + 4096) : (size_t)-1;
    int rc = bz2::bz2_decompress(pc, (size_t)h.ctrl_len, ctrl, ctrl_max);
    if (rc != bz2::kOk && rc != bz2::kTooLong) return kPatchCorrupt;
    rc = bz2::bz2_decompress(pd, (size_t)h.diff_len, diff, (size_t)h.new_size);
    if (rc != bz2::kOk && rc != bz2::kTooLong) return kPatchCorrupt;
    rc = bz2::bz2_decompress(pe, elen, extra, (size_t)h.new_size);
    if (rc != bz2::kOk && rc != bz2::kTooLong) return kPatchCorrupt;
    return apply_streams(old, n, ctrl, diff, extra, h.new_size, out) == 0 ? kPatchOk : kPatchCorrupt;
}

}  // namespace bsdiff
}  // namespace dq
