// Test harness: the product's scan loop (deltaq_amd/csrc/dq_bsdiff.h) with Search answers read from a table the
// test filled, so that the loop's decisions can be compared with the oracle's restatement without a GPU.
#include "../../deltaq_amd/csrc/dq_bsdiff.h"

extern "C" int64_t t_scan_loop(const uint8_t *old, int64_t n, const uint8_t *nw, int64_t m, const int64_t *pos,
                               const int64_t *len, uint8_t *ctrl, int64_t *ctrl_len, uint8_t *diff, int64_t *diff_len,
                               uint8_t *extra, int64_t *extra_len)
{
    dq::bsdiff::RawStreams rs;
    const int rc = dq::bsdiff::scan_loop(old, n, nw, m, [&](int64_t scan, int64_t *p, int64_t *l) {
        *p = pos[scan];
        *l = len[scan];
        return 0;
    }, rs);
    if (rc != 0) return -1;
    memcpy(ctrl, rs.ctrl.data(), rs.ctrl.size());
    memcpy(diff, rs.diff.data(), rs.diff.size());
    memcpy(extra, rs.extra.data(), rs.extra.size());
    *ctrl_len = (int64_t)rs.ctrl.size();
    *diff_len = (int64_t)rs.diff.size();
    *extra_len = (int64_t)rs.extra.size();
    return rs.searches;
}

extern "C" int64_t t_packed_roundtrip(int64_t v)
{
    uint8_t b[8];
    dq::bsdiff::write_packed_long(b, v);
    return dq::bsdiff::read_packed_long(b);
}

// the anchors of the reference's loop (the positions it emits a triple at, with the match position in old) and the streams
// the product's emitter makes of them: what the device's anchor search hands over (dq_anchor_scan.h)
extern "C" int64_t t_scan_from_anchors(const uint8_t *old, int64_t n, const uint8_t *nw, int64_t m, const int64_t *pairs,
                                       int64_t count, uint8_t *ctrl, int64_t *ctrl_len, uint8_t *diff, int64_t *diff_len,
                                       uint8_t *extra, int64_t *extra_len)
{
    dq::bsdiff::RawStreams rs;
    dq::bsdiff::TripleEmitter em(old, n, nw, m, rs);
    dq::bsdiff::scan_from_anchors(em, pairs, count / 2);
    dq::bsdiff::scan_from_anchors(em, pairs + 2 * (count / 2), count - count / 2);      // (two batches)
    memcpy(ctrl, rs.ctrl.data(), rs.ctrl.size());
    memcpy(diff, rs.diff.data(), rs.diff.size());
    memcpy(extra, rs.extra.data(), rs.extra.size());
    *ctrl_len = (int64_t)rs.ctrl.size();
    *diff_len = (int64_t)rs.diff.size();
    *extra_len = (int64_t)rs.extra.size();
    return 0;
}
