#!/usr/bin/env python3
"""CPU-only simulation for DESIGN section 2c, "several anchors per window": how many windows the device scan would
need if a window were laid out as RUNS of 64 positions at PREDICTED bases (the alignment resumes behind a small edit:
same shift after a replacement, shift - x at position x after an insertion of x bytes, shift + x at once after a
deletion), and what that is worth under the per-phase times DQ_TRACE reports (search 8 us, one prediction level 3 us,
one evaluation 2.5 us, a plain window 11 us).  Nothing here runs on a GPU: Search answers come from the oracle, the
loop is Diff.cs:100-125 transcribed (tests/anchor_model.py::literal_anchors with its per-iteration state recorded).

usage: sim_anchor_runs.py [MiB of old file = 2]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle
from tools import datagen

SEG, NSEG, WAVE_WINS, WAVE_WIN, LANE_WIN = 64, 8, 3, 512, 32768


def edited(rng, old, edits, span):
    new = bytearray(old.tobytes())
    for _ in range(edits):
        k = int(rng.integers(0, 4)); a = int(rng.integers(0, max(1, len(new)))); ln = int(rng.integers(1, span))
        if k == 0: new[a:a] = rng.integers(0, 256, ln, dtype=np.uint8).tobytes()
        elif k == 1: del new[a:a + ln]
        elif k == 2: new[a:a + ln] = rng.integers(0, 256, min(ln, len(new) - a), dtype=np.uint8).tobytes()
        else: new[a:a] = new[max(0, a - 3 * ln):max(0, a - 2 * ln)]
    return np.frombuffer(bytes(new), dtype=np.uint8)


def iterations(old, new, pos, ln):
    """The scan loop's outer iterations: (base, break position, hit_pos, hit_len, shift before, emitted)."""
    n, m = len(old), len(new)
    cursor = hit_pos = hit_len = shift = 0
    out = []
    while cursor < m:
        carried = 0
        cursor += hit_len
        base = counted = cursor
        while cursor < m:
            hit_pos, hit_len = int(pos[cursor]), int(ln[cursor])
            if counted < cursor + hit_len:
                end = cursor + hit_len
                upto = min(end, n - shift)
                if upto > counted:
                    carried += int(np.count_nonzero(old[counted + shift:upto + shift] == new[counted:upto]))
                counted = end
            if (hit_len == carried and hit_len != 0) or hit_len > carried + 8:
                break
            carried -= int(cursor + shift < n and old[cursor + shift] == new[cursor])
            cursor += 1
        emitted = not (hit_len == carried and cursor != m)
        out.append((base, cursor, hit_pos, hit_len, shift, emitted))
        if emitted:
            shift = hit_pos - cursor
    return out


def common_prefix(a, b):
    k = min(len(a), len(b))
    if k == 0:
        return 0
    ne = np.flatnonzero(a[:k] != b[:k])
    return int(ne[0]) if ne.size else k


def predict(old, new, b, sh, positions, dmax=64, min_len=16):
    """Where the loop is expected to break behind base b under alignment sh: (j, p, L) or None.  Three 8-byte tests per
    window position -- the same alignment (a replacement ends here), the old file's next bytes (an insertion ends
    here), and, at the base only, the old file d bytes further on (a deletion of d bytes)."""
    n, m = len(old), len(new)
    A = min(positions, m - b - 8)
    if A <= 0 or b + sh < 0 or b + sh + 8 > n:
        return None
    nw = np.lib.stride_tricks.sliding_window_view(new[b:b + A + 7], 8)                      # nw[a] = new[b + a .. + 8)
    cands = []
    hi = min(A, n - (b + sh) - 8)
    if hi > 0:
        ow = np.lib.stride_tricks.sliding_window_view(old[b + sh:b + sh + hi + 7], 8)
        same = np.flatnonzero((nw[:hi] == ow[:hi]).all(axis=1))
        cands += [(int(a), 0, int(a) + b + sh) for a in same[:4]]
        d = min(dmax, hi)
        dele = np.flatnonzero((ow[:d] == nw[0]).all(axis=1))
        cands += [(0, 2, b + sh + int(x)) for x in dele[:4] if x > 0]
    ins = np.flatnonzero((nw[:A] == old[b + sh:b + sh + 8]).all(axis=1))
    cands += [(int(a), 1, b + sh) for a in ins[:4] if a > 0]
    for a, _, p in sorted(cands):
        L = common_prefix(new[b + a:], old[p:])
        if L >= min_len:
            return b + a, p, L
    return None


def simulate(old, new, its):
    m = len(new)
    plain = spec = spec_plain = 0
    runs_used = levels = hits = misses = 0
    t_spec = 0.0
    k = 0

    def extra_windows(start, brk):          # normal windows from `start` until the one that holds position brk
        w, i, passed = 0, start, 0
        while True:
            size = LANE_WIN if passed >= WAVE_WINS else WAVE_WIN
            w += 1
            if brk < i + size or i + size >= m:
                return w
            i += size
            passed += 1

    for base, brk, _, _, _, _ in its:
        plain += extra_windows(base, brk)
    while k < len(its):
        base, brk, hit_pos, hit_len, shift, emitted = its[k]
        # ---- the layout of this window: runs at predicted bases, as long as segments are left ----
        layout, b, sh, left = [], base, shift, NSEG
        while left > 0:
            pr = predict(old, new, b, sh, SEG * left) if left > 1 else None
            if pr is None:
                layout.append((b, sh, left)); left = 0
                break
            j, p, L = pr
            need = (j - b) // SEG + 1
            layout.append((b, sh, need)); left -= need
            b, sh = j + L, p - j
            levels += 1
        # ---- the loop's iterations against it ----
        spec += 1
        evals = 0
        r = 0
        while True:
            rb, rsh, rseg = layout[r]
            base, brk, hit_pos, hit_len, shift, emitted = its[k]
            assert (rb, rsh) == (base, shift)
            evals += 1
            covered = rb + SEG * rseg
            if brk >= covered and covered < m:          # the run went by without a break: plain windows take over
                spec_plain += extra_windows(covered, brk)
                k += 1
                break
            k += 1
            if k >= len(its):
                break
            nb, nsh = its[k][0], its[k][4]
            if r + 1 < len(layout) and (layout[r + 1][0], layout[r + 1][1]) == (nb, nsh):
                r += 1; hits += 1
            else:
                misses += r + 1 < len(layout)
                break
        runs_used += evals
        t_spec += 8.0 + 3.0 * (len(layout) - 1) + 2.5 * evals
    return dict(iterations=len(its), plain_windows=plain, laid_out_windows=spec, plain_windows_behind_them=spec_plain,
                runs_evaluated=runs_used, prediction_levels=levels, predicted_right=hits, predicted_wrong=misses,
                plain_ms=round(plain * 11e-3, 2), laid_out_ms=round((t_spec + 11.0 * spec_plain) * 1e-3, 2))


def main():
    mib = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
    n = int(mib * (1 << 20))
    rng = np.random.default_rng(3)
    cases = [("random, edits of <= 40 bytes, one per 840 bytes", datagen.gen_uniform(n, 8), n // 840, 40),
             ("random, edits of <= 400 bytes, one per 8400 bytes", datagen.gen_uniform(n, 5), n // 8400, 400),
             ("text, edits of <= 400 bytes, one per 8400 bytes", datagen.gen_enwik_like(n, 3, 64 * 1024), n // 8400, 400)]
    for name, old, edits, span in cases:
        new = edited(rng, old, edits, span)
        t0 = time.perf_counter()
        sa = oracle.divsufsort(old)
        pos, ln = oracle.bsdiff_search(old, sa, new)
        its = iterations(old, new, pos, ln)
        st = simulate(old, new, its)
        print(f"{name}: {mib:g} MiB, {edits} edits: {st}  ({time.perf_counter() - t0:.1f} s)", flush=True)


if __name__ == "__main__":
    main()
