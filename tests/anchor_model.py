"""CPU model of the device's anchor search (deltaq_amd/csrc/dq_anchor_scan.h): step 1 of the scan loop (Diff.cs:100-125)
evaluated window by window, the way the persistent kernel does it.

The reference's loop keeps two running numbers while it walks from `base` (the end of the last match): `oldscore`
(here: carried) and `scsc` (counted).  With agree(k) = "the previous alignment still gets byte k right"
(k + shift < n and old[k + shift] == new[k]) and  M_j = max(base, max_{k <= j} (k + len_k))  they satisfy, at the break
test of position j,
        carried_j = #{ k in [j, M_j) : agree(k) } = cnt(base, M_j) - cnt(base, j),
and every search also reports cw_j = cnt(j, j + len_j), so that cnt(base, j + len_j) = S_j + cw_j with S_j = cnt(base, j)
-- a prefix count over the window's positions.  The tests of a whole window are then a prefix maximum of the match
ends that carries cnt(base, end), minus S_j.  A position whose answer is not exact yet (the search hit its cap) is a
STOP POINT: the window is evaluated up to it, then it is searched again exactly and taken on its own, and a new
window starts behind it.  test_models_cpu.py checks the anchors this gives against a literal transcription of the loop and the
streams the product's emitter makes of them against the oracle's."""
import numpy as np


def literal_anchors(old, new, pos, ln):
    """Diff.cs:100-125 as written (dq_bsdiff.h::scan_loop, step 1): the (cursor, hit_pos) of every emitted triple."""
    n, m = len(old), len(new)
    cursor = hit_pos = hit_len = shift = 0
    out = []
    while cursor < m:
        carried = 0
        cursor += hit_len
        counted = cursor
        while cursor < m:
            hit_pos, hit_len = int(pos[cursor]), int(ln[cursor])
            if counted < cursor + hit_len:
                end = cursor + hit_len
                upto = min(end, n - shift)
                for k in range(counted, upto):
                    carried += old[k + shift] == new[k]
                counted = end
            if (hit_len == carried and hit_len != 0) or hit_len > carried + 8:
                break
            carried -= int(cursor + shift < n and old[cursor + shift] == new[cursor])
            cursor += 1
        if hit_len == carried and cursor != m:
            continue
        out.append((cursor, hit_pos))
        shift = hit_pos - cursor
    return out


def windowed_anchors(old, new, pos, ln, capped, w_first=8, w_next=32, extra=0, rng=None, loose=None):
    """The kernel's evaluation.  capped(j) -> True: the window's answer for position j comes back undecided (len < 0)
    unless it is the window's first position; the exact answer (pos[j], ln[j]) is fetched when the position is taken on
    its own.  Every answer comes with cw = cnt(j, j + len), counted by the wave that searched (`extra` is unused: the
    first form of the kernel covered a few bytes behind the window instead).
    loose(j) -> slack >= 0: the count of a match that lies under another alignment comes as an UPPER BOUND (the wave
    counted the front of a long match and took the rest as agreeing): cw + slack, at most len.  Such a position breaks
    for certain if its length beats even the bound; otherwise -- if the bound would be carried on -- it is a stop point."""
    n, m = len(old), len(new)
    cursor = hit_pos = hit_len = shift = 0
    out = []
    stats = dict(windows=0, stops=0)

    def agree(k):
        return int(k + shift < n and old[k + shift] == new[k])

    def cnt(a, b):
        return sum(agree(k) for k in range(a, b))

    while cursor < m:
        cursor += hit_len
        base = cursor
        i, M, C, S = base, base, 0, 0                      # C = cnt(base, M), S = cnt(base, i); M >= i
        found = False
        first_window = True
        streak = 0                                         # stop points in a row that did not break
        last = None                                        # last position walked over (exact answer)
        while i < m and not found:
            W = w_first if first_window else w_next
            first_window = False
            c = min(W, m - i)
            stats["windows"] += 1
            agp = [0]
            for k in range(i, i + c):
                agp.append(agp[-1] + agree(k))             # agp[x] = cnt(i, i + x)
            lens, cws, bound = [], [], []
            for j in range(i, i + c):
                exact = (j == i) or streak >= 2 or not capped(j)
                l = int(ln[j]) if exact else -1
                lens.append(l)
                cw = cnt(j, j + l) if l >= 0 else 0
                slack = 0
                if loose is not None and l > 0 and int(pos[j]) - j != shift:
                    slack = min(int(loose(j)), l - cw)
                cws.append(cw + slack)
                bound.append(slack > 0)
            s = next((t for t, l in enumerate(lens) if l < 0), None)          # stop point: first capped position
            upto = c if s is None else s
            f = None
            Mj, Cj = M, C
            for t in range(upto):                          # prefix maximum of the ends, carrying cnt(base, end)
                j = i + t
                Sj = S + agp[t]
                e = j + lens[t]
                rests_on_bound = bound[t] and e > Mj       # (an end behind the running maximum: its count is not used)
                if e > Mj:
                    Mj, Cj = e, Sj + cws[t]
                carried = Cj - Sj
                if rests_on_bound:
                    if lens[t] > carried + 8:              # carried <= this bound: it breaks whatever the true count is
                        f = t
                    else:                                  # undecided: this position is taken on its own, exactly
                        s, upto = t, t
                    break
                if (lens[t] == carried and lens[t] != 0) or lens[t] > carried + 8:
                    f = t
                    break
            if f is not None:
                cursor, hit_pos, hit_len, carried_at = i + f, int(pos[i + f]), lens[f], carried
                found = True
                break
            # (what the positions before the stop point / the whole window left)
            Mj, Cj = M, C
            for t in range(upto):
                if i + t + lens[t] > Mj:
                    Mj, Cj = i + t + lens[t], S + agp[t] + cws[t]
            if s is None:                                  # the whole window went by
                last = i + c - 1
                S += agp[c]
                C, M = Cj, Mj
                i += c
            else:                                          # the stop point, on its own
                stats["stops"] += 1
                j = i + s
                l = int(ln[j])                             # exact now
                last = j
                Sj = S + agp[s]
                C, M = Cj, Mj
                if j + l > M:
                    M, C = j + l, Sj + cnt(j, j + l)
                carried = C - Sj
                if (l == carried and l != 0) or l > carried + 8:
                    cursor, hit_pos, hit_len, carried_at = j, int(pos[j]), l, carried
                    found = True
                    break
                streak += 1
                S = Sj + agree(j)
                i = j + 1
            if M < i:                                      # keep M >= i (every later M_j is >= j >= i anyway)
                C += cnt(M, i)
                M = i
        if not found:
            cursor = m
            if last is not None:
                hit_pos, hit_len = int(pos[last]), int(ln[last])
            out.append((cursor, hit_pos))
            shift = hit_pos - cursor
            continue
        if hit_len == carried_at and cursor != m:
            continue
        out.append((cursor, hit_pos))
        shift = hit_pos - cursor
    return out, stats
