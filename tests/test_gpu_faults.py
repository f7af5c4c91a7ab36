"""Error paths under test (failure detection / recovery): DQ_FAULT=alloc:K | hip:K | spin makes
the K-th allocation / checked HIP call of ONE call fail, or every bounded device spin give up at once (dq_runtime.h).
What must hold: the documented error code and a message, nothing written into the caller's output, the next call on
the same thread bit-exact, and -- after dq_sufsort_hip_release -- no allocation left behind.

And the device's anchor scan (dq_anchor_scan.h) on adversarial schedules: one workgroup of the persistent grid made the
straggler of every window, grids of 8 / 48 / 128 workgroups, pollers that sleep 1 or 32, new files whose last window
ends exactly on / one past a buffer of answers -- every patch against the oracle's restatement of the reference loop;
a starved grid (spin bound hit) must hand the file to the host loop, say so in dq_last_diff_info, and still produce
the oracle's streams."""
import numpy as np
import pytest

from test_gpu_match_search import edited

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sorter(backend_lib):
    from deltaq_amd import HipSuffixSort
    assert backend_lib.dq_device_count() >= 1
    return HipSuffixSort(0)


def free_hbm():
    import torch
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info(0)[0]


SENTINEL = 0x5A5A5A5A


@pytest.mark.parametrize("n", [5000, 300_000, 3_000_000])
def test_injected_allocation_failures_leave_the_library_usable(backend_lib, sorter, oracle_mod, monkeypatch, n):
    from deltaq_amd import SuffixSortError, _abi
    T = oracle_mod.gen_enwik_like(n, 77, 4096)
    ref = oracle_mod.divsufsort(T)
    seen = 0
    for k in range(1, 8):
        backend_lib.dq_sufsort_hip_release()           # every allocation of the call is made again: context, pinned areas, workspace
        out = np.full(n, SENTINEL, np.int32)
        monkeypatch.setenv("DQ_FAULT", f"alloc:{k}")
        try:
            sorter.Sort(T, out)
            failed = False
        except SuffixSortError as e:
            failed = True
            assert e.code == _abi.DQ_ERR_OOM, (k, e)
            assert "out of memory" in str(e).lower() or "hipMalloc" in str(e) or "context" in str(e), str(e)
        monkeypatch.delenv("DQ_FAULT")
        if failed:
            seen += 1
            assert (out == SENTINEL).all(), f"alloc:{k}: the failed call wrote into the caller's suffix array"
        else:
            assert np.array_equal(out, ref)            # fewer than k allocations in this call: it simply succeeded
        out2 = np.full(n, SENTINEL, np.int32)
        sorter.Sort(T, out2)                           # the next call on the same thread
        assert np.array_equal(out2, ref), f"after alloc:{k}"
    assert seen >= 2, "the fault plan never fired: DQ_FAULT is not being honoured"


@pytest.mark.parametrize("n,every", [(5000, 1), (200_000, 1), (2_000_000, 7)])
def test_injected_hip_failures_at_every_step(backend_lib, sorter, oracle_mod, monkeypatch, n, every):
    """The K-th checked HIP call of a sort fails, for K = 1, 1 + every, ... until a call gets through: each failure is
    DQ_ERR_HIP with the injected message, leaves the output alone, and the next sort is bit-exact."""
    from deltaq_amd import SuffixSortError, _abi
    T = oracle_mod.gen_enwik_like(n, 5, 8192) if n > 5000 else oracle_mod.net_random_bytes(n)
    ref = oracle_mod.divsufsort(T)
    sorter.Sort(T)                                     # buffers grown: from here on the calls differ only in the fault
    fired = 0
    k = 1
    while k < 4000:
        out = np.full(n, SENTINEL, np.int32)
        monkeypatch.setenv("DQ_FAULT", f"hip:{k}")
        try:
            sorter.Sort(T, out)
            monkeypatch.delenv("DQ_FAULT")
            assert np.array_equal(out, ref)            # past the last checked call of this sort
            break
        except SuffixSortError as e:
            monkeypatch.delenv("DQ_FAULT")
            fired += 1
            assert e.code == _abi.DQ_ERR_HIP and "injected fault" in str(e), (k, str(e))
            assert (out == SENTINEL).all(), f"hip:{k}: the failed call wrote into the caller's suffix array"
        if fired % 5 == 1:
            out2 = np.empty(n, np.int32)
            sorter.Sort(T, out2)
            assert np.array_equal(out2, ref), f"after hip:{k}"
        k += every
    assert fired >= 3
    assert np.array_equal(sorter.Sort(T), ref)


def test_look_back_spin_bound_is_an_error_not_a_hang(backend_lib, sorter, oracle_mod, monkeypatch):
    """DQ_FAULT=spin: the first empty poll of a look-back raises the kernels' error word; the host reports DQ_ERR_HIP
    ("... timed out"), and the production bound is back for the next call."""
    from deltaq_amd import SuffixSortError, _abi
    T = oracle_mod.gen_uniform(48 << 20, 0xFA17)       # thousands of tiles: some tile meets an unpublished predecessor
    out = np.full(T.size, SENTINEL, np.int32)
    monkeypatch.setenv("DQ_FAULT", "spin")
    with pytest.raises(SuffixSortError) as ei:
        sorter.Sort(T, out)
    monkeypatch.delenv("DQ_FAULT")
    assert ei.value.code == _abi.DQ_ERR_HIP and "timed out" in str(ei.value), str(ei.value)
    assert (out == SENTINEL).all()
    sorter.Sort(T, out)
    assert oracle_mod.sufcheck_mt(T, out) == oracle_mod.CHECK_DONE
    assert oracle_mod.verify_sampled(T, out, 200_000, 3) == -1


def test_release_leaves_no_allocation_behind(backend_lib, sorter, oracle_mod, monkeypatch):
    from deltaq_amd import Diff, SuffixSortError
    import torch
    T = oracle_mod.gen_uniform(8 << 20, 3)
    sorter.Sort(T)                                     # (code objects loaded, runtime pools grown: not what is measured)
    sorter.Sort(T[:3000])
    backend_lib.dq_sufsort_hip_release()
    torch.cuda.empty_cache()
    before = free_hbm()
    sorter.Sort(T)
    sorter.Sort(T[:3000])
    old = oracle_mod.gen_uniform(1 << 20, 4)
    Diff.CreateBytes(old, edited(np.random.default_rng(1), old, 20), 0)
    for fault in ("hip:3", "hip:9", "hip:18"):         # (this sort makes ~25 checked calls)
        monkeypatch.setenv("DQ_FAULT", fault)
        with pytest.raises(SuffixSortError):
            sorter.Sort(T)
        monkeypatch.delenv("DQ_FAULT")
    big = oracle_mod.gen_uniform(24 << 20, 5)           # a larger text: the workspace must grow, and its allocation fails
    monkeypatch.setenv("DQ_FAULT", "alloc:1")
    with pytest.raises(SuffixSortError):
        sorter.Sort(big)
    monkeypatch.delenv("DQ_FAULT")
    sorter.Sort(T)
    assert free_hbm() < before                         # (the cached workspace is there ...)
    backend_lib.dq_sufsort_hip_release()
    assert abs(free_hbm() - before) <= (8 << 20), (before, free_hbm())      # ... and gone: no buffer of a failed call was leaked


def test_flags_are_ignored_without_the_debug_gate(backend_lib, sorter, oracle_mod, monkeypatch):
    """A stray DQ_* variable in a production environment changes nothing: only DQ_DEBUG_FLAGS=1 arms the overrides."""
    T = oracle_mod.gen_uniform(200_000, 9)
    ref = oracle_mod.divsufsort(T)
    monkeypatch.setenv("DQ_DEBUG_FLAGS", "0")
    monkeypatch.setenv("DQ_FAULT", "hip:1")
    assert np.array_equal(sorter.Sort(T), ref)


# ---------------------------------------------------------------------------------- the device's anchor scan
def scan_pairs(oracle_mod):
    rng = np.random.default_rng(0xA5C)
    old = oracle_mod.gen_uniform(700_000, 31)
    text = oracle_mod.gen_enwik_like(600_000, 8, 8192)
    out = [(old, edited(rng, old, 60)), (text, edited(rng, text, 80)), (old, oracle_mod.gen_uniform(70_000, 32))]
    # new files whose length puts the last window of either kind exactly on / one past / one short of a buffer of
    # answers (512 positions per wave window, 32768 per lane window)
    for m in (512, 513, 1023, 1024, 1025, 32768, 32769, 3 * 512 + 32768, 3 * 512 + 32768 + 1, 3 * 512 + 2 * 32768 - 1):
        out.append((old, oracle_mod.gen_uniform(m, 40 + m)))               # unrelated: wave windows, then lane windows
        x = old[1000:1000 + m].copy()
        if m > 600:
            x[300] ^= 0x55
        out.append((old, x))                                             # one long match that ends with the file
    return out


def check_pair(oracle_mod, old, new, tag):
    from deltaq_amd import Diff, _abi
    sa = oracle_mod.divsufsort(old)
    wc, wd, we, _ = oracle_mod.bsdiff_scan(old, sa, new)
    ctrl, diff, extra, stats = Diff.Scan(old, new, 0)
    assert np.array_equal(ctrl.reshape(-1), np.asarray(wc).reshape(-1)) and np.array_equal(diff, wd) and np.array_equal(extra, we), tag
    return stats, _abi.last_diff_info()


@pytest.mark.parametrize("groups", [8, 48, 128])
@pytest.mark.parametrize("poll", [1, 32])
def test_anchor_scan_with_a_straggler_in_every_window(backend_lib, oracle_mod, monkeypatch, groups, poll):
    monkeypatch.setenv("DQ_SCAN_GROUPS", str(groups))
    monkeypatch.setenv("DQ_SCAN_POLL_SLEEP", str(poll))
    pairs = scan_pairs(oracle_mod)
    for slow in (1, groups // 2 + 1, groups):          # the first, a middle and the last workgroup ~50 us late, every window
        monkeypatch.setenv("DQ_SCAN_SLOW_GROUP", str(slow))
        for i, (old, new) in enumerate(pairs if slow == 1 else pairs[:6]):
            stats, info = check_pair(oracle_mod, old, new, (groups, poll, slow, i))
            assert info["host_loop_fallbacks"] == 0 and info["scan_groups"] == groups, info


def test_starved_anchor_scan_falls_back_loudly(backend_lib, oracle_mod, monkeypatch):
    """Spin bound of 2 polls: the grid gives up, the host loop produces the oracle's streams, and the call says so."""
    pairs = scan_pairs(oracle_mod)[:3]
    monkeypatch.setenv("DQ_SCAN_SPIN_LOG2", "1")       # (the scan's bound alone: DQ_FAULT=spin would also fail the sort of the old file)
    fell = 0
    for i, (old, new) in enumerate(pairs):
        stats, info = check_pair(oracle_mod, old, new, ("spin", i))
        fell += info["host_loop_fallbacks"]
        assert stats["host_loop_fallbacks"] == info["host_loop_fallbacks"]
    assert fell >= 1, "a spin bound of 2 polls never made the device scan give up"
    monkeypatch.delenv("DQ_SCAN_SPIN_LOG2")
    stats, info = check_pair(oracle_mod, *pairs[0], "after")
    assert info["host_loop_fallbacks"] == 0


def test_starved_chains_fall_back_loudly(backend_lib, oracle_mod, monkeypatch):
    """Several grids on one file, a spin bound of 2 polls: whichever grid gives up, every launch is taken off the stream,
    the emitter threads end, the host loop produces the oracle's streams, and the next call works."""
    pairs = scan_pairs(oracle_mod)[:3]
    monkeypatch.setenv("DQ_SCAN_CHAINS", "6")
    monkeypatch.setenv("DQ_SCAN_MIN_SEG", "512")
    monkeypatch.setenv("DQ_SCAN_GROUPS", "16")
    monkeypatch.setenv("DQ_SCAN_SPIN_LOG2", "1")
    fell = 0
    for i, (old, new) in enumerate(pairs):
        stats, info = check_pair(oracle_mod, old, new, ("chains, spin", i))
        fell += info["host_loop_fallbacks"]
    assert fell >= 1, "a spin bound of 2 polls never made the grids give up"
    monkeypatch.delenv("DQ_SCAN_SPIN_LOG2")
    for i, (old, new) in enumerate(pairs):
        stats, info = check_pair(oracle_mod, old, new, ("chains, after", i))
        assert info["host_loop_fallbacks"] == 0, info
        if new.size >= 2048:
            assert info["chains_launched"] >= 2, info


def test_a_device_that_holds_too_few_workgroups_takes_the_host_loop(backend_lib, oracle_mod, monkeypatch):
    """Occupancy clamp (advisor, round 4): a part that holds fewer than 8 workgroups of the persistent grid never
    launches it; between 8 and 128 the grid shrinks to what is resident."""
    old, new = scan_pairs(oracle_mod)[0]
    monkeypatch.setenv("DQ_SCAN_GROUPS_CAP", "5")
    stats, info = check_pair(oracle_mod, old, new, "cap5")
    assert info["host_loop_fallbacks"] == 1
    monkeypatch.setenv("DQ_SCAN_GROUPS_CAP", "24")
    stats, info = check_pair(oracle_mod, old, new, "cap24")
    assert info["host_loop_fallbacks"] == 0 and info["scan_groups"] == 24
