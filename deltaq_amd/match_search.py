"""Host-side mirror of Diff.Create's match search for the HIP backend.

Reference (jzebedee/deltaq, src/DeltaQ.BsDiff/Diff.cs):
    :267-298  private static int Search(ReadOnlySpan<int> I, ReadOnlySpan<byte> oldData,
                                        ReadOnlySpan<byte> newData, int start, int end, out int pos)
    :106      len = Search(I, oldData, newData[scan..], 0, oldData.Length, out pos);   (the scan loop's call)

``HipMatchSearch.Search`` keeps the reference's argument meaning -- I (the suffix array of oldData; the zeroed
sentinel slot I[n] of Diff.cs:78 is implied), oldData, newData -- but answers a BATCH of scan positions at once:
for every scan it returns exactly the (pos, len) the reference's Search returns for ``newData[scan..]``.
All compute happens in libdq_sufsort_hip.so (``dq_bsdiff_search_*``); this file only marshals buffers.
torch CUDA tensors stay on the device (the suffix array left there by ``HipSuffixSort.Sort``), numpy / bytes
go through the host entry points.
"""
from __future__ import annotations

import numpy as np

from . import _abi
from .suffix_sort import _as_text, _is_torch_tensor


class HipMatchSearch:
    def __init__(self, device: int = -1):
        self.device = int(device)
        self._lib = _abi.load()

    def Search(self, I, oldData, newData, scans=None, *, scan0: int = 0, count=None, cap: int = 0):
        """(pos, len) of ``Search(I, oldData, newData[scan..], 0, len(oldData))`` for scan in ``scans`` (an
        int64 array / tensor) or for ``scan0 .. scan0 + count - 1``.  ``cap > 0``: a query whose comparison
        would run more than cap bytes beyond what is known to match comes back with len = -1."""
        if _is_torch_tensor(I) or _is_torch_tensor(oldData) or _is_torch_tensor(newData):
            return self._search_device(I, oldData, newData, scans, scan0, count, cap)
        O, N = _as_text(oldData), _as_text(newData)
        I = np.ascontiguousarray(I)
        if I.dtype not in (np.int32, np.int64):
            raise TypeError("I must be an int32 or int64 suffix array")
        if I.size not in (O.size, O.size + 1):
            raise ValueError("I must hold len(oldData) entries (+ optionally the sentinel slot)")
        if scans is not None:
            scans = np.ascontiguousarray(scans, dtype=np.int64)
            count = scans.size
        elif count is None:
            count = N.size - scan0
        pos = np.empty(count, dtype=I.dtype)
        ln = np.empty(count, dtype=I.dtype)
        fn = self._lib.dq_bsdiff_search_i32 if I.dtype == np.int32 else self._lib.dq_bsdiff_search_i64
        p = lambda a: a.ctypes.data if a is not None and a.size else None
        _abi.check(fn(p(O), O.size, p(I), p(N), N.size, p(scans), scan0, count, cap, p(pos), p(ln), self.device))
        return pos, ln

    def _search_device(self, I, oldData, newData, scans, scan0, count, cap):
        import torch

        for t, name in ((I, "I"), (oldData, "oldData"), (newData, "newData")):
            if not _is_torch_tensor(t) or not t.is_cuda or not t.is_contiguous():
                raise TypeError(f"{name} must be a contiguous CUDA tensor when any argument is one")
        if oldData.dtype != torch.uint8 or newData.dtype != torch.uint8:
            raise TypeError("oldData / newData must be uint8 tensors")
        if I.dtype not in (torch.int32, torch.int64):
            raise TypeError("I must be an int32 or int64 tensor")
        n, m = oldData.numel(), newData.numel()
        if I.numel() not in (n, n + 1):
            raise ValueError("I must hold len(oldData) entries (+ optionally the sentinel slot)")
        if scans is not None:
            if not _is_torch_tensor(scans):
                scans = torch.as_tensor(np.ascontiguousarray(scans, dtype=np.int64), device=I.device)
            if scans.dtype != torch.int64 or not scans.is_contiguous():
                raise TypeError("scans must be a contiguous int64 tensor")
            count = scans.numel()
        elif count is None:
            count = m - scan0
        pos = torch.empty(count, dtype=I.dtype, device=I.device)
        ln = torch.empty(count, dtype=I.dtype, device=I.device)
        fn = self._lib.dq_bsdiff_search_dev_i32 if I.dtype == torch.int32 else self._lib.dq_bsdiff_search_dev_i64
        dev = I.device.index if I.device.index is not None else torch.cuda.current_device()
        cur = torch.cuda.current_stream(I.device)
        stream = cur.cuda_stream
        if not stream:
            cur.synchronize()
        ptr = lambda t: t.data_ptr() if t is not None and t.numel() else None
        _abi.check(fn(ptr(oldData), n, ptr(I), ptr(newData), m, ptr(scans), scan0, count, cap, ptr(pos), ptr(ln), dev,
                      stream))
        return pos, ln
