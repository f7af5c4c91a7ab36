"""Host-side mirror of the reference's ISuffixSort plugin interface for the HIP backend.

Reference interface (jzebedee/deltaq):
    src/DeltaQ.SuffixSorting.Abstractions/ISuffixSort.cs:9-28
        IMemoryOwner<int> Sort(ReadOnlySpan<byte> text);
        void Sort(ReadOnlySpan<byte> text, Span<int> suffixes);
    src/DeltaQ.SuffixSorting.LibDivSufSort/LibDivSufSort.cs:10-32  (the provider this one replaces)

``HipSuffixSort`` keeps the reference's names, argument meaning and error behaviour
(``ValueError`` with the reference's message where C# throws ``ArgumentException``) so the
parity tests read like LibDivSufSortTests.cs.  All compute happens in
libdq_sufsort_hip.so; this file only marshals buffers.
"""
from __future__ import annotations

import numpy as np

from . import _abi

LENGTH_MISMATCH_MESSAGE = "Text and suffix buffers should have the same length"  # LibDivSufSort.cs:31
INT_MAX = 0x7FFFFFFF


def _as_text(text) -> np.ndarray:
    if isinstance(text, np.ndarray):
        if text.dtype != np.uint8:
            raise TypeError("text must be bytes-like or a uint8 array")
        return np.ascontiguousarray(text)
    return np.frombuffer(memoryview(text).cast("B"), dtype=np.uint8)


def _is_torch_tensor(x) -> bool:
    return type(x).__module__.startswith("torch") and hasattr(x, "data_ptr")


class HipSuffixSort:
    """MI355X suffix sorting provider: drop-in for ``new LibDivSufSort()``.

    ``device`` is the HIP device ordinal (-1: ``DQ_HIP_DEVICE`` or device 0).
    Instances are stateless and may be shared between threads, like the reference's
    providers (SuffixSortingBenchmarks.cs:59-61).
    """

    def __init__(self, device: int = -1):
        self.device = int(device)
        self._lib = _abi.load()          # raises BackendMissingError when not built

    # -- IMemoryOwner<int> Sort(ReadOnlySpan<byte> text)   (ISuffixSort.cs:18) ----------------
    def Sort(self, text, suffixes=None, *, index_dtype=None):
        """``Sort(text)`` returns a new suffix array; ``Sort(text, suffixes)`` fills the caller's.

        Host buffers (bytes / bytearray / numpy uint8) go through ``dq_sufsort_hip_i32``;
        torch CUDA tensors stay on the device (``dq_sufsort_hip_dev_i32``).  ``index_dtype``
        of ``np.int64`` selects the 64-bit entry points (inputs beyond the reference's int limit).
        """
        if _is_torch_tensor(text):
            return self._sort_device(text, suffixes, index_dtype)
        T = _as_text(text)
        n = T.size
        if suffixes is None:
            dtype = np.dtype(index_dtype or (np.int32 if n <= INT_MAX else np.int64))
            sa = np.empty(n, dtype=dtype)    # uncleared, like MemoryOwner<int>.Allocate (LibDivSufSort.cs:14)
            self._sort_host(T, sa)
            return sa
        # -- void Sort(ReadOnlySpan<byte> text, Span<int> suffixes)   (ISuffixSort.cs:27) ------
        if not isinstance(suffixes, np.ndarray) or suffixes.dtype not in (np.int32, np.int64):
            raise TypeError("suffixes must be a numpy int32 or int64 array")
        if suffixes.ndim != 1 or not suffixes.flags.c_contiguous:
            raise TypeError("suffixes must be a contiguous 1-D array")
        if suffixes.size != n:
            raise ValueError(LENGTH_MISMATCH_MESSAGE)            # LibDivSufSort.cs:23-31
        self._sort_host(T, suffixes)
        return None

    sort = Sort

    def _sort_host(self, T: np.ndarray, sa: np.ndarray) -> None:
        fn = self._lib.dq_sufsort_hip_i32 if sa.dtype == np.int32 else self._lib.dq_sufsort_hip_i64
        tp = T.ctypes.data if T.size else None
        sp = sa.ctypes.data if sa.size else None
        _abi.check(fn(tp, T.size, sp, self.device))

    def _sort_device(self, text, suffixes, index_dtype):
        import torch

        if text.dtype != torch.uint8 or text.dim() != 1 or not text.is_contiguous():
            raise TypeError("device text must be a contiguous 1-D uint8 tensor")
        if not text.is_cuda:
            raise TypeError("torch text tensors must live on the GPU; pass host data as bytes/numpy")
        n = text.numel()
        ret = None
        if suffixes is None:
            tdt = torch.int64 if (index_dtype in (np.int64, torch.int64) or n > INT_MAX) else torch.int32
            suffixes = torch.empty(n, dtype=tdt, device=text.device)
            ret = suffixes
        else:
            if suffixes.dtype not in (torch.int32, torch.int64) or not suffixes.is_contiguous():
                raise TypeError("suffixes must be a contiguous int32 or int64 tensor")
            if suffixes.device != text.device:
                raise TypeError("text and suffixes must be on the same device")
            if suffixes.numel() != n:
                raise ValueError(LENGTH_MISMATCH_MESSAGE)
        fn = (self._lib.dq_sufsort_hip_dev_i32 if suffixes.dtype == torch.int32
              else self._lib.dq_sufsort_hip_dev_i64)
        dev = text.device.index if text.device.index is not None else torch.cuda.current_device()
        cur = torch.cuda.current_stream(text.device)
        stream = cur.cuda_stream
        if not stream:
            # torch's default stream is the legacy null stream: the library then works on its own
            # (non-blocking) stream, so whatever produced `text` has to be finished first
            cur.synchronize()
        _abi.check(fn(text.data_ptr() if n else None, n, suffixes.data_ptr() if n else None, dev, stream))
        return ret


def device_count() -> int:
    return int(_abi.load().dq_device_count())
