"""Host-side mirror of the reference's delta codec entry points for the HIP backend.

Reference (jzebedee/deltaq, src/DeltaQ.BsDiff):
    Diff.cs:27    public static void Create(ReadOnlySpan<byte> oldData, ReadOnlySpan<byte> newData,
                                            Stream output, ISuffixSort suffixSort)
    Patch.cs:34   public static void Apply(ReadOnlySpan<byte> input, ReadOnlySpan<byte> diff, Stream output)
    Patch.cs:43   public static void Apply(Stream input, OpenPatchStream openPatchStream, Stream output)

``Diff.Create`` / ``Patch.Apply`` keep the reference's names and argument meaning (streams are Python file
objects; the reference's argument checks become ``ValueError``).  All compute happens in libdq_sufsort_hip.so
(``dq_bsdiff_create``: suffix array and match search on the MI355X, the scan loop and the bzip2 framing on the
host around them; ``dq_bspatch_apply``: host code); this file only marshals buffers.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _abi
from .suffix_sort import _as_text


def _bytes_of(x) -> np.ndarray:
    if hasattr(x, "read"):
        x = x.read()
    return _as_text(x)


class Diff:
    @staticmethod
    def Create(oldData, newData, output, suffixSort=None) -> None:
        """Writes a BSDIFF40 patch that turns ``oldData`` into ``newData`` to the stream ``output``.
        ``suffixSort``: a ``HipSuffixSort`` (its device is used) or None."""
        if output is None:
            raise ValueError("output")                               # ArgumentNullException(nameof(output)), Diff.cs:31
        if not (hasattr(output, "write") and (not hasattr(output, "writable") or output.writable())):
            raise ValueError("Output stream must be writable.")      # Diff.cs:50
        if hasattr(output, "seekable") and not output.seekable():
            raise ValueError("Output stream must be seekable.")      # Diff.cs:45
        output.write(Diff.CreateBytes(oldData, newData, getattr(suffixSort, "device", -1)))

    @staticmethod
    def CreateBytes(oldData, newData, device: int = -1) -> bytes:
        L = _abi.load()
        O, N = _as_text(oldData), _as_text(newData)
        cap = L.dq_bsdiff_patch_bound(O.size, N.size)
        buf = np.empty(cap, dtype=np.uint8)
        ln = ctypes.c_int64()
        p = lambda a: a.ctypes.data if a.size else None
        _abi.check(L.dq_bsdiff_create(p(O), O.size, p(N), N.size, buf.ctypes.data, cap, ctypes.byref(ln), device))
        return buf[:ln.value].tobytes()

    @staticmethod
    def Scan(oldData, newData, device: int = -1):
        """The raw streams of the scan loop (before bzip2): (ctrl triples [k, 3] int64, diff bytes, extra bytes,
        {searches, windows, exact})."""
        L = _abi.load()
        O, N = _as_text(oldData), _as_text(newData)
        m = N.size
        ctrl = np.empty(3 * (m + 1), dtype=np.int64)
        diff = np.empty(max(m, 1), dtype=np.uint8)
        extra = np.empty(max(m, 1), dtype=np.uint8)
        nc, nd, ne = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        stats = (ctypes.c_int64 * 3)()
        p = lambda a: a.ctypes.data if a.size else None
        _abi.check(L.dq_bsdiff_scan_i32(p(O), O.size, p(N), m, ctrl.ctypes.data, m + 1, ctypes.byref(nc), diff.ctypes.data,
                                        ctypes.byref(nd), extra.ctypes.data, ctypes.byref(ne), stats, device))
        return (ctrl[:3 * nc.value].reshape(-1, 3).copy(), diff[:nd.value].copy(), extra[:ne.value].copy(),
                {"searches": stats[0], "windows": stats[1], "exact": stats[2],
                 "host_loop_fallbacks": _abi.last_diff_info()["host_loop_fallbacks"]})


class DiffIndex:
    """One old file on one device, ready to be diffed against many new files: what ``Diff.Create`` computes from
    ``oldData`` alone (the suffix array, Diff.cs:89-90) is computed once.  ``Create`` returns the patch
    ``Diff.CreateBytes(oldData, newData)`` returns.

    ``device_text`` / ``device_sa``: CUDA tensors (uint8 / int32) that already hold the text and its suffix array
    on this device -- a rank that received them by broadcast -- instead of sorting here."""

    def __init__(self, oldData, device: int = -1, device_text=None, device_sa=None):
        L = _abi.load()
        self._lib = L
        self._old = np.ascontiguousarray(_as_text(oldData))          # the scan loop reads it: kept alive with the index
        self._keep = (device_text, device_sa)
        h = ctypes.c_void_p()
        d_old = d_sa = None
        if device_text is not None:
            if device_sa is None or int(device_text.numel()) != self._old.size or int(device_sa.numel()) != self._old.size:
                raise ValueError("device_text and device_sa must both be given, with one entry per byte of oldData")
            d_old, d_sa = device_text.data_ptr(), device_sa.data_ptr()
            device = device_text.device.index if device < 0 and device_text.device.index is not None else device
        p = self._old.ctypes.data if self._old.size else None
        _abi.check(L.dq_bsdiff_index_create(p, self._old.size, d_old, d_sa, device, ctypes.byref(h)))
        self._h = h

    def clone(self, device: int = -1) -> "DiffIndex":
        """One more copy of this index on ``device`` (dq_bsdiff_index_clone): device-to-device copies -- xGMI between the
        devices of a node -- instead of a second sort.  The copy shares this index's host text and is closed on its own."""
        other = object.__new__(DiffIndex)
        other._lib, other._old, other._keep = self._lib, self._old, ()
        h = ctypes.c_void_p()
        _abi.check(self._lib.dq_bsdiff_index_clone(self._h, device, ctypes.byref(h)))
        other._h = h
        return other

    def buffers(self):
        """(device pointer of the text, device pointer of the suffix array, n)"""
        a, b, n = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_int64()
        _abi.check(self._lib.dq_bsdiff_index_buffers(self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(n)))
        return a.value, b.value, n.value

    def Create(self, newData) -> bytes:
        N = _as_text(newData)
        cap = self._lib.dq_bsdiff_patch_bound(self._old.size, N.size)
        buf = np.empty(cap, dtype=np.uint8)
        ln = ctypes.c_int64()
        _abi.check(self._lib.dq_bsdiff_index_diff(self._h, N.ctypes.data if N.size else None, N.size, buf.ctypes.data, cap,
                                                  ctypes.byref(ln)))
        return buf[:ln.value].tobytes()

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.dq_bsdiff_index_free(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Patch:
    @staticmethod
    def Apply(input, diff, output=None):
        """Applies the BSDIFF40 patch ``diff`` to ``input``; writes the new file to the stream ``output`` or, without
        one, returns it as bytes.  A patch the reference rejects raises ``ValueError("Corrupt patch")``."""
        L = _abi.load()
        O, P = _bytes_of(input), _bytes_of(diff)
        size = ctypes.c_int64()
        p = lambda a: a.ctypes.data if a.size else None
        rc = L.dq_bspatch_apply(p(O), O.size, P.ctypes.data if P.size else ctypes.c_char_p(b"").value, P.size, None, 0, ctypes.byref(size))
        if rc != 0:
            raise ValueError(_abi.last_error())
        out = np.empty(size.value, dtype=np.uint8)
        rc = L.dq_bspatch_apply(p(O), O.size, P.ctypes.data, P.size, out.ctypes.data if out.size else P.ctypes.data, size.value,
                                ctypes.byref(size))
        if rc != 0:
            raise ValueError(_abi.last_error())                      # InvalidOperationException("Corrupt patch")
        if output is None:
            return out.tobytes()
        output.write(out.tobytes())
        return None
