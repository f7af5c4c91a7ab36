"""deltaq_amd -- MI355X (gfx950) suffix-sorting backend for DeltaQ's ISuffixSort plugin point.

    from deltaq_amd import HipSuffixSort
    sa = HipSuffixSort().Sort(text)            # == new LibDivSufSort().Sort(text)

The package is a thin host layer over libdq_sufsort_hip.so (hand-written HIP kernels, C ABI in
include/dq_sufsort.h).  Importing the package does not load the library; constructing a
provider does, and raises if it has not been built.  There is no CPU fallback.
"""
from ._abi import BackendMissingError, SuffixSortError  # noqa: F401
from .suffix_sort import HipSuffixSort, device_count, LENGTH_MISMATCH_MESSAGE  # noqa: F401
from .match_search import HipMatchSearch  # noqa: F401
from .bsdiff import Diff, DiffIndex, Patch  # noqa: F401

__all__ = ["HipSuffixSort", "HipMatchSearch", "Diff", "DiffIndex", "Patch", "device_count", "BackendMissingError", "SuffixSortError",
           "LENGTH_MISMATCH_MESSAGE"]
