"""ctypes binding of the C ABI declared in include/dq_sufsort.h.

This is the same surface the C# P/Invoke shim binds (bindings/csharp/HipSuffixSort.cs).
The library is loaded on first use and the load FAILS LOUDLY if the HIP backend has not
been built: there is no CPU fallback anywhere in this package.
"""
from __future__ import annotations

import ctypes
import os

from . import build as _build

DQ_OK = 0
DQ_ERR_BAD_ARGS = -1
DQ_ERR_OOM = -2
DQ_ERR_HIP = -3
DQ_ERR_TOO_LARGE = -4
DQ_ERR_NO_DEVICE = -5

K_RADIX_RANK = 2          # DQ_K_RADIX_RANK: the dominant kernel's profile category

# every symbol include/dq_sufsort.h declares
EXPORTS = (
    "dq_abi_version", "dq_device_count", "dq_last_error",
    "dq_sufsort_hip_i32", "dq_sufsort_hip_i64",
    "dq_sufsort_hip_dev_i32", "dq_sufsort_hip_dev_i64",
    "dq_sufsort_hip_batch_i32",
    "dq_bsdiff_search_dev_i32", "dq_bsdiff_search_dev_i64", "dq_bsdiff_search_i32", "dq_bsdiff_search_i64",
    "dq_bsdiff_create", "dq_bsdiff_patch_bound", "dq_bsdiff_scan_i32", "dq_bspatch_apply",
    "dq_bsdiff_index_create", "dq_bsdiff_index_clone", "dq_bsdiff_index_buffers", "dq_bsdiff_index_diff", "dq_bsdiff_index_free",
    "dq_sufsort_hip_workspace_bytes", "dq_sufsort_hip_release",
    "dq_profile_enable", "dq_profile_reset", "dq_profile_get", "dq_profile_kernel_name",
    "dq_profile_category_count",
    "dq_last_sort_info", "dq_last_diff_info", "dq_last_batch_info", "dq_device_numa_node",
)


class BackendMissingError(RuntimeError):
    """libdq_sufsort_hip.so is absent or unloadable -- the product cannot run."""


_lib = None


def lib_path() -> str:
    return os.environ.get("DQ_SUFSORT_LIB", _build.LIB_PATH)


def _preload_torch_hip_runtime() -> None:
    """PyTorch-ROCm wheels bundle their own libamdhip64.so.7.  Two HIP runtimes in one
    process cannot both own the GPU, so when torch is installed its runtime is mapped
    first and libdq_sufsort_hip.so (NEEDED libamdhip64.so.7) binds to that same copy.
    A host without torch (the C# shim, a C program) uses the system ROCm runtime."""
    if os.environ.get("DQ_NO_TORCH_PRELOAD"):
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.origin:
            return
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
    except Exception:  # pragma: no cover - best effort
        pass


def load() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise BackendMissingError(
            f"{path} not found: build the MI355X backend first "
            "(python -m deltaq_amd.build, or __graft_entry__.build()). There is no CPU fallback.")
    _preload_torch_hip_runtime()
    try:
        L = ctypes.CDLL(path)
    except OSError as e:  # pragma: no cover - depends on the machine
        raise BackendMissingError(f"cannot load {path}: {e}") from e
    vp, i32, i64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64
    L.dq_abi_version.restype = i32
    L.dq_abi_version.argtypes = []
    L.dq_device_count.restype = i32
    L.dq_device_count.argtypes = []
    L.dq_last_error.restype = ctypes.c_char_p
    L.dq_last_error.argtypes = []
    for name in ("dq_sufsort_hip_i32", "dq_sufsort_hip_i64"):
        getattr(L, name).restype = i32
        getattr(L, name).argtypes = [vp, i64, vp, i32]
    for name in ("dq_sufsort_hip_dev_i32", "dq_sufsort_hip_dev_i64"):
        getattr(L, name).restype = i32
        getattr(L, name).argtypes = [vp, i64, vp, i32, vp]
    L.dq_sufsort_hip_batch_i32.restype = i32
    L.dq_sufsort_hip_batch_i32.argtypes = [i32, vp, vp, vp, i32, vp]
    for name in ("dq_bsdiff_search_dev_i32", "dq_bsdiff_search_dev_i64"):
        getattr(L, name).restype = i32
        getattr(L, name).argtypes = [vp, i64, vp, vp, i64, vp, i64, i64, i64, vp, vp, i32, vp]
    for name in ("dq_bsdiff_search_i32", "dq_bsdiff_search_i64"):
        getattr(L, name).restype = i32
        getattr(L, name).argtypes = [vp, i64, vp, vp, i64, vp, i64, i64, i64, vp, vp, i32]
    L.dq_bsdiff_create.restype = i32
    L.dq_bsdiff_create.argtypes = [vp, i64, vp, i64, vp, i64, ctypes.POINTER(i64), i32]
    L.dq_bsdiff_patch_bound.restype = i64
    L.dq_bsdiff_patch_bound.argtypes = [i64, i64]
    L.dq_bsdiff_scan_i32.restype = i32
    L.dq_bsdiff_scan_i32.argtypes = [vp, i64, vp, i64, vp, i64, ctypes.POINTER(i64), vp, ctypes.POINTER(i64), vp,
                                     ctypes.POINTER(i64), vp, i32]
    L.dq_bsdiff_index_create.restype = i32
    L.dq_bsdiff_index_create.argtypes = [vp, i64, vp, vp, i32, ctypes.POINTER(vp)]
    L.dq_bsdiff_index_clone.restype = i32
    L.dq_bsdiff_index_clone.argtypes = [vp, i32, ctypes.POINTER(vp)]
    L.dq_bsdiff_index_buffers.restype = i32
    L.dq_bsdiff_index_buffers.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(i64)]
    L.dq_bsdiff_index_diff.restype = i32
    L.dq_bsdiff_index_diff.argtypes = [vp, vp, i64, vp, i64, ctypes.POINTER(i64)]
    L.dq_bsdiff_index_free.restype = None
    L.dq_bsdiff_index_free.argtypes = [vp]
    L.dq_bspatch_apply.restype = i32
    L.dq_bspatch_apply.argtypes = [vp, i64, vp, i64, vp, i64, ctypes.POINTER(i64)]
    L.dq_sufsort_hip_workspace_bytes.restype = i64
    L.dq_sufsort_hip_workspace_bytes.argtypes = [i64, i32]
    L.dq_sufsort_hip_release.restype = None
    L.dq_sufsort_hip_release.argtypes = []
    L.dq_profile_enable.restype = i32
    L.dq_profile_enable.argtypes = [i32]
    L.dq_profile_reset.restype = None
    L.dq_profile_reset.argtypes = []
    L.dq_profile_get.restype = i32
    L.dq_profile_get.argtypes = [i32, ctypes.POINTER(i64), ctypes.POINTER(ctypes.c_double),
                                 ctypes.POINTER(i64), ctypes.POINTER(i64)]
    L.dq_profile_kernel_name.restype = ctypes.c_char_p
    L.dq_profile_kernel_name.argtypes = [i32]
    L.dq_profile_category_count.restype = i32
    L.dq_profile_category_count.argtypes = []
    L.dq_last_sort_info.restype = i32
    L.dq_last_sort_info.argtypes = [ctypes.POINTER(i64)] * 3
    L.dq_last_diff_info.restype = i32
    L.dq_last_diff_info.argtypes = [ctypes.POINTER(i64), i32]
    L.dq_last_batch_info.restype = i32
    L.dq_last_batch_info.argtypes = [ctypes.POINTER(i64), i32]
    L.dq_device_numa_node.restype = i32
    L.dq_device_numa_node.argtypes = [i32]
    _lib = L
    return L


def last_error() -> str:
    return load().dq_last_error().decode("utf-8", "replace")


class SuffixSortError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"dq_sufsort_hip failed ({code}): {message}")
        self.code = code


def check(code: int) -> None:
    if code != DQ_OK:
        raise SuffixSortError(code, last_error())


def profile_snapshot() -> dict:
    """{kernel name: {launches, ms, elements, alg_bytes}} accumulated since dq_profile_reset."""
    L = load()
    out = {}
    for cat in range(L.dq_profile_category_count()):
        n, ms, el, by = ctypes.c_int64(), ctypes.c_double(), ctypes.c_int64(), ctypes.c_int64()
        L.dq_profile_get(cat, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(el), ctypes.byref(by))
        out[L.dq_profile_kernel_name(cat).decode()] = {
            "launches": n.value, "ms": ms.value, "elements": el.value, "alg_bytes": by.value}
    return out


def category_of(kernel_name: str) -> int:
    """Profile category (DQ_K_*) of a kernel name."""
    L = load()
    for cat in range(L.dq_profile_category_count()):
        if L.dq_profile_kernel_name(cat).decode() == kernel_name:
            return cat
    raise KeyError(kernel_name)


def last_diff_info() -> dict:
    """Shape of the last Diff.Create / index diff on this thread (dq_last_diff_info)."""
    L = load()
    v = (ctypes.c_int64 * 9)()
    L.dq_last_diff_info(v, 9)
    return {"searches": v[0], "windows": v[1], "exact": v[2], "host_loop_fallbacks": v[3], "scan_groups": v[4],
            "chains_launched": v[5], "chains_joined": v[6], "chains_dropped": v[7], "triples_from_chain_emitters": v[8]}


def last_batch_info() -> dict:
    """Shape of the last dq_sufsort_hip_batch_i32 on this thread (dq_last_batch_info)."""
    L = load()
    v = (ctypes.c_int64 * 6)()
    L.dq_last_batch_info(v, 6)
    return {"pipelined": v[0], "copy_in_ms": v[1] / 1e3, "sort_ms": v[2] / 1e3, "copy_out_ms": v[3] / 1e3,
            "slowest_share_ms": v[4] / 1e3, "shares_bound_to_numa_node": v[5]}


def bind_process_to_device_numa_node(device: int) -> int | None:
    """One process per GPU (bench.py's ranks, deltaq_amd.batch): run this process on the CPUs of the NUMA node the
    device hangs off, so that its staged host copies do not cross the socket link.  Returns the node, or None where the
    platform does not say / DQ_NUMA_BIND=0 / the node's CPUs are outside what the process may use (nothing is changed)."""
    if os.environ.get("DQ_NUMA_BIND") == "0":
        return None
    node = load().dq_device_numa_node(device)
    if node < 0:
        return None
    try:
        with open(f"/sys/devices/system/node/node{node}/cpulist") as f:
            spec = f.read().strip()
        cpus = set()
        for part in spec.split(","):
            if not part:
                continue
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        cpus &= os.sched_getaffinity(0)
        if not cpus:
            return None
        os.sched_setaffinity(0, cpus)
        return node
    except (OSError, ValueError):
        return None


def last_sort_info() -> dict:
    L = load()
    a, b, c = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
    L.dq_last_sort_info(ctypes.byref(a), ctypes.byref(b), ctypes.byref(c))
    return {"rounds": a.value, "initial_active": b.value, "sum_active": c.value}
