"""Build recipe for libdq_sufsort_hip.so (hipcc, gfx950 only).

The shared library is built IN-TREE (deltaq_amd/libdq_sufsort_hip.so) so that it
travels with the repository snapshot; it is git-ignored.

Four translation units, compiled side by side and linked into one library:
    dq_sorter_i32.hip / dq_sorter_i64.hip   the suffix sorter and its kernels per index width
    dq_diff.hip                             match search, Diff.Create / Patch.Apply
    dq_abi.hip                              the C ABI and the batch pipeline (host code only)
Objects live in deltaq_amd/csrc/obj/ with the compiler's own dependency files, so an edit
rebuilds only the units that include what changed.
"""
from __future__ import annotations

import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "obj")
LIB_NAME = "libdq_sufsort_hip.so"
LIB_PATH = os.path.join(HERE, LIB_NAME)
SOURCES = ["dq_sorter_i32.hip", "dq_sorter_i64.hip", "dq_diff.hip", "dq_abi.hip"]
ARCH = "gfx950"
FLAGS = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-pthread"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the MI355X backend cannot be built")


def _obj(src: str) -> str:
    return os.path.join(OBJ, os.path.splitext(src)[0] + ".o")


def _deps(src: str) -> list[str]:
    """Files the object was compiled from, as recorded by the compiler (-MD); none recorded = unknown."""
    dep = _obj(src)[:-2] + ".d"
    if not os.path.exists(dep):
        return []
    text = open(dep).read().replace("\\\n", " ")
    root = os.path.dirname(HERE)
    out = []
    for p in text.split(":", 1)[-1].split():
        # the list holds absolute paths of the tree the object was compiled in; the tree may have been copied since
        # (the GPU box gets a snapshot under another path): re-root what belongs to the repository, drop system headers
        # (only what really maps onto a file of this tree: a toolchain installed elsewhere -- ~/rocm/include/hip/... --
        # also has "/include/" in its paths, and a re-rooted name that does not exist would make every object stale
        # on every import; anything else is a system header)
        for mark, base in (("/deltaq_amd/csrc/", CSRC), ("/include/", os.path.join(root, "include"))):
            k = p.rfind(mark)
            if k >= 0:
                mapped = os.path.join(base, p[k + len(mark):])
                if os.path.exists(mapped):
                    out.append(mapped)
                    break
    return out


def _obj_digest(src: str) -> str | None:
    """sha256 over the contents of every file the object was compiled from (the compiler's own list, -MD) and the flags;
    None when that list is missing or names a file that is gone."""
    import hashlib
    deps = _deps(src)
    if not deps:
        return None
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for p in sorted(set(deps + [os.path.join(CSRC, src)])):
        if not os.path.exists(p):
            return None
        h.update(os.path.relpath(p, os.path.dirname(HERE)).encode())
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _obj_stale(src: str) -> bool:
    """By content, like the library itself: the object is current iff the digest stamped beside it when it was compiled
    names its sources as they are now.  (File times order nothing in a copied tree: an object that travelled to the
    GPU box with an edited header would be relinked as it is and the manifest stamped over it.)"""
    obj = _obj(src)
    stamp = obj[:-2] + ".digest"
    if not os.path.exists(obj) or not os.path.exists(stamp):
        return True
    d = _obj_digest(src)
    try:
        return d is None or open(stamp).read().strip() != d
    except OSError:
        return True


MANIFEST = os.path.join(HERE, "libdq_sufsort_hip.manifest")


def _source_digest() -> str:
    """sha256 over every file the library is built from (names and contents)."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hip")))
    files.append(os.path.join(os.path.dirname(HERE), "include", "dq_sufsort.h"))
    for p in files:
        h.update(os.path.basename(p).encode())
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def is_stale() -> bool:
    """The library is current iff the manifest written beside it at build time names the sources as they are now.
    (By content, not by time stamps: the GPU box gets a copy of the tree whose file times say nothing.)"""
    if not os.path.exists(LIB_PATH) or not os.path.exists(MANIFEST):
        return True
    try:
        return open(MANIFEST).read().strip() != _source_digest()
    except OSError:
        return True


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 into deltaq_amd/libdq_sufsort_hip.so."""
    if not force and not is_stale():
        return LIB_PATH
    # one builder at a time (bench.py ranks, pytest-xdist workers): the others wait, then find it current
    import fcntl
    with open(os.path.join(HERE, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not is_stale():
            return LIB_PATH
        return _build_locked(force, verbose)


def _compile(src: str, verbose: bool) -> None:
    obj = _obj(src)
    cmd = [_hipcc(), *FLAGS, "-c", os.path.join(CSRC, src), "-MD", "-MF", obj[:-2] + ".d", "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    stamp = obj[:-2] + ".digest"
    if os.path.exists(stamp):
        os.remove(stamp)                    # (an interrupted compile leaves no stamp behind)
    subprocess.run(cmd, check=True, cwd=CSRC)
    d = _obj_digest(src)
    if d is not None:
        with open(stamp + ".tmp", "w") as f:
            f.write(d + "\n")
        os.replace(stamp + ".tmp", stamp)


def _build_locked(force: bool, verbose: bool) -> str:
    os.makedirs(OBJ, exist_ok=True)
    todo = [s for s in SOURCES if force or _obj_stale(s)]
    with ThreadPoolExecutor(max_workers=max(1, min(len(todo), os.cpu_count() or 1))) as pool:
        list(pool.map(lambda s: _compile(s, verbose), todo))
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-pthread", *[_obj(s) for s in SOURCES], "-o", LIB_PATH + ".tmp"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=CSRC)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    with open(MANIFEST + ".tmp", "w") as f:
        f.write(_source_digest() + "\n")
    os.replace(MANIFEST + ".tmp", MANIFEST)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
