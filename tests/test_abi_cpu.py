"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol the
header declares, validates arguments, and the host mirror keeps the reference's error
behaviour.  No compute is attempted without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def header_symbols():
    text = open(os.path.join(ROOT, "include", "dq_sufsort.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dq_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(backend_lib):
    from deltaq_amd import _abi
    syms = header_symbols()
    assert set(syms) == set(_abi.EXPORTS)
    for s in syms:
        assert hasattr(backend_lib, s), f"{s} declared in include/dq_sufsort.h but not exported"


def test_abi_version(backend_lib):
    assert backend_lib.dq_abi_version() == 1
    assert backend_lib.dq_device_count() >= 0


def test_workspace_query(backend_lib):
    n = 1 << 20
    b4 = backend_lib.dq_sufsort_hip_workspace_bytes(n, 4)
    b8 = backend_lib.dq_sufsort_hip_workspace_bytes(n, 8)
    assert 40 * n <= b4 < 41 * n + (16 << 20)      # 28 B/byte + the third list buffer (12 B/byte, round 5) + fixed tables
    assert 52 * n <= b8 < 53 * n + (16 << 20)      # 36 + 16
    assert backend_lib.dq_sufsort_hip_workspace_bytes(n, 3) == -1
    assert backend_lib.dq_sufsort_hip_workspace_bytes(-5, 4) == -1


def test_argument_validation_precedes_device_use(backend_lib):
    from deltaq_amd import _abi
    buf = np.zeros(8, np.uint8)
    sa = np.zeros(8, np.int32)
    assert backend_lib.dq_sufsort_hip_i32(buf.ctypes.data, -1, sa.ctypes.data, 0) == _abi.DQ_ERR_BAD_ARGS
    assert backend_lib.dq_sufsort_hip_i32(None, 8, sa.ctypes.data, 0) == _abi.DQ_ERR_BAD_ARGS
    assert backend_lib.dq_sufsort_hip_i32(buf.ctypes.data, 8, None, 0) == _abi.DQ_ERR_BAD_ARGS
    assert backend_lib.dq_sufsort_hip_i32(buf.ctypes.data, 1 << 31, sa.ctypes.data, 0) == _abi.DQ_ERR_TOO_LARGE
    assert b"2^31" in backend_lib.dq_last_error()
    assert backend_lib.dq_sufsort_hip_batch_i32(-1, None, None, None, 1, None) == _abi.DQ_ERR_BAD_ARGS
    assert backend_lib.dq_sufsort_hip_batch_i32(0, None, None, None, 1, None) == _abi.DQ_OK


def test_no_cpu_fallback_without_a_device(backend_lib):
    """On a machine without a GPU the product must fail loudly, never compute on the CPU."""
    from deltaq_amd import HipSuffixSort, SuffixSortError, _abi
    if backend_lib.dq_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(SuffixSortError) as ei:
        HipSuffixSort().Sort(b"banana")
    assert ei.value.code == _abi.DQ_ERR_NO_DEVICE


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from deltaq_amd import _abi
    monkeypatch.setattr(_abi, "_lib", None)
    monkeypatch.setenv("DQ_SUFSORT_LIB", str(tmp_path / "nope.so"))
    with pytest.raises(_abi.BackendMissingError):
        _abi.load()


def test_length_mismatch_matches_reference_message(backend_lib):
    # LibDivSufSort.cs:23-31: ArgumentException("Text and suffix buffers should have the same length")
    from deltaq_amd import HipSuffixSort
    with pytest.raises(ValueError, match="Text and suffix buffers should have the same length"):
        HipSuffixSort().Sort(b"abcdef", np.zeros(5, np.int32))
    with pytest.raises(TypeError):
        HipSuffixSort().Sort(b"abcdef", np.zeros(6, np.float32))


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under deltaq_amd/ may reference it."""
    pkg = os.path.join(ROOT, "deltaq_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert "import oracle" not in src and "from oracle" not in src, f
                assert "dq_oracle" not in src, f


# ---- the never-compiled C# shim (no dotnet in any image): its [DllImport] signatures against the header
C_WIDTH = {"int32_t": "i32", "int64_t": "i64", "void": "void", "double": "f64"}
CS_WIDTH = {"int": "i32", "long": "i64", "void": "void", "double": "f64", "uint": "i32", "ulong": "i64"}


def c_kind(decl: str) -> str:
    """'i32' / 'i64' for integers by value, 'ptr' for any pointer, 'void'."""
    decl = decl.replace("const", " ").strip()
    if "*" in decl:
        return "ptr"
    base = decl.split()[0]
    assert base in C_WIDTH, decl
    return C_WIDTH[base]


def cs_kind(decl: str) -> str:
    decl = decl.strip()
    if "*" in decl or decl.split()[0] in ("IntPtr", "UIntPtr") or decl.startswith(("ref ", "out ")):
        return "ptr"
    base = decl.split()[0]
    assert base in CS_WIDTH, decl
    return CS_WIDTH[base]


def header_signatures():
    text = open(os.path.join(ROOT, "include", "dq_sufsort.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    sigs = {}
    for ret, name, args in re.findall(r"^\s*([A-Za-z_][\w \*]*?)\s*\b(dq_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", text, flags=re.M):
        params = [] if args.strip() in ("", "void") else [c_kind(a) for a in args.split(",")]
        sigs[name] = (c_kind(ret), params)
    return sigs


def csharp_signatures():
    sigs = {}
    base = os.path.join(ROOT, "bindings", "csharp")
    for dirpath, _, files in os.walk(base):
        for f in files:
            if not f.endswith(".cs"):
                continue
            text = re.sub(r"//[^\n]*", "", open(os.path.join(dirpath, f)).read())
            for attrs, ret, name, args in re.findall(
                    r"\[DllImport\(([^\]]*)\)\]\s*(?:(?:internal|private|public|static|extern|unsafe)\s+)+([\w\*]+)\s+(dq_\w+)\s*\(([^)]*)\)\s*;",
                    text, flags=re.S):
                assert "CallingConvention.Cdecl" in attrs, f"{f}: {name} must be Cdecl"
                assert "EntryPoint" not in attrs, f"{f}: {name} renames its entry point"
                params = [] if not args.strip() else [cs_kind(a) for a in args.split(",")]
                sigs.setdefault(name, []).append((f, cs_kind(ret), params))
    return sigs


def test_csharp_dllimports_match_the_header():
    """Every [DllImport] of bindings/csharp/**/*.cs names an export of include/dq_sufsort.h with the same arity, the same
    integer widths by value (C# int = int32_t, long = int64_t) and pointers where the header has pointers."""
    hdr, cs = header_signatures(), csharp_signatures()
    assert len(hdr) == len(header_symbols())            # the signature parser sees every declaration
    assert cs, "no [DllImport] found under bindings/csharp"
    assert {"dq_sufsort_hip_i32", "dq_bsdiff_create", "dq_bspatch_apply", "dq_bsdiff_search_i32", "dq_bsdiff_index_create",
            "dq_bsdiff_index_diff", "dq_bsdiff_index_free"} <= set(cs)
    for name, uses in cs.items():
        assert name in hdr, f"{name}: imported by {uses[0][0]} but not declared in include/dq_sufsort.h"
        ret, params = hdr[name]
        for f, cret, cparams in uses:
            # `const char *` comes back as IntPtr
            assert cret == ret, f"{f}: {name} returns {cret}, header says {ret}"
            assert len(cparams) == len(params), f"{f}: {name} takes {len(cparams)} arguments, header says {len(params)}"
            assert cparams == params, f"{f}: {name} argument kinds {cparams} differ from the header's {params}"
