#!/usr/bin/env python3
"""Where does DQ_FAULT=hip:K fire?  (diagnostic for tests/test_gpu_faults.py)"""
import os, sys
os.environ.setdefault("DQ_DEBUG_FLAGS", "1")
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle
from deltaq_amd import HipSuffixSort, Diff, SuffixSortError, _abi

s = HipSuffixSort(0)
T = oracle.gen_uniform(8 << 20, 3)
s.Sort(T)


def probe(tag):
    for k in (1, 2, 5, 9, 20, 40, 80):
        os.environ["DQ_FAULT"] = f"hip:{k}"
        try:
            s.Sort(T)
            print(tag, k, "no error", flush=True)
        except SuffixSortError as e:
            print(tag, k, "->", e, flush=True)
        del os.environ["DQ_FAULT"]


probe("before")
old = oracle.gen_uniform(1 << 20, 4)
new = old.copy(); new[5000:5010] = 7
Diff.CreateBytes(old, new, 0)
probe("after diff")
_abi.load().dq_sufsort_hip_release()
probe("after release")
