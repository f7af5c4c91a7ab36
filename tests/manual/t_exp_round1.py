#!/usr/bin/env python3
"""Timing experiment: round 0 + the FIRST doubling round only (DQ_EXP_STOP_ROUNDS=1; the suffix array is left unfinished),
per-kernel profile -- run once per library variant (tools/exp/build_variant.sh; DQ_SUFSORT_LIB selects it):
    DQ_SUFSORT_LIB=tools/exp/libdq_nowalk.so python tests/manual/t_exp_round1.py enwik256"""
import os, sys
os.environ.setdefault("DQ_DEBUG_FLAGS", "1")      # the library honours its DQ_* overrides only under this gate
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "manual"))
os.environ["DQ_EXP_STOP_ROUNDS"] = os.environ.get("DQ_EXP_STOP_ROUNDS", "1")
import numpy as np
import torch
from deltaq_amd import HipSuffixSort, _abi
import importlib.util
spec = importlib.util.spec_from_file_location("t_case_mod", os.path.join(ROOT, "tests", "manual", "t_case.py"))
src = open(spec.origin).read()
ns = {}
exec(src[src.index("def load_case"):src.index("L = _abi.load()")], {"np": np, "os": os, "glob": __import__("glob"),
                                                                       "datagen": __import__("tools.datagen", fromlist=["x"])}, ns)
T = np.ascontiguousarray(ns["load_case"](sys.argv[1])); n = T.size
L = _abi.load(); s = HipSuffixSort(0)
dT = torch.from_numpy(T).cuda(); out = torch.empty(n, dtype=torch.int32, device="cuda")
for _ in range(2): s.Sort(dT, out)
torch.cuda.synchronize()
L.dq_profile_enable(1); L.dq_profile_reset()
for _ in range(3): s.Sort(dT, out)
torch.cuda.synchronize(); L.dq_profile_enable(0)
print(f"== {sys.argv[1]} lib={os.environ.get('DQ_SUFSORT_LIB', 'default')} stop after {os.environ['DQ_EXP_STOP_ROUNDS']} round(s)")
for k, p in _abi.profile_snapshot().items():
    if p["launches"]:
        print(f"   {k:26s} launches/sort={p['launches']/3:5.1f} ms/sort={p['ms']/3:8.3f}", flush=True)
