import os, sys, numpy as np
os.environ.setdefault("DQ_DEBUG_FLAGS", "1")      # the library honours its DQ_* overrides only under this gate
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle
from deltaq_amd import HipSuffixSort, _abi
from structured_inputs import structured_text
env_len, target = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(0xD17A + env_len)
sizes = [int(x) for x in rng.integers(1, 2000, 120)] + [int(x) for x in rng.integers(2000, 20000, 120)] + \
        [int(x) for x in rng.integers(20000, 100000, 100)] + [int(x) for x in rng.integers(100000, 300000, 50)] + \
        [int(x) for x in rng.integers(300000, 2000000, 10)]
for i, n in enumerate(sizes):
    T = structured_text(rng, n)
    if i == target:
        break
print("n", T.size, "distinct", len(np.unique(T)))
ref = oracle.divsufsort(T)
s = HipSuffixSort(0)
for env in ({}, {"DQ_NO_BINNED_ISA": "1"}):
    for k, v in env.items(): os.environ[k] = v
    sa = s.Sort(T)
    bad = np.nonzero(sa != ref)[0]
    print(env, "mismatches", bad.size, "first", bad[:5], _abi.last_sort_info())
    if bad.size:
        p = bad[0]; print(" sa", sa[p-1:p+3], "ref", ref[p-1:p+3], "perm?", np.array_equal(np.sort(sa), np.arange(T.size)))
