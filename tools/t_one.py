"""One text-like sort of the given size under the profiler (kernel timeline of the last sort)."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deltaq_amd import HipSuffixSort
from tools import datagen
n = int(sys.argv[1])
T = datagen.gen_enwik_like(n, 0xD17A0, min(n // 4, 64 * 1024))
s = HipSuffixSort(0); sa = np.empty(n, np.int32)
for _ in range(3): s.Sort(T, sa)
