import sys, time, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deltaq_amd import HipSuffixSort, _abi, workload
L=_abi.load(); s=HipSuffixSort(0)
n=64<<20
T=torch.from_numpy(workload.gen_uniform(n,0x5EED0002)).cuda(); out=torch.empty(n,dtype=torch.int32,device='cuda')
try:
    s.Sort(T,out)
except Exception as e: print('err', e)
L.dq_profile_enable(1); L.dq_profile_reset()
for _ in range(3):
    try: s.Sort(T,out)
    except Exception as e: pass
torch.cuda.synchronize()
for k,v in _abi.profile_snapshot().items():
    if v['launches']: print(k, v['launches'], round(v['ms']/v['launches']*1e3,1),'us')
