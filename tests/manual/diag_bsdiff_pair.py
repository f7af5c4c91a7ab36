import os, sys
os.environ.setdefault("DQ_DEBUG_FLAGS", "1")      # the library honours its DQ_* overrides only under this gate
import numpy as np
ROOT='/root/repo'
sys.path.insert(0, ROOT)
import oracle
from deltaq_amd import Diff, HipMatchSearch
old=np.load(os.path.join(ROOT,'tests/golden/regress/bsdiff_373_7459_old.npy')); new=np.load(os.path.join(ROOT,'tests/golden/regress/bsdiff_373_7459_new.npy'))
sa=oracle.divsufsort(old)
wc,wd,we,ns=oracle.bsdiff_scan(old,sa,new)
print('n',old.size,'m',new.size,'oracle triples',wc.shape, 'searches', ns)
for env in ({}, {'DQ_NO_RESUME':'1'}, {'DQ_WALK_ON':'0'}, {'DQ_NO_SECOND_STAGE':'1'}, {'DQ_NO_WAVE_WINDOWS':'1'}, {'DQ_NO_POLL':'1'}):
    for k,v in env.items(): os.environ[k]=v
    c,d,e,st=Diff.Scan(old,new)
    same=np.array_equal(c,wc) and np.array_equal(d,wd) and np.array_equal(e,we)
    first=None
    if not same:
        k=min(len(c),len(wc))
        neq=np.flatnonzero((c[:k]!=wc[:k]).any(axis=1))
        first=int(neq[0]) if neq.size else k
    print(env, 'same' if same else f'DIFFERENT at triple {first}: got {c[first].tolist() if first is not None and first<len(c) else None} want {wc[first].tolist() if first is not None and first<len(wc) else None}', st)
    for k in env: del os.environ[k]
# the first anchor: what does the reference's Search say around the position where the loop breaks?
c,d,e,st=Diff.Scan(old,new)
print('got triples[:3]', c[:3].tolist(), 'want', wc[:3].tolist())
# every position: exact device search vs the oracle, with and without the prefix table
for tab in (None, '2'):
    if tab: os.environ['DQ_SEARCH_PTAB']=tab
    ms0=HipMatchSearch(0)
    pos0,ln0=ms0.Search(sa, old, new, scan0=0, count=new.size)
    wp0,wl0=oracle.bsdiff_search(old, sa, new, scan0=0, count=new.size)
    bad0=np.flatnonzero((pos0!=wp0)|(ln0!=wl0))
    print('lane kernel, ptab', tab, 'mismatches:', bad0.size, bad0[:10].tolist())
    os.environ['DQ_SEARCH_WAVE']='1'
    tot=0
    for s0 in range(0,new.size,4096):
        cnt=min(4096,new.size-s0)
        p2,l2=ms0.Search(sa, old, new, scan0=s0, count=cnt, cap=0)
        b=np.flatnonzero((p2!=wp0[s0:s0+cnt])|(l2!=wl0[s0:s0+cnt]))
        if b.size:
            tot+=b.size
            if tot<=12:
                for i in b[:4]:
                    q=s0+int(i); print('  wave exact mismatch at scan', q, 'got', int(p2[i]), int(l2[i]), 'want', int(wp0[q]), int(wl0[q]), 'new bytes', new[q:q+6].tolist())
    print('wave kernel exact, ptab', tab, 'mismatches:', tot)
    del os.environ['DQ_SEARCH_WAVE']
    if tab: del os.environ['DQ_SEARCH_PTAB']
# every position: exact device search vs the oracle
ms=HipMatchSearch(0)
pos,ln=ms.Search(sa, old, new, scan0=0, count=new.size)
wp,wl=oracle.bsdiff_search(old, sa, new, scan0=0, count=new.size)
bad=np.flatnonzero((pos!=wp)|(ln!=wl))
print('exact lane-kernel search mismatches:', bad.size, bad[:10].tolist())
os.environ['DQ_SEARCH_WAVE']='1'
tot=0
for s0 in range(0,new.size,4096):
    cnt=min(4096,new.size-s0)
    p2,l2=ms.Search(sa, old, new, scan0=s0, count=cnt, cap=64)
    m=(l2>=0)
    b=np.flatnonzero(m & ((p2!=wp[s0:s0+cnt])|(l2!=wl[s0:s0+cnt])))
    if b.size:
        tot+=b.size
        if tot<=20: print('wave kernel mismatch at', (s0+b[:5]).tolist(), 'got', p2[b[:5]].tolist(), l2[b[:5]].tolist(), 'want', wp[s0+b[:5]].tolist(), wl[s0+b[:5]].tolist())
print('wave kernel (cap 64) mismatching decided answers:', tot)
