import sys, ctypes, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from foldseek_amd import api, synth
q3, qa = synth.make_queries(6, seed=5, mean_len=250, lo=20, hi=500)
rng = np.random.default_rng(77)
for i, L in enumerate((20, 64, 130, 260, 390, 512)):
    q3[i] = rng.choice(20, size=L).astype(np.uint8); qa[i] = rng.choice(20, size=L).astype(np.uint8)
db = synth.make_db(2500, (q3, qa), seed=7, homologs_per_query=40, mask_frac=0.02)
ctx=api.Context(0); ctx.load_db(db)
ctypes.CDLL('/root/repo/tools/segv_bt.so').segv_bt_install()
m = api.Matrix(0, 2.0)
for qi in range(6):
    pssm, cap = api.prefilter_profile(m, q3[qi], True, 0.15)
    ctx.gapless_scan(pssm, cap, min_score=30, identity=-1, max_res=300); ctx.gapless_scores()
for max_res in (1, 7, 50, 100000):
    ctx.gapless_scan(pssm, cap, min_score=30, identity=5, max_res=max_res)
for qi in range(6):
  for atype in (2,0):
    mAA=api.Matrix(1,1.4 if atype==2 else 0.0); m3=api.Matrix(0,2.1)
    pAf,p3f,_,_=api.align_profiles(mAA,m3,qa[qi],q3[qi]); pAr,p3r,_,_=api.align_profiles(mAA,m3,qa[qi][::-1].copy(),q3[qi][::-1].copy())
    r2 = np.random.default_rng(qi)
    ids = np.unique(np.concatenate([r2.integers(0, db.n, size=150), np.arange(db.n - 20, db.n), np.arange(0, 20)])).astype(np.uint32)
    print("query",qi,atype,len(ids),flush=True)
    f,r=ctx.sw_batch(pAf if atype==2 else None,p3f,pAr if atype==2 else None,p3r,ids)
    print(f[:2],flush=True)
