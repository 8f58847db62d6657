import sys, time, torch, os
sys.path.insert(0, '.')
from tests.parity_utils import run_parity_case
for kw in [dict(n_env=2,img=64,seed=2,mesh='synthetic'), dict(n_env=2,img=64,seed=5,mesh='synthetic', az_range=3.0), dict(n_env=2,img=128,seed=6,mesh='mixed', az_range=3.0)]:
    t=time.time()
    try:
        r=run_parity_case(**kw)
        print(os.environ.get('OCC_HIP_LIB'), kw, {k: float('%.3g'%v) for k,v in r.items()}, 'time %.1f'%(time.time()-t), flush=True)
    except Exception as e:
        import traceback; traceback.print_exc()
        print(kw, 'FAILED', e, flush=True)
