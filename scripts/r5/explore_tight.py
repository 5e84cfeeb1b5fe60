import sys; sys.path[:0]=['/root/repo','/root/repo/tests','/root/repo/tests/golden']
import numpy as np
from test_gpu_round5 import _traces, _parting
for tight in (1000., 3000., 10000., 30000.):
    for seed in (7,):
        tr_d, tr_o, P = _traces('m4b_sg', 32, 500, 2, 200, 7, tight, seed)
        t_star, before, err = _parting(tr_d, tr_o)
        print('tight', tight, 't_star', t_star.ravel().tolist(), 'mean nleap', tr_o[...,1].mean(), 'max err', err.max(), flush=True)
        m = tr_o[0,0,:,6]; ch=np.nonzero(m[1:]!=m[:-1])[0]+1; print('  metric changes at', ch.tolist())
