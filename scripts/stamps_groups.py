"""Diagnostic (stamped build): where a leapfrog of the grouped one-workgroup-per-chain kernel goes,
at the reference's default experiment shape (J = 64 groups on K = 32 sites, D = 16, 20 rows per group)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from epstan_amd import models, _lib
from epstan_amd.engine import HipEngine
from epstan_amd.method import Master
from epstan_amd.util import distribute_groups
J, K, D, npg = 64, 32, 16, 20
mod = models.m4b(J, D, npg)
data = mod.simulate_data(Sigma_x='rand', rng=100)
_, _, Q0, r0 = mod.get_prior()
Nk, Nj_k, j_ind_k = distribute_groups(J, K, data.Nj)
M = Master('m4b', data.X, data.y, site_sizes=Nk, A_k={'J': Nj_k}, A_n={'j_ind': j_ind_k + 1},
           prior={'Q': Q0, 'r': r0}, chains=4, iter=60)
stats, ms = M.engine.sample_batch(np.arange(K) + 1, HipEngine.sampler_opts(chains=4, iter=60, init='random', layout=2))
lib = _lib.load()
buf = np.zeros((4096, 8), dtype=np.uint64)
lib.epx_dbg_get_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
nb = lib.epx_dbg_get_stamps(M.engine.ctx, buf.ctypes.data, 4096)
st = buf[:nb].astype(np.float64)
per = np.median(st[:, :7] / st[:, 7:8], axis=0)
names = ['exp + shared scalars', 'groups: prep, rows, butterfly, records', '-', 'Omega share', 'exchange', 'chain rule', 'bookkeeping wave busy (concurrent)']
print('layout', M.engine.last_layout(), '%.1f ms' % ms, 'units per leapfrog', per[:6].sum())
for n, v in zip(names, per):
    print('   %-45s %8.0f %5.1f%%' % (n, v, 100 * v / per[:6].sum()))
