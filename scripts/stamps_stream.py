"""Diagnostic: where one leapfrog of the streaming sampler spends its time (stamped build:
EPX_STAMPS=1 ./ep-stan_amd/csrc/build.sh; run with EPX_LIB=<...>/libepx_stamps.so)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from epstan_amd import _lib
from epstan_amd.engine import HipEngine, QI
NAMES = ['A: transforms', 'Omega pass', 'row stream', 'D: chain rule', 'bookkeeping (other)', 'top barrier', 'bookkeeping (plain leaf)']
K = int(sys.argv[1]) if len(sys.argv) > 1 else 256
D = int(sys.argv[3]) if len(sys.argv) > 3 else 128
n = int(sys.argv[4]) if len(sys.argv) > 4 else 2000
layout = int(sys.argv[5]) if len(sys.argv) > 5 else 0
rng = np.random.RandomState(0)
X = rng.randn(K * n, D) * 0.3
y = (rng.rand(K * n) < 0.5).astype(int)
eng = HipEngine('m4b_sg', X, y, np.arange(K + 1) * n)
d = eng.d
eng.set_prior(np.eye(d), np.zeros(d)); eng.set_global(np.eye(d) * 2.0, np.zeros(d))
assert np.all(eng.cavity_batch(QI))
stats, ms = eng.sample_batch(np.arange(K) + 1, HipEngine.sampler_opts(chains=4, iter=int(sys.argv[2]) if len(sys.argv) > 2 else 8, init="random", layout=layout))
lib = _lib.load()
buf = np.zeros((4096, 8), dtype=np.uint64)
lib.epx_dbg_get_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
nb = lib.epx_dbg_get_stamps(eng.ctx, buf.ctypes.data, 4096)
st = buf[:min(nb, K)].astype(np.float64)        # (first record of every workgroup: the chain wave's shares)
st = st[st[:, 7] > 0]
per = st[:, :7] / st[:, 7:8]
med = np.median(per, axis=0)
tot_ticks = st[:, 7].max()
print('K=%d: %.1f ms, slowest block %d leapfrogs -> %.1f us each; s_memtime units per leapfrog (median over %d blocks): %.0f'
      % (K, ms, tot_ticks, ms * 1e3 / tot_ticks, nb, med.sum()))
for nm, v in zip(NAMES, med):
    print('    %-14s %8.0f  %5.1f%%' % (nm, v, 100 * v / med.sum()))
