"""Diagnostic: where one leapfrog of the sampler kernel spends its cycles.
Needs the stamped build: EPX_STAMPS=1 ./ep-stan_amd/csrc/build.sh, run with
EPX_LIB=<...>/libepx_stamps.so (built on the GPU box into gpurun_out/)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from epstan_amd import models, _lib
from epstan_amd.engine import HipEngine
from epstan_amd.method import Master

NAMES = ['prep(beta)', 'row loop', 'butterfly', 'Omega matvec', 'exchange', 'chain rule',
         'state machine (layout 2: busy time of the bookkeeping wave, concurrent)']

def run(name, J, D, n, layout, it=30):
    mod = models.MODELS[name](J, D, n)
    data = mod.simulate_data(Sigma_x='rand', rng=100)
    _, _, Q0, r0 = mod.get_prior()
    M = Master(mod.site_model, data.X, data.y, site_sizes=data.Nj, prior={'Q': Q0, 'r': r0}, chains=4, iter=it)
    eng = M.engine
    opts = HipEngine.sampler_opts(chains=4, iter=it, init='random', layout=layout)
    stats, ms = eng.sample_batch(np.arange(J) + 1, opts)
    lib = _lib.load()
    buf = np.zeros((4096, 8), dtype=np.uint64)
    lib.epx_dbg_get_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    nb = lib.epx_dbg_get_stamps(eng.ctx, buf.ctypes.data, 4096)
    st = buf[:nb].astype(np.float64)
    per = st[:, :7] / st[:, 7:8]
    med = np.median(per, axis=0)
    print('%s J=%d D=%d n=%d layout=%d: %.1f ms; cycles per leapfrog (median over %d blocks), total %.0f'
          % (name, J, D, n, layout, ms, nb, med.sum()))
    for nm, v in zip(NAMES, med):
        print('    %-14s %8.0f  %5.1f%%' % (nm, v, 100 * v / med.sum()))

if __name__ == '__main__':
    run('m4b', 64, 16, 200, 2)
    run('m4b', 64, 16, 200, 1)
    run('m4b', 256, 32, 500, 1, it=16)
