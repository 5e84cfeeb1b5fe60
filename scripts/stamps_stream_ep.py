"""Diagnostic: where one leapfrog of the streaming sampler goes in the regime the C5-shard bench times (real m4b
sites, EP iteration >= 3: deep trees), not in a synthetic first launch (scripts/stamps_stream.py).  Stamped build:
EPX_LIB=variants/libepx_stamps.so python3 scripts/stamps_stream_ep.py [sites] [ep_iters] [D] [n]"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from epstan_amd import models, _lib
from epstan_amd.method import Master

NAMES = ['A: transforms', 'Omega pass', 'row stream', 'D: chain rule', 'bookkeeping (other)', 'top barrier', 'bookkeeping (plain leaf)']


def main():
    J = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    nit = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    D = int(sys.argv[3]) if len(sys.argv) > 3 else 128
    n = int(sys.argv[4]) if len(sys.argv) > 4 else 2000
    mod = models.m4b(J, D, n)
    data = mod.simulate_data(rng=100)
    _, _, Q0, r0 = mod.get_prior()
    M = Master(mod.site_model, data.X, data.y, site_sizes=data.Nj, prior={'Q': Q0, 'r': r0},
               chains=4, iter=200, df0=models.default_df0(J), sync_sites=False)
    info = M.run(nit, verbose=False, seed=1)[0]
    eng = M.engine
    lib = _lib.load()
    buf = np.zeros((4096, 8), dtype=np.uint64)
    lib.epx_dbg_get_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    nb = lib.epx_dbg_get_stamps(eng.ctx, buf.ctypes.data, 4096)
    st = buf[:min(nb, J)].astype(np.float64)
    st = st[st[:, 7] > 0]
    per = st[:, :7] / st[:, 7:8]
    med = np.median(per, axis=0)
    lf = eng.get_chain_stats(4)[:, :, 3]
    passes = lf.max(axis=1)
    print('J=%d D=%d n=%d, layout %d, info %d, EP iteration %d: sampling launches (ms) %s' % (J, D, n, eng.last_layout(), info, nit, np.round(M.sampling_ms, 1)))
    print('leapfrogs per transition %.0f; passes per site: mean %.0f max %.0f; launch / max passes = %.1f us per pass of the slowest site'
          % (lf.mean() / 200, passes.mean(), passes.max(), M.sampling_ms[-1] * 1e3 / passes.max()))
    print('s_memtime units per leapfrog (median over %d workgroups): %.0f' % (st.shape[0], med.sum()))
    for nm, v in zip(NAMES, med):
        print('    %-26s %8.0f  %5.1f%%' % (nm, v, 100 * v / med.sum()))


if __name__ == '__main__':
    main()
