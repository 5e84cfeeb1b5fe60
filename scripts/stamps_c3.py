"""Diagnostic: cycle shares of one leapfrog at the C3 site size (D=32, n_j=500, m4b) in the
steady state of EP (tight cavities, deep trees).  Needs the stamped build:
EPX_LIB=variants/libepx_stamps.so python3 scripts/stamps_c3.py [sites] [ep_iters] [layout] [D] [n]"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from epstan_amd import models, _lib
from epstan_amd.method import Master

NAMES = ['prep(beta)', 'row loop', 'butterfly', 'Omega matvec', 'exchange', 'chain rule', 'state machine']
NAMES_DUO = ['S: kick/drift/transforms/publish', 'S: tree bookkeeping', 'S: cavity term + gathers', 'S: waiting for the row waves',
             'S: chain rule + finish', 'R: waiting for a job', 'R: row pass + butterfly + publish']

def main():
    J = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    nit = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    layout = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    D = int(sys.argv[4]) if len(sys.argv) > 4 else 32
    n = int(sys.argv[5]) if len(sys.argv) > 5 else 500
    mod = models.m4b(J, D, n)
    data = mod.simulate_data(Sigma_x='rand', rng=100)
    _, _, Q0, r0 = mod.get_prior()
    M = Master(mod.site_model, data.X, data.y, site_sizes=data.Nj, prior={'Q': Q0, 'r': r0},
               chains=4, iter=200, df0=models.default_df0(J), layout=layout, sync_sites=False)
    info = M.run(nit, verbose=False, seed=1)[0]
    eng = M.engine
    lib = _lib.load()
    buf = np.zeros((8192, 8), dtype=np.uint64)
    lib.epx_dbg_get_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    nb = lib.epx_dbg_get_stamps(eng.ctx, buf.ctypes.data, 8192)
    st = buf[:nb].astype(np.float64)
    det = None
    waves = None
    hist = None
    if (nb - 2) % 3 == 0 and eng.last_layout() == 7:    # second third: the phases of the row team's pass; third: per row wave; then a histogram
        hist = st[nb - 2:].reshape(-1)
        nb = (nb - 2) // 3
        det = st[nb:2 * nb]; waves = st[2 * nb:3 * nb]; st = st[:nb]
    per = st[:, :7] / st[:, 7:8]
    med = np.median(per, axis=0)
    lf = eng.get_chain_stats(4)[:, :, 3]
    print('site size D=%d n=%d, J=%d,' % (D, n, J)); print('J=%d, layout %d, info %d, EP iteration %d: sampling launches (ms) %s'
          % (J, eng.last_layout(), info, nit, np.round(M.sampling_ms, 1)))
    print('leapfrogs per chain: mean %.0f max %.0f; per transition %.0f' % (lf.mean(), lf.max(), lf.mean() / 200))
    print('cycles per leapfrog (median over %d blocks), total %.0f' % (nb, med.sum()))
    for nm, v in zip(NAMES_DUO if eng.last_layout() >= 5 else NAMES, med):
        print('    %-36s %8.0f  %5.1f%%' % (nm, v, 100 * v / med.sum()))
    if det is not None:
        ok = det[:, 7] > 0
        pd = np.median(det[ok, :7] / det[ok, 7:8], axis=0)
        print('row team, cycles per pass (median over %d blocks), total %.0f' % (ok.sum(), pd.sum()))
        for nm, v in zip(['waiting for the jobs', 'operands + cavity term', 'forward products', 'LDS requests', 'logistic terms',
                          'backward products', 'sums + publication'], pd):
            print('    %-36s %8.0f  %5.1f%%' % (nm, v, 100 * v / pd.sum()))
    if waves is not None:
        ok = det[:, 7] > 0
        w = waves[ok] / det[ok, 7:8]
        print('row waves 0..3, cycles per pass (median): waiting %s, working %s' % (np.round(np.median(w[:, :4], axis=0)), np.round(np.median(w[:, 4:], axis=0))))
    if hist is not None:
        print('row wave 0: waits for the jobs by length, bins of 512 cycles (last: >= 7680), share of the passes: %s' % np.round(hist / max(hist.sum(), 1), 3))
        print('   ... share of the waiting TIME (bin centre x count): %s' % np.round((np.arange(16) + 0.5) * hist / max(((np.arange(16) + 0.5) * hist).sum(), 1), 3))
    big = int(np.argmax(st[:, 7]))
    print('the workgroup with the most leapfrogs (%d): cycles per leapfrog %s, total S %.0f, total R %.0f'
          % (st[big, 7], np.round(per[big]).astype(int), per[big, :5].sum(), per[big, 5:7].sum()))
    wg = lf.max(axis=1)
    print('us per leapfrog of the slowest chain of a workgroup ~ %.2f' % (M.sampling_ms[-1] * 1e3 / wg.max()))

if __name__ == '__main__':
    main()
