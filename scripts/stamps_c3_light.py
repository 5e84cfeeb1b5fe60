"""Diagnostic: the row team's pass (layout 7) in its own kernel with FOUR stamps per pass and wave instead of a stamp per
phase (scripts/stamps_c3.py: every stamp drains the wave's LDS reads, which serialises the pass it measures).
Per pass and row wave: waiting for the jobs (barrier 1) | the pass | waiting for "the results are in" (barrier 2) | work behind it;
per leapfrog and state wave: the five slots of stamps_c3.py.
Needs the light stamped build (csrc/nuts_duo.hip compiled with -DEPX_STAMPS: EPX_STAMPS_LIGHT is on by default there):
EPX_LIB=variants/libepx_stamps.so python3 scripts/stamps_c3_light.py [sites] [ep_iters]"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from epstan_amd import models, _lib
from epstan_amd.method import Master


def main():
    J = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    nit = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    D, n = 32, 500
    mod = models.m4b(J, D, n)
    data = mod.simulate_data(Sigma_x='rand', rng=100)
    _, _, Q0, r0 = mod.get_prior()
    M = Master(mod.site_model, data.X, data.y, site_sizes=data.Nj, prior={'Q': Q0, 'r': r0},
               chains=4, iter=200, df0=models.default_df0(J), layout=7, sync_sites=False)
    info = M.run(nit, verbose=False, seed=1)[0]
    eng = M.engine
    lib = _lib.load()
    buf = np.zeros((8192, 8), dtype=np.uint64)
    lib.epx_dbg_get_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    nb = lib.epx_dbg_get_stamps(eng.ctx, buf.ctypes.data, 8192)
    st = buf[:nb].astype(np.float64)
    nw = (nb - 8) // 3
    first, second, third = st[:nw], st[nw:2 * nw], st[2 * nw:3 * nw]
    lf = np.maximum(first[:, 7], 1.0)[:, None]
    S = np.median(first[:, :5] / lf, axis=0)
    names = ['S: loop top .. job out (first half / publish)', 'S: tree bookkeeping', 'S: results in -> job out (view update)',
             'S: waiting for the results (barrier 2)', 'S: relay / chain rule + finish']
    print('J=%d layout %d info %d; sampling launches (ms) %s' % (J, eng.last_layout(), info, np.round(M.sampling_ms, 1)))
    print('state wave of chain 0, cycles per leapfrog (median over %d workgroups), total %.0f' % (nw, S.sum()))
    for nm, v in zip(names, S):
        print('    %-48s %7.0f  %5.1f%%' % (nm, v, 100 * v / S.sum()))
    w1 = np.median(third[:, :4] / lf, axis=0); wk = np.median(third[:, 4:] / lf, axis=0)
    w2 = np.median(second[:, :4] / lf, axis=0); po = np.median(second[:, 4:] / lf, axis=0)
    print('row waves 0..3, cycles per pass (median):')
    print('    waiting for the jobs (barrier 1)  %s' % np.round(w1))
    print('    the pass                          %s' % np.round(wk))
    print('    waiting at barrier 2              %s' % np.round(w2))
    print('    behind barrier 2                  %s' % np.round(po))
    print('    sum                               %s' % np.round(w1 + wk + w2 + po))
    def hist(rec):
        raw = buf[3 * nw + 2 + rec].astype(np.uint64)
        h = np.zeros(16)
        h[0::2] = (raw & np.uint64(0xFFFFFFFF)).astype(np.float64); h[1::2] = (raw >> np.uint64(32)).astype(np.float64)
        return h / max(h.sum(), 1.0)
    for rec, nm in ((0, 'state wave 0: tree bookkeeping of a leapfrog'), (1, 'state wave 0: its wait at barrier 2'), (2, 'row wave 0: its wait at barrier 2')):
        h = hist(rec)
        print('%s, bins of 1024 cycles (last: >= 15360), share of the passes:\n    %s\n    share of the TIME: %s'
              % (nm, np.round(h, 3), np.round((np.arange(16) + 0.5) * h / max(((np.arange(16) + 0.5) * h).sum(), 1e-30), 3)))
    gl = eng.get_chain_stats(4)[:, :, 3]
    print('us per lock-step pass (launch / max passes of a site): %.3f' % (M.sampling_ms[-1] * 1e3 / gl.max()))


if __name__ == '__main__':
    main()
