"""Bring-up check of the pieced launch (epx_set_piece_queue): cut sites give the draws of the uncut run."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from epstan_amd import models
from epstan_amd.engine import HipEngine
from epstan_amd.method import Master

def run(J, D, n, it, pieces):
    mod = models.m4b(J, D, n)
    data = mod.simulate_data(Sigma_x='rand', rng=100)
    _, _, Q0, r0 = mod.get_prior()
    M = Master(mod.site_model, data.X, data.y, site_sizes=data.Nj, prior={'Q': Q0, 'r': r0}, chains=4, iter=it)
    eng = M.engine
    seeds = np.arange(J) + 11
    opts = HipEngine.sampler_opts(chains=4, iter=it, init='random', layout=5)
    out = []
    for p in (None, 'queue', None, 'queue'):
        eng.set_piece_queue(max(1, it // 5) if p == 'queue' else 0, None)
        stats, ms = eng.sample_batch(seeds, opts)
        dr = np.stack([eng.get_draws(k, all_params=True) for k in range(J)])
        out.append((dr, eng.get_chain_stats(4).copy(), stats.copy(), ms, eng.last_segments(), eng.last_layout()))
    a, b = out[0], out[1]
    print('J=%d D=%d n=%d it=%d: plain %.1f ms (layout %d), from the piece queue %.1f ms (%d): draws identical %s, chain stats identical %s, site stats identical %s; third run plain again identical %s'
          % (J, D, n, it, a[3], a[5], b[3], b[4], np.array_equal(a[0], b[0]), np.array_equal(a[1], b[1]), np.array_equal(a[2], b[2]),
             np.array_equal(out[2][0], a[0])))
    q = out[3]
    print('   piece queue: %.1f ms, last_segments %d: draws identical %s, chain stats identical %s' % (q[3], q[4], np.array_equal(q[0], a[0]), np.array_equal(q[1], a[1])))
    eng.set_piece_queue(0)
    if not np.array_equal(a[1], b[1]):
        bad = np.argwhere(a[1] != b[1])
        print('   first differing chain stats', bad[:5], a[1][tuple(bad[0])], b[1][tuple(bad[0])])

if __name__ == '__main__':
    it = 40
    pieces = [[(0, 0, it)],
              [(1, 0, 15), (2, 0, it)],
              [(3, 0, 20), (1, 15, it)],
              [(4, 0, 7), (3, 20, it)],
              [(5, 0, it), (4, 7, 30), (4, 30, it)]]
    run(6, 16, 200, it, pieces)
    run(6, 32, 500, it, pieces)
