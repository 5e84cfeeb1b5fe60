"""Helper of tests/test_gpu_round4.py (not a test): the piece hand-off litmus run in a process of its own, so that the
library under test can be chosen with EPX_LIB (the default build, or variants/libepx_fence.so = -DEPX_PIECE_FENCE).

Many small sites in pieces of ONE transition: every site is claimed `iter` times per launch, by whichever workgroup is
free -- it changes CU and XCD dozens of times, and every boundary's checkpoint record crosses between L2s
(csrc/epx_pieces.h).  The launch is repeated; every repetition must reproduce the uncut launch bit for bit (draws, last
states, chain statistics).  Prints one line: `litmus <layout> <sites> <reps> <mismatching reps> <sha256 of the uncut draws>`."""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np                                   # noqa: E402


def main():
    layout, D, n, K, it, reps = (int(a) for a in sys.argv[1:7])
    from epstan_amd.engine import HipEngine
    rng = np.random.RandomState(5)
    X = rng.randn(K * n, D)
    y = (rng.rand(K * n) < 0.5).astype(int)
    eng = HipEngine('m4b_sg', X, y, np.arange(K + 1) * n)
    d = eng.d
    eng.set_prior(np.eye(d), np.zeros(d))
    eng.set_global(np.eye(d) * 2.0, np.zeros(d))
    assert np.all(eng.cavity_batch(0))
    seeds = np.arange(K, dtype=np.int64) * 13 + 5
    opts = HipEngine.sampler_opts(chains=4, iter=it, init='random', max_depth=5, layout=layout)

    def state():
        dr = np.stack([eng.get_draws(k, all_params=True) for k in range(0, K, 7)])
        return dr, eng.get_chain_stats(4).copy()

    eng.sample_batch(seeds, opts)
    assert eng.last_segments() >= 0
    lay = eng.last_layout()
    assert layout == 0 or lay == layout, (lay, layout)
    dr0, cs0 = state()
    bad = 0
    # uneven predicted rates: the claims do not go round robin, sites wander over the workgroups
    rate = 1.0 + (np.arange(K) % 5)
    for rep in range(reps):
        eng.set_piece_queue(1, rate if rep % 2 else None)
        eng.sample_batch(seeds, opts)
        assert eng.last_segments() == -it and eng.last_layout() == lay
        dr, cs = state()
        if not (np.array_equal(dr, dr0) and np.array_equal(cs, cs0)):
            bad += 1
    eng.set_piece_queue(0)
    eng.close()
    print('litmus %d %d %d %d %s' % (lay, K, reps, bad, hashlib.sha256(np.ascontiguousarray(dr0).tobytes()).hexdigest()), flush=True)


if __name__ == '__main__':
    main()
