import sys, os, hashlib
sys.path.insert(0, os.getcwd())
import numpy as np
from epstan_amd.engine import HipEngine
K, D, n, it, reps = 64, 16, 200, 120, int(sys.argv[1])
rng = np.random.RandomState(12)
X = rng.randn(K * n, D); y = (rng.rand(K * n) < 0.5).astype(int)
eng = HipEngine('m4b_sg', X, y, np.arange(K + 1) * n)
d = eng.d
eng.set_prior(np.eye(d), np.zeros(d)); eng.set_global(np.eye(d) * 3.0, np.zeros(d))
assert np.all(eng.cavity_batch(0))
seeds = np.arange(K, dtype=np.int64) * 11 + 7
opts = HipEngine.sampler_opts(chains=4, iter=it, init='random', layout=6)
def state():
    return np.stack([eng.get_draws(k, all_params=True) for k in range(K)]), eng.get_chain_stats(4).copy()
eng.sample_batch(seeds, opts); assert eng.last_layout() == 6
dr0, cs0 = state(); bad = 0
for rep in range(reps):
    eng.sample_batch(seeds, opts); dr, cs = state()
    bad += int(not (np.array_equal(dr, dr0) and np.array_equal(cs, cs0)))
print('layout6 stress: %d sites x 4 chains, %d transitions, %d repetitions, %d differing, leapfrogs per launch %d, sha256 %s' % (K, it, reps, bad, int(cs0[:, :, 3].sum()), hashlib.sha256(dr0.tobytes()).hexdigest()[:16]))
