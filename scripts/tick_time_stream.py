"""Per-leapfrog time and HBM rate of the streaming sampler at the C5 site size."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from epstan_amd.engine import HipEngine, QI
K, D, n = int(sys.argv[1]) if len(sys.argv) > 1 else 256, 128, 2000
rng = np.random.RandomState(0)
X = rng.randn(K * n, D) * 0.3
y = (rng.rand(K * n) < 0.5).astype(int)
eng = HipEngine('m4b_sg', X, y, np.arange(K + 1) * n)
d, P = eng.d, eng.P
eng.set_prior(np.eye(d), np.zeros(d))
eng.set_global(np.eye(d) * 2.0, np.zeros(d))
assert np.all(eng.cavity_batch(QI))
for it in [int(v) for v in sys.argv[2:]] or (4, 8):
    opts = HipEngine.sampler_opts(chains=4, iter=it, init='random')
    stats, ms = eng.sample_batch(np.arange(K) + 1, opts)
    cs = eng.get_chain_stats(4)
    ticks = cs[:, :, 3].max(axis=1)                 # lock-step leapfrogs per site
    bytes_tick = n * D * 8 + n + d * d * 8
    print('K=%d iter=%d: %.1f ms, ticks/site max %d mean %.0f -> %.1f us/tick (slowest site), '
          'HBM algorithmic %.2f TB/s (%.1f GB in %.1f ms), gradients %.3g'
          % (K, it, ms, ticks.max(), ticks.mean(), ms * 1e3 / ticks.max(),
             ticks.sum() * bytes_tick / (ms * 1e-3) / 1e12, ticks.sum() * bytes_tick / 1e9, ms, cs[:, :, 3].sum()))
