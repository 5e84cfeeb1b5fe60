"""Seed derivation of the reference, vectorised over sites.

The reference draws `seeds = RandomState(seed).randint(0, MAX_UINT, (niter, K))`
(method.py:956-960) and, per site update, the Stan seed
`RandomState(seeds[i,k]).randint(0, MAX_UINT)` (method.py:342-346).  Building K
RandomState objects per iteration costs ~20 us each; `stan_seeds` computes the
same numbers with array operations: MT19937 `init_genrand`, the first tempered
output, and NumPy's masked rejection for `randint(0, 2**31-1)` (draw & 0x7fffffff,
rejected only when it equals 2**31-1; such seeds fall back to RandomState).
"""

import numpy as np

MAX_UINT = 2**31 - 1      # pystan.constants.MAX_UINT (method.py:40)


def run_seeds(seed, niter, K):
    """method.py:956-960."""
    if isinstance(seed, np.random.RandomState):
        rng = seed
    else:
        rng = np.random.RandomState(seed=seed)
    return rng.randint(0, MAX_UINT, size=(niter, K))


def _first_mt_output(seeds):
    s = np.asarray(seeds, dtype=np.uint64) & np.uint64(0xFFFFFFFF)
    M = np.uint64(0xFFFFFFFF)
    # init_genrand: mt[i] = 1812433253 * (mt[i-1] ^ (mt[i-1] >> 30)) + i; keep mt[0], mt[1], mt[397]
    mt = s.copy()
    mt0 = mt.copy()
    mt1 = None
    for i in range(1, 398):
        mt = (np.uint64(1812433253) * (mt ^ (mt >> np.uint64(30))) + np.uint64(i)) & M
        if i == 1:
            mt1 = mt.copy()
    mt397 = mt
    y = (mt0 & np.uint64(0x80000000)) | (mt1 & np.uint64(0x7FFFFFFF))
    v = mt397 ^ (y >> np.uint64(1)) ^ np.where(y & np.uint64(1), np.uint64(0x9908B0DF), np.uint64(0))
    v ^= v >> np.uint64(11)
    v ^= (v << np.uint64(7)) & np.uint64(0x9D2C5680)
    v ^= (v << np.uint64(15)) & np.uint64(0xEFC60000)
    v ^= v >> np.uint64(18)
    return v & M


def stan_seeds(seeds):
    """Vector form of `RandomState(s).randint(0, MAX_UINT)` for integer seeds."""
    seeds = np.asarray(seeds, dtype=np.int64)
    out = (_first_mt_output(seeds.ravel()) & np.uint64(0x7FFFFFFF)).astype(np.int64)
    bad = np.nonzero(out > MAX_UINT - 1)[0]          # masked rejection (p = 2**-31)
    for i in bad:
        out[i] = np.random.RandomState(int(seeds.ravel()[i])).randint(0, MAX_UINT)
    return out.reshape(seeds.shape)


def stan_seed(seed):
    """method.py:342-346 for one site (accepts a RandomState like the reference)."""
    if isinstance(seed, np.random.RandomState):
        return int(seed.randint(0, MAX_UINT))
    if seed is None:
        return int(np.random.RandomState(None).randint(0, MAX_UINT))
    return int(stan_seeds(np.array([seed]))[0])
