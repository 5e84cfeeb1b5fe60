"""Round-5 device tests: the sampler's per-transition TRACE (epx_set_trace: every transition of a site update, the warm-up
included, which the draws the reference's caller sees -- method.py:88-104, post-warm-up only -- never show) against the
same trace of oracle/nuts_oracle.c.

 * test_trace_follows_the_oracle_until_a_decision_parts_them -- VERDICT round 4, item 4b (scripts/errgrowth.py promoted):
   at generic (funnel-shaped) cavities of the C3 site shape the device's chain and the oracle's agree transition by
   transition -- draw, leapfrog count, accept statistic, the step size dual averaging hands to the next transition, the
   metric after the variance window -- up to the first transition whose leapfrog count or draw differs, and the error in
   front of that transition is still at rounding level: the two runs do not drift apart, one DECISION parts them.
 * test_warmup_adaptation_state_matches_the_oracle_transition_by_transition -- item 4a at cavities where the chains stay
   together through the whole warm-up: every warm-up transition, the end of the metric window (transition 89 of 100:
   Stan's windows rescaled to a 100-transition warm-up) and complete_adaptation (99) included, leaves the same step size (1e-9), metric (1e-9) and sample (1e-6) on both sides.

Everything goes through the C ABI (ctypes)."""

import numpy as np
import pytest

from epstan_amd.engine import HipEngine
from oracle import nuts_oracle as no
from test_gpu_parity import _engine_with_cavity, _site_problem

pytestmark = pytest.mark.gpu


def _traces(model, D, n, K, it, layout, tight, seed, chains=4):
    X, y, k_lim, Oms, mus, d, P = _site_problem(model, D, n, seed, K=K, tight=tight)
    eng, Om_dev, mu_dev = _engine_with_cavity(model, X, y, k_lim, Oms, mus)
    seeds = (np.arange(K, dtype=np.int64) * 7 + 3 + seed)
    opts = HipEngine.sampler_opts(chains=chains, iter=it, warmup=None, init='random', layout=layout)
    eng.set_trace(K)
    eng.sample_batch(seeds, opts)
    assert layout == 0 or eng.last_layout() == layout
    tr_d = eng.get_trace(chains, it)
    kept = np.stack([eng.get_draws(k, all_params=True).reshape(chains, it - it // 2, P) for k in range(K)])
    # the trace's tail IS the kept draws (same stores' source)
    assert np.array_equal(tr_d[:, :, it // 2:, 8:], kept)
    _, _, st_o, tr_o = no.nuts_sites(model, X, y, k_lim, mu_dev, Om_dev, seeds, chains=chains, iter=it, trace_sites=K)
    return tr_d, tr_o, P


def _parting(tr_d, tr_o):
    """Per chain: the first transition whose leapfrog count differs or whose draw differs by more than 1e-6 (relative to
    the draw's scale); `iter` when the chains agree to the end.  Returns (t_star, error in front of t_star)."""
    K, C, T, _ = tr_d.shape
    scale = np.maximum(1.0, np.abs(tr_o[..., 8:]).max(axis=3))
    err = np.abs(tr_d[..., 8:] - tr_o[..., 8:]).max(axis=3) / scale            # (K, C, T)
    differ = (tr_d[..., 1] != tr_o[..., 1]) | (err > 1e-6)
    t_star = np.where(differ.any(axis=2), differ.argmax(axis=2), T)
    before = np.zeros((K, C))
    for k in range(K):
        for c in range(C):
            if t_star[k, c] > 0:
                before[k, c] = err[k, c, :t_star[k, c]].max()
    return t_star, before, err


@pytest.mark.parametrize('layout', [7, 5, 3])
def test_trace_follows_the_oracle_until_a_decision_parts_them(layout):
    """Generic cavities (tight = 1: the funnels of the hierarchical scales are felt, trees go deep), the C3 site shape."""
    it = 60
    tr_d, tr_o, P = _traces('m4b_sg', 32, 500, 3, it, layout, 1.0, 41)
    t_star, before, err = _parting(tr_d, tr_o)
    K, C, T, _ = tr_d.shape
    n_checked = 0
    for k in range(K):
        for c in range(C):
            ts = int(t_star[k, c])
            # everything in front of the parting transition agrees: no drift, the error is still at rounding level there
            assert before[k, c] < 1e-6, (layout, k, c, ts, before[k, c])
            for t in range(ts):
                a, b = tr_d[k, c, t], tr_o[k, c, t]
                assert a[1] == b[1] and a[3] == b[3] and a[4] == b[4], (layout, k, c, t, a[:8], b[:8])
                # (the adaptation state follows the trajectories: dual averaging divides the accept statistics' differences by
                # gamma = 0.05, and on these posteriors a relative difference of 1e-7 in a state is 1e-4 in its energy -- so
                # the bounds widen with the error the draws have reached; at the first transitions they are rounding level)
                tol = 1e-9 + 1e3 * (err[k, c, :t].max() if t > 0 else 0.0)
                assert abs(a[0] - b[0]) <= tol * abs(b[0]), (layout, k, c, t, 'step size used', a[0], b[0], tol)
                assert abs(a[2] - b[2]) <= 10 * tol + 1e-7 * abs(b[2]), (layout, k, c, t, 'accept', a[2], b[2], tol)
                assert abs(a[5] - b[5]) <= (1e3 * tol + 1e3 * err[k, c, t]) * abs(b[5]), (layout, k, c, t, 'adapted step size', a[5], b[5], tol)
                assert abs(a[6] - b[6]) <= (1e-9 + 1e2 * err[k, c, :t + 1].max()) * abs(b[6]), (layout, k, c, t, 'metric', a[6], b[6])
                n_checked += 1
            if ts < T:
                # the transition that parts them started from states that agree: its inputs (the sample and the step size
                # in front of it) are the same to rounding -- a decision inside it flipped, nothing drifted
                assert abs(tr_d[k, c, ts, 0] - tr_o[k, c, ts, 0]) <= 1e-7 * abs(tr_o[k, c, ts, 0])
    # the first transition -- the step-size search from eps = 1 included -- agrees to rounding: same problem, same arithmetic
    assert err[:, :, 0].max() < 1e-9 and np.abs(tr_d[:, :, 0, 5] / tr_o[:, :, 0, 5] - 1.0).max() < 1e-9
    # the comparison is not vacuous: most chains get through the step-size search and several deep transitions together
    assert np.median(t_star) >= 3 and n_checked >= 3 * K * C, (t_star, n_checked)
    deep = [tr_o[k, c, t, 1] for k in range(K) for c in range(C) for t in range(int(t_star[k, c]))]
    assert max(deep) >= 63, 'no transition of depth >= 6 was compared: %s' % (sorted(deep)[-5:],)


@pytest.mark.parametrize('layout', [7, 5, 3])
def test_warmup_adaptation_state_matches_the_oracle_transition_by_transition(layout):
    """iter = 200 (warm-up 100: init buffer 15, ONE variance window ending with transition 89, term buffer 10 -- Stan's
    rescaled windows).  At cavities that dominate the likelihood the chains stay together, so EVERY
    warm-up transition is compared: learn_stepsize after each, learn_variance at the window's end (the metric changes and
    a new step-size search follows), complete_adaptation at transition 99, and the sampling phase behind it."""
    it = 200
    tr_d, tr_o, P = _traces('m4b_sg', 32, 500, 2, it, layout, 300.0, 7)
    t_star, before, err = _parting(tr_d, tr_o)
    K, C, T, _ = tr_d.shape
    together = t_star >= T
    assert together.sum() >= K * C - 1, t_star              # (one near-threshold decision may flip somewhere in 200 transitions)
    for k in range(K):
        for c in range(C):
            ts = int(t_star[k, c])
            a, b = tr_d[k, c, :ts], tr_o[k, c, :ts]
            assert np.array_equal(a[:, 1], b[:, 1]) and np.array_equal(a[:, 3], b[:, 3])
            np.testing.assert_allclose(a[:, 0], b[:, 0], rtol=1e-9)            # step size used
            np.testing.assert_allclose(a[:, 5], b[:, 5], rtol=1e-9)            # ... handed on by learn_stepsize / complete_adaptation
            np.testing.assert_allclose(a[:, 6], b[:, 6], rtol=1e-9)            # metric (changes once: at the window's end)
            np.testing.assert_allclose(a[:, 2], b[:, 2], rtol=1e-7, atol=1e-9)  # accept statistic
            np.testing.assert_allclose(a[:, 7], b[:, 7], rtol=1e-9, atol=1e-7)  # log density of the new sample
            assert before[k, c] < 1e-6
    # the window really ended inside the warm-up and changed the metric; the step size was re-searched behind it
    m = tr_o[0, 0, :, 6]
    change = np.nonzero(m[1:] != m[:-1])[0] + 1
    assert len(change) == 1 and 15 <= change[0] < 100, change
    assert tr_o[0, 0, change[0] + 1, 0] != tr_o[0, 0, change[0], 5]        # (the next transition runs at a SEARCHED step size, not the learned one)
