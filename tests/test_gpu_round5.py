"""Round-5 device tests: the sampler's per-transition TRACE (epx_set_trace: every transition of a site update, the warm-up
included, which the draws the reference's caller sees -- method.py:88-104, post-warm-up only -- never show) against the
same trace of oracle/nuts_oracle.c.

 * test_trace_follows_the_oracle_until_a_decision_parts_them -- VERDICT round 4, item 4b (scripts/errgrowth.py promoted):
   at generic (funnel-shaped) cavities of the C3 site shape the device's chain and the oracle's agree transition by
   transition -- draw, leapfrog count, accept statistic, the step size dual averaging hands to the next transition, the
   metric after the variance window -- up to the first transition whose leapfrog count or draw differs, and the error in
   front of that transition is still at rounding level: the two runs do not drift apart, one DECISION parts them.
 * test_teacher_forced_warmup_transitions_at_generic_cavities -- item 4a: from the ORACLE's state in front of warm-up
   transition t in {3, 40, 89 (the end of the metric window), 99 (complete_adaptation)} -- injected as the checkpoint record
   of a pieced launch (epx_sample_piece) -- the device takes that one transition; draw, leapfrog count, accept statistic
   and the UPDATED step size, dual-averaging state, Welford sums, window counters and metric must be the oracle's.  Deep
   trees, no need for the two runs to have stayed together up to t.
   (Free-running chains cannot reach those transitions together: whatever the cavity's tightness, the rounding differences of
   the two summation orders double per transition on these posteriors -- NUTS trajectories are as long as the posterior is
   wide -- and reach 1e-6 around transition 30-50 of a 200-transition run: gpurun_out/r5/explore_tight.txt.)

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


@pytest.mark.parametrize('layout,D,n', [(7, 32, 500), (5, 32, 500), (3, 32, 500), (6, 16, 200), (1, 16, 200), (2, 16, 200)])
def test_trace_follows_the_oracle_until_a_decision_parts_them(layout, D, n):
    """Generic cavities (tight = 1: the funnels of the hierarchical scales are felt, trees go deep); the C3 site shape for
    the layouts of C3 / C4 / C5, the C2 site shape for C2's layout 6 (a workgroup per chain; it has no pieced form, so this
    free-running trace is its warm-up check) and for the generic kernels 1 / 2."""
    it = 60
    tr_d, tr_o, P = _traces('m4b_sg', D, n, 3, it, layout, 1.0, 41)
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
                assert abs(tr_d[k, c, ts, 0] - tr_o[k, c, ts, 0]) <= (1e-9 + 1e3 * before[k, c]) * abs(tr_o[k, c, ts, 0])
    # the first transition -- the step-size search from eps = 1 included -- agrees to rounding: same problem, same arithmetic
    assert err[:, :, 0].max() < 1e-9 and np.abs(tr_d[:, :, 0, 5] / tr_o[:, :, 0, 5] - 1.0).max() < 1e-9
    # the comparison is not vacuous: most chains get through the step-size search and several deep transitions together
    assert np.median(t_star) >= 3 and n_checked >= 3 * K * C, (t_star, n_checked)
    deep = [tr_o[k, c, t, 1] for k in range(K) for c in range(C) for t in range(int(t_star[k, c]))]
    assert max(deep) >= 63, 'no transition of depth >= 6 was compared: %s' % (sorted(deep)[-5:],)


@pytest.mark.parametrize('layout', [7, 5, 3])
def test_teacher_forced_warmup_transitions_at_generic_cavities(layout):
    """iter = 200: warm-up 100, the variance window ends with transition 89 (metric update + step-size search), transition
    99 completes the adaptation.  The oracle runs freely and leaves its state in front of transitions t and t + 1; the
    device starts every chain from the oracle's state at t (epx_sample_piece) and must arrive at the oracle's state at t + 1."""
    _teacher_forced('m4b_sg', 32, 500, layout, 4)


@pytest.mark.parametrize('model,D,n', [('m1b_sg', 32, 300), ('m5b_sg', 21, 333), ('m4b_sg', 16, 200)])
def test_teacher_forced_warmup_transitions_other_models_and_shapes(model, D, n):
    """The same check for the model without per-coefficient scales (m1b: the plain path of the state wave), the one with a
    third hierarchy level (m5b, odd sizes: ragged row tiles) and the C2 site shape, on the layout the headline uses."""
    _teacher_forced(model, D, n, 7, 0)


def _teacher_forced(model, D, n, layout, min_deep):
    K, it, chains = 2, 200, 4
    X, y, k_lim, Oms, mus, d, P = _site_problem(model, D, n, 23, K=K, tight=1.0)
    eng, Om_dev, mu_dev = _engine_with_cavity(model, X, y, k_lim, Oms, mus)
    seeds = np.array([77, 1234], dtype=np.int64)
    ts = [3, 40, 89, 99]
    dump_at = sorted(set(ts + [t + 1 for t in ts]))
    _, _, st_o, tr_o, du = no.nuts_sites(model, X, y, k_lim, mu_dev, Om_dev, seeds, chains=chains, iter=it,
                                         trace_sites=K, dump_at=dump_at)
    opts = HipEngine.sampler_opts(chains=chains, iter=it, warmup=None, init='random', layout=layout)
    eng.set_trace(K)
    n_deep = 0
    for t in ts:
        i0, i1 = dump_at.index(t), dump_at.index(t + 1)
        s0 = du[:, :, i0]
        rec_in = eng.pack_records(s0[..., :20], s0[..., 20:20 + P], s0[..., 20 + P:20 + 2 * P],
                                  s0[..., 20 + 2 * P:20 + 3 * P], s0[..., 20 + 3 * P:20 + 4 * P])
        rec_out = eng.sample_piece(seeds, opts, t, rec_in)
        assert eng.last_layout() == layout
        tr_d = eng.get_trace(chains, it)[:, :, t]
        sc, qs, wmean, wm2, inv_e = eng.unpack_records(rec_out)
        s1 = du[:, :, i1]
        names = HipEngine.CK_SCALARS
        for k in range(K):
            for c in range(chains):
                a, b = tr_d[k, c], tr_o[k, c, t]
                ctx = (layout, t, k, c)
                # the transition itself: same tree, same draw
                assert a[1] == b[1] and a[3] == b[3] and a[4] == b[4], (ctx, a[:8], b[:8])
                scale = max(1.0, np.abs(b[8:]).max())
                assert np.abs(a[8:] - b[8:]).max() / scale < 1e-6, (ctx, np.abs(a[8:] - b[8:]).max())
                assert abs(a[0] - b[0]) <= 1e-12 * abs(b[0])                      # the injected step size was used
                assert abs(a[2] - b[2]) <= 1e-9 + 1e-9 * abs(b[2]), (ctx, 'accept', a[2], b[2])
                n_deep += int(b[1] >= 127)
                # the UPDATED adaptation state (the record the piece leaves) against the oracle's state in front of t + 1
                got = dict(zip(names, sc[k, c]))
                want = dict(zip(names, s1[k, c, :20]))
                for key in ('eps', 'da_mu', 's_bar', 'x_bar'):
                    assert abs(got[key] - want[key]) <= 1e-9 * max(1.0, abs(want[key])), (ctx, key, got[key], want[key])
                for key in ('da_count', 'va_n', 't', 'va_counter', 'va_wsize', 'va_next', 'ndiv', 'npost', 'kept', 'failed'):
                    assert got[key] == want[key], (ctx, key, got[key], want[key])
                assert abs(got['lps'] - want['lps']) <= 1e-7 * max(1.0, abs(want['lps'])), (ctx, 'lps', got['lps'], want['lps'])
                np.testing.assert_allclose(qs[k, c], s1[k, c, 20:20 + P], rtol=0, atol=1e-6 * scale)
                np.testing.assert_allclose(wmean[k, c], s1[k, c, 20 + P:20 + 2 * P], rtol=1e-9, atol=1e-6 * scale)
                np.testing.assert_allclose(wm2[k, c], s1[k, c, 20 + 2 * P:20 + 3 * P], rtol=1e-7, atol=1e-6 * scale)
                np.testing.assert_allclose(inv_e[k, c], s1[k, c, 20 + 3 * P:20 + 4 * P], rtol=1e-7, atol=1e-12)
        if t == 89:
            # the window's end: the metric changed, the dual averaging restarted around a SEARCHED step size
            assert not np.allclose(inv_e[0, 0], s0[0, 0, 20 + 3 * P:20 + 4 * P])
            assert sc[0, 0, names.index('da_count')] == 0 and sc[0, 0, names.index('va_n')] == 0
    assert n_deep >= min_deep, 'the teacher-forced transitions were shallow (%d of depth >= 7)' % n_deep


def test_team_passes_count_what_the_row_team_did():
    """epx_get_team_passes (layout 7): a device-side count of the lock-step passes, the ones a chain yielded included --
    never fewer than the gradients of the site's longest chain, and not many more; zero under the other layouts."""
    X, y, k_lim, Oms, mus, d, P = _site_problem('m4b_sg', 32, 500, 5, K=3, tight=1.0)
    eng, Om_dev, mu_dev = _engine_with_cavity('m4b_sg', X, y, k_lim, Oms, mus)
    seeds = np.arange(3, dtype=np.int64) + 11
    for layout in (7, 5):
        opts = HipEngine.sampler_opts(chains=4, iter=40, warmup=None, init='random', layout=layout)
        eng.sample_batch(seeds, opts)
        assert eng.last_layout() == layout
        tp, rp = eng.team_passes(), eng.row_passes(4)
        if layout == 7:
            longest = eng.get_chain_stats(4)[:, :, 3].max(axis=1)
            assert np.all(tp >= longest) and np.all(tp <= 1.3 * longest + 8), (tp, longest)
            assert np.array_equal(rp, longest)
        else:
            assert np.all(tp == 0)
