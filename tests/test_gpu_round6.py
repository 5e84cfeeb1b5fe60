"""Round-6 device tests (all through the C ABI).

 * the test hooks of the sampler under the launch forms round 5 did not cover: epx_sample_piece with more
   sites than the device holds workgroups (a site is released as FINISHED behind its one transition, so a late workgroup
   cannot take it for a second one and leave another site untouched), and epx_set_trace under a split launch
   (epx_set_site_split: the lead sites run in a second launch, which has to carry the trace as well).
References: the work these hooks check is the tilted-distribution sampling of /root/reference/epstan/method.py:338-408."""

import numpy as np
import pytest

from epstan_amd.engine import HipEngine
from oracle import nuts_oracle as no
from test_gpu_parity import _engine_with_cavity, _site_problem

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('layout,K', [(7, 2304), (5, 600)])
def test_teacher_forced_transition_with_more_sites_than_resident_workgroups(layout, K):
    """epx_sample_piece launches one workgroup per site; with K above what the device holds (8 workgroups per CU at most:
    2 048) some start only when others have ended.  Every site must still take exactly ONE transition, its own: the record
    at boundary t0 + 1 of every site is the oracle's state in front of t0 + 1."""
    it, chains, t0 = 16, 4, 3
    X, y, k_lim, Oms, mus, d, P = _site_problem('m4b_sg', 16, 24, 9, K=K, tight=4.0)
    eng, Om_dev, mu_dev = _engine_with_cavity('m4b_sg', X, y, k_lim, Oms, mus)
    seeds = np.arange(K, dtype=np.int64) * 3 + 5
    _, _, st_o, tr_o, du = no.nuts_sites('m4b_sg', X, y, k_lim, mu_dev, Om_dev, seeds, chains=chains, iter=it,
                                         trace_sites=K, dump_at=[t0, t0 + 1])
    opts = HipEngine.sampler_opts(chains=chains, iter=it, warmup=None, init='random', layout=layout)
    s0, s1 = du[:, :, 0], du[:, :, 1]
    rec_in = eng.pack_records(s0[..., :20], s0[..., 20:20 + P], s0[..., 20 + P:20 + 2 * P],
                              s0[..., 20 + 2 * P:20 + 3 * P], s0[..., 20 + 3 * P:20 + 4 * P])
    rec_out = eng.sample_piece(seeds, opts, t0, rec_in)
    assert eng.last_layout() == layout
    sc, qs, wmean, wm2, inv_e = eng.unpack_records(rec_out)
    names = HipEngine.CK_SCALARS
    it_ = names.index('t')
    # no site was left out (an untouched record is all zero) and none went two transitions
    assert np.all(sc[..., it_] == t0 + 1), np.unique(sc[..., it_], return_counts=True)
    for key in ('da_count', 'va_n', 'va_counter', 'va_next', 'kept', 'failed', 'ngrad', 'nleap_tot'):
        j = names.index(key)
        assert np.array_equal(sc[..., j], s1[..., j]), key
    scale = np.maximum(1.0, np.abs(s1[..., 20:20 + P]).max(axis=2))[..., None]
    assert (np.abs(qs - s1[..., 20:20 + P]) / scale).max() < 1e-6
    j = names.index('eps')
    np.testing.assert_allclose(sc[..., j], s1[..., j], rtol=1e-8)


def test_trace_of_a_split_launch_covers_the_lead_sites():
    """epx_set_site_split + epx_set_trace: the lead sites (second launch, layout 2) leave their trace records too, keyed by
    the real site -- the kept tail of every site's trace is its draws, and every transition has a leapfrog count."""
    K, it, chains = 330, 24, 4               # (enough sites for the library to pick one workgroup per site by itself, as the split needs)
    X, y, k_lim, Oms, mus, d, P = _site_problem('m1b_sg', 4, 30, 5, K=K, tight=30.0)
    eng, _, _ = _engine_with_cavity('m1b_sg', X, y, k_lim, Oms, mus)
    seeds = np.arange(K, dtype=np.int64) + 11
    opts = HipEngine.sampler_opts(chains=chains, iter=it, init='random', max_depth=6)
    order = np.random.RandomState(1).permutation(K)
    eng.set_site_order(order)
    eng.set_site_split(6)
    eng.set_trace(K)
    eng.sample_batch(seeds, opts)
    m = eng.last_split()
    assert m >= 1
    tr = eng.get_trace(chains, it)
    kept = np.stack([eng.get_draws(k, all_params=True).reshape(chains, it - it // 2, P) for k in range(K)])
    assert np.array_equal(tr[:, :, it // 2:, 8:], kept)
    assert np.all(tr[order[:m]][..., 1] >= 1), 'the lead sites of the split launch left no trace'
    assert np.all(tr[..., 1] >= 1)
    # the same records as without the split, for the sites the split does not touch
    eng.set_site_split(0)
    eng.sample_batch(seeds, opts)
    tr0 = eng.get_trace(chains, it)
    assert np.array_equal(tr0[order[m:]], tr[order[m:]])
    eng.set_trace(0)


def test_get_trace_without_set_trace_fails_with_the_librarys_error():
    """(round 5: an AttributeError from the wrapper, because `_trace_sites` only existed after set_trace)"""
    from epstan_amd._lib import EpxError
    X, y, k_lim, Oms, mus, d, P = _site_problem('m1b_sg', 4, 30, 5, K=2)
    eng, _, _ = _engine_with_cavity('m1b_sg', X, y, k_lim, Oms, mus)
    with pytest.raises(EpxError, match='no trace'):
        eng.get_trace(4, 20)
