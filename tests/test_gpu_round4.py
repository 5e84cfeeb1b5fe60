"""Round-4 device tests: the piece hand-off's litmus run under both protocols (the default build and the
-DEPX_PIECE_FENCE one shipped as variants/libepx_fence.so), the out-of-memory fallback of the pieced launch, Stan's retry of a
random start, the row team's yielded passes (the oracle bounds of the headline kernel are tightened in place:
test_gpu_parity.py, test_gpu_round3.py).

Everything goes through the C ABI (ctypes)."""

import os
import subprocess
import sys

import numpy as np
import pytest

from epstan_amd.engine import HipEngine
from test_gpu_parity import _engine_with_cavity, _site_problem

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FENCE_LIB = os.path.join(ROOT, 'variants', 'libepx_fence.so')
pytestmark = pytest.mark.gpu


def _litmus(layout, D, n, K, it, reps, lib=None):
    env = dict(os.environ)
    env.pop('EPX_PIECE_GRID', None)
    if lib is not None:
        env['EPX_LIB'] = lib
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'piece_litmus_helper.py')] +
                         [str(v) for v in (layout, D, n, K, it, reps)], env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, timeout=900)
    assert res.returncode == 0, res.stderr.decode('utf-8', 'replace')[-3000:]
    line = [l for l in res.stdout.decode().splitlines() if l.startswith('litmus ')][-1].split()
    return int(line[1]), int(line[4]), line[5]


# (the last case: the C3 site size, where ONE workgroup fills a CU -- the occupancy MI355X_MICROARCH.md measured the
# fence-free form at; the small sites of the other cases put several workgroups on a CU)
@pytest.mark.parametrize('layout,D,n,K,it,reps', [(7, 16, 48, 320, 12, 50), (5, 16, 48, 320, 12, 25), (0, 72, 40, 96, 8, 25),
                                                  (7, 32, 500, 288, 8, 8)])
def test_piece_handoff_litmus_default_and_fence_builds_agree(layout, D, n, K, it, reps):
    """csrc/epx_pieces.h hands a site's checkpoint from one XCD to another with write-through stores + vmcnt(0) + barrier +
    a relaxed flag store (no L2 write-back fence).  Litmus: hundreds of sites cut into pieces of ONE transition (every site
    is claimed `it` times per launch by whichever of the device's looping workgroups is free, so it crosses XCDs dozens of
    times), repeated `reps` times: every repetition bit-equal to the uncut launch.  The same run under the build that keeps
    the memory model's release fence (variants/libepx_fence.so) must give the same draws."""
    lay, bad, digest = _litmus(layout, D, n, K, it, reps)
    assert (lay == layout or layout == 0) and bad == 0, (lay, bad)
    assert os.path.exists(FENCE_LIB), 'variants/libepx_fence.so is missing: run __graft_entry__.build()'
    lay_f, bad_f, digest_f = _litmus(layout, D, n, K, it, max(5, reps // 5), lib=FENCE_LIB)
    assert lay_f == lay and bad_f == 0 and digest_f == digest


def test_pieced_launch_falls_back_to_the_uncut_one_when_the_checkpoints_cannot_be_allocated(monkeypatch):
    """run_sampler re-allocates the tree stack for the looping workgroups and then the checkpoint records; when the second
    allocation is refused the launch runs uncut -- on the NEW stack buffer (round 3 left the launch arguments pointing at
    the freed one).  EPX_TEST_FAIL_CKPT makes the allocation count as refused."""
    K, it, chains = 5, 16, 4
    X, y, k_lim, Oms, mus, d, P = _site_problem('m4b_sg', 32, 500, 11, K=K, tight=1000.)
    eng, Om_dev, mu_dev = _engine_with_cavity('m4b_sg', X, y, k_lim, Oms, mus)
    seeds = np.arange(K, dtype=np.int64) * 3 + 2
    opts = HipEngine.sampler_opts(chains=chains, iter=it, init='random', layout=7)
    monkeypatch.setenv('EPX_TEST_FAIL_CKPT', '1')
    eng.set_piece_queue(2, None)
    eng.sample_batch(seeds, opts)                           # first call of the context: stack grown, checkpoints refused
    assert eng.last_layout() == 7 and eng.last_segments() == 0
    fell_back = [eng.get_draws(k, all_params=True).copy() for k in range(K)]
    monkeypatch.delenv('EPX_TEST_FAIL_CKPT')
    eng.sample_batch(seeds, opts)
    assert eng.last_segments() == -8
    for k in range(K):
        np.testing.assert_array_equal(eng.get_draws(k, all_params=True), fell_back[k])
    eng.set_piece_queue(0)
    eng.sample_batch(seeds, opts)
    for k in range(K):
        np.testing.assert_array_equal(eng.get_draws(k, all_params=True), fell_back[k])


@pytest.mark.parametrize('model,D,n,layout', [('m1b_sg', 3, 20, 0), ('m4b_sg', 16, 48, 7), ('m4b_sg', 16, 48, 5), ('m4b_sg', 16, 48, 1),
                                              ('m4b_sg', 40, 30, 0)])
def test_random_init_retry_follows_the_oracle(model, D, n, layout):
    """init='random' draws the start again (up to 100 times) until log density and gradient are finite, as Stan's
    initialize() behind /root/reference/epstan/util.py:716 does.  A cavity that is finite only for |phi_0| < 1 rejects about
    half of the first draws; device and oracle continue from the same later draw of the chain's Philox stream (the posterior
    is so narrow that no chain moves by more than 1e-100 afterwards), and a start that was GIVEN is not replaced."""
    from oracle import nuts_oracle as no
    K = 4
    rng = np.random.RandomState(3)
    X = rng.randn(K * n, D)
    y = (rng.rand(K * n) < 0.5).astype(int)
    eng = HipEngine(model, X, y, np.arange(K + 1) * n)
    d, P = eng.d, eng.P
    Om = np.eye(d); Om[0, 0] = np.finfo(float).max
    for k in range(K):
        assert eng.cavity_site(k, Om + np.eye(d), np.zeros(d), np.eye(d), np.zeros(d))
    Om_dev = np.stack([eng.get_cavity(k)[0] for k in range(K)])
    mu_dev = np.stack([eng.get_cavity(k)[1] for k in range(K)])
    assert Om_dev[0, 0, 0] == np.finfo(float).max
    seeds = np.arange(K, dtype=np.int64) + 40
    stats, _ = eng.sample_batch(seeds, HipEngine.sampler_opts(chains=4, iter=4, warmup=2, init='random', layout=layout))
    assert layout == 0 or eng.last_layout() == layout
    draws_o, _, st_o = no.nuts_sites(model, X, y, np.arange(K + 1) * n, mu_dev, Om_dev, seeds, chains=4, iter=4, warmup=2)
    first = np.array([[-2.0 + 4.0 * no.rng_probe(int(s), c, 0, 0, 0, 0)[0] for c in range(4)] for s in seeds])
    assert (np.abs(first) >= 1.0).any() and np.all(st_o[:, :, 7] == 0)
    cs = eng.get_chain_stats(4)
    assert np.all(cs[:, :, 7] == 0) and stats[:, 7].sum() == 0
    for k in range(K):
        np.testing.assert_allclose(eng.get_draws(k, all_params=True), draws_o[k].reshape(-1, P), rtol=0, atol=1e-100)
    eng.close()


@pytest.mark.parametrize('chains', [4, 3])
def test_yielded_passes_do_not_change_the_draws(monkeypatch, chains):
    """Layout 7: a state wave whose subtree / transition bookkeeping would outlast the team's pass lets the pass go by (two
    barriers, its job untouched: csrc/nuts_duo.hip sm_yield).  Only the number of passes depends on the clock: never
    (EPX_YIELD=0), by the default budget, and at EVERY yield point (EPX_YIELD=1: the barrier alternation under the most
    yields there can be, chains leaving at different times and a workgroup with a chain missing included) give the same
    draws, last states and statistics -- unpieced and from the piece queue."""
    K, it = 6, 24
    X, y, k_lim, Oms, mus, d, P = _site_problem('m4b_sg', 32, 500, 11, K=K, tight=1000.)
    eng, Om_dev, mu_dev = _engine_with_cavity('m4b_sg', X, y, k_lim, Oms, mus)
    seeds = np.arange(K, dtype=np.int64) * 5 + 1
    opts = HipEngine.sampler_opts(chains=chains, iter=it, init='random', layout=7)

    def run(pieces):
        eng.set_piece_queue(pieces, None)
        eng.sample_batch(seeds, opts)
        assert eng.last_layout() == 7
        return [eng.get_draws(k, all_params=True).copy() for k in range(K)], eng.get_chain_stats(chains).copy()

    ref = {}
    for y_cycles in ('0', None, '1'):
        if y_cycles is None:
            monkeypatch.delenv('EPX_YIELD', raising=False)
        else:
            monkeypatch.setenv('EPX_YIELD', y_cycles)
        for pieces in (0, 4):
            dr, cs = run(pieces)
            if not ref:
                ref['dr'], ref['cs'] = dr, cs
                continue
            for k in range(K):
                np.testing.assert_array_equal(dr[k], ref['dr'][k])
            np.testing.assert_array_equal(cs, ref['cs'])
    monkeypatch.delenv('EPX_YIELD', raising=False)
    eng.set_piece_queue(0)
    eng.close()


def test_layout6_handoffs_repeat_bit_for_bit():
    """Layout 6 (one chain per workgroup, C2's kernel): a look at a hand-off word asks for the data behind it as well, and
    words are stored without waiting for the data stores of the same wave -- both lean on the LDS performing one wave's
    operations in issue order (csrc/nuts_duo.hip, duo_publish_c).  A read that overtook its word would show as a draw
    that depends on timing: 64 sites x 4 chains (every CU busy, as in the C2 launch), 40 repetitions of the launch, every
    repetition bit-equal to the first (layout 6 against the oracle: test_gpu_round2.py, test_gpu_parity.py)."""
    K, D, n, it = 64, 16, 200, 80
    rng = np.random.RandomState(12)
    X = rng.randn(K * n, D)
    y = (rng.rand(K * n) < 0.5).astype(int)
    eng = HipEngine('m4b_sg', X, y, np.arange(K + 1) * n)
    d, P = eng.d, eng.P
    eng.set_prior(np.eye(d), np.zeros(d))
    eng.set_global(np.eye(d) * 3.0, np.zeros(d))
    assert np.all(eng.cavity_batch(0))
    seeds = np.arange(K, dtype=np.int64) * 11 + 7
    opts = HipEngine.sampler_opts(chains=4, iter=it, init='random', layout=6)

    def state():
        return np.stack([eng.get_draws(k, all_params=True) for k in range(K)]), eng.get_chain_stats(4).copy()

    eng.sample_batch(seeds, opts)
    assert eng.last_layout() == 6
    dr0, cs0 = state()
    assert np.all(np.isfinite(dr0)) and np.all(cs0[:, :, 7] == 0)
    for rep in range(40):
        eng.sample_batch(seeds, opts)
        dr, cs = state()
        assert np.array_equal(dr, dr0) and np.array_equal(cs, cs0), rep
    eng.close()
