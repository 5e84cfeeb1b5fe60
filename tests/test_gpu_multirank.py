"""The N>1 path on the device: the in-library RCCL communicator (dist.EpxComm, epx_comm_init)
and the host-buffer transport (dist.TorchComm over gloo) in front of the HIP engine.

  * world size 1 over RCCL runs on any GPU box: the all-reduces of epx_update_trial execute in
    stream order and must leave every result bit-equal to the single-process path;
  * two ranks SHARING device 0 over gloo drive HipEngine's host-buffer branch (RCCL itself
    refuses two ranks on one device);
  * the same two ranks with dist.HostComm: the library's OWN multi-rank code (epx_update_trial with its
    per-rank statistics slots, site offsets and flag reductions -- what runs over RCCL on 8 GPUs) with every
    collective handed to gloo through epx_comm_init_host;
  * two ranks over RCCL need two devices: skipped on a 1-GPU box.
References: /root/reference/epstan/method.py:1073-1074 (the reduction), :1145 (the flags)."""

import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.fixture(scope='module')
def runs():
    return np.load(os.path.join(ROOT, 'tests', 'golden', 'master_run.npz'))


def _g6_master(runs, scenario, df0, nsites, comm=None, **kw):
    import injectors
    from epstan_amd.method import Master
    Nj = runs['g6_Nj'][:nsites]
    nrow = int(Nj.sum())
    M = Master('m1b_sg', runs['g6_X'][:nrow], runs['g6_y'][:nrow], site_sizes=Nj,
               prior={'Q': runs['g6_Q0'], 'r': runs['g6_r0']}, A_k={'site_id': np.arange(nsites)},
               chains=4, iter=200, df0=df0, comm=comm, **kw)
    M._sample_injector = injectors.GaussianTilted(scenario)
    return M


def _wide_problem(nsites=16, n=40, D=4):
    """A synthetic problem with enough sites for eight ranks (the goldens have four)."""
    rng = np.random.RandomState(77)
    X = rng.randn(nsites * n, D)
    y = (rng.rand(nsites * n) < 0.55).astype(int)
    return X, y, np.full(nsites, n)


def _wide_master(scenario, df0, comm=None, **kw):
    import injectors
    from epstan_amd.method import Master
    X, y, Nj = _wide_problem()
    M = Master('m1b_sg', X, y, site_sizes=Nj, prior={'Q': np.eye(5) * 0.25, 'r': np.zeros(5)},
               A_k={'site_id': np.arange(len(Nj))}, chains=4, iter=200, df0=df0, comm=comm, **kw)
    M._sample_injector = injectors.GaussianTilted(scenario)
    return M


@pytest.mark.parametrize('tag,scenario,niter,df0,nsites', [('smooth', 'smooth', 12, 0.5, 4),
                                                           ('decay', 'wide_first', 4, 1.0, 3)])
def test_rccl_world_of_one_equals_the_local_path_and_the_goldens(runs, tag, scenario, niter, df0, nsites):
    from epstan_amd import dist
    comm = dist.EpxComm(rank=0, world=1)
    M = _g6_master(runs, scenario, df0, nsites, comm=comm)
    assert comm.size() == (0, 1)                       # RCCL's own view of the communicator
    info, (m_s, S_s) = M.run(niter, verbose=False, seed=1)
    L = _g6_master(runs, scenario, df0, nsites)
    info_l, (m_l, S_l) = L.run(niter, verbose=False, seed=1)
    assert info == info_l == int(runs['g6_%s_info' % tag])
    for a, b in ((m_s, m_l), (S_s, S_l), (M.Qi, L.Qi), (M.ri, L.ri), (M.Q, L.Q), (M.r, L.r)):
        np.testing.assert_array_equal(a, b)            # a one-rank all-reduce must not change a bit
    np.testing.assert_allclose(m_s, runs['g6_%s_m' % tag], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(S_s, runs['g6_%s_S' % tag], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(M.Qi, runs['g6_%s_Qi' % tag], rtol=1e-8, atol=1e-9)
    assert M.df_log == L.df_log
    comm.close()


def test_rccl_world_of_one_with_the_real_sampler():
    from epstan_amd import dist, models
    from epstan_amd.method import Master
    mod = models.m4b(6, 3, 60)
    data = mod.simulate_data(Sigma_x='rand', rng=100)
    _, _, Q0, r0 = mod.get_prior()
    kw = dict(site_sizes=data.Nj, prior={'Q': Q0, 'r': r0}, chains=4, iter=120, df0=0.4)
    comm = dist.EpxComm(rank=0, world=1)
    M = Master('m4b_sg', data.X, data.y, comm=comm, **kw)
    L = Master('m4b_sg', data.X, data.y, **kw)
    out = [m.run(2, verbose=False, return_analytics=True, seed=3) for m in (M, L)]
    assert out[0][0] == out[1][0] == 0
    np.testing.assert_array_equal(out[0][1][0], out[1][1][0])
    np.testing.assert_array_equal(out[0][1][1], out[1][1][1])
    np.testing.assert_array_equal(M.Qi, L.Qi)
    for a, b in zip(out[0][2][:3], out[1][2][:3]):     # stimes differ (clock), msteps / mrhats must not
        if a is not out[0][2][0]:
            np.testing.assert_array_equal(a, b)
    # the small host-side collectives of the communicator
    assert comm.allreduce_min_int(1) == 1
    np.testing.assert_array_equal(comm.allreduce_max(np.array([1.5, -2.0])), [1.5, -2.0])
    np.testing.assert_array_equal(comm.allgather_sites(M.Qi, M.K), M.Qi)
    comm.close()


# ------------------------------------------------------------------ two processes
def _worker(rank, world, port, mode, outdir, transport):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ['RANK'] = str(rank)
    os.environ['WORLD_SIZE'] = str(world)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    if transport in ('gloo', 'host'):
        # several PROCESSES on the box's one device: the driver may park a workgroup of a pieced launch that holds a site
        # for as long as another process's launch lasts (tens of seconds at the C5 shard), and the claim's lost-piece
        # limit is wall time -- 60 s on a device of one's own (epx_pieces.h); seen once in ~10 runs of the two-shard test
        os.environ.setdefault('EPX_PIECE_WAIT_S', '900')
    for p in (ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')):
        if p not in sys.path:
            sys.path.insert(0, p)
    from epstan_amd import dist, models
    from epstan_amd.method import Master
    if transport in ('gloo', 'host'):
        import torch.distributed as tdist
        tdist.init_process_group('gloo', rank=rank, world_size=world)
        comm = dist.TorchComm()
        if transport == 'host':
            # the library's own multi-rank code (fused update, per-rank slots, site offsets) over gloo
            comm = dist.HostComm(comm)
        device = 0                                     # both ranks share the one GPU
    else:
        comm = dist.EpxComm(rank=rank, world=world, port=port)
        device = rank
    runs = np.load(os.path.join(ROOT, 'tests', 'golden', 'master_run.npz'))
    if mode == 'injected':
        M = _g6_master(runs, 'smooth', 0.5, 4, comm=comm, device=device)
        info, (m_s, S_s) = M.run(12, verbose=False, seed=1)
        np.savez(os.path.join(outdir, 'r%d.npz' % rank), info=info, m=m_s, S=S_s, Qi=M.Qi, ri=M.ri, Q=M.Q,
                 klo=M.k_lo, khi=M.k_hi)
    elif mode == 'decay':
        M = _g6_master(runs, 'wide_first', 1.0, 3, comm=comm, device=device)
        info, (m_s, S_s) = M.run(4, verbose=False, seed=1)
        np.savez(os.path.join(outdir, 'r%d.npz' % rank), info=info, m=m_s, S=S_s, Qi=M.Qi, ri=M.ri, Q=M.Q,
                 klo=M.k_lo, khi=M.k_hi)
    elif mode in ('wide', 'wide_decay'):
        M = _wide_master('smooth' if mode == 'wide' else 'wide_first', 0.5 if mode == 'wide' else 1.0, comm=comm, device=device)
        info, (m_s, S_s) = M.run(6, verbose=False, seed=1)
        np.savez(os.path.join(outdir, 'r%d.npz' % rank), info=info, m=m_s, S=S_s, Qi=M.Qi, ri=M.ri, Q=M.Q,
                 klo=M.k_lo, khi=M.k_hi, df=np.array(M.df_log))
    elif mode in ('c4', 'c5x2'):
        # BASELINE config C4 at its OWN size (J = 4096, D = 32, n_j = 500: eight shards of 512 sites), or two of C5's
        # shards (J = 1024, D = 128, n_j = 2000), every rank on device 0: the real sampler, the timed kernels and launch
        # forms (layout 7 / 3 from the piece queue), the library's own multi-rank update
        J, D, n, cor, estim, it = (4096, 32, 500, True, 'sample', 1) if mode == 'c4' else (1024, 128, 2000, False, 'olse', 1)
        mod = models.m4b(J, D, n)
        data = mod.simulate_data(Sigma_x='rand', rng=100) if cor else mod.simulate_data(rng=100)
        _, _, Q0, r0 = mod.get_prior()
        M = Master(mod.site_model, data.X, data.y, site_sizes=data.Nj, prior={'Q': Q0, 'r': r0}, chains=4, iter=200,
                   prec_estim=estim, df0=models.default_df0(J), comm=comm, device=device, sync_sites=False)
        info, (m_s, S_s), an = M.run(it, verbose=False, return_analytics=True, seed=1)
        lo, hi = M.k_lo, M.k_hi
        eng = M.engine
        stats = M.last_site_stats
        k = 137
        samp = eng.get_draws(k)
        Mat, vec, nsamp = eng.get_tilted(k)
        c = samp - samp.mean(axis=0)
        np.savez(os.path.join(outdir, 'r%d.npz' % rank), info=info, m=m_s, S=S_s, Q=M.Q, r=M.r, Q0=M.Q0, r0=M.r0,
                 Qi_sum=M.Qi[:, :, lo:hi].sum(axis=2), ri_sum=M.ri[:, lo:hi].sum(axis=1), klo=lo, khi=hi,
                 fails=stats[:, 7].sum(), min_leapfrogs=stats[:, 2].min(), layout=eng.last_layout(), pieces=eng.last_segments(),
                 nsamp=nsamp, mean_err=np.abs(vec - samp.mean(axis=0)).max(), scat_err=np.abs(Mat - c.T.dot(c)).max() / np.abs(Mat).max(),
                 ms=M.sampling_ms[-1], msteps=an[1], mrhats=an[2], stimes=an[0])
    else:
        nsite = 16 if mode == 'nuts16' else 6
        mod = models.m4b(nsite, 3, 60)
        data = mod.simulate_data(Sigma_x='rand', rng=100)
        _, _, Q0, r0 = mod.get_prior()
        M = Master('m4b_sg', data.X, data.y, site_sizes=data.Nj, prior={'Q': Q0, 'r': r0},
                   chains=4, iter=120, df0=0.4, comm=comm, device=device)
        info, (m_s, S_s), an = M.run(1, verbose=False, return_analytics=True, seed=3)
        np.savez(os.path.join(outdir, 'r%d.npz' % rank), info=info, m=m_s, S=S_s, Qi=M.Qi, ri=M.ri, Q=M.Q,
                 klo=M.k_lo, khi=M.k_hi, msteps=an[1], mrhats=an[2])
    if transport == 'host':
        assert M._fused and comm.size() == (rank, world)
    comm.barrier()
    if transport != 'gloo':
        comm.close()
    if transport in ('gloo', 'host'):
        import torch.distributed as tdist
        tdist.destroy_process_group()


def _spawn(mode, tmp_path, transport, world=2):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(world, port, mode, str(tmp_path), transport), nprocs=world, join=True)
    return [np.load(os.path.join(str(tmp_path), 'r%d.npz' % r)) for r in range(world)]


def _transports():
    from epstan_amd import _lib
    out = [pytest.param('gloo', id='gloo-shared-gpu'), pytest.param('host', id='library-collectives-over-gloo-shared-gpu')]
    two = _lib.device_count() >= 2
    out.append(pytest.param('rccl', id='rccl-2gpu',
                            marks=pytest.mark.skipif(not two, reason='RCCL needs one device per rank')))
    return out


@pytest.mark.parametrize('transport', _transports())
@pytest.mark.parametrize('mode,tag', [('injected', 'smooth'), ('decay', 'decay')])
def test_two_ranks_reproduce_reference_trajectory_on_gpu(runs, tmp_path, mode, tag, transport):
    res = _spawn(mode, tmp_path, transport)
    K = runs['g6_%s_Qi' % tag].shape[2]
    assert [(int(r['klo']), int(r['khi'])) for r in res] == [(0, K // 2), (K // 2, K)]
    for r in res:
        assert int(r['info']) == int(runs['g6_%s_info' % tag])
        np.testing.assert_allclose(r['m'], runs['g6_%s_m' % tag], rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(r['S'], runs['g6_%s_S' % tag], rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(r['Qi'], runs['g6_%s_Qi' % tag], rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(r['ri'], runs['g6_%s_ri' % tag], rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(r['Q'], runs['g6_%s_Q' % tag], rtol=1e-8, atol=1e-9)
    np.testing.assert_array_equal(res[0]['Qi'], res[1]['Qi'])      # gathered site arrays agree bit for bit


@pytest.mark.parametrize('transport', _transports())
def test_sharding_does_not_change_the_first_iteration_on_gpu(tmp_path, transport):
    """Seeds are indexed by GLOBAL site id: one rank and two ranks sample the same draws in the
    first iteration (identical cavities), so the site updates agree to reduction-order rounding."""
    from epstan_amd import models
    from epstan_amd.method import Master
    res = _spawn('nuts', tmp_path, transport)
    mod = models.m4b(6, 3, 60)
    data = mod.simulate_data(Sigma_x='rand', rng=100)
    _, _, Q0, r0 = mod.get_prior()
    M = Master('m4b_sg', data.X, data.y, site_sizes=data.Nj, prior={'Q': Q0, 'r': r0}, chains=4, iter=120, df0=0.4)
    info, (m_s, S_s), an = M.run(1, verbose=False, return_analytics=True, seed=3)
    for r in res:
        assert int(r['info']) == info == 0
        np.testing.assert_allclose(r['m'], m_s, rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(r['Qi'], M.Qi, rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(r['msteps'], an[1], rtol=1e-12)
        np.testing.assert_allclose(r['mrhats'], an[2], rtol=1e-12)


# ------------------------------------------------------------------ the driver's world size on one device
def test_four_ranks_one_site_each_reproduce_the_reference_trajectory(runs, tmp_path):
    """The four-site golden trajectory with one site per rank: four processes share device 0 and run the library's own
    multi-rank code (epx_update_trial with per-rank statistics slots, site offsets, flag reductions) over gloo."""
    res = _spawn('injected', tmp_path, 'host', world=4)
    assert [(int(r['klo']), int(r['khi'])) for r in res] == [(0, 1), (1, 2), (2, 3), (3, 4)]
    for r in res:
        assert int(r['info']) == int(runs['g6_smooth_info'])
        np.testing.assert_allclose(r['m'], runs['g6_smooth_m'], rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(r['S'], runs['g6_smooth_S'], rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(r['Qi'], runs['g6_smooth_Qi'], rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(r['Q'], runs['g6_smooth_Q'], rtol=1e-8, atol=1e-9)


@pytest.mark.parametrize('mode,scenario,df0', [('wide', 'smooth', 0.5), ('wide_decay', 'wide_first', 1.0)])
def test_eight_ranks_equal_one_rank_with_the_injected_sampler(tmp_path, mode, scenario, df0):
    """World size 8 (the driver's node) on one device: 16 sites, two per rank, the deterministic injected sampler --
    a smooth run and one whose first site drives the damping decay (non-positive-definite cavities on some ranks only:
    the flag reduction decides for all).  Trajectory, site parameters and damping factors equal the one-rank run's."""
    res = _spawn(mode, tmp_path, 'host', world=8)
    L = _wide_master(scenario, df0)
    info, (m_s, S_s) = L.run(6, verbose=False, seed=1)
    assert [(int(r['klo']), int(r['khi'])) for r in res] == [(2 * i, 2 * i + 2) for i in range(8)]
    if mode == 'wide_decay':
        assert min(L.df_log) < 1.0                      # the decay branch was taken
    for r in res:
        assert int(r['info']) == info
        np.testing.assert_allclose(r['df'], L.df_log, rtol=0, atol=0)
        np.testing.assert_allclose(r['m'], m_s, rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(r['S'], S_s, rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(r['Qi'], L.Qi, rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(r['Q'], L.Q, rtol=1e-9, atol=1e-10)
    for r in res[1:]:
        np.testing.assert_array_equal(r['Qi'], res[0]['Qi'])


def test_eight_ranks_equal_one_rank_with_the_real_sampler(tmp_path):
    """The same with the device sampler: seeds are indexed by global site id, so eight ranks sample the draws one
    rank samples in the first iteration."""
    from epstan_amd import models
    from epstan_amd.method import Master
    res = _spawn('nuts16', tmp_path, 'host', world=8)
    mod = models.m4b(16, 3, 60)
    data = mod.simulate_data(Sigma_x='rand', rng=100)
    _, _, Q0, r0 = mod.get_prior()
    M = Master('m4b_sg', data.X, data.y, site_sizes=data.Nj, prior={'Q': Q0, 'r': r0}, chains=4, iter=120, df0=0.4)
    info, (m_s, S_s), an = M.run(1, verbose=False, return_analytics=True, seed=3)
    for r in res:
        assert int(r['info']) == info == 0
        np.testing.assert_allclose(r['m'], m_s, rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(r['Qi'], M.Qi, rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(r['msteps'], an[1], rtol=1e-12)
        np.testing.assert_allclose(r['mrhats'], an[2], rtol=1e-12)


def _own_size_checks(res, world, J, layout):
    per = J // world
    assert [(int(r['klo']), int(r['khi'])) for r in res] == [(per * i, per * (i + 1)) for i in range(world)]
    Qsum = sum(r['Qi_sum'] for r in res)
    rsum = sum(r['ri_sum'] for r in res)
    for r in res:
        assert int(r['info']) == 0 and int(r['layout']) == layout and int(r['pieces']) < 0        # the timed kernel, from the piece queue
        assert float(r['fails']) == 0 and float(r['min_leapfrogs']) > 0 and int(r['nsamp']) == 400
        assert float(r['mean_err']) < 1e-10 and float(r['scat_err']) < 1e-8
        m, S = r['m'][-1], r['S'][-1]
        assert np.all(np.isfinite(m)) and np.all(np.isfinite(S))
        np.testing.assert_allclose(S, S.T, rtol=1e-9, atol=1e-13)
        assert np.linalg.eigvalsh(S)[0] > 0
        # the replicated global approximation is the SAME on every rank (one all-reduce, method.py:1073-1074) ...
        for key in ('Q', 'r', 'm', 'S', 'msteps', 'mrhats'):
            np.testing.assert_array_equal(r[key], res[0][key])
        # ... and is the prior plus the sum of every rank's accepted site parameters
        np.testing.assert_allclose(r['Q'], r['Q0'] + Qsum, rtol=1e-9, atol=1e-8)
        np.testing.assert_allclose(r['r'], r['r0'] + rsum, rtol=1e-9, atol=1e-7)


def test_c4_at_its_own_size_eight_ranks_on_one_device(tmp_path):
    """BASELINE config C4: J = 4096 sites, D = 32, n_j = 500, sharded over 8 ranks (512 sites each, = C3 per rank), one EP
    iteration with the real sampler -- all eight ranks on the one device of the box, the library's own multi-rank code
    (epx_update_trial: per-rank statistics slots, global site offsets, flag reductions) with every collective handed to
    gloo: everything that runs on the 8-GPU node except ncclAllReduce itself.  Invariants of the C3-size test on every rank,
    bit-equal replicated Q, r, S, m, and Q = Q0 + the sum over ALL ranks' sites (method.py:1073-1074, 1145;
    experiment/fit.py:326-335)."""
    res = _spawn('c4', tmp_path, 'host', world=8)
    _own_size_checks(res, 8, 4096, 7)
    print('C4 on one device: sampling launch per rank %s ms (eight launches share the device)' % np.round([float(r['ms']) for r in res], 0))


def test_two_c5_shards_two_ranks_on_one_device(tmp_path):
    """Two of BASELINE config C5's eight shards (2 x 512 sites, D = 128, n_j = 2000, d = 258, prec_estim='olse') as two
    ranks on the one device: the streaming sampler from the piece queue and the d = 258 update phase through the library's
    multi-rank code.  (C5's own J = 4096 is eight such shards: 8 x 35 s of sampling per EP iteration on one device plus
    8 x 8.4 GB of host data do not fit this suite's time; the per-rank code path is the one exercised here.)"""
    res = _spawn('c5x2', tmp_path, 'host', world=2)
    _own_size_checks(res, 2, 1024, 3)



def test_bench_main_two_ranks_on_one_device_with_the_parity_and_cpu_legs():
    """`bench.py --gpus 2` end to end on the device, at its default form (parity iteration + CPU leg ON): two ranks share
    GPU 0, the library's own multi-rank update with its collectives over gloo (tests/bench_gpu_hostcomm.py).  The parity
    iteration behind the timed region is a collective EP iteration: both ranks must still be there for it (VERDICT round 5:
    rank 0 ran it alone), and rank 0's CPU leg behind it must need nobody.  One line, every record filled in."""
    import json
    import subprocess
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(ROOT, 'tests', 'bench_gpu_hostcomm.py'),
           '--gpus', '2', '--sites', '6', '--D', '8', '--rows', '80', '--siter', '60', '--steps', '2', '--warmup', '1',
           '--cpu-sites', '4', '--parity-sites', '4', '--cpu-seq-sites', '1', '--no-secondary']
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('EPX_PIECE_WAIT_S', '900')          # (two processes on one device: see _worker)
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, text=True)
    assert res.returncode == 0, res.stderr[-4000:]
    lines = [l for l in res.stdout.splitlines() if l.strip().startswith('{')]
    assert len(lines) == 1, res.stdout[-2000:]
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['config']['rccl_world_size'] == 2 and out['steps'] == 2
    np.testing.assert_allclose(out['value'], 12 * 2 / (out['ms_per_step'] * 1e-3 * 2), rtol=1e-9)
    assert out['roofline']['frac'] > 0 and out['roofline']['gradients_per_launch'] > 0
    assert out['cpu_baseline']['value'] is not None and out['cpu_baseline']['value'] > 0, out['cpu_baseline']
    assert 1 <= out['cpu_baseline']['cores'] <= out['cpu_baseline']['threads_used']
    par = out['parity']
    assert par is not None and par['sites'] == 4, res.stderr[-3000:]
    assert par['site_delta_vs_numpy_moment_stage_max_rel_err'] < 1e-7
    assert par['first_draws'] is None or par['first_draws']['max_rel_err'] < 1e-6
    by_t = par['transition_by_transition']
    assert by_t is None or by_t['chains_equal_through_the_first_transition'] >= 0.9 * by_t['chains']
