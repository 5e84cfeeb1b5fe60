"""The N>1 path on CPU: world_size-2 gloo process group, sites sharded over the
ranks, ONE all-reduce of the packed site sums per EP iteration.  The oracle
stands in for the device engine; the host logic (epstan_amd.method / dist) is
the code under test."""

import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, mode, outdir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ['RANK'] = str(rank)
    os.environ['WORLD_SIZE'] = str(world)
    for p in (ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as tdist
    from epstan_amd import dist, models
    from epstan_amd.method import Master
    from oracle.engine_oracle import OracleEngine
    import injectors
    tdist.init_process_group('gloo', rank=rank, world_size=world)
    comm = dist.TorchComm()
    fac = lambda m, X, y, kl, **g: OracleEngine(m, X, y, kl, nthreads=2, **g)
    runs = np.load(os.path.join(ROOT, 'tests', 'golden', 'master_run.npz'))
    if mode == 'injected':
        M = Master('m1b_sg', runs['g6_X'], runs['g6_y'], site_sizes=runs['g6_Nj'],
                   prior={'Q': runs['g6_Q0'], 'r': runs['g6_r0']}, A_k={'site_id': np.arange(4)},
                   chains=4, iter=200, df0=0.5, comm=comm, _engine_factory=fac)
        M._sample_injector = injectors.GaussianTilted('smooth')
        info, (m_s, S_s) = M.run(12, verbose=False, seed=1)
        np.savez(os.path.join(outdir, 'r%d.npz' % rank), info=info, m=m_s, S=S_s, Qi=M.Qi, ri=M.ri,
                 Q=M.Q, klo=M.k_lo, khi=M.k_hi)
    elif mode == 'decay':
        Nj = runs['g6_Nj'][:3]; nrow = int(Nj.sum())
        M = Master('m1b_sg', runs['g6_X'][:nrow], runs['g6_y'][:nrow], site_sizes=Nj,
                   prior={'Q': runs['g6_Q0'], 'r': runs['g6_r0']}, A_k={'site_id': np.arange(3)},
                   chains=4, iter=200, df0=1.0, comm=comm, _engine_factory=fac)
        M._sample_injector = injectors.GaussianTilted('wide_first')
        info, (m_s, S_s) = M.run(4, verbose=False, seed=1)
        np.savez(os.path.join(outdir, 'r%d.npz' % rank), info=info, m=m_s, S=S_s, Qi=M.Qi, ri=M.ri,
                 Q=M.Q, klo=M.k_lo, khi=M.k_hi)
    elif mode == 'groups':
        # K < J: 7 groups on 3 sites (sharded 1 + 2), with a damping sweep and the pooled moments
        from epstan_amd.util import distribute_groups
        mod = models.m4b(7, 2, 25)
        data = mod.simulate_data(Sigma_x='rand', rng=100)
        _, _, Q0, r0 = mod.get_prior()
        Nk, Nj_k, j_ind_k = distribute_groups(7, 3, data.Nj)
        M = Master('m4b', data.X, data.y, site_sizes=Nk, A_k={'J': Nj_k}, A_n={'j_ind': j_ind_k + 1},
                   prior={'Q': Q0, 'r': r0}, chains=4, iter=100, df0=0.4, comm=comm, _engine_factory=fac)
        sw = dict(damps=np.array([0.1, 0.4, 0.9, -50.0]), m_target=np.zeros(6), S_target=np.eye(6))
        info, (m_s, S_s) = M.run(1, verbose=False, seed=3, sweep=sw)
        Sm, mm = M.mix_phi()
        np.savez(os.path.join(outdir, 'r%d.npz' % rank), info=info, m=m_s, S=S_s, Qi=M.Qi, ri=M.ri, Q=M.Q,
                 klo=M.k_lo, khi=M.k_hi, kls=M.sweep_log[0]['kls'], cav=M.sweep_log[0]['cav_pd'], Sm=Sm, mm=mm)
    else:   # real sampler (C oracle NUTS), 6 sites of m4b
        mod = models.m4b(6, 3, 60)
        data = mod.simulate_data(Sigma_x='rand', rng=100)
        _, _, Q0, r0 = mod.get_prior()
        M = Master('m4b_sg', data.X, data.y, site_sizes=data.Nj, prior={'Q': Q0, 'r': r0},
                   chains=4, iter=120, df0=0.4, comm=comm, _engine_factory=fac)
        info, (m_s, S_s), an = M.run(1, verbose=False, return_analytics=True, seed=3)
        Qi1 = M.Qi.copy()
        info2 = M.run(1, verbose=False, calc_moments=False, seed=4)
        np.savez(os.path.join(outdir, 'r%d.npz' % rank), info=info, info2=info2, m=m_s, S=S_s, Qi=Qi1,
                 ri=M.ri, Q=M.Q, klo=M.k_lo, khi=M.k_hi, msteps=an[1], mrhats=an[2])
    tdist.barrier()
    tdist.destroy_process_group()


def _spawn(mode, tmp_path, world=2):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(world, port, mode, str(tmp_path)), nprocs=world, join=True)
    return [np.load(os.path.join(str(tmp_path), 'r%d.npz' % r)) for r in range(world)]


@pytest.mark.parametrize('mode,tag', [('injected', 'smooth'), ('decay', 'decay')])
def test_two_ranks_reproduce_reference_trajectory(tmp_path, mode, tag):
    runs = np.load(os.path.join(ROOT, 'tests', 'golden', 'master_run.npz'))
    res = _spawn(mode, tmp_path)
    K = runs['g6_%s_Qi' % tag].shape[2]
    assert [(int(r['klo']), int(r['khi'])) for r in res] == [(0, K // 2), (K // 2, K)]
    for r in res:       # every rank holds the same global state and the gathered site arrays
        assert int(r['info']) == int(runs['g6_%s_info' % tag])
        np.testing.assert_allclose(r['m'], runs['g6_%s_m' % tag], rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(r['S'], runs['g6_%s_S' % tag], rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(r['Qi'], runs['g6_%s_Qi' % tag], rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(r['ri'], runs['g6_%s_ri' % tag], rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(r['Q'], runs['g6_%s_Q' % tag], rtol=1e-8, atol=1e-9)


def test_sharding_does_not_change_results_with_real_sampler(tmp_path):
    """Seeds are indexed by GLOBAL site id, so 1 rank and 2 ranks sample the same
    draws in the first iteration (identical cavities).  Later iterations see
    cavities that differ by the summation order of the site reduction (1e-16),
    which chaotic HMC trajectories amplify, so only iteration 1 is compared
    exactly."""
    sys.path.insert(0, ROOT)
    from epstan_amd import models
    from epstan_amd.method import Master
    from oracle.engine_oracle import OracleEngine
    res = _spawn('nuts', tmp_path)
    mod = models.m4b(6, 3, 60)
    data = mod.simulate_data(Sigma_x='rand', rng=100)
    _, _, Q0, r0 = mod.get_prior()
    M = Master('m4b_sg', data.X, data.y, site_sizes=data.Nj, prior={'Q': Q0, 'r': r0},
               chains=4, iter=120, df0=0.4,
               _engine_factory=lambda m, X, y, kl: OracleEngine(m, X, y, kl, nthreads=2))
    info, (m_s, S_s), an = M.run(1, verbose=False, return_analytics=True, seed=3)
    for r in res:
        assert int(r['info']) == info == 0 and int(r['info2']) == 0
        np.testing.assert_allclose(r['m'], m_s, rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(r['Qi'], M.Qi, rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(r['msteps'], an[1], rtol=1e-12)     # max over ALL sites (method.py:1044)
        np.testing.assert_allclose(r['mrhats'], an[2], rtol=1e-12)


def test_two_ranks_with_groups_sweep_and_mix(tmp_path):
    """K < J sharded over two ranks (the group limits are cut at the rank boundary), the damping
    sweep's flags min-reduced over the ranks, mix_phi's sums all-reduced: all equal to one rank."""
    sys.path.insert(0, ROOT)
    from epstan_amd import models
    from epstan_amd.method import Master
    from epstan_amd.util import distribute_groups
    from oracle.engine_oracle import OracleEngine
    res = _spawn('groups', tmp_path)
    mod = models.m4b(7, 2, 25)
    data = mod.simulate_data(Sigma_x='rand', rng=100)
    _, _, Q0, r0 = mod.get_prior()
    Nk, Nj_k, j_ind_k = distribute_groups(7, 3, data.Nj)
    M = Master('m4b', data.X, data.y, site_sizes=Nk, A_k={'J': Nj_k}, A_n={'j_ind': j_ind_k + 1},
               prior={'Q': Q0, 'r': r0}, chains=4, iter=100, df0=0.4,
               _engine_factory=lambda m, X, y, kl, **g: OracleEngine(m, X, y, kl, nthreads=2, **g))
    sw = dict(damps=np.array([0.1, 0.4, 0.9, -50.0]), m_target=np.zeros(6), S_target=np.eye(6))
    info, (m_s, S_s) = M.run(1, verbose=False, seed=3, sweep=sw)
    Sm, mm = M.mix_phi()
    assert [(int(r['klo']), int(r['khi'])) for r in res] == [(0, 1), (1, 3)]
    for r in res:
        assert int(r['info']) == info == 0
        np.testing.assert_allclose(r['m'], m_s, rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(r['Qi'], M.Qi, rtol=1e-9, atol=1e-10)
        np.testing.assert_array_equal(r['cav'], M.sweep_log[0]['cav_pd'])
        np.testing.assert_allclose(r['kls'], M.sweep_log[0]['kls'], rtol=1e-9, equal_nan=True)
        np.testing.assert_allclose(r['mm'], mm, rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(r['Sm'], Sm, rtol=1e-9, atol=1e-12)
    assert M.sweep_log[0]['cav_pd'][0] and not M.sweep_log[0]['cav_pd'][-1]


# ------------------------------------------------------------------ the driver's launch, on CPU
def _run_torchrun(n, script, args, timeout=600):
    import subprocess
    port = _free_port()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
           '--master-addr', '127.0.0.1', '--master-port', str(port), script] + args
    env = dict(os.environ)
    env['OMP_NUM_THREADS'] = '1'
    return subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout, text=True)


def _bench_line(res):
    import json
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.strip().startswith('{')]
    assert len(lines) == 1, res.stdout
    return json.loads(lines[0])


@pytest.mark.parametrize('world', [2, 8])
def test_bench_runs_end_to_end_on_cpu_with_the_parity_and_cpu_legs(world):
    """`bench.py --gpus N` as the driver launches it (torch.distributed.run, one process per rank), with the oracle
    engine and a gloo transport in place of the device, at its DEFAULT form: the parity EP iteration and the CPU leg
    behind the timed region are ON.  The parity iteration is collective (the all-reduces of the update phase), so every
    rank has to stay for it; only behind it may ranks != 0 leave.  (Round 5 shipped the opposite order: rank 0 ran the
    collective alone, 'Connection closed by peer' over gloo, a hang over RCCL -- and this test passed --cpu-sites 0.)
    ONE JSON line from rank 0 with the whole-job value, cpu_baseline and parity filled in, clean exit of every rank."""
    res = _run_torchrun(world, os.path.join(ROOT, 'tests', 'bench_cpu_smoke.py'),
                        ['--gpus', str(world), '--sites', '2', '--D', '3', '--rows', '30', '--siter', '20', '--steps', '2',
                         '--warmup', '1', '--cpu-sites', '2'])
    out = _bench_line(res)
    assert 'Connection closed' not in res.stderr and 'Traceback' not in res.stderr, res.stderr[-3000:]
    assert out['n_gpus'] == world and out['steps'] == 2 and out['warmup'] == 1 and out['scaling'] == 'weak'
    assert out['config']['rccl_world_size'] == world
    assert out['metric'] == 'site-updates/sec' and out['unit'] == 'site-updates/s'
    np.testing.assert_allclose(out['value'], 2 * world * 2 / (out['ms_per_step'] * 1e-3 * 2), rtol=1e-9)     # whole job
    assert 'roofline' in out and out['roofline']['frac'] > 0
    cb = out['cpu_baseline']
    assert cb['value'] is not None and cb['value'] > 0 and cb['kind'] == 'port'
    # the stated core count is what the process can run at once, and the timings have to support it
    assert 1 <= cb['cores'] <= cb['threads_used'] <= (cb['affinity'] or cb['host_threads'])
    assert cb['cgroup_quota_cpus'] is None or cb['threads_used'] <= max(1, int(cb['cgroup_quota_cpus']))
    assert cb['cpus_delivered'] > 0 and cb['us_per_gradient_and_thread'] > 0
    assert out['parity'] is not None and out['parity']['sites'] == 2
    assert out['parity']['site_delta_vs_numpy_moment_stage_max_rel_err'] < 1e-7


def test_bench_without_the_cpu_leg_at_world_size_8_on_cpu():
    """--cpu-sites 0: no parity iteration, no CPU leg; the ranks leave right behind the timed region."""
    res = _run_torchrun(8, os.path.join(ROOT, 'tests', 'bench_cpu_smoke.py'),
                        ['--gpus', '8', '--sites', '2', '--D', '3', '--rows', '30', '--siter', '20', '--steps', '2',
                         '--warmup', '1', '--cpu-sites', '0'])
    out = _bench_line(res)
    assert out['n_gpus'] == 8 and 'cpu_baseline' not in out
    np.testing.assert_allclose(out['value'], 16 * 2 / (out['ms_per_step'] * 1e-3 * 2), rtol=1e-9)


def test_no_rank_leaves_bench_before_the_last_collective():
    """Source-level guard of the same property: in bench.main() every `comm.close()` lies BEHIND the parity iteration
    (`M.run(1, ... seed=PARITY_SEED)`), the last collective of the run."""
    src = open(os.path.join(ROOT, 'bench.py')).read()
    main = src[src.index('def main():'):]
    last_collective = main.index('seed=PARITY_SEED')
    closes = [i for i in range(len(main)) if main.startswith('comm.close()', i)]
    assert closes and all(i > last_collective for i in closes)
    assert 'M.run(' not in main[last_collective + 20:]             # nothing collective behind it on rank 0


def test_cpu_width_is_the_affinity_capped_by_the_cgroup_quota(tmp_path, monkeypatch):
    """bench.cpu_width / host_cpu_limits: 256 hardware threads with a cgroup quota of 16 CPUs (this pool's GPU boxes,
    profiles/r06_cpu_probe.txt) are 16 usable threads; no quota -> the affinity mask; --cpu-threads wins."""
    sys.path.insert(0, ROOT)
    import bench
    lim = {'host_threads': 256, 'affinity': 256, 'cgroup_quota_cpus': 16.0, 'cgroup_source': '/sys/fs/cgroup/cpu.max', 'loadavg_1min': 0.0}
    assert bench.cpu_width(lim)[0] == 16 and 'quota' in bench.cpu_width(lim)[1]
    assert bench.cpu_width(dict(lim, cgroup_quota_cpus=None))[0] == 256
    assert bench.cpu_width(dict(lim, affinity=8))[0] == 8
    assert bench.cpu_width(dict(lim, cgroup_quota_cpus=0.5))[0] == 1
    assert bench.cpu_width(lim, requested=4)[0] == 4
    here = bench.host_cpu_limits()
    assert here['host_threads'] == os.cpu_count() and 1 <= here['affinity'] <= here['host_threads']
    assert here['cgroup_quota_cpus'] is None or here['cgroup_quota_cpus'] > 0


def test_rccl_id_travels_through_torchruns_store():
    """dist.EpxComm hands the RCCL id from rank 0 to the others through the key-value store torchrun's agent serves on
    MASTER_PORT (no second port): four ranks, two binds in a row, every rank ends up with rank 0's bytes."""
    helper = os.path.join(ROOT, 'tests', 'uid_exchange_helper.py')
    res = _run_torchrun(4, helper, [], timeout=300)
    assert res.returncode == 0, res.stderr[-3000:]
    got = sorted(l for l in res.stdout.splitlines() if l.startswith('uid '))
    assert len(got) == 4 and len(set(l.split(' ', 2)[2] for l in got)) == 1, res.stdout


def test_rccl_id_socket_exchange_counts_a_peer_only_after_its_acknowledgement():
    """Without torchrun's store the id goes over a TCP connection to EPX_COMM_PORT; rank 0 keeps serving until every
    peer has acknowledged (a peer that drops the connection before that is served again)."""
    import threading
    from epstan_amd import dist
    port = _free_port()
    uid = bytes(range(128))
    out = {}

    def rank0():
        out[0] = dist.EpxComm(rank=0, world=3, addr='127.0.0.1', port=port)._exchange_id(uid)

    def peer(r, flaky):
        import socket as sk, struct, time
        if flaky:                           # first attempt: ask, then hang up before taking the id
            for _ in range(200):
                try:
                    c = sk.create_connection(('127.0.0.1', port), timeout=5.0)
                    c.sendall(struct.pack('<i', r))
                    c.close()
                    break
                except OSError:
                    time.sleep(0.02)
        out[r] = dist.EpxComm(rank=r, world=3, addr='127.0.0.1', port=port)._exchange_id(b'')

    env_keep = os.environ.pop('TORCHELASTIC_USE_AGENT_STORE', None)
    try:
        ts = [threading.Thread(target=rank0), threading.Thread(target=peer, args=(1, True)), threading.Thread(target=peer, args=(2, False))]
        for t in ts:
            t.start()
        for t in ts:
            t.join(60)
    finally:
        if env_keep is not None:
            os.environ['TORCHELASTIC_USE_AGENT_STORE'] = env_keep
    assert out.get(0) == uid and out.get(1) == uid and out.get(2) == uid
