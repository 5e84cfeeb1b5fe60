#!/usr/bin/env python3
"""Headline benchmark: site-updates/sec (and EP iters/sec) of the EP inner loop on synthetic
hierarchical logistic regression, model m4b, chains=4, iter=200 (S=400 draws per site update).

    python bench.py --gpus N --steps K --warmup W [--config c3|c2|c5shard]

Default workload = BASELINE.json configs[2], the largest single-GPU configuration: 512 sites per
GPU, D=32, n_j=500 (N=1 is C3; N=8 is exactly C4: J=4096 sharded over 8 GPUs, weak scaling).
One process per GPU; a step is one outer EP iteration over all sites: batched NUTS site updates ->
moment stage -> packed site sums -> ONE RCCL all-reduce (inside libepx.so) -> damped update +
cavities -> moments.  Launched by torchrun (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* in the
environment) every rank joins; a plain `python bench.py --gpus N` with N > 1 starts the N ranks
itself (fresh child processes through torch.distributed.run, before anything touches a GPU).
Rank 0 prints ONE JSON line (contract in the task statement).
"""

import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# MI355X_MICROARCH.md: HBM3E 8 TB/s peak; LDS 160 KiB/CU, ~150 TB/s aggregate for ds_read_b64/b128.
# FP64 vector = FP64 matrix peak 78.6 TFLOP/s (AMD MI355X datasheet)
FP64_PEAK_TFLOPS = 78.6
HBM_PEAK_GBS = 8000.0
LDS_PEAK_TBS = 150.0
CLOCK_GHZ = 2.4          # MI355X_MICROARCH.md: max clock (the C3 kernel holds it: SQ_WAVE_CYCLES agree with wall time x 2.4 GHz)

# Hooks for tests/test_multirank_gloo.py only (None = the product path: HipEngine behind Master, RCCL inside libepx.so):
# a CPU run of THIS file at the driver's world size executes the rank-0 JSON assembly, the barriers and the reductions
# around the timed region before the driver depends on them.  Nothing in this file sets them.
_ENGINE_FACTORY = None
_COMM_FACTORY = None

CONFIGS = {
    # name: (sites per GPU, D, n_j, correlated covariates, default steps, default warm-up)
    'c2': (64, 16, 200, 1, 20, 5),
    'c3': (512, 32, 500, 1, 20, 5),      # the driver's command: --steps 20 --warmup 5
    'c5shard': (512, 128, 2000, 0, 2, 2),      # (the secondary record of the default run uses these too: README's number = the driver's)
}


def workload(J, D, n, model, cor_input=True):
    from epstan_amd import models
    mod = models.MODELS[model](J, D, n)
    if cor_input:
        data = mod.simulate_data(Sigma_x='rand', rng=100)      # seed_data=100, cor_input=True (fit.py:157, 235-238)
    else:
        # fit.py's cor_input=False branch; the random vine correlation matrix of the reference
        # (common.py:33-78) is numerically not positive definite from D ~ 120 on
        data = mod.simulate_data(rng=100)
    _, _, Q0, r0 = mod.get_prior()
    return mod, data, Q0, r0


def cpu_model_name():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def host_cpu_limits():
    """What this process may really use of the host's processors: the affinity mask, the cgroup CPU quota (v2 cpu.max, v1
    cfs_quota_us / cfs_period_us; the process's own cgroup first, then the root of the mounted hierarchy) and the load
    other processes already put on the box.  os.cpu_count() sees none of these."""
    lim = {'host_threads': os.cpu_count(), 'affinity': None, 'cgroup_quota_cpus': None, 'cgroup_source': None,
           'loadavg_1min': None}
    try:
        lim['affinity'] = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        pass
    try:
        lim['loadavg_1min'] = float(open('/proc/loadavg').read().split()[0])
    except (OSError, ValueError):
        pass
    rel = []
    try:
        for line in open('/proc/self/cgroup'):
            parts = line.strip().split(':', 2)
            if len(parts) == 3 and (parts[1] == '' or 'cpu' in parts[1].split(',')):
                rel.append(parts[2].lstrip('/'))
    except OSError:
        pass
    cands = []
    for r in rel + ['']:
        # walk up from the process's cgroup to the mounted root: the tightest quota on the way binds
        parts = [x for x in r.split('/') if x]
        for i in range(len(parts), -1, -1):
            sub = '/'.join(parts[:i])
            cands.append(os.path.join('/sys/fs/cgroup', sub, 'cpu.max'))
            cands.append(os.path.join('/sys/fs/cgroup/cpu', sub, 'cpu.cfs_quota_us'))
            cands.append(os.path.join('/sys/fs/cgroup/cpu,cpuacct', sub, 'cpu.cfs_quota_us'))
    best = None
    for path in cands:
        try:
            txt = open(path).read().split()
            if path.endswith('cpu.max'):
                if txt[0] == 'max':
                    continue
                q = float(txt[0]) / float(txt[1])
            else:
                quota = float(txt[0])
                if quota <= 0:
                    continue
                q = quota / float(open(os.path.join(os.path.dirname(path), 'cpu.cfs_period_us')).read().split()[0])
            if best is None or q < best[0]:
                best = (q, path)
        except (OSError, ValueError, IndexError):
            continue
    if best is not None:
        lim['cgroup_quota_cpus'], lim['cgroup_source'] = best
    return lim


def cpu_width(lim, requested=0):
    """Threads of the CPU legs: what the process can really run at once -- the affinity mask capped by the cgroup quota
    (16 CPUs of the 256 hardware threads on this pool's GPU boxes, profiles/r06_cpu_probe.txt: beyond the quota more
    threads only take turns, and gradient throughput FALLS, 1.22e6/s on 64 threads -> 0.75e6/s on 256)."""
    if requested > 0:
        return int(requested), 'requested (--cpu-threads)'
    w = lim['affinity'] or lim['host_threads'] or 1
    why = 'affinity mask'
    q = lim['cgroup_quota_cpus']
    if q is not None and q < w:
        w, why = max(1, int(q)), 'cgroup CPU quota %.4g (%s)' % (q, lim['cgroup_source'])
    return int(w), why


PARITY_SEED = 3      # seed of the one EP iteration behind the timed region that the CPU leg re-does (Master.run(1, seed=...))


def _ess(x):
    """Effective sample size of the chains x (chains, n) for the mean: Geyer's initial positive sequence on the
    chain-averaged autocorrelations."""
    c, n = x.shape
    xc = x - x.mean(axis=1, keepdims=True)
    var = (xc * xc).mean()
    if not var > 0:
        return float(c * n)
    f = np.fft.rfft(xc, 2 * n, axis=1)
    acov = np.fft.irfft(f * np.conj(f), 2 * n, axis=1)[:, :n].mean(axis=0) / n
    rho = acov / acov[0]
    tau, t = 1.0, 1
    while t + 1 < n:
        pair = rho[t] + rho[t + 1]
        if pair <= 0:
            break
        tau += 2.0 * pair
        t += 2
    return float(c * n / max(tau, 1.0 / np.log10(max(c * n, 10))))


def _chi2_sf(x, k):
    """P(chi-square with k degrees of freedom > x)."""
    try:
        from scipy import stats
        return float(stats.chi2.sf(x, k))
    except Exception:                       # Wilson-Hilferty
        from math import erfc, sqrt
        z = ((x / k) ** (1.0 / 3.0) - (1.0 - 2.0 / (9.0 * k))) / sqrt(2.0 / (9.0 * k))
        return 0.5 * erfc(z / sqrt(2.0))


def pooled_z(z, what):
    """The distribution of signed z-scores z (sites, coordinates) of two independent estimates of the same quantities:
    under agreement they scatter like N(0, 1) -- mean 0, variance ~ 1 -- instead of merely staying within 4.  The
    coordinates of a site are correlated (one posterior), the sites are independent: the test of the MEAN is made on the
    per-site means (a t-statistic over the sites), which a bias shared by all coordinates cannot hide in (0.3 standard
    errors on every coordinate give t ~ 4 with 32 sites; `share_within_4` never sees it).  The chi-square of the pooled
    sum of squares is given with its nominal degrees of freedom (indicative: it ignores the within-site correlation and the
    noise of the effective-sample-size estimates in the denominators, both of which fatten the tails)."""
    z = np.asarray(z, dtype=float)
    K, n = z.shape
    site_mean = z.mean(axis=1)
    t = float(site_mean.mean() / (site_mean.std(ddof=1) / np.sqrt(K))) if K > 1 and site_mean.std(ddof=1) > 0 else 0.0
    ss = float((z * z).sum())
    return {'what': what, 'n': int(z.size), 'sites': int(K),
            'mean': float(z.mean()), 'variance': float(z.var(ddof=1)) if z.size > 1 else 0.0,
            'mean_of_site_means_t_statistic': t,
            'chi2': ss, 'chi2_dof': int(z.size), 'chi2_p_value_nominal': _chi2_sf(ss, z.size),
            'share_beyond_2': float(np.mean(np.abs(z) > 2.0)), 'share_beyond_3': float(np.mean(np.abs(z) > 3.0)),
            'expected_under_N01': {'mean': 0.0, 'variance': 1.0, 'share_beyond_2': 0.0455, 'share_beyond_3': 0.0027}}


def snapshot_for_cpu_leg(M, n_all, chains, siter):
    """What the CPU leg starts from, taken BEFORE the parity iteration: the cavities, the chains' last draws and the
    global approximation the device holds after its timed iterations."""
    eng = M.engine
    P = eng.P
    nkeep = siter - siter // 2
    n_all = min(n_all, M.K_local)
    mus = np.stack([eng.get_cavity(k)[1] for k in range(n_all)])
    Oms = np.stack([eng.get_cavity(k)[0] for k in range(n_all)])
    last = np.stack([eng.get_draws(k, all_params=True).reshape(chains, nkeep, P)[:, -1, :] for k in range(n_all)])
    Q, r = eng.get_global()
    return {'n_all': n_all, 'mus': mus, 'Oms': Oms, 'last': last, 'Q': Q, 'r': r}


def cpu_leg(M, snap, estim, chains, siter, n_seq, threads, df, n_par, limits=None):
    """The CPU port (oracle/, kind 'port') on this box's host cores re-does, for the first sites of the workload, the site
    updates of ONE EP iteration that the device has just run behind the timed region (`Master.run(1, seed=PARITY_SEED)`):
    same cavities, same starting draws, same per-site Stan seeds (method.py:342-346), the C restatement of the sampler +
    the NumPy moment stage.  That run is timed (`cpu_baseline`) AND compared with the device's results (`parity`:
    north_star's same-run agreement on the posterior mean / covariance -- the tilted moments of method.py:413-437 that the
    sites contribute, and the global moments of method.py:1211-1219 they add up to).
    TWO BUILDS of the C restatement run (oracle/Makefile): the TIMED one is the fast build (-O3 -march=native, contraction
    allowed, vectorised logistic terms and reductions: the strongest honest CPU number) on all
    `n_all` sites; the COMPARED one is the strict build (-O2, no contraction: the checker of every parity test) on the first
    `n_par` sites -- its time is stated beside the fast one's.  tests/test_nuts_oracle.py holds the two builds together
    (gradients at 1e-9, a site update statistically).
    Two schedules are timed:
      all-cores           (site, chain) pairs handed out dynamically to as many threads as the process can really run at
                          once (cpu_width: affinity mask capped by the cgroup quota) -- `cores` is that number, and the record
                          carries the CPUs the kernel actually delivered (process CPU time / wall time of the leg) and the
                          per-thread cost beside the 4-thread leg's, so the count is checked by the timings;
      reference-faithful  sites strictly one after the other, the 4 chains of a site on 4 threads
                          (PyStan n_jobs=-1 inside method.py:1005-1023), plus the reference's own
                          "limiting sampling time" = max over sites (method.py:1043)."""
    from epstan_amd.engine import DQI
    from epstan_amd.seeds import run_seeds, stan_seeds
    from oracle import ep_oracle as eo
    from oracle import nuts_oracle as no
    no.build()
    eng = M.engine
    P, d = eng.P, eng.d
    nkeep = siter - siter // 2
    n_all = snap['n_all']
    n_seq = min(n_seq, n_all)
    mus, Oms, last, Q, r = snap['mus'], snap['Oms'], snap['last'], snap['Q'], snap['r']
    lim = np.asarray(M.k_lim[:n_all + 1], dtype=np.int64)
    X, y = M.X[:lim[-1]], M.y[:lim[-1]]
    # the Stan seeds the device used for these sites in the parity iteration
    seeds = stan_seeds(run_seeds(PARITY_SEED, 1, M.K)[0, M.k_lo:M.k_hi])[:n_all].astype(np.int64)
    limits = limits or host_cpu_limits()
    nthr, width_why = cpu_width(limits, threads)

    trace_c = [None]

    def site_update(ks, nt, trace=0):
        sl = slice(ks[0], ks[-1] + 1)
        l = lim[ks[0]:ks[-1] + 2]
        t0, c0 = time.perf_counter(), time.process_time()
        res = no.nuts_sites(M.model_name, X[l[0]:l[-1]], y[l[0]:l[-1]], l - l[0], mus[sl], Oms[sl],
                            seeds[sl], chains=chains, iter=siter, init=last[sl], nthreads=nt, trace_sites=trace)
        busy = (time.process_time() - c0) / max(time.perf_counter() - t0, 1e-9)      # CPUs the kernel delivered to the sampler
        draws, stats = res[0], res[2]
        if trace:
            trace_c[0] = res[3]
        mom = [eo.tilted_moments(np.asfortranarray(draws[j].reshape(-1, P)[:, :d]), Q, r, estim) for j in range(len(ks))]
        return time.perf_counter() - t0, float(stats[:, :, 3].sum()), draws, stats, mom, busy

    # the timed legs: the fast build.  Reference schedule first (a site's chains on 4 threads: the per-thread cost every
    # other leg is held against), then all sites' (site, chain) pairs over the full width
    with no.timing_build():
        seq = [site_update([k], min(chains, nthr)) for k in range(n_seq)]
        t_seq = [q[0] for q in seq]
        us_seq = float(np.sum(t_seq)) * 1e6 * min(chains, nthr) / max(float(np.sum([q[1] for q in seq])), 1.0)
        t_all, g_all, draws_f, _, _, busy_all = site_update(list(range(n_all)), nthr)
    # the compared leg: the strict build (the checker), first n_par sites
    n_par = min(n_par, n_all)
    t_par, g_par, draws_c, stats_c, mom_c, busy_par = site_update(list(range(n_par)), nthr, trace=n_par if snap.get('trace') is not None else 0)
    work_items = n_all * chains
    used = int(min(nthr, work_items))
    us_all = t_all * 1e6 * used / max(g_all, 1.0)
    # the count is held against the timings: `cores` never exceeds what they support.  A leg whose threads cost more than
    # twice the 4-thread leg's per gradient did not have `used` processors; it is then stated as what was delivered.
    slowdown = us_all / max(us_seq, 1e-12) if n_seq > 0 else None
    supported = slowdown is None or slowdown <= 2.0
    cores = used if supported else int(max(1, round(min(busy_all, used / slowdown))))
    base = {'value': n_all / t_all, 'unit': 'site-updates/s', 'cores': cores, 'kind': 'port',
            'cpu': cpu_model_name(), 'host_threads': limits['host_threads'], 'affinity': limits['affinity'],
            'cgroup_quota_cpus': limits['cgroup_quota_cpus'], 'cgroup_source': limits['cgroup_source'],
            'loadavg_1min_before': limits['loadavg_1min'],
            'threads_used': used, 'width_from': width_why, 'work_items': int(work_items),
            'cpus_delivered': busy_all,
            'us_per_gradient_and_thread': us_all, 'us_per_gradient_and_thread_of_the_4_thread_leg': us_seq if n_seq > 0 else None,
            'per_thread_slowdown_vs_the_4_thread_leg': slowdown,
            'cores_note': ('`cores` = the threads the leg ran on = what the process can run at once (%s); the kernel delivered %.1f '
                           'CPUs over the leg (process CPU time / wall time) and a thread cost %.2f x the 4-thread leg\'s per '
                           'gradient' % (width_why, busy_all, slowdown if slowdown is not None else float('nan')))
                          if supported else
                          ('the leg ran %d threads (%s) but they cost %.1f x the 4-thread leg\'s per gradient: `cores` is what the '
                           'timings support (%.1f CPUs delivered), not the thread count' % (used, width_why, slowdown, busy_all)),
            'build': 'fast: gcc -O3 -march=native -ffp-contract=fast, vectorised logistic terms (oracle/Makefile FAST_LIB); '
                     'timing only, never the checker',
            'sample': 'one site update (C-oracle NUTS + NumPy moment stage) of the first %d sites of this workload: the EP '
                      'iteration the device ran behind its timed ones, from the same cavities, last draws and Stan seeds; '
                      '%d (site, chain) pairs handed out dynamically to %d threads: %.1f s wall, %.3g gradients, %.1f us per gradient and thread'
                      % (n_all, work_items, used, t_all, g_all, us_all),
            'strict_build': {'what': 'the checker (gcc -O2 -ffp-contract=off, libm) on the first %d sites, %d pairs over %d '
                                     'threads: the run the parity record compares with' % (n_par, n_par * chains, min(nthr, n_par * chains)),
                             'site_updates_per_s': n_par / t_par, 'seconds': t_par, 'gradients': g_par, 'cpus_delivered': busy_par,
                             'us_per_gradient_and_thread': t_par * 1e6 * min(nthr, n_par * chains) / max(g_par, 1.0)},
            'reference_schedule': {
                'what': 'sites one after the other, the %d chains of a site on %d threads (method.py:1005-1023); fast build'
                        % (chains, min(chains, nthr)),
                'sites_timed': n_seq, 'site_updates_per_s': (n_seq / float(np.sum(t_seq))) if n_seq > 0 else None,
                'seconds_per_site': [float(t) for t in t_seq],
                'max_over_sites_s': float(np.max(t_seq)) if n_seq > 0 else None,
                'us_per_gradient_and_thread': us_seq if n_seq > 0 else None,
                'note': 'the reference reports max over sites as its per-iteration "sampling time" '
                        '(method.py:1043), i.e. the time if every site had its own 4 cores'},
            'extrapolation': 'none: rates are per site update; an EP iteration over J sites costs J / rate'}
    n_all = n_par                                        # (from here on: the compared sites)
    lim = lim[:n_all + 1]
    X, y, seeds = X[:lim[-1]], y[:lim[-1]], seeds[:n_all]

    # ---- parity: the device's results of the same site updates
    cs = eng.get_chain_stats(chains)[:n_all]
    first_err, n_first, n_end, n_stats = 0.0, 0, 0, 0
    z_mean, rel_cov, tol_cov, dq_rel = [], [], [], []
    zs_mean, zs_var = np.zeros((n_all, d)), np.zeros((n_all, d))      # signed z-scores, pooled below
    for k in range(n_all):
        dev = eng.get_draws(k, all_params=True).reshape(chains, nkeep, P)
        ref = draws_c[k]
        scale = max(1.0, float(np.abs(ref).max()))
        err = np.abs(dev - ref).max(axis=2) / scale                      # (chains, nkeep)
        for c in range(chains):
            if err[c, 0] < 1e-4:                                         # (a decision that differs gives O(0.1 - 1))
                n_first += 1
                first_err = max(first_err, float(err[c, :5].max()) if np.all(err[c, :5] < 1e-4) else float(err[c, 0]))
            if np.all(err[c] < 1e-4):
                n_end += 1
                n_stats += int(cs[k, c, 2] == stats_c[k, c, 2] and cs[k, c, 3] == stats_c[k, c, 3])
        # tilted moments (method.py:413-437): both sets of draws are samples of the same tilted distribution; where the
        # chains have parted (chaotic trajectories amplify the last-bit differences of the summation order) they are
        # independent ones, and the tolerance is the Monte-Carlo error of both
        g, c_ = dev[:, :, :d], ref[:, :, :d]
        mg, mc = g.reshape(-1, d).mean(axis=0), c_.reshape(-1, d).mean(axis=0)
        vg, vc = g.reshape(-1, d).var(axis=0, ddof=1), c_.reshape(-1, d).var(axis=0, ddof=1)
        for i in range(d):
            eg, ec = _ess(g[:, :, i]), _ess(c_[:, :, i])
            vp = 0.5 * (vg[i] + vc[i])
            z_mean.append(abs(mg[i] - mc[i]) / np.sqrt(vp * (1.0 / eg + 1.0 / ec)))
            rel_cov.append(abs(vg[i] - vc[i]) / vp)
            tol_cov.append(4.0 * np.sqrt(2.0 / eg + 2.0 / ec))
            zs_mean[k, i] = (mg[i] - mc[i]) / np.sqrt(vp * (1.0 / eg + 1.0 / ec))
            zs_var[k, i] = np.log(vg[i] / vc[i]) / np.sqrt(2.0 / eg + 2.0 / ec)
        # the site delta the device formed from ITS draws against the NumPy moment stage on the same draws (deterministic)
        dQ_dev, dr_dev = eng.get_site(DQI, k)
        dQ_o = eo.tilted_moments(np.asfortranarray(dev.reshape(-1, P)[:, :d]), Q, r, estim)[0]
        dq_rel.append(float(np.abs(dQ_dev - dQ_o).max() / np.abs(dQ_o).max()))
    z_mean, rel_cov, tol_cov = np.array(z_mean), np.array(rel_cov), np.array(tol_cov)
    # the global moments (method.py:1211-1219) the iteration leads to, with the CPU's deltas in place of the device's for
    # the sampled sites: Q(df) = Q + df sum_k dQi (method.py:1071-1074)
    dQ_all, dr_all = eng.get_sites(DQI)
    glob = None
    for damp in (df, 0.5 * df, 0.25 * df):
        try:
            Qg = Q + damp * dQ_all.sum(axis=2)
            rg = r + damp * dr_all.sum(axis=1)
            Qc, rc = Qg.copy(), rg.copy()
            for k in range(n_all):
                Qc += damp * (mom_c[k][0] - dQ_all[:, :, k])
                rc += damp * (mom_c[k][1] - dr_all[:, k])
            Sg, Sc = np.linalg.inv(Qg), np.linalg.inv(Qc)
            np.linalg.cholesky(Sg), np.linalg.cholesky(Sc)
            m_g, m_c = Sg.dot(rg), Sc.dot(rc)
            sd = np.sqrt(np.diag(Sg))
            glob = {'df': float(damp), 'mean_shift_in_sd_max': float(np.max(np.abs(m_g - m_c) / sd)),
                    'cov_rel_err_max': float(np.abs(Sg - Sc).max() / np.abs(Sg).max())}
            break
        except np.linalg.LinAlgError:
            continue
    # one transition from the SAME state with the same step size and metric: the arithmetic itself, no chaos in between
    q0 = np.stack([eng.get_draws(k, all_params=True).reshape(chains, nkeep, P)[:, -1, :] for k in range(n_all)])
    eps = cs[:, :, 1]
    inv_e = np.stack([np.repeat(eng.get_adapt(k, chains)[1][None, :], chains, axis=0) for k in range(n_all)])
    inv_e = np.where(inv_e > 0, inv_e, 1.0)
    mus2 = np.stack([eng.get_cavity(k)[1] for k in range(n_all)])
    Oms2 = np.stack([eng.get_cavity(k)[0] for k in range(n_all)])
    lay = eng.last_layout()
    tf = None
    if lay in (1, 2, 5, 6, 7):
        ref_t, st_t = no.nuts_transitions(M.model_name, X, y, lim, mus2, Oms2, seeds, q0, eps, inv_e, nt=1, t_offset=7)
        out_t, cs_t = eng.nuts_transitions(seeds, q0, eps, inv_e, nt=1, t_offset=7, layout=lay if lay in (5, 7) else 0, k0=0)
        e_t = np.abs(out_t - ref_t).max(axis=(2, 3)) / np.maximum(1.0, np.abs(ref_t).max(axis=(2, 3)))
        tf = {'chains': int(e_t.size), 'layout': int(eng.last_layout()), 'max_rel_err': float(e_t.max()),
              'chains_within_1e-6': int(np.sum(e_t < 1e-6)),
              'leapfrog_counts_equal': int(np.sum(cs_t[:, :, 3] == st_t[:, :, 3])),
              'mean_leapfrogs': float(st_t[:, :, 3].mean())}
    # transition by transition (the device's trace of the parity iteration, epx_set_trace, against the oracle's): how long the
    # two runs of a chain stay together, and whether anything drifted in front of the transition that parted them
    by_t = None
    if snap.get('trace') is not None and trace_c[0] is not None:
        td, to = snap['trace'][:n_all], trace_c[0][:n_all]
        T = td.shape[2]
        sc_ = np.maximum(1.0, np.abs(to[..., 8:]).max(axis=3))
        e_t = np.abs(td[..., 8:] - to[..., 8:]).max(axis=3) / sc_
        differ = (td[..., 1] != to[..., 1]) | (e_t > 1e-6)
        t_star = np.where(differ.any(axis=2), differ.argmax(axis=2), T)
        # error growth: the factor by which a chain's error grows per transition while the two runs are together
        growth = []
        for k in range(td.shape[0]):
            for c in range(td.shape[1]):
                e = np.maximum(e_t[k, c, :max(int(t_star[k, c]), 1)], 1e-17)
                if len(e) >= 3:
                    growth.append(float(np.exp(np.mean(np.diff(np.log(e))))))
        first_eps = np.abs(td[:, :, 0, 5] / to[:, :, 0, 5] - 1.0)
        edges = [0, 1, 2, 4, 8, 16, 32, 64, 128, T, T + 1]
        by_t = {'chains': int(t_star.size),
                'transitions_until_parting_histogram': {('%d-%d' % (edges[i], edges[i + 1] - 1)) if edges[i + 1] <= T else 'never':
                                                        int(np.sum((t_star >= edges[i]) & (t_star < edges[i + 1])))
                                                        for i in range(len(edges) - 1)},
                'median_transitions_until_parting': float(np.median(t_star)),
                'chains_equal_through_the_first_transition': int(np.sum(t_star >= 1)),
                'first_transition_max_rel_err_of_the_draws': float(e_t[:, :, 0].max()),
                'first_transition_max_rel_err_of_the_adapted_step_size': float(first_eps.max()),
                'median_error_growth_factor_per_transition': float(np.median(growth)) if growth else None,
                'leapfrogs_of_the_compared_transitions': float(sum(to[k, c, :t_star[k, c], 1].sum()
                                                                   for k in range(td.shape[0]) for c in range(td.shape[1]))),
                'note': 'parting = the first transition whose leapfrog count differs or whose draw differs by more than 1e-6. '
                        'The CPU leg starts every chain from the device\'s own cavity, last draw and Stan seed: a leg on a different '
                        'problem parts at transition 0.  On these funnel-shaped posteriors (hundreds of leapfrogs per transition) '
                        'the rounding differences of the two summation orders grow by the factor above per transition until they '
                        'reach 1e-6 or flip a decision: chaos, not a difference of the algorithm -- the first transition, step-size '
                        'search included, agrees to rounding'}
    # The null reference of those scatters: the FAST CPU build's draws of the same site updates (the timed leg above) against
    # the strict build's -- two CPU runs of one algorithm that differ in rounding only (contraction, polynomial exp), part
    # like the device's chains do, and share the random stream with them (common random numbers keep parted chains
    # coupled: that, not the ESS estimate, is why all these z-scores scatter by LESS than 1).  The device agrees with the
    # CPU port in distribution if device-vs-strict scatters like fast-vs-strict.
    def pair_z(A, B):
        zm, zv = np.zeros((n_all, d)), np.zeros((n_all, d))
        for k in range(n_all):
            a_, b_ = A[k][:, :, :d], B[k][:, :, :d]
            ma, mb = a_.reshape(-1, d).mean(axis=0), b_.reshape(-1, d).mean(axis=0)
            va, vb = a_.reshape(-1, d).var(axis=0, ddof=1), b_.reshape(-1, d).var(axis=0, ddof=1)
            for i in range(d):
                ea, eb = _ess(a_[:, :, i]), _ess(b_[:, :, i])
                zm[k, i] = (ma[i] - mb[i]) / np.sqrt(0.5 * (va[i] + vb[i]) * (1.0 / ea + 1.0 / eb))
                zv[k, i] = np.log(va[i] / vb[i]) / np.sqrt(2.0 / ea + 2.0 / eb)
        return zm, zv
    null_ref = None
    if draws_f is not None and len(draws_f) >= n_all and n_all >= 8:
        zm0, zv0 = pair_z([np.asarray(draws_f[k]) for k in range(n_all)], [np.asarray(draws_c[k]) for k in range(n_all)])
        null_ref = {'what': 'CPU fast build against CPU strict build, same sites, cavities, starting draws and seeds: what two runs of '
                            'ONE algorithm that differ in rounding only look like in these statistics',
                    'tilted_mean_z_pooled': pooled_z(zm0, 'fast-build mean - strict-build mean, in the same units'),
                    'tilted_log_variance_ratio_z_pooled': pooled_z(zv0, 'log(fast-build variance / strict-build variance), in the same units')}
    pz_m = pooled_z(zs_mean, 'signed (device mean - CPU mean) / sqrt(pooled variance (1/ESS_dev + 1/ESS_cpu)), every coordinate of every compared site')
    pz_v = pooled_z(zs_var, 'log(device variance / CPU variance) / sqrt(2/ESS_dev + 2/ESS_cpu), every coordinate of every compared site')
    # agreement of the two samplers in distribution: no shared bias (|t| of the per-site means), and a scatter that is
    # neither much wider than the Monte-Carlo error (a real difference) nor much narrower (a leg that is not independent)
    pooled_ok = all(abs(q['mean_of_site_means_t_statistic']) <= 4.5 and q['variance'] <= 2.5 for q in (pz_m, pz_v)) if n_all >= 8 else True
    scatter_ratio = None
    if null_ref is not None:
        # device-vs-CPU scatter over the CPU-vs-CPU scatter (variances of the pooled z-scores): ~1 when the device is "one
        # more run of the same algorithm" (0 when it equals the strict build draw for draw); F-like with ~sites x (effective
        # coordinates) degrees of freedom each -- a factor of 2 is far outside its noise
        scatter_ratio = {'means': pz_m['variance'] / max(null_ref['tilted_mean_z_pooled']['variance'], 1e-300),
                         'log_variance_ratios': pz_v['variance'] / max(null_ref['tilted_log_variance_ratio_z_pooled']['variance'], 1e-300)}
        pooled_ok = pooled_ok and all(v <= 2.0 for v in scatter_ratio.values()) \
            and all(abs(null_ref[q]['mean_of_site_means_t_statistic']) <= 4.5 for q in ('tilted_mean_z_pooled', 'tilted_log_variance_ratio_z_pooled'))
    parity = {
        'what': 'the EP iteration behind the timed ones, sites 0..%d: device (the timed kernel, piece queue and all) against the '
                'CPU port from the same cavities, starting draws and Stan seeds' % (n_all - 1),
        'sites': int(n_all), 'chains': int(n_all * chains),
        # draw by draw: ONE transition of every chain from the same state, step size and metric (no chaos in between)
        'first_draws_max_abs_err': None if tf is None else tf['max_rel_err'],
        'first_draws': tf,
        # the whole site update (100 warm-up + 100 kept transitions of up to 1 023 leapfrogs): HMC amplifies the last-bit
        # differences of the summation order, so chains are compared until they part -- on these funnel-shaped posteriors
        # most part during the warm-up, and from there on the two runs are independent samples of the same tilted distribution
        'whole_update': {'chains_equal_at_first_kept_draw': int(n_first), 'chains_equal_to_the_last_draw': int(n_end),
                         'equal_chains_with_equal_leapfrog_counts': int(n_stats),
                         'max_rel_err_of_their_first_kept_draws': first_err},
        'tilted_mean_err_in_mcse': {'max': float(z_mean.max()), 'median': float(np.median(z_mean)),
                                    'share_within_4': float(np.mean(z_mean <= 4.0))},
        'tilted_cov_rel_err': {'max': float(rel_cov.max()), 'median': float(np.median(rel_cov)),
                               'share_within_tolerance': float(np.mean(rel_cov <= tol_cov))},
        # the whole distribution of the z-scores, not only their tails
        'tilted_mean_z_pooled': pz_m, 'tilted_log_variance_ratio_z_pooled': pz_v,
        'null_reference_cpu_fast_vs_cpu_strict': null_ref, 'scatter_device_vs_cpu_over_cpu_vs_cpu': scatter_ratio,
        'site_delta_vs_numpy_moment_stage_max_rel_err': float(np.max(dq_rel)),
        'global_moments_with_cpu_deltas_for_these_sites': glob,
        'transition_by_transition': by_t,
        'tolerance': 'first draws (one transition from the same state): 1e-6 relative; tilted mean within 4 MCSE per coordinate, '
                     'tilted variances within 4 sqrt(2/ESS_dev + 2/ESS_cpu) relative (SURVEY.md section 8c; ESS by Geyer\'s '
                     'initial positive sequence over the %d chains); pooled over all coordinates of all compared sites the signed '
                     'z-scores of means and log variance ratios must have |t| <= 4.5 for the mean of the per-site means and a '
                     'variance <= 2.5 (N(0, 1) expected; from 8 sites on), and their variances must not exceed twice the '
                     'same statistics between the two CPU builds (the null reference: rounding-only differences); site delta from the device\'s own draws against the NumPy '
                     'moment stage: 1e-7; transition by transition: >= 90 %% of the chains equal through the first transition (step-size '
                     'search included), its draws and adapted step sizes within 1e-6' % chains,
        'ok': bool((tf is None or tf['max_rel_err'] < 1e-6) and np.mean(z_mean <= 4.0) >= 0.99 and pooled_ok
                   and np.mean(rel_cov <= tol_cov) >= 0.99 and np.max(dq_rel) < 1e-7
                   # the same problem, and no drift: nearly every chain gets through its first transition (step-size search
                   # included) with the oracle's, and what lies in front of a parting agrees to rounding
                   and (by_t is None or (by_t['chains_equal_through_the_first_transition'] >= 0.9 * by_t['chains']
                                         and by_t['first_transition_max_rel_err_of_the_draws'] < 1e-6
                                         and by_t['first_transition_max_rel_err_of_the_adapted_step_size'] < 1e-6)))}
    return base, parity


def spawn_ranks(n, argv):
    """Parent of a plain `bench.py --gpus N`: start N fresh ranks, relay their output."""
    # two verified free ports: torchrun's rendezvous, and the one the ranks fall back to for the RCCL id when
    # torchrun's store is not usable (dist.EpxComm); both sockets stay open until both numbers are known
    with socket.socket() as s, socket.socket() as s2:
        s.bind(('127.0.0.1', 0))
        s2.bind(('127.0.0.1', 0))
        port, port2 = s.getsockname()[1], s2.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault('EPX_COMM_PORT', str(port2))
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')     # (the pool's driver only supports dmabuf IPC: RCCL across processes needs it)
    return subprocess.call(cmd, env=env)


def measure(args, cfg_name, sizes, comm, rank, world, local_rank, on_gpu, custom=False):
    """One configuration: workload, Master, `warm` untimed EP iterations, then EXACTLY `steps` timed ones bracketed by
    barrier + device synchronisation on both sides.  Returns (record, Master) on rank 0, (None, Master) elsewhere: every rank
    keeps its Master, because the parity iteration behind the timed region is a collective EP iteration like any other."""
    from epstan_amd import _lib as elib, models
    from epstan_amd.method import Master
    sites, D, n, cor, steps, warm = sizes
    J = sites * world
    mod, data, Q0, r0 = workload(J, D, n, args.model, bool(cor))
    M = Master(mod.site_model, data.X, data.y, site_sizes=data.Nj, prior={'Q': Q0, 'r': r0},
               chains=args.chains, iter=args.siter, prec_estim=args.prec_estim,
               df0=models.default_df0(J), comm=comm, device=local_rank, layout=args.layout,
               adapt=args.adapt, sync_sites=False, **({} if on_gpu else {'_engine_factory': _ENGINE_FACTORY}))
    rccl_rank, rccl_world = comm.size() if hasattr(comm, 'size') else (comm.rank, comm.world)

    def sync():
        if on_gpu:
            elib.device_synchronize(local_rank)
        comm.barrier()
        if on_gpu:
            elib.device_synchronize(local_rank)

    # (EPX_BENCH_SEED_SHIFT: a diagnostic of how far the leapfrog counts of the late iterations depend on the random
    # streams -- HISTORY.md section 6; the driver's command does not set it and the line says so when it is set)
    shift = int(os.environ.get('EPX_BENCH_SEED_SHIFT', '0'))
    if warm > 0:
        info = M.run(warm, verbose=False, seed=1 + shift)[0]
        assert info == 0, 'warm-up EP iterations failed with info %d' % info
    # BASELINE configs[4] is "damped EP (find_damp path)": every EP iteration scores the reference's 31 damping factors
    # (find_damp.py:105-173) before its own damped update -- epx_damp_sweep, inside the timed region.  find_damp scores
    # against a full-posterior fit, which needs Stan; here the target is the prior's moments shrunk (any positive
    # definite target runs the same 31 x (global Cholesky + cavities of all sites + criteria)).
    sweep = None
    if cfg_name == 'c5shard':
        from epstan_amd import find_damp
        sweep = dict(damps=find_damp.default_damps(), m_target=np.zeros(M.dphi), S_target=np.linalg.inv(Q0) * 0.25,
                     samp_target=None)
    n_launch0 = len(M.sampling_ms)
    sync()
    t0 = time.perf_counter()
    res = M.run(steps, verbose=False, return_analytics=True, seed=2 + shift, **({'sweep': sweep} if sweep else {}))
    sync()
    dt = time.perf_counter() - t0
    info = res[0]
    tmax = float(comm.allreduce_max(np.array([dt]))[0])
    if rank != 0:
        return None, M
    assert info == 0, 'EP failed with info %d' % info

    # dominant kernel: the sampler, timed with HIP events on the library's stream
    ms = np.array(M.sampling_ms[n_launch0:])
    ngrad = np.array(M.ngrad_log[n_launch0:])
    n_rows = float(n)
    F_g = 4.0 * n_rows * D + 12.0 * n_rows             # SURVEY.md §8d flops per gradient
    B_g = n_rows * D * 8 + n_rows                      # ... and bytes swept per gradient (from LDS when resident)
    flops_per_launch = float(ngrad.mean()) * F_g
    t_kernel = float(ms.mean()) * 1e-3
    achieved_tf = flops_per_launch / t_kernel / 1e12
    layout = M.engine.last_layout()
    wg_per_site = args.chains if layout in (2, 6) else 1
    n_cu = M.engine.cu_count() if hasattr(M.engine, 'cu_count') else 256
    P = M.engine.P
    # HBM bytes one sampler launch has to move: X, y and the cavity in (once per workgroup), draws
    # and last states out
    hbm_alg = sites * wg_per_site * (n_rows * D * 8 + n_rows + (M.dphi**2 + M.dphi) * 8) \
        + sites * args.chains * ((args.siter - args.siter // 2) * P * 8 + P * 8)
    key = [sites, D, n, args.model, args.chains, args.siter]

    def measured_traffic(names, alg_bytes_this_run):
        """HBM bytes of THIS run's launches as the counters see such launches: the ratio counter bytes / algorithmic bytes
        of a committed rocprofv3 --pmc run of this command (FETCH_SIZE x 2 + WRITE_SIZE, MI355X_MICROARCH.md: counters
        cannot be collected inside the timed run) times the algorithmic bytes of this run's launches -- never the
        absolute bytes of another run's launches, whose trajectories made a different number of passes.
        Returns (bytes per launch, ratio, source)."""
        for name in names:
            path = os.path.join(ROOT, 'profiles', name)
            if os.path.exists(path):
                pj = json.load(open(path))
                if pj.get('workload_key', pj.get('workload')) == key:
                    ratio = pj.get('traffic_over_algorithmic')
                    if ratio is None and pj.get('algorithmic_bytes_per_launch'):
                        ratio = pj['hbm_bytes_per_launch_corrected'] / pj['algorithmic_bytes_per_launch']
                    if ratio is not None:
                        return float(ratio) * alg_bytes_this_run, float(ratio), 'profiles/' + name
        return None, None, None

    if layout == 3:
        # streaming sampler: the site rows (and the cavity precision) come from HBM once per
        # leapfrog of a workgroup's chains in lock step -> the HBM roofline is the one that binds
        passes = np.array([p.sum() for p in M.pass_log[n_launch0:]])
        B_pass = n_rows * D * 8 + n_rows * 4 + M.dphi**2 * 8     # X, y (int32), Omega
        hbm_alg = float(passes.mean()) * B_pass
        gbs = hbm_alg / t_kernel / 1e9
        tr, tr_ratio, src = measured_traffic(tuple('r%02d_stream_pmc_hbm.json' % r for r in (6, 5, 4, 2, 1)), hbm_alg)
        roof = {'kernel': 'k_nuts_stream (sampler)', 'bound': 'hbm', 'achieved': gbs, 'peak': HBM_PEAK_GBS,
                'unit': 'GB/s', 'frac': gbs / HBM_PEAK_GBS, 'traffic': tr, 'traffic_over_algorithmic': tr_ratio,
                'traffic_source': src, 'algorithmic_bytes_per_launch': hbm_alg,
                'traffic_note': 'counter bytes per algorithmic byte of the committed --pmc run of this command x the algorithmic bytes of THIS run\'s launches',
                'note': 'algorithmic bytes = row passes x (n D 8 + n 4 + d^2 8) over the HIP-event duration '
                        'of the sampler launch; includes the tail where few sites are still sampling',
                'launch_ms': float(ms.mean()), 'row_passes_per_launch': float(passes.mean()),
                'gradients_per_launch': float(ngrad.mean()),
                'passes_max_over_mean_site': float(np.mean([p.max() / p.mean() for p in M.pass_log[n_launch0:]])),
                'fp64_tflops': achieved_tf,
                # build-comparable scalars (site-updates/s moves with the trajectories EP happens to take, these do not):
                'bytes_per_row_pass': B_pass,
                'ns_per_row_pass_per_cu': t_kernel * 1e9 * n_cu / float(passes.mean())}
    else:
        tr, tr_ratio, src = measured_traffic(tuple('r%02d_%s_pmc_hbm.json' % (r, cfg_name) for r in (6, 5, 4, 3, 2)), hbm_alg)
        team = layout == 7
        # bytes the kernel reads from LDS for the rows: one sweep of the site per gradient in the one-wave-per-chain forms;
        # layout 7 reads the rows TWICE per pass (forward and transposed product) for the FOUR gradients of a site's chains
        passes = np.array([p.sum() for p in M.pass_log[n_launch0:]])
        tp = [x for x in getattr(M, 'team_pass_log', [])[n_launch0:] if x]
        team_passes = float(np.mean(tp)) if team and tp else None
        lds_bytes = (team_passes or float(passes.mean())) * 2.0 * B_g if team else float(ngrad.mean()) * B_g     # (the team's own count when the library has one: yielded passes sweep the rows too)
        lds_tbs = lds_bytes / t_kernel / 1e12
        roof = {'kernel': 'NUTS sampler (site rows resident in LDS)', 'bound': 'mfma' if team else 'fp64-valu',
                'achieved': achieved_tf, 'peak': FP64_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                'frac': achieved_tf / FP64_PEAK_TFLOPS, 'traffic': tr, 'traffic_over_algorithmic': tr_ratio, 'traffic_source': src,
                'traffic_note': 'counter bytes per algorithmic byte (hbm_algorithmic_bytes) of the committed --pmc run of this command x this run\'s algorithmic bytes',

                'note': ('FP64 flops of the gradient sweeps (G x (4 n D + 12 n)) over the HIP-event duration of the '
                         'sampler launch.  Layout 7: the two products of a gradient run on v_mfma_f64_4x4x4 for the four '
                         'chains of a site in lock step (dense FP64 matrix peak = FP64 vector peak = 78.6 TFLOP/s); X is '
                         'LDS resident, so HBM does not bound it: lds_frac / hbm_frac below (lds_swept: two reads of the '
                         'rows per lock-step pass of four chains)') if team else
                        ('FP64 vector flops of the gradient sweeps (G x (4 n D + 12 n)) over the HIP-event '
                         'duration of the sampler launch.  The kernel issues no MFMA (a wave owns one chain: '
                         'matrix-vector work) and X is LDS resident, so neither the matrix pipes nor HBM '
                         'bound it: lds_frac / hbm_frac below'),
                'launch_ms': float(ms.mean()), 'gradients_per_launch': float(ngrad.mean()),
                'row_passes_per_launch': float(passes.mean()),
                # build-comparable scalars (site-updates/s moves with the trajectories EP happens to take, these do not):
                # time per gradient evaluation, and the cycles of one lock-step pass of a workgroup at the nominal clock --
                # launch time x 2.4 GHz / (passes of all sites / CUs): every CU is busy for the whole launch when the sites
                # come from the piece queue, so this is the average pass of a CU, waits and piece changes included
                'ns_per_gradient': t_kernel * 1e9 / float(ngrad.mean()),
                'pass_cycles': (t_kernel * CLOCK_GHZ * 1e9 * min(n_cu, sites) / max(float(passes.mean()), 1.0)) if team else None,
                'pass_cycles_note': 'layout 7 only (lock-step passes): launch_ms x %.1f GHz x min(CUs, sites) / row_passes_per_launch' % CLOCK_GHZ,
                # what the row team really did (a device-side count; epx_get_team_passes): row_passes_per_launch counts the
                # gradients of every site's longest chain, the team also makes the passes that chain sat out (yields)
                'team_passes_per_launch': team_passes,
                'team_pass_cycles': (t_kernel * CLOCK_GHZ * 1e9 * min(n_cu, sites) / team_passes) if team_passes else None,
                'passes_lost_to_yields_share': (1.0 - float(passes.mean()) / team_passes) if team_passes else None,
                'chains_per_team_pass': (float(ngrad.mean()) / team_passes) if team_passes else None,
                'lds_swept_TBps': lds_tbs, 'lds_peak_TBps': LDS_PEAK_TBS, 'lds_frac': lds_tbs / LDS_PEAK_TBS,
                'hbm_algorithmic_bytes': hbm_alg,
                'hbm_frac': hbm_alg / t_kernel / 1e9 / HBM_PEAK_GBS}
    out = {
        'metric': 'site-updates/sec', 'value': J * steps / tmax, 'unit': 'site-updates/s',
        'n_gpus': world, 'steps': steps, 'warmup': warm,
        'ms_per_step': tmax / steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'ep_iters_per_sec': steps / tmax,
        'config': {'workload': 'hierarchical logistic regression %s, J=%d sites (=%d/GPU), D=%d, n_j=%d, '
                               'chains=%d, iter=%d (S=%d draws/site), prec_estim=%s, df0=default_df0(K)%s%s%s'
                               % (args.model, J, sites, D, n, args.chains, args.siter,
                                  args.chains * (args.siter - args.siter // 2), args.prec_estim,
                                  '' if cor else ', uncorrelated covariates',
                                  ', the 31-factor damping sweep of find_damp.py in every timed iteration' if sweep else '',
                                  '' if args.adapt == 'fresh' else ', adapt=carry (NOT the reference\'s per-update re-adaptation)'),
                   'adapt': args.adapt, **({'seed_shift_DIAGNOSTIC': shift} if shift else {}),
                   'name': 'custom' if custom else cfg_name,
                   'parallelism': 'sites sharded over %d GPU(s), 1 RCCL all-reduce/iter inside libepx.so' % world,
                   'rccl_world_size': rccl_world},
        'roofline': roof,
        'sampler_share_of_step': float(ms.sum() * 1e-3 / dt),
        'update_phase_ms_per_step': float(np.mean(M.othertime_log[-steps:]) * 1e3),
        **({'damping_sweep': {'factors': int(len(sweep['damps'])), 'admissible_last_iteration': int(np.sum(np.isfinite(M.sweep_log[-1]['kls']))),
                              'df_taken_last_iteration': float(M.df_log[-1]) if getattr(M, 'df_log', None) else None}} if sweep else {}),
        'mean_leapfrogs_per_transition': float(M.last_site_stats[:, 2].sum()
                                               / (sites * args.chains * args.siter)),
    }
    # what a "site update" is made of here: most transitions of the late iterations build a tree of max_treedepth
    # (1 023 leapfrogs); the headline is a statement about such trees
    max_lf = float(2 ** M.max_treedepth - 1)
    out['tree_depth_note'] = ('%.0f %% of the possible %d leapfrogs per transition over the timed launches (first %.0f %%, '
                              'last %.0f %%): these funnel-shaped site posteriors drive Stan\'s NUTS to max_treedepth in most '
                              'transitions, so site-updates/s here is a rate of depth-%d trees'
                              % (100 * float(ngrad.mean()) / (sites * args.chains * args.siter) / max_lf, int(max_lf),
                                 100 * float(ngrad[0]) / (sites * args.chains * args.siter) / max_lf,
                                 100 * float(ngrad[-1]) / (sites * args.chains * args.siter) / max_lf, M.max_treedepth))
    # the launch ends with its slowest chain (one workgroup per chain) / slowest site (chains in lock step):
    # how far that is from the average, last launch of this rank
    lf = M.engine.get_chain_stats(args.chains)[:, :, 3]
    out['launch_ms_timed'] = [round(float(x), 1) for x in ms]
    out['leapfrogs_per_transition_timed'] = [round(float(g) / (sites * args.chains * args.siter), 1) for g in ngrad]
    out['gradients_all_launches'] = float(np.sum(M.ngrad_log))      # warm-up + timed: what a profiler pass over this command counts
    out['launch_tail'] = {'slowest_chain_leapfrogs': float(lf.max()), 'mean_chain_leapfrogs': float(lf.mean()),
                          'last_launch_us_per_leapfrog_of_the_slowest_chain': float(ms[-1]) * 1e3 / max(float(lf.max()), 1.0),
                          'max_over_mean': float(lf.max() / max(lf.mean(), 1.0)), 'layout': int(layout),
                          'lead_sites_of_a_split_launch': int(M.engine.last_split()),
                          # < 0: the launch ran from the piece queue (persistent workgroups), -pieces per site
                          'pieces': int(M.engine.last_segments()) if hasattr(M.engine, 'last_segments') else 0,
                          # (negative `pieces`: pieces per site of a launch from the piece queue; who takes them)
                          'piece_form': 'one workgroup per piece (EPX_PIECE_GRID)' if os.environ.get('EPX_PIECE_GRID')
                          else 'looping workgroups, as many as the device holds'}
    return out, M


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=None)
    ap.add_argument('--warmup', type=int, default=None)
    ap.add_argument('--config', default='c3', choices=sorted(CONFIGS))
    ap.add_argument('--sites', type=int, default=None, help='sites per GPU (J = sites * gpus)')
    ap.add_argument('--D', type=int, default=None)
    ap.add_argument('--n', '--rows', dest='n', type=int, default=None, help='rows per site')
    ap.add_argument('--model', default='m4b')
    ap.add_argument('--chains', type=int, default=4)
    ap.add_argument('--siter', type=int, default=200)
    ap.add_argument('--prec-estim', default='sample')
    ap.add_argument('--layout', type=int, default=0)
    ap.add_argument('--adapt', default='fresh', choices=['fresh', 'carry'],
                    help="'fresh' = the reference's behaviour (default, the headline); 'carry' = opt-in carried adaptation")
    ap.add_argument('--cor-input', type=int, default=None, help='0: uncorrelated covariates (fit.py cor_input=False)')
    ap.add_argument('--cpu-sites', type=int, default=None, help='sites of the all-cores cpu_baseline leg (fast build); 0 disables it; default: two per thread the process can really run (affinity capped by the cgroup quota), at most 64')
    ap.add_argument('--parity-sites', type=int, default=None, help='sites the strict build re-does for the parity record (default: --cpu-sites, at most 32)')
    ap.add_argument('--cpu-seq-sites', type=int, default=3, help='sites of the reference-faithful schedule')
    ap.add_argument('--cpu-threads', type=int, default=0)
    ap.add_argument('--no-secondary', dest='secondary', action='store_false',
                    help='skip the C2 / C5-shard records behind the headline (default: measured when the headline is C3 on one GPU)')
    ap.add_argument('--dry-run', action='store_true',
                    help='launch + rendezvous only (no GPU work): checks that --gpus N starts N ranks')
    args = ap.parse_args()
    sites, D, n, cor, steps, warm = CONFIGS[args.config]
    sites = args.sites if args.sites is not None else sites
    D = args.D if args.D is not None else D
    n = args.n if args.n is not None else n
    cor = args.cor_input if args.cor_input is not None else cor
    steps = args.steps if args.steps is not None else steps
    warm = args.warmup if args.warmup is not None else warm

    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # before anything touches the GPU (see spawn_ranks)
    # the CPU legs are sized by what the host really gives this process (they are a bounded sample, and the default run has
    # to stay within minutes): two sites per usable thread's worth of chains, i.e. 8 (site, chain) work items per thread ->
    # 32 sites on this pool's GPU boxes (256 hardware threads, cgroup quota 16 CPUs), handed out dynamically
    limits = host_cpu_limits()
    width = cpu_width(limits, args.cpu_threads)[0]
    if args.cpu_sites is None:
        args.cpu_sites = int(min(64, max(4, 2 * width)))
    if args.parity_sites is None:
        args.parity_sites = int(min(32, max(2, args.cpu_sites))) if args.cpu_sites > 0 else 0
    if args.gpus > 1 and 'RANK' not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world and rank == 0:
        print('warning: --gpus %d but WORLD_SIZE %d' % (args.gpus, world), file=sys.stderr)

    if args.dry_run:
        # rendezvous of the ranks over gloo, nothing else: what `--gpus N` launches, without a GPU
        import torch.distributed as tdist
        if world > 1 or 'MASTER_ADDR' in os.environ:
            tdist.init_process_group('gloo')
            tdist.barrier()
            seen = tdist.get_world_size()
            tdist.destroy_process_group()
        else:
            seen = 1
        if rank == 0:
            print(json.dumps({'metric': 'site-updates/sec', 'value': None, 'unit': 'site-updates/s', 'n_gpus': world,
                              'ranks_seen': seen, 'dry_run': True, 'steps': steps, 'warmup': warm}))
        return

    # stdout carries exactly ONE line, the JSON record: whatever native libraries print meanwhile
    # (RCCL writes its version banner to fd 1) goes to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    # (no PyTorch in a rank: the library's own entry brackets the timed region, so the process holds ONE HIP / RCCL
    # runtime, the ROCm installation's that libepx.so is linked against -- the one the parity tests run on)
    from epstan_amd import _lib as elib, dist as edist, models
    from epstan_amd.method import Master
    on_gpu = _ENGINE_FACTORY is None
    # every rank, also a single one, goes through the in-library RCCL communicator
    comm = edist.EpxComm(rank=rank, world=world) if _COMM_FACTORY is None else _COMM_FACTORY(rank, world)

    sizes = (sites, D, n, cor, steps, warm)
    out, M = measure(args, args.config, sizes, comm, rank, world, local_rank, on_gpu,
                     custom=(args.sites, args.D, args.n) != (None, None, None))
    # ---- behind the timed region.  ORDER MATTERS: the parity iteration is an EP iteration, i.e. it
    # contains the all-reduces of epx_update_trial, so EVERY rank runs it; only behind it do ranks != 0 leave, and from
    # there on rank 0 calls nothing collective (cpu_leg: engine getters, device-local test hooks, the CPU oracle).
    # A watchdog guarantees the headline line whatever happens to this part: if the parity iteration or the CPU leg has
    # not finished by its deadline (a peer died inside a collective: RCCL would spin for ever), rank 0 writes the line
    # without cpu_baseline / parity and the process exits.
    emitted = threading.Lock()

    def emit(rec):
        if emitted.acquire(False):
            sys.stdout.flush()
            os.write(json_fd, (json.dumps(rec) + '\n').encode())

    watchdog = None
    if args.cpu_sites > 0 and rank == 0:
        step_s = out['ms_per_step'] * 1e-3
        deadline = float(os.environ.get('EPX_BENCH_PARITY_DEADLINE_S', max(900.0, 30.0 * step_s)))

        def give_up():
            rec = dict(out)
            rec['cpu_baseline'] = {'value': None, 'unit': 'site-updates/s', 'cores': 0, 'kind': 'port',
                                   'sample': 'failed: the parity iteration / CPU leg did not finish within %.0f s' % deadline}
            rec['parity'] = None
            emit(rec)
            os._exit(0)
        watchdog = threading.Timer(deadline, give_up)
        watchdog.daemon = True
        watchdog.start()
    snap, err_p = None, None
    if args.cpu_sites > 0:
        if rank == 0:
            try:
                # what the CPU port starts from (cavities, last draws, global approximation): rank-local getters
                snap = snapshot_for_cpu_leg(M, args.cpu_sites, args.chains, args.siter)
                if hasattr(M.engine, 'set_trace'):
                    M.engine.set_trace(min(args.parity_sites, snap['n_all']))     # every transition of the compared sites' chains, warm-up included
            except Exception as ex:                  # (rank 0 still joins the collective iteration below)
                import traceback
                traceback.print_exc()
                err_p = ex
        try:
            # one more EP iteration on the device(s), behind the timed region -- ALL ranks: it holds the all-reduces of the
            # update phase -- whose first sites the CPU port re-does on rank 0 from the same state and seeds
            info_p = M.run(1, verbose=False, seed=PARITY_SEED)[0]
            if info_p != 0:
                err_p = err_p or RuntimeError('parity EP iteration failed with info %d' % info_p)
        except Exception as ex:
            import traceback
            traceback.print_exc()
            err_p = err_p or ex
    if rank != 0:
        # the last collective of this process is behind it: rank 0's CPU leg needs nothing from the others, and they must
        # not spin in a barrier beside it (an RCCL barrier busy-waits a host thread per rank) -- ncclCommDestroy needs no peer
        if hasattr(comm, 'close'):
            comm.close()
        return
    if args.cpu_sites > 0:
        try:
            if err_p is not None:
                raise err_p
            snap['trace'] = None
            if hasattr(M.engine, 'set_trace'):
                snap['trace'] = M.engine.get_trace(args.chains, args.siter)
                M.engine.set_trace(0)
            # collective-free from here on (the other ranks have gone)
            out['cpu_baseline'], out['parity'] = cpu_leg(M, snap, args.prec_estim, args.chains, args.siter,
                                                         args.cpu_seq_sites, args.cpu_threads,
                                                         M.df_log[-1] if getattr(M, 'df_log', None) else M.df0(M.iter),
                                                         args.parity_sites, limits)
        except Exception as ex:                      # the baseline must not void the GPU measurement
            import traceback
            traceback.print_exc()
            out['cpu_baseline'] = {'value': None, 'unit': 'site-updates/s', 'cores': 0, 'kind': 'port',
                                   'sample': 'failed: %r' % (ex,)}
            out['parity'] = None
    if watchdog is not None:
        watchdog.cancel()
    # ---- secondary records: the other two single-GPU configurations of BASELINE.json, driver-timed in the same process
    # (configs[1] = C2: 64 sites, D = 16, n = 200, layout 6; one 512-site shard of configs[4] = C5: D = 128, n = 2000, the
    # streaming sampler against the HBM roofline).  Headline keys stay the default workload's.  One GPU only: a rank of a
    # multi-GPU run has nothing to add to them.
    if args.secondary and world == 1 and on_gpu and args.config == 'c3' and (args.sites, args.D, args.n) == (None, None, None):
        import gc
        out['secondary'] = []
        try:
            M.engine.close()
        except Exception:
            pass
        del M
        gc.collect()
        for name, (st_, wm_) in (('c2', (20, 5)), ('c5shard', (2, 2))):
            try:
                t_s = time.perf_counter()
                sz = CONFIGS[name][:4] + (st_, wm_)
                comm2 = edist.EpxComm(rank=0, world=1)       # (a communicator of its own: the headline's is bound to the engine just closed)
                rec, M2 = measure(args, name, sz, comm2, rank, world, local_rank, on_gpu)
                rf = rec['roofline']
                keep = ('kernel', 'bound', 'achieved', 'peak', 'unit', 'frac', 'launch_ms', 'row_passes_per_launch',
                        'gradients_per_launch', 'ns_per_gradient', 'pass_cycles', 'bytes_per_row_pass',
                        'ns_per_row_pass_per_cu', 'traffic', 'traffic_over_algorithmic', 'traffic_source', 'algorithmic_bytes_per_launch')
                out['secondary'].append({
                    'config': rec['config'], 'value': rec['value'], 'unit': rec['unit'], 'steps': st_, 'warmup': wm_,
                    'ms_per_step': rec['ms_per_step'], 'ep_iters_per_sec': rec['ep_iters_per_sec'],
                    'roofline': {k: rf[k] for k in keep if k in rf},
                    'launch_tail': rec['launch_tail'], 'mean_leapfrogs_per_transition': rec['mean_leapfrogs_per_transition'],
                    **({'damping_sweep': rec['damping_sweep']} if 'damping_sweep' in rec else {}),
                    'update_phase_ms_per_step': rec['update_phase_ms_per_step'],
                    'wall_s_with_setup': time.perf_counter() - t_s})
                try:
                    if hasattr(comm2, 'close'):
                        comm2.close()
                    M2.engine.close()
                except Exception:
                    pass
                del M2
                gc.collect()
            except Exception as ex:
                import traceback
                traceback.print_exc()
                out['secondary'].append({'config': {'name': name}, 'value': None, 'error': repr(ex)})
    if hasattr(comm, 'close'):
        comm.close()
    emit(out)


if __name__ == '__main__':
    main()
