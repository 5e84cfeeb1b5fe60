#!/usr/bin/env python3
"""Headline benchmark: site-updates/sec (and EP iters/sec) of the EP inner loop on
synthetic hierarchical logistic regression (BASELINE.json configs[1]:
J=64 sites, D=16, n_j=200, model m4b, chains=4, iter=200 -> S=400 draws/site).

    python bench.py --gpus N --steps K --warmup W

One process per GPU (torchrun sets RANK/LOCAL_RANK/WORLD_SIZE); a step is one
outer EP iteration over all sites: batched NUTS site updates -> moment stage ->
packed site sums -> all-reduce -> damped update + cavities -> moments.  Weak
scaling: every rank owns `--sites` sites (64), so J = 64 * N.
Rank 0 prints ONE JSON line (contract in the task statement).
"""

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6     # MI355X FP64 vector = FP64 matrix peak (AMD spec)
HBM_PEAK_GBS = 8000.0


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


def cpu_baseline(mod, data, Q0, r0, nsites, chains, siter, threads):
    """The oracle (kind 'port') timed on this box's host cores: one EP iteration
    (cavity -> NUTS -> moment stage -> damped update) over the first `nsites` sites."""
    from epstan_amd import models
    from epstan_amd.method import Master
    from oracle.engine_oracle import OracleEngine
    from oracle import nuts_oracle as no
    no.build()
    nrow = int(data.j_lim[nsites])
    t_threads = threads if threads > 0 else no.lib().epo_num_threads()
    M = Master(mod.site_model, data.X[:nrow], data.y[:nrow], site_sizes=data.Nj[:nsites],
               prior={'Q': Q0, 'r': r0}, chains=chains, iter=siter,
               df0=models.default_df0(max(nsites, 2)),
               _engine_factory=lambda m, X, y, kl: OracleEngine(m, X, y, kl, nthreads=t_threads))
    t0 = time.time()
    info = M.run(1, verbose=False, calc_moments=True, seed=1)[0]
    dt = time.time() - t0
    return {'value': nsites / dt, 'unit': 'site-updates/s', 'cores': int(t_threads), 'kind': 'port',
            'sample': '1 EP iteration over the first %d sites of the same workload '
                      '(C oracle NUTS + NumPy moment/cavity stages), %.1f s wall' % (nsites, dt),
            'info': int(info)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--sites', type=int, default=64, help='sites per GPU (J = sites * gpus)')
    ap.add_argument('--D', type=int, default=16)
    ap.add_argument('--n', type=int, default=200)
    ap.add_argument('--model', default='m4b')
    ap.add_argument('--chains', type=int, default=4)
    ap.add_argument('--siter', type=int, default=200)
    ap.add_argument('--prec-estim', default='sample')
    ap.add_argument('--layout', type=int, default=0)
    ap.add_argument('--cor-input', type=int, default=1, help='0: uncorrelated covariates (fit.py cor_input=False)')
    ap.add_argument('--cpu-sites', type=int, default=32, help='0 disables the cpu_baseline leg')
    ap.add_argument('--cpu-threads', type=int, default=0)
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    import torch
    from epstan_amd import dist as edist, models
    from epstan_amd.method import Master
    comm = None
    under_torchrun = 'RANK' in os.environ and 'MASTER_ADDR' in os.environ
    if world > 1 or under_torchrun:
        # one rank per GPU over RCCL; also taken at world_size 1 under torchrun so that the
        # collective path (device-resident packed sums, all-reduce) is the one exercised
        import torch.distributed as tdist
        torch.cuda.set_device(local_rank)
        tdist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        comm = edist.TorchComm(device=torch.device('cuda', local_rank))
    if args.gpus != world and rank == 0 and world > 1:
        print('warning: --gpus %d but WORLD_SIZE %d' % (args.gpus, world), file=sys.stderr)

    J = args.sites * world
    mod, data, Q0, r0 = workload(J, args.D, args.n, args.model, bool(args.cor_input))
    M = Master(mod.site_model, data.X, data.y, site_sizes=data.Nj, prior={'Q': Q0, 'r': r0},
               chains=args.chains, iter=args.siter, prec_estim=args.prec_estim,
               df0=models.default_df0(J), comm=comm, device=local_rank, layout=args.layout,
               sync_sites=False)

    def sync():
        torch.cuda.synchronize()
        if comm is not None:
            tdist.barrier()

    if args.warmup > 0:
        info = M.run(args.warmup, verbose=False, seed=1)[0]
        assert info == 0, 'warm-up EP iterations failed with info %d' % info
    n_launch0 = len(M.sampling_ms)
    sync()
    t0 = time.perf_counter()
    res = M.run(args.steps, verbose=False, return_analytics=True, seed=2)
    sync()
    dt = time.perf_counter() - t0
    info = res[0]
    if comm is not None:
        tmax = comm.allreduce_max(np.array([dt]))[0]
        tdist.barrier()
    else:
        tmax = dt
    if rank != 0:
        tdist.destroy_process_group()
        return
    assert info == 0, 'EP failed with info %d' % info

    # dominant kernel: k_nuts, timed with HIP events on the library's stream
    ms = np.array(M.sampling_ms[n_launch0:])
    ngrad = np.array(M.ngrad_log[n_launch0:])
    n_rows = float(args.n)
    F_g = 4.0 * n_rows * args.D + 12.0 * n_rows             # SURVEY.md §8d flops per gradient
    flops_per_launch = float(ngrad.mean()) * F_g
    t_kernel = float(ms.mean()) * 1e-3
    achieved_tf = flops_per_launch / t_kernel / 1e12
    sites_local = args.sites
    # HBM bytes one sampler launch has to move: X, y and the cavity in (once per workgroup: one
    # workgroup per (site, chain) in layout 2, per site in layout 1), draws and last states out
    wg_per_site = 1 if M.engine.last_layout() == 1 else args.chains
    P = M.engine.P
    hbm_alg = sites_local * wg_per_site * (n_rows * args.D * 8 + n_rows + (M.dphi**2 + M.dphi) * 8) \
        + sites_local * args.chains * ((args.siter - args.siter // 2) * P * 8 + P * 8)
    traffic = None
    for pmc in ('r01_c2_pmc_hbm.json', 'r01_c3_pmc_hbm.json'):
        # measured separately with rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE on this workload
        # (scripts/profile_round.sh, scripts/profile_summarise.py)
        pmc = os.path.join(ROOT, 'profiles', pmc)
        if os.path.exists(pmc):
            pj = json.load(open(pmc))
            if pj.get('workload_key') == [args.sites, args.D, args.n, args.model, args.chains, args.siter]:
                traffic = pj['hbm_bytes_per_launch_corrected']
    roof = {'kernel': 'k_nuts (sampler)', 'bound': 'mfma', 'achieved': achieved_tf,
            'peak': FP64_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': achieved_tf / FP64_PEAK_TFLOPS,
            'traffic': traffic,
            'note': 'FP64 flops of the gradient sweeps (G x (4 n D + 12 n)) over the HIP-event '
                    'duration of the sampler launch; X is LDS resident, so HBM is not the bound: '
                    'hbm_frac below',
            'launch_ms': float(ms.mean()), 'gradients_per_launch': float(ngrad.mean()),
            'hbm_algorithmic_bytes': hbm_alg,
            'hbm_frac': hbm_alg / t_kernel / 1e9 / HBM_PEAK_GBS}
    if M.engine.last_layout() == 3:
        # streaming sampler: the site rows (and the cavity precision) come from HBM once per
        # leapfrog of a workgroup's chains in lock step -> the HBM roofline is the one that binds
        passes = np.array([p.sum() for p in M.pass_log[n_launch0:]])
        B_pass = n_rows * args.D * 8 + n_rows * 4 + M.dphi**2 * 8     # X, y (int32), Omega
        hbm_alg = float(passes.mean()) * B_pass
        gbs = hbm_alg / t_kernel / 1e9
        tr = None
        pmc3 = os.path.join(ROOT, 'profiles', 'r01_stream_pmc_hbm.json')
        if os.path.exists(pmc3):
            pj = json.load(open(pmc3))
            if pj.get('workload_key', pj.get('workload')) == [args.sites, args.D, args.n, args.model, args.chains, args.siter]:
                tr = pj['hbm_bytes_per_launch_corrected']
        roof = {'kernel': 'k_nuts_stream (sampler)', 'bound': 'hbm', 'achieved': gbs, 'peak': HBM_PEAK_GBS,
                'unit': 'GB/s', 'frac': gbs / HBM_PEAK_GBS, 'traffic': tr,
                'note': 'algorithmic bytes = row passes x (n D 8 + n 4 + d^2 8) over the HIP-event duration '
                        'of the sampler launch; includes the tail where few sites are still sampling',
                'launch_ms': float(ms.mean()), 'row_passes_per_launch': float(passes.mean()),
                'gradients_per_launch': float(ngrad.mean()),
                'passes_max_over_mean_site': float(np.mean([p.max() / p.mean() for p in M.pass_log[n_launch0:]])),
                'fp64_tflops': achieved_tf}
    out = {
        'metric': 'site-updates/sec', 'value': J * args.steps / tmax, 'unit': 'site-updates/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': tmax / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'ep_iters_per_sec': args.steps / tmax,
        'config': {'workload': 'hierarchical logistic regression %s, J=%d sites (=%d/GPU), D=%d, n_j=%d, '
                               'chains=%d, iter=%d (S=%d draws/site), prec_estim=%s, df0=default_df0(K)%s'
                               % (args.model, J, args.sites, args.D, args.n, args.chains, args.siter,
                                  args.chains * (args.siter - args.siter // 2), args.prec_estim,
                                  '' if args.cor_input else ', uncorrelated covariates'),
                   'parallelism': 'sites sharded over %d GPU(s), 1 all-reduce/iter' % world},
        'roofline': roof,
        'sampler_share_of_step': float(ms.sum() * 1e-3 / dt),
        'mean_leapfrogs_per_transition': float(M.last_site_stats[:, 2].sum()
                                               / (sites_local * args.chains * args.siter)),
    }
    # the launch ends with its slowest chain (one workgroup per chain) / slowest site (chains in lock step):
    # how far that is from the average, last launch of this rank
    lf = M.engine.get_chain_stats(args.chains)[:, :, 3]
    out['launch_tail'] = {'slowest_chain_leapfrogs': float(lf.max()), 'mean_chain_leapfrogs': float(lf.mean()),
                          'max_over_mean': float(lf.max() / max(lf.mean(), 1.0)), 'layout': int(M.engine.last_layout()),
                          'lead_sites_of_a_split_launch': int(M.engine.last_split())}
    if comm is not None:
        tdist.destroy_process_group()
    if args.cpu_sites > 0:
        try:
            out['cpu_baseline'] = cpu_baseline(mod, data, Q0, r0, min(args.cpu_sites, J), args.chains,
                                               args.siter, args.cpu_threads)
        except Exception as ex:                      # the baseline must not void the GPU measurement
            out['cpu_baseline'] = {'value': None, 'unit': 'site-updates/s', 'cores': 0, 'kind': 'port',
                                   'sample': 'failed: %r' % (ex,)}
    print(json.dumps(out))


if __name__ == '__main__':
    main()
