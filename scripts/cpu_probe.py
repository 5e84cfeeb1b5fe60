#!/usr/bin/env python3
"""What the host CPUs of the box really give this process: the limits bench.host_cpu_limits()
reads, then the fast oracle build on k sites x 4 chains with 4 k threads for k = 1, 2, 4, ... -- gradients per second
and per thread of every rung.  No GPU involved.  python3 scripts/cpu_probe.py [max_threads] [omp_env...]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                          # noqa: E402
from oracle import nuts_oracle as no                  # noqa: E402


def main():
    lim = bench.host_cpu_limits()
    print(json.dumps(lim))
    for f in ('/proc/self/cgroup', '/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us', '/sys/fs/cgroup/cpu.stat',
              '/sys/fs/cgroup/cpu/cpu.stat', '/proc/pressure/cpu'):
        try:
            print(f, '->', open(f).read().strip().replace('\n', ' | ')[:400])
        except OSError as ex:
            print(f, '-> n/a', ex.__class__.__name__)
    print('OMP env:', {k: v for k, v in os.environ.items() if k.startswith(('OMP_', 'GOMP_', 'KMP_'))})
    J, D, n = 64, 32, 500
    mod, data, Q0, r0 = bench.workload(J, D, n, 'm4b', True)
    d, P = no.dims('m4b_sg', D)
    lim_rows = np.concatenate(([0], np.cumsum(data.Nj))).astype(np.int64)
    # cavities of the first EP iteration: the prior (method.py:875-880 -- every site starts from Q0, r0)
    mus = np.repeat(np.linalg.solve(Q0, r0)[None, :], J, axis=0)
    Oms = np.repeat(Q0[None, :, :], J, axis=0)
    seeds = np.arange(1, J + 1, dtype=np.int64) * 7919
    top = int(sys.argv[1]) if len(sys.argv) > 1 else (os.cpu_count() or 4)
    with no.timing_build():
        no.nuts_sites('m4b_sg', data.X[:lim_rows[1]], data.y[:lim_rows[1]], lim_rows[:2], mus[:1], Oms[:1], seeds[:1], chains=4, iter=10,
                      nthreads=4)                     # (builds and loads the library, starts the thread pool)
        k = 1
        base = None
        while 4 * k <= max(top, 4) and k <= J:
            l = lim_rows[:k + 1]
            t0 = time.perf_counter()
            c0 = time.process_time()
            res = no.nuts_sites('m4b_sg', data.X[:l[-1]], data.y[:l[-1]], l, mus[:k], Oms[:k], seeds[:k], chains=4, iter=200,
                                nthreads=4 * k)
            t = time.perf_counter() - t0
            cpu = time.process_time() - c0
            g = float(res[2][:, :, 3].sum())
            gmax = float(res[2][:, :, 3].max())
            rate = g / t
            base = base or rate / 4
            print('sites %3d threads %3d: %.2f s wall, %.2f s cpu (%.1f busy threads), %.3g gradients (longest chain %.3g = %.0f %% of the mean), '
                  '%.3g gradients/s, %.1f us per gradient and thread, %.1f threads\' worth of the first rung'
                  % (k, 4 * k, t, cpu, cpu / t, g, gmax, 100 * gmax * 4 * k / g, rate, t * 1e6 * 4 * k / g, rate / base), flush=True)
            k *= 2


if __name__ == '__main__':
    main()
