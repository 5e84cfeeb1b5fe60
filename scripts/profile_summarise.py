"""Turns the rocprofv3 outputs of scripts/profile_round.sh (gpurun_out/round/) into the tracked
summaries under profiles/: per-kernel time statistics (CSV, as --stats prints them) and the HBM
bytes of the sampler kernels from the FETCH_SIZE / WRITE_SIZE passes (JSON; KiB units, FETCH_SIZE x2
for wide reads on gfx950 as MI355X_MICROARCH.md prescribes, WRITE_SIZE as reported).
Usage: python scripts/profile_summarise.py [round_tag]"""
import glob, json, os, shutil, sqlite3, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'gpurun_out', 'round')
tag = sys.argv[1] if len(sys.argv) > 1 else 'r03'
PROF = os.environ.get('PROFILES_OUT', os.path.join(ROOT, 'profiles'))     # (on the GPU box: a directory under gpurun_out/, which is what travels back)
os.makedirs(PROF, exist_ok=True)


def dbs(d):
    return sorted(glob.glob(os.path.join(SRC, d, '**', '*.db'), recursive=True))


def kernel_rows(db):
    con = sqlite3.connect(db)
    tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table' and name like 'rocpd_kernel_dispatch%'")]
    if not tabs:
        return None, None
    suf = tabs[0].replace('rocpd_kernel_dispatch', '')
    n = con.execute('select count(*) from rocpd_kernel_dispatch%s' % suf).fetchone()[0]
    return (con, suf) if n else (None, None)


def pmc_sums(d, warmup, timed):
    """Counter value and duration of the sampler kernels per EP iteration of the profiled command
    (a split launch runs two sampler kernels side by side: both are booked to their iteration);
    returns the per-kernel totals and the list of per-iteration sums of the TIMED iterations."""
    for db in dbs(d):
        con, suf = kernel_rows(db)
        if con is None:
            continue
        q = """select s.kernel_name, d.start, d.end, sum(e.value) from rocpd_pmc_event%s e
               join rocpd_kernel_dispatch%s d on e.event_id = d.event_id
               join rocpd_info_kernel_symbol%s s on d.kernel_id = s.id
               where s.kernel_name like '%%k_nuts%%' or s.kernel_name like '%%k_moments%%' group by d.id order by d.start""" % (suf, suf, suf)
        rows = con.execute(q).fetchall()
        if not any('k_nuts' in r[0] for r in rows):
            continue
        # EP iterations: every iteration ends with ONE k_moments dispatch behind its sampler launch(es) -- a split launch
        # runs two sampler kernels, side by side in the bench and one after the other under --pmc (counter collection
        # serialises dispatches), so overlap in time cannot delimit them.  The first `warmup` ITERATIONS are dropped,
        # whatever kernels they ran, and so is everything behind the `timed` ones (bench.py's parity iteration)
        groups, cur = [], None
        kern = {}
        for name, t0, t1, val in rows:
            if 'k_moments' in name:
                cur = None
                continue
            if cur is None:
                cur = {'val': 0.0, 'ms': 0.0}
                groups.append(cur)
            cur['val'] += val; cur['ms'] += (t1 - t0) / 1e6
            gi = len(groups) - 1
            k = kern.setdefault(name, {'dispatches': 0, 'sum_KiB': 0.0, 'timed_dispatches': 0})
            k['dispatches'] += 1; k['sum_KiB'] += val
            k['timed_dispatches'] += int(warmup <= gi < warmup + timed)
        per_iter = [g['val'] for g in groups]
        dur = [g['ms'] for g in groups]
        return kern, per_iter[warmup:warmup + timed], dur[warmup:warmup + timed]
    return {}, [], []


for name in ('c2', 'c3', 'stream'):
    if not os.path.exists(os.path.join(SRC, name + '_bench.json')):
        continue
    line = [l for l in open(os.path.join(SRC, name + '_bench.json')) if l.startswith('{')]
    if line:
        open(os.path.join(PROF, '%s_%s_bench.json' % (tag, name)), 'w').write(line[-1])
    for db in dbs(name + '_trace'):
        if kernel_rows(db)[0] is not None:
            subprocess.check_call([sys.executable, os.path.join(ROOT, 'scripts', 'rocpd_summary.py'), db,
                                   os.path.join(PROF, '%s_%s_kernel_stats.csv' % (tag, name))])
    warmup, timed, key = {'c2': (1, 3, [64, 16, 200, 'm4b', 4, 200]), 'c3': (5, 20, [512, 32, 500, 'm4b', 4, 200]),
                          'stream': (2, 2, [512, 128, 2000, 'm4b', 4, 200])}[name]        # (scripts/profile_round.sh's --warmup / --steps)
    f, f_it, f_ms = pmc_sums(name + '_fetch', warmup, timed)
    w, w_it, w_ms = pmc_sums(name + '_write', warmup, timed)
    for kind in ('fetch', 'write'):
        for db in dbs(name + '_' + kind):
            if kernel_rows(db)[0] is not None:
                subprocess.check_call([sys.executable, os.path.join(ROOT, 'scripts', 'rocpd_summary.py'), db,
                                       os.path.join(PROF, '%s_%s_pmc_%s_size.csv' % (tag, name, kind))])
    bench = json.loads(line[-1]) if line else {}
    mean = lambda v: sum(v) / len(v) if v else None
    fk, wk = mean(f_it), mean(w_it)
    summary = {
        'commands': 'scripts/profile_round.sh: rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python3 bench.py <workload>, '
                    'and separately --pmc WRITE_SIZE; kernel-trace summary of the same command in %s_%s_kernel_stats.csv' % (tag, name),
        'workload_key': key,
        'sampler_kernels_fetch_pass': f, 'sampler_kernels_write_pass': w,
        'timed_launches': len(f_it),
        'FETCH_SIZE_KiB_per_timed_launch': fk, 'WRITE_SIZE_KiB_per_timed_launch': wk,
        'hbm_bytes_per_launch_corrected': (2 * fk + wk) * 1024 if fk is not None and wk is not None else None,
        'launch_ms_pmc_passes': [mean(f_ms), mean(w_ms)],
        'note': 'per EP iteration of the timed region (warm-up iterations dropped), all sampler kernels of the '
                'iteration summed (a split launch runs k_nuts and k_nuts_spec side by side). Units: rocprofv3 '
                'reports KiB; gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts 64 B per '
                '128-B request for wide coalesced reads -> x2; WRITE_SIZE as reported.',
    }
    if bench:
        roof = bench['roofline']
        summary['workload'] = bench['config']['workload']
        summary['bench_launch_ms'] = roof.get('launch_ms')
        summary['algorithmic_bytes_per_launch'] = roof.get('hbm_algorithmic_bytes')
        if roof.get('bound') == 'hbm':
            summary['algorithmic_bytes_per_launch'] = roof['achieved'] * 1e9 * roof['launch_ms'] * 1e-3
        if summary['hbm_bytes_per_launch_corrected'] and summary['algorithmic_bytes_per_launch']:
            summary['traffic_over_algorithmic'] = summary['hbm_bytes_per_launch_corrected'] / summary['algorithmic_bytes_per_launch']
        # the bench line was printed before these passes ran: its traffic field is (re)filled from them
        roof['traffic'] = summary['hbm_bytes_per_launch_corrected']
        roof['traffic_source'] = 'profiles/%s_%s_pmc_hbm.json (this capture)' % (tag, name)
        open(os.path.join(PROF, '%s_%s_bench.json' % (tag, name)), 'w').write(json.dumps(bench) + '\n')
    json.dump(summary, open(os.path.join(PROF, '%s_%s_pmc_hbm.json' % (tag, name)), 'w'), indent=1)
    print(name, json.dumps({k: summary.get(k) for k in ('timed_launches', 'hbm_bytes_per_launch_corrected', 'bench_launch_ms', 'launch_ms_pmc_passes', 'algorithmic_bytes_per_launch')}))


# ---- instruction mix of the C3 sampler (SQ counters over the driver's command), per leapfrog of a chain
mix = {}
for part in ('c3_mix_a', 'c3_mix_b'):
    for db in dbs(part):
        con, suf = kernel_rows(db)
        if con is None:
            continue
        rows = con.execute("""select s.kernel_name, p.name, sum(e.value) from rocpd_pmc_event%s e
                              join rocpd_info_pmc%s p on e.pmc_id = p.id
                              join rocpd_kernel_dispatch%s d on e.event_id = d.event_id
                              join rocpd_info_kernel_symbol%s s on d.kernel_id = s.id
                              where s.kernel_name like '%%k_nuts%%' group by s.kernel_name, p.name""" % (suf, suf, suf, suf)).fetchall()
        for kname, cname, val in rows:
            mix.setdefault(kname[:100], {})[cname] = val
bench_path = os.path.join(PROF, '%s_c3_bench.json' % tag)
if mix and os.path.exists(bench_path):
    bench = json.loads(open(bench_path).read())
    # gradients of ALL launches of the command (warm-up + timed): the mix passes count every launch
    logs = [open(os.path.join(SRC, 'c3_mix_a.log')).read() if os.path.exists(os.path.join(SRC, 'c3_mix_a.log')) else '']
    out = {'command': 'rocprofv3 --pmc <SQ counters> --kernel-trace -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sites 0 (two passes)',
           'kernels': mix,
           'note': 'counter sums over ALL sampler launches of the command (5 warm-up + 20 timed); per chain-leapfrog figures need the '
                   'gradient count of the same launches: bench.py reports the timed ones (gradients_per_launch x 20), the warm-up '
                   'launches are lighter, so the ratios below use the timed share = timed duration / total duration of the trace'}
    G = bench.get('gradients_all_launches')
    if G:
        out['gradients_of_all_launches'] = G
        out['per_chain_leapfrog'] = {k: {c: v / G for c, v in m.items()} for k, m in mix.items()}
        out['note'] = ('counter sums over ALL sampler launches of the command (5 warm-up + 20 timed) divided by the gradient '
                       'evaluations of the same launches (bench.py: gradients_all_launches; the run is deterministic, so the '
                       'profiled passes make the same ones)')
    json.dump(out, open(os.path.join(PROF, '%s_c3_instruction_mix.json' % tag), 'w'), indent=1)
    print('c3 instruction mix:', json.dumps({k: {c: '%.4g' % v for c, v in m.items()} for k, m in mix.items()}))
