#!/bin/bash
# HBM traffic of the sampler launches of a short C3 run under a given build: two rocprofv3 passes (FETCH_SIZE, WRITE_SIZE,
# separately, as MI355X_MICROARCH.md prescribes), summed per EP iteration.  Usage (on the GPU box):
#   scripts/pmc_traffic.sh <tag> [bench args...]      (EPX_LIB selects the build)
set -u
REPO=$PWD; TAG=$1; shift
OUT=$REPO/gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace -d $OUT/$C -o run -- python3 $REPO/bench.py "$@" --cpu-sites 0 > $OUT/$C.log 2>&1
done
python3 - $OUT <<'PY'
import glob, sqlite3, sys, json
out = sys.argv[1]
res = {}
for C in ('FETCH_SIZE', 'WRITE_SIZE'):
    for db in sorted(glob.glob(out + '/' + C + '/**/*.db', recursive=True)):
        con = sqlite3.connect(db)
        tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table' and name like 'rocpd_kernel_dispatch%'")]
        if not tabs:
            continue
        suf = tabs[0].replace('rocpd_kernel_dispatch', '')
        rows = con.execute("""select s.kernel_name, d.start, d.end, sum(e.value) from rocpd_pmc_event%s e
               join rocpd_kernel_dispatch%s d on e.event_id = d.event_id
               join rocpd_info_kernel_symbol%s s on d.kernel_id = s.id
               where s.kernel_name like '%%k_nuts%%' group by d.id order by d.start""" % (suf, suf, suf)).fetchall()
        if rows:
            res[C] = [(r[0][:40], (r[2] - r[1]) / 1e6, r[3]) for r in rows]
            break
for C, rows in res.items():
    print(C, 'KiB per sampler launch:', [round(v / 1e6, 3) for _, _, v in rows], '(x 1e6 KiB) ms:', [round(ms, 1) for _, ms, _ in rows])
if len(res) == 2:
    f = [v for _, _, v in res['FETCH_SIZE']][-3:]; w = [v for _, _, v in res['WRITE_SIZE']][-3:]
    print('last 3 launches: HBM bytes per launch (2 x FETCH + WRITE) = %.2f GB' % ((2 * sum(f) / len(f) + sum(w) / len(w)) * 1024 / 1e9))
PY
