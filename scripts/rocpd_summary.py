"""Summarise rocprofv3 rocpd (SQLite) outputs: per-kernel time statistics (as --stats would
print) and per-kernel PMC counter sums.  Usage: rocpd_summary.py <results.db> [out.csv]"""
import sqlite3, sys, csv
db = sys.argv[1]
con = sqlite3.connect(db)
suf = [r[0] for r in con.execute("select name from sqlite_master where type='table' and name like 'rocpd_kernel_dispatch%'")][0].replace('rocpd_kernel_dispatch', '')
rows = con.execute("""select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start),
                      max(s.arch_vgpr_count), max(s.accum_vgpr_count), max(d.group_segment_size), max(d.private_segment_size)
                      from rocpd_kernel_dispatch%s d join rocpd_info_kernel_symbol%s s on d.kernel_id = s.id
                      group by s.kernel_name order by 3 desc""" % (suf, suf)).fetchall()
tot = sum(r[2] for r in rows)
out = [['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs', 'arch_vgpr', 'accum_vgpr', 'lds_bytes', 'scratch_bytes']]
for r in rows:
    out.append([r[0][:120], r[1], r[2], '%.1f' % r[3], '%.2f' % (100.0 * r[2] / tot), r[4], r[5], r[6], r[7], r[8], r[9]])
pm = con.execute("""select s.kernel_name, p.name, count(*), sum(e.value) from rocpd_pmc_event%s e
                    join rocpd_info_pmc%s p on e.pmc_id = p.id
                    join rocpd_kernel_dispatch%s d on e.event_id = d.event_id
                    join rocpd_info_kernel_symbol%s s on d.kernel_id = s.id
                    group by s.kernel_name, p.name order by 4 desc""" % (suf, suf, suf, suf)).fetchall()
w = csv.writer(open(sys.argv[2], 'w') if len(sys.argv) > 2 else sys.stdout)
w.writerows(out)
if pm:
    w.writerow([])
    w.writerow(['Name', 'Counter', 'Dispatches', 'Sum', 'PerDispatch'])
    for r in pm:
        w.writerow([r[0][:120], r[1], r[2], r[3], r[3] / r[2]])
