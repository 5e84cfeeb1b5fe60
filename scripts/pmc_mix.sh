#!/bin/bash
# Instruction mix of the sampler kernel per chain-leapfrog: one rocprofv3 --pmc pass (counters only, with the kernel
# trace) over scripts/ab_duo.py.  usage: AB_LAYOUT=7 scripts/pmc_mix.sh [sites] ; prints the summary.
R=$PWD
J=${1:-64}
mkdir -p $R/gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc/m
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --kernel-trace -d $R/gpurun_out/pmc/m -o run -- python3 $R/scripts/ab_duo.py $J > $R/gpurun_out/pmc/m.log 2>&1
cd $R
python3 - <<'PY'
import sqlite3, re
log = open('gpurun_out/pmc/m.log').read()
m = re.search(r"gradients \[(.*?)\]", log)
G = sum(float(x.strip(" '")) for x in m.group(1).split(','))
print(log.strip().split('\n')[-1])
con = sqlite3.connect('gpurun_out/pmc/m/run_results.db')
suf = [r[0] for r in con.execute("select name from sqlite_master where type='table' and name like 'rocpd_kernel_dispatch%'")][0].replace('rocpd_kernel_dispatch', '')
pm = con.execute("""select p.name, sum(e.value) from rocpd_pmc_event%s e join rocpd_info_pmc%s p on e.pmc_id = p.id
                    join rocpd_kernel_dispatch%s d on e.event_id = d.event_id join rocpd_info_kernel_symbol%s s on d.kernel_id = s.id
                    where s.kernel_name like '%%k_nuts%%' group by p.name order by 2 desc""" % (suf, suf, suf, suf)).fetchall()
print('per chain-leapfrog (%.4g gradients): ' % G + ', '.join('%s %.1f' % (n.replace('SQ_', ''), v / G) for n, v in pm))
PY
