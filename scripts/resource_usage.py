"""Parse `hipcc -Rpass-analysis=kernel-resource-usage` remarks (stderr saved to a file) into one line per kernel."""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
for b in re.split(r'remark: (?:\S+: )?Function Name: ', txt)[1:]:
    name = b.split()[0]
    dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    g = lambda k: re.search(k + r': (\d+)', b).group(1)
    print('%-62s V %3s A %3s S %3s scratch %4s occ %s lds %s' % (
        dem.replace('epx::', '').replace('(NutsArgs)', '').replace('void ', '')[:62], g('VGPRs'), g('AGPRs'), g('TotalSGPRs'),
        g(r'ScratchSize \[bytes/lane\]'), g(r'Occupancy \[waves/SIMD\]'), g(r'LDS Size \[bytes/block\]')) +
          ' sgpr-spill %s vgpr-spill %s' % (g('SGPRs Spill'), g('VGPRs Spill')))
