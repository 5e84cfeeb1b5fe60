"""EP iterations at the C3 shape with and without the split launch (lead sites one workgroup per
chain): device time of each sampling launch and the slowest chain."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from epstan_amd import models
from epstan_amd.method import Master
J, D, n = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (512, 32, 500)))
fracs = [float(f) for f in sys.argv[4:]] or [2.0, 0.4]
mod = models.MODELS['m4b'](J, D, n)
data = mod.simulate_data(Sigma_x='rand', rng=100)
_, _, Q0, r0 = mod.get_prior()
for frac in fracs:
    Master.LEAD_FRACTION = frac
    M = Master(mod.site_model, data.X, data.y, site_sizes=data.Nj, prior={'Q': Q0, 'r': r0}, chains=4, iter=200,
               df0=(models.default_df0(J) if os.environ.get('AB_DF0') else None))
    sched = {'order': None}
    eng = M.engine
    _set = eng.set_site_order
    def spy(order=None, _set=_set, sched=sched):
        sched['next'] = None if order is None else np.array(order)
        _set(order)
    eng.set_site_order = spy
    for it in range(int(os.environ.get('AB_ITERS', '6'))):
        sched['order'] = sched.get('next')
        M.run(1, verbose=False)
        cs = M.engine.get_chain_stats(4)[:, :, 3]
        m = eng.last_split()
        if m:
            o = sched['order']
            print('      lead sites: slowest chain %d (x5.64us = %.0f ms); other sites: slowest chain %d (x10.3us = %.0f ms), %d of them > 77K'
                  % (cs[o[:m]].max(), cs[o[:m]].max() * 5.64e-3, cs[o[m:]].max(), cs[o[m:]].max() * 10.3e-3,
                     (cs[o[m:]].max(axis=1) > 77000).sum()))
        print('LEAD_FRACTION %.2f iter %d: %7.1f ms, lead sites %3d, slowest chain %d, total leapfrogs %.1fM'
              % (frac, it, M.sampling_ms[-1], M.engine.last_split(), cs.max(), cs.sum() / 1e6), flush=True)
