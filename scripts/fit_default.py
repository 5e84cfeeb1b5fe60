"""The reference's default experiment (`python fit.py m4b --run_ep true`: J = 64 groups on K = 32 sites,
D = 16, 20 rows per group, 4 x 200 NUTS iterations per site update, fit.py:134-168) end to end on the
device; iterations from argv (the reference's default is max(4K, 20) = 128)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from epstan_amd import fit
niter = int(sys.argv[1]) if len(sys.argv) > 1 else 40
model = sys.argv[2] if len(sys.argv) > 2 else 'm4b'
K = int(sys.argv[3]) if len(sys.argv) > 3 else 32
conf = fit.configurations(run_ep=True, iter=niter, save_res=False, K=K)
t0 = time.time()
from epstan_amd import method
made = []
_init = method.Master.__init__
def spy(self, *a, **k):
    _init(self, *a, **k); made.append(self)
method.Master.__init__ = spy
res = fit.main(model, conf, verbose=False)
M = made[-1]
dt = time.time() - t0
m, S = res['m_s_ep'], res['S_s_ep']
phi = res['phi_true']
mse = np.mean((m - phi)**2, axis=1)
sd = np.sqrt(np.diagonal(S, axis1=1, axis2=2))
z = np.abs(m[-1] - phi) / sd[-1]
print('%s: %d EP iterations in %.1f s (%.0f ms per iteration, sampling %.0f ms), layout %d, P = %d'
      % (model, niter, dt, dt / niter * 1e3, np.mean(M.sampling_ms), M.engine.last_layout(), M.engine.P))
print('MSE(mean, phi_true) at iterations 0, 5, 10, 20, last:', np.array2string(mse[[0, min(5, niter), min(10, niter), min(20, niter), -1]], precision=4))
print('largest |z|: coordinates', np.argsort(-z)[:6], 'z', np.array2string(np.sort(z)[::-1][:6], precision=1), 'mean', np.array2string(m[-1][np.argsort(-z)[:6]], precision=2), 'true', np.array2string(phi[np.argsort(-z)[:6]], precision=2))
print('|mean - phi_true| / posterior sd at the end: median %.2f, max %.2f; accepted damping factors: first %.3f last %.3f'
      % (np.median(z), z.max(), M.df_log[0], M.df_log[-1]))
