"""Where a pieced launch of the streaming sampler (C5 shard) loses time: per-piece timeline from the diagnostic build
(EPX_STAMPS=1 ep-stan_amd/csrc/build.sh; EPX_LIB=variants/libepx_stamps.so): every workgroup records when it started,
when it had claimed a site and when its piece was sampled (100 MHz clock common to all CUs).
   python3 scripts/piece_timeline.py [sites] [D] [n] [ep_iterations] [out.json]"""
import sys, os, json, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from epstan_amd import models, _lib
from epstan_amd.method import Master

J = int(sys.argv[1]) if len(sys.argv) > 1 else 512
D = int(sys.argv[2]) if len(sys.argv) > 2 else 128
n = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
nit = int(sys.argv[4]) if len(sys.argv) > 4 else 3
out = sys.argv[5] if len(sys.argv) > 5 else None
mod = models.m4b(J, D, n)
data = mod.simulate_data(rng=100)
_, _, Q0, r0 = mod.get_prior()
M = Master(mod.site_model, data.X, data.y, site_sizes=data.Nj, prior={'Q': Q0, 'r': r0}, chains=4, iter=200,
           prec_estim='olse', df0=models.default_df0(J), sync_sites=False)
info = M.run(nit, verbose=False, calc_moments=False, seed=1)
eng = M.engine
lib = _lib.load()
buf = np.zeros((40000, 8), dtype=np.uint64)
lib.epx_dbg_get_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
nb = lib.epx_dbg_get_stamps(eng.ctx, buf.ctypes.data, 40000)
nwg = (nb - 2) // 3
tl = buf[nwg:2 * nwg].astype(np.float64)
ok = tl[:, 2] > 0
tl = tl[ok]
t0 = tl[:, 0].min()
entry, claim, end = (tl[:, 0] - t0) / 100.0, (tl[:, 1] - t0) / 100.0, (tl[:, 2] - t0) / 100.0      # microseconds
span = end.max()
busy = (end - claim).sum()
wait = (claim - entry).sum()
ncu = eng.cu_count()
# CUs still sampling over time: the tail is where fewer than all of them are
edges = np.linspace(0, span, 201)
act = np.array([np.sum((claim <= t) & (end > t)) for t in edges])
full = act >= 0.98 * min(ncu, act.max())
res = {
    'workload': 'C5 shard: %d sites, D = %d, n_j = %d, EP iteration %d (pieced launch of k_nuts_stream, layout %d)' % (J, D, n, nit, eng.last_layout()),
    'form': 'one workgroup per piece' if os.environ.get('EPX_PIECE_GRID') or not os.environ.get('EPX_PIECE_LOOP') else 'looping workgroups',
    'launch_ms_by_events': float(M.sampling_ms[-1]), 'span_ms_by_piece_stamps': span / 1e3, 'pieces': int(len(tl)), 'cus': int(ncu),
    'sum_of_piece_sampling_ms_over_cus': busy / 1e3 / ncu, 'sum_of_claim_waits_ms_over_cus': wait / 1e3 / ncu,
    'share_of_the_span_with_every_cu_sampling': float(full.mean()),
    'mean_cus_sampling': float(act.mean()),
    'piece_ms': {'mean': float((end - claim).mean() / 1e3), 'p10': float(np.percentile(end - claim, 10) / 1e3),
                 'p90': float(np.percentile(end - claim, 90) / 1e3), 'max': float((end - claim).max() / 1e3)},
    'claim_wait_ms': {'mean': float((claim - entry).mean() / 1e3), 'p90': float(np.percentile(claim - entry, 90) / 1e3),
                      'max': float((claim - entry).max() / 1e3)},
    'us_per_leapfrog_of_a_piece': {'median': float(np.median((end - claim) / np.maximum(tl[:, 5] / 4.0, 1.0))),
                                   'p10': float(np.percentile((end - claim) / np.maximum(tl[:, 5] / 4.0, 1.0), 10)),
                                   'p90': float(np.percentile((end - claim) / np.maximum(tl[:, 5] / 4.0, 1.0), 90))},
    'cus_sampling_over_the_span_200_bins': [int(x) for x in act],
}
# why CUs idle in mid-launch: sites that still have pieces left (a site runs on one CU at a time), and the pieces in
# flight by blockIdx % 8 (blocks are dealt round-robin over the 8 XCDs: a free CU gets work only when the dispatcher's
# round-robin reaches its XCD)
sites = tl[:, 3].astype(int)
site_end = np.zeros(J)
np.maximum.at(site_end, sites, end)
res['unfinished_sites_over_the_span_200_bins'] = [int(np.sum(site_end > t)) for t in edges]
xcc_of = tl[:, 7].astype(np.int64) & 0xF              # (the XCD a piece ran on: blockIdx % 8 with one workgroup per piece)
res['pieces_in_flight_by_xcd_at_25_50_75_90_percent'] = [
    [int(np.sum((claim <= t) & (end > t) & (xcc_of == x))) for x in range(8)] for t in (0.25 * span, 0.5 * span, 0.75 * span, 0.9 * span)]
# per physical CU (HW_ID: cu_id bits 11:8, sh_id bit 12, se_id bits 15:13; XCC_ID bits 3:0): the gaps between the end of
# one piece and the claim of the next one on the same CU
hw = tl[:, 6].astype(np.int64); xcc = tl[:, 7].astype(np.int64) & 0xF
cu_key = (xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xF)
gaps = []
per_cu_busy = []
for key in np.unique(cu_key):
    idx = np.nonzero(cu_key == key)[0]
    o = idx[np.argsort(claim[idx])]
    gaps.extend(list(claim[o][1:] - end[o][:-1]))
    per_cu_busy.append(float((end[o] - claim[o]).sum()))
gaps = np.array(gaps)
res['physical_cus_seen'] = int(len(np.unique(cu_key)))
res['gap_between_pieces_on_a_cu_ms'] = {'mean': float(gaps.mean() / 1e3), 'median': float(np.median(gaps) / 1e3), 'p90': float(np.percentile(gaps, 90) / 1e3),
                                        'max': float(gaps.max() / 1e3), 'sum_over_cus_ms_per_cu': float(gaps.sum() / 1e3 / max(len(per_cu_busy), 1))}
res['busy_ms_per_physical_cu'] = {'min': float(np.min(per_cu_busy) / 1e3), 'median': float(np.median(per_cu_busy) / 1e3), 'max': float(np.max(per_cu_busy) / 1e3)}
# the tail: time after the last moment every CU was busy
last_full = edges[np.nonzero(full)[0].max()] if full.any() else 0.0
res['tail_ms_after_the_last_full_moment'] = (span - last_full) / 1e3
print(json.dumps({k: v for k, v in res.items() if not k.endswith('200_bins')}, indent=1)); print('unfinished sites', res['unfinished_sites_over_the_span_200_bins'][::10])
if out:
    json.dump(res, open(out, 'w'), indent=1)
