"""Damping-factor study of /root/reference/experiment/find_damp.py on the GPU path.

SURVEY.md §8(f) rank 4 / BASELINE config C5's "find_damp path": every EP iteration scores
N_DAMP damping factors for the pending site updates against a target posterior (mean squared
error of the mean, KL divergence, log-likelihood of target samples), then applies the
preselected schedule `fit.default_df0` with the reference's decay rule.  The reference runs the
31 trials one after the other in NumPy (31 x (K cavities + a Cholesky), find_damp.py:146-173);
here they are one batched device call (`Master.damp_sweep`), the proposal being affine in df.
"""

import os

import numpy as np

from . import fit
from .method import Master

CHAINS = 8              # find_damp.py:27-29
SITER = 200
N_DAMP = 31

kl_mvn = fit.kl_mvn


def default_damps():
    return np.linspace(0, 1, N_DAMP + 2)[1:-1]                      # find_damp.py:107


def main(model_name, K=None, iters=None, target=None, conf=None, seed=None, verbose=True,
         save=True, **master_kwargs):
    """Mirror of find_damp.main (find_damp.py:58-257).

    `target`: dict with `m_target`, `S_target` (and optionally `samp_target`, `conf`) or the
    path of a `target_<model>.npz` written by the reference's `fit.py --run_target`; by default
    `results/target_<model>.npz` next to the working directory, as in the reference.
    Returns the dict the reference saves to `find_damp_K<K>.npz`."""
    if target is None:
        target = os.path.join(fit.RES_PATH, 'target_{}.npz'.format(model_name))
    if isinstance(target, str):
        with np.load(target, allow_pickle=True) as tf:
            target = {k: tf[k] for k in tf.files}
        if 'conf' in target:
            target['conf'] = target['conf'][()]
    m_target = np.asarray(target['m_target'], dtype=np.float64)
    S_target = np.asarray(target['S_target'], dtype=np.float64)
    samp_target = target.get('samp_target')
    if conf is None:
        tconf = target['conf']
        J, D = tconf['J'], tconf['D']
        if K is None:
            K = J
        conf = fit.configurations(J=J, D=D, K=K, chains=CHAINS, siter=SITER, save_true=False)   # :83-84
    else:
        K = conf.K
    if iters is None:
        iters = fit.EP_DEFAULT_ITERS_TO_RUN(K)                      # :80-81
    master = fit.main(model_name, conf, ret_master=True, verbose=verbose, **master_kwargs)      # :85
    damps = default_damps()
    mses = np.full((iters, N_DAMP), np.nan)
    lls = np.full((iters, N_DAMP), np.nan)
    kls = np.full((iters, N_DAMP), np.nan)
    damps_selected = np.full(iters, np.nan)
    mses_selected = np.full(iters + 1, np.nan)
    lls_selected = np.full(iters + 1, np.nan)
    kls_selected = np.full(iters + 1, np.nan)

    ld_target = np.linalg.slogdet(S_target)[1]

    def score(i):
        S, m = master.cur_approx()
        mses_selected[i] = np.mean((m - m_target)**2)               # :121-129
        Q1 = master.engine.invert_normal_params(np.asfortranarray(S), np.zeros(len(m)))[0]      # kl_mvn, :38-56
        dm = m - m_target
        kls_selected[i] = (0.5 * (np.sum(Q1 * S_target) + dm.dot(Q1.dot(dm)) - len(m))
                           - 0.5 * ld_target + 0.5 * np.linalg.slogdet(S)[1])
        if samp_target is not None:
            from scipy import stats
            lls_selected[i] = np.sum(stats.multivariate_normal.logpdf(samp_target, mean=m, cov=S.T))

    score(0)
    sweep = dict(damps=damps, m_target=m_target, S_target=S_target, samp_target=samp_target)
    rng = np.random.RandomState(seed)
    for it in range(iters):
        if verbose:
            print("Iteration {}/{}".format(it + 1, iters))
        # one EP iteration: tilted for every site, the sweep, then the preselected df0 with the
        # decay rule of :180-228 -- which is Master.run's own damping loop
        info = master.run(1, calc_moments=False, verbose=False, seed=rng, sweep=sweep)
        if info != Master.INFO_OK:
            if verbose:
                print("    stopped with info {}".format(info))
            break
        res = master.sweep_log[-1]
        mses[it], lls[it], kls[it] = res['mses'], res['lls'], res['kls']
        damps_selected[it] = master.df_log[-1]
        score(it + 1)
    out = dict(damps=damps, mses=mses, lls=lls, kls=kls, damps_selected=damps_selected,
               mses_selected=mses_selected, lls_selected=lls_selected, kls_selected=kls_selected)
    if save:
        os.makedirs(fit.RES_PATH, exist_ok=True)
        np.savez(os.path.join(fit.RES_PATH, 'find_damp_K{}.npz'.format(K)), **out)      # :246-256
    return out
