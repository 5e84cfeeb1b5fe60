"""Experiment plumbing of the EP branch of /root/reference/experiment/fit.py.

SURVEY.md §8(f) rank 1: `configurations`, the default damping schedule and
iteration count, `main(model_name, conf)` for `run_ep` with K == J (one group
per site, the `_sg` densities), the result `.npz` schema
(`m_s_ep, S_s_ep, time_s_ep, mstepsize_s_ep, mrhat_s_ep, othertimes`) with the
initial approximation prepended, and `kl_mvn` (plot_res.py:41-60) to score it.
The competing methods of fit.py (full model, consensus MC, target run) need
Stan itself and are out of scope; asking for them raises NotImplementedError.
"""

import os

import numpy as np

from . import models
from .method import Master
from .util import invert_normal_params, distribute_groups

CONFS = [
    'J', 'D', 'npg', 'cor_input',
    'run_all', 'run_ep', 'run_full', 'run_consensus', 'run_target',
    'iter', 'siter', 'target_siter', 'chains',
    'K', 'damp', 'mix', 'prec_estim',
    'seed_data', 'seed_ep', 'seed_full', 'seed_cons', 'seed_target',
    'id', 'save_true', 'save_res', 'save_target_samp',
]

# fit.py:134-168
CONF_DEFAULT = dict(
    J=64, D=16, K=32, npg=20, cor_input=True,
    run_all=False, run_ep=False, run_full=False, run_consensus=False, run_target=False,
    iter=None, siter=200, target_siter=10000, chains=4,
    damp=None, mix=False, prec_estim='sample',
    seed_data=100, seed_ep=1, seed_full=2, seed_cons=3, seed_target=4,
    id=None, save_true=True, save_res=True, save_target_samp=False,
)

EP_DEFAULT_ITERS_TO_RUN = lambda K: int(max(4*K, 20))        # fit.py:174
default_df0 = models.default_df0                              # fit.py:176-186

RES_PATH = os.path.join(os.getcwd(), 'results')


class configurations(object):
    """Configuration container for `main` (fit.py:189-207)."""

    def __init__(self, **kwargs):
        for k, v in kwargs.items():
            if k not in CONF_DEFAULT:
                raise ValueError("Invalid option `{}`".format(k))
            setattr(self, k, v)
        for k, v in CONF_DEFAULT.items():
            if k not in kwargs:
                setattr(self, k, v)

    def __str__(self):
        conf_dict = self.__dict__
        opts = ['{!s} = {!r}'.format(opt, conf_dict[opt]) for opt in CONFS if opt in conf_dict]
        return '\n'.join(opts)

    __repr__ = __str__


def kl_mvn(m0, S0, m1, S1):
    """KL(p||q), p ~ N(m0,S0), q ~ N(m1,S1) (plot_res.py:41-60); the inverse and
    the Cholesky log-determinants come from the device routines."""
    d = len(m0)
    Q1, _ = invert_normal_params(np.asfortranarray(S1, dtype=np.float64))
    dm = np.asarray(m1) - np.asarray(m0)
    # log det via the eigenvalues of the SPD matrices (host, d x d once per evaluation)
    ld0 = np.linalg.slogdet(S0)[1]
    ld1 = np.linalg.slogdet(S1)[1]
    return 0.5 * (np.trace(Q1.dot(S0)) + dm.dot(Q1.dot(dm)) - d) - 0.5 * ld0 + 0.5 * ld1


def main(model_name, conf, ret_master=False, verbose=True, _engine_factory=None, **master_kwargs):
    """The `run_ep` branch of fit.py:210-459 for K == J.

    Returns the dict that is saved to `res_d_<model>.npz` (or the Master when
    `ret_master`)."""
    if not isinstance(conf, configurations):
        raise ValueError("Invalid arg. `conf`, use class fit.configurations")
    if conf.run_full or conf.run_consensus or conf.run_target or conf.run_all:
        raise NotImplementedError("only the distributed EP method (`run_ep`) is built; the full, "
                                  "consensus and target runs need Stan itself")
    J, D, K = conf.J, conf.D, conf.K
    if model_name not in models.MODELS:
        raise ValueError("unknown model {!r}; available: {}".format(model_name, sorted(models.MODELS)))
    model = models.MODELS[model_name](J, D, conf.npg)
    if conf.cor_input:
        data = model.simulate_data(Sigma_x='rand', rng=conf.seed_data)      # fit.py:235-238
    else:
        data = model.simulate_data(rng=conf.seed_data)
    S0, m0, Q0, r0 = model.get_prior()
    prior = {'Q': Q0, 'r': r0}
    iters_to_run = EP_DEFAULT_ITERS_TO_RUN(K) if conf.iter is None else conf.iter      # fit.py:284-287
    df0 = default_df0(K) if conf.damp is None else conf.damp                             # fit.py:289-293
    epstan_options = dict(prior=prior, prec_estim=conf.prec_estim, df0=df0, init_site=None,
                          chains=conf.chains, iter=conf.siter, warmup=None, thin=1)      # fit.py:296-305
    if K < 2:
        raise ValueError("K should be at least 2.")
    elif K > J:
        raise NotImplementedError("Splitting the groups not implemented.")               # fit.py:339-341
    if _engine_factory is not None:
        master_kwargs['_engine_factory'] = _engine_factory
    if K < J:
        # several groups per site (fit.py:310-324): the multi-group program m*b.stan
        Nk, Nj_k, j_ind_k = distribute_groups(J, K, data.Nj)
        epstan_master = Master(model_name, data.X, data.y, A_k={'J': Nj_k}, A_n={'j_ind': j_ind_k + 1},
                               site_sizes=Nk, **epstan_options, **master_kwargs)
    else:
        epstan_master = Master(model.site_model, data.X, data.y, site_sizes=data.Nj,
                               **epstan_options, **master_kwargs)                        # fit.py:326-335
    if ret_master:
        return epstan_master
    S_ep_init, m_ep_init = epstan_master.cur_approx()                                    # fit.py:351
    info, (m_s_ep, S_s_ep), (time_s_ep, mstepsize_s_ep, mrhat_s_ep, othertimes) = epstan_master.run(
        iters_to_run, return_analytics=True, seed=conf.seed_ep, verbose=verbose)         # fit.py:358-369
    time_s_ep = time_s_ep.cumsum()                                                       # fit.py:372-380
    S_s_ep = np.concatenate((S_ep_init[None, :, :], S_s_ep), axis=0)
    m_s_ep = np.concatenate((m_ep_init[None, :], m_s_ep), axis=0)
    time_s_ep = np.insert(time_s_ep, 0, 0.0)
    mstepsize_s_ep = np.insert(mstepsize_s_ep, 0, np.nan)
    mrhat_s_ep = np.insert(mrhat_s_ep, 0, np.nan)
    res = dict(conf=conf.__dict__, m_s_ep=m_s_ep, S_s_ep=S_s_ep, time_s_ep=time_s_ep,
               mstepsize_s_ep=mstepsize_s_ep, mrhat_s_ep=mrhat_s_ep, othertimes=othertimes)
    if info:
        res['last_iter'] = epstan_master.iter                                            # fit.py:391-403
    elif conf.mix:
        # fit.py:408-411, 440-441: the final approximation from the last samples of all the sites.  `mix_pred`
        # (fit.py:414-421) reads worker.fit, which the reference's own child-process sampler never keeps: out of scope
        S_ep, m_ep = epstan_master.mix_phi()
        res['m_phi_ep'] = m_ep
        res['S_phi_ep'] = S_ep
    if conf.save_res:
        os.makedirs(RES_PATH, exist_ok=True)
        fname = 'res_d_{}_{}.npz'.format(model_name, conf.id) if conf.id else 'res_d_{}.npz'.format(model_name)
        np.savez(os.path.join(RES_PATH, fname), **res)
    if info:
        raise RuntimeError('epstan algorithm failed with error code: {}'.format(info))   # fit.py:405-408
    res['phi_true'] = data.phi_true
    return res
