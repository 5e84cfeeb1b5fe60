"""Named parameters of the built-in site models, computed from the sampled coordinates.

The reference saves the draws of arbitrary Stan parameters by name
(`fit.extract(pars=par)[par]`, /root/reference/epstan/method.py:387-392, asked for by
`Worker.tilted(save_samples=...)` / `Master.run(save_last_param=...)`; experiment/fit.py:366 passes
the models' `('alpha', 'beta')`).  The device sampler keeps the unconstrained coordinates
`theta = [phi | eta (groups) | etb (groups x D)]`; this module restates the `parameters` and
`transformed parameters` blocks of experiment/models/m{1..5}{a,b}[_sg].stan on top of them.

Draws come back chain-major (all post-warm-up draws of chain 0, then chain 1, ...), not in
PyStan's random permutation.
"""

import numpy as np

# sampled blocks behind phi, per b-model id: does the program have the `etb` block?
_HAS_ETB = {0: False, 1: True, 2: True, 3: True, 4: True}


def layout(model_id, D, ng, gauss):
    """Index slices of theta for a site with `ng` groups."""
    o = 1 if gauss else 0
    d = o + {0: D + 1, 1: 2, 2: D + 1, 3: 2 * D + 2, 4: 2 * D + 2}[model_id]
    eta = slice(d, d + ng)
    etb = slice(d + ng, d + ng + ng * D) if _HAS_ETB[model_id] else None
    return o, d, eta, etb


def names(model_id, gauss):
    base = ['phi', 'eta', 'alpha', 'beta', 'sigma_a']
    if _HAS_ETB[model_id]:
        base += ['etb', 'sigma_b']
    if model_id >= 3:
        base += ['mu_a', 'mu_b']
    if gauss:
        base += ['sigma']
    return base


def named_draws(model_id, D, ng, gauss, single_group, theta, wanted):
    """theta: (S, P) draws of all sampled coordinates of one site -> {name: draws}.

    Scalars of a single-group program (`real alpha`) come back as (S,), vectors as (S, D); the
    multi-group programs declare `vector[J] alpha`, `vector[D] beta[J]`: (S, J) and (S, J, D)."""
    theta = np.asarray(theta, dtype=np.float64)
    S = theta.shape[0]
    o, d, sl_eta, sl_etb = layout(model_id, D, ng, gauss)
    phi = theta[:, :d]
    b = phi[:, o:]                                       # the b-model's phi
    eta = theta[:, sl_eta]                               # (S, ng)
    etb = theta[:, sl_etb].reshape(S, ng, D) if sl_etb is not None else None
    val = {'phi': phi, 'eta': eta}
    if gauss:
        val['sigma'] = np.exp(phi[:, 0])
    if model_id == 0:                                    # phi = [log sigma_a, beta]
        sig_a = np.exp(b[:, 0])
        val['alpha'] = eta * sig_a[:, None]
        val['beta'] = b[:, 1:1 + D]                      # `vector[D] beta` is shared by the groups (m1b.stan:27-31)
    elif model_id == 1:                                  # phi = [log sigma_a, log sigma_b]
        sig_a = np.exp(b[:, 0])
        val['sigma_b'] = np.exp(b[:, 1])
        val['alpha'] = eta * sig_a[:, None]
        val['beta'] = etb * val['sigma_b'][:, None, None]
    elif model_id == 2:                                  # phi = [log sigma_a, log sigma_b (D)]
        sig_a = np.exp(b[:, 0])
        val['sigma_b'] = np.exp(b[:, 1:1 + D])
        val['alpha'] = eta * sig_a[:, None]
        val['beta'] = etb * val['sigma_b'][:, None, :]
    else:                                                # phi = [mu_a, log sigma_a, mu_b (D), log sigma_b (D)]
        sig_a = np.exp(b[:, 1])
        val['mu_a'] = b[:, 0]
        val['mu_b'] = b[:, 2:2 + D]
        val['sigma_b'] = np.exp(b[:, 2 + D:2 + 2 * D])
        val['alpha'] = val['mu_a'][:, None] + eta * sig_a[:, None]
        val['beta'] = val['mu_b'][:, None, :] + etb * val['sigma_b'][:, None, :]
    val['sigma_a'] = sig_a
    if etb is not None:
        val['etb'] = etb
    out = {}
    for name in wanted:
        if name not in val:
            raise ValueError("parameter {!r} is not defined by this site model (known: {})"
                             .format(name, sorted(val)))
        v = np.array(val[name])
        if single_group and name in ('eta', 'alpha'):
            v = v[:, 0]
        elif single_group and name in ('etb', 'beta') and v.ndim == 3:
            v = v[:, 0, :]
        out[name] = v
    return out
