"""GPU forms of the hot-path helpers of /root/reference/epstan/util.py.

Same names, argument meaning and error behaviour as the reference's
`invert_normal_params` (util.py:51-125) and `olse` (util.py:128-194); the
arithmetic runs in the batched HIP kernels `k_invert` / `k_olse`
(csrc/dense.hip) through the C ABI (epx_invert_normal_params, epx_olse).
"""

__all__ = ['invert_normal_params', 'olse']

import numpy as np
from numpy.linalg import LinAlgError

from . import _lib
from ._lib import check, dptr

DEVICE = 0          # device of the stand-alone calls when none is given


def _destination(src, out, what):
    """Array the kernel works in: `out` may be None (a new F-ordered copy of `src`), the string
    'in-place' (`src` itself) or an ndarray (receives a copy of `src`) -- the three spellings of
    the reference's `out_*` arguments."""
    if isinstance(out, str):
        if out != 'in-place':
            raise ValueError("{} has to be None, an ndarray or 'in-place'".format(what))
        return src
    if out is None:
        return src.copy(order='F')
    np.copyto(out, src)
    return out


def _column_major(M, message):
    """A view of the square matrix that the kernels can address column by column.  A C-ordered
    SYMMETRIC matrix is its own transpose, so its transposed view serves."""
    if M.flags['FARRAY']:
        return M
    Mt = M.T
    if not Mt.flags['FARRAY'] and M.shape[0] > 1:
        raise ValueError(message)
    return Mt


def _device(device):
    return DEVICE if device is None else int(device)


def invert_normal_params(A, b=None, out_A=None, out_b=None, cho_form=False, device=None):
    """Invert moment parameters into natural parameters or vice versa.

    (S, m) -> (Q, r) and back: returns (A^-1, A^-1 b).  `out_A`/`out_b` may be
    None (new arrays), an ndarray, or the string 'in-place'.  With
    `cho_form=True`, `A` holds the UPPER Cholesky factor of the real matrix
    (whatever sits below the diagonal is ignored).  Raises LinAlgError if A is
    not positive definite (util.py:82-86).  `device`: HIP device the kernel runs
    on (not in the reference; default `util.DEVICE`).
    """
    lib = _lib.load()
    work = _column_major(_destination(A, out_A, 'out_A'), 'Provided array A is inappropriate')
    if work.dtype != np.float64:
        raise ValueError('Provided array A has to be float64')
    vec = None if b is None else _destination(b, out_b, 'out_b')
    staged = vec
    if vec is not None and not (vec.flags['C_CONTIGUOUS'] and vec.dtype == np.float64):
        staged = np.ascontiguousarray(vec, dtype=np.float64)
    info = np.zeros(1, dtype=np.int32)
    check(lib.epx_invert_normal_params(_device(device), work.shape[0], 1, dptr(work), dptr(staged),
                                       1 if cho_form else 0, info.ctypes.data_as(_lib.c_int32_p)))
    if info[0]:
        raise LinAlgError('matrix is not positive definite')
    if staged is not vec:
        np.copyto(vec, staged)
    return work, vec


def olse(S, n, P=None, out=None, device=None):
    """Optimal linear shrinkage estimator of the precision matrix (Bodnar,
    Gupta, Parolya, arXiv:1308.0931) from a sample covariance `S` of `n`
    draws, shrinking towards `P` (None: the naive I/d prior)."""
    lib = _lib.load()
    work = _column_major(_destination(S, out, 'out'), 'Provided array should be in F-order')
    if not work.flags['FARRAY']:
        raise ValueError('Provided array should be in F-order')
    prior = None if P is None else np.require(P, dtype=np.float64, requirements=['F', 'A'])
    info = np.zeros(1, dtype=np.int32)
    check(lib.epx_olse(_device(device), work.shape[0], 1, dptr(work), int(n), dptr(prior),
                       info.ctypes.data_as(_lib.c_int32_p)))
    if info[0]:
        raise LinAlgError('matrix is not positive definite')
    return work


def distribute_groups(J, K, Nj):
    """Distribute `J` groups of sizes `Nj` to `K` sites (util.py:541-640 of the reference).

    K < J: consecutive groups are combined, always the adjacent pair with the smallest joint
    size first (ties: the first such pair), until K sites are left.  Returns `(Nk, Nj_k,
    j_ind_k)`: rows per site, groups per site, and for every row its 0-based group index
    WITHIN its site.  K == J: `(Nj, None, None)`.  K > J (splitting groups) is not built."""
    Nj = np.asarray(Nj)
    if Nj.shape[0] != J:
        raise ValueError("J does not match the provided group sizes")
    if np.any(Nj <= 0):
        raise ValueError("Every group must have at least one item")
    if K < 2:
        raise ValueError("K should be at least 2.")
    if K == J:
        return Nj, None, None
    if K > J:
        raise NotImplementedError("Splitting the groups (K > J) is not built.")
    sizes = [int(n) for n in Nj]            # rows of every (merged) site
    groups = [1] * J                        # groups of every (merged) site
    while len(sizes) > K:
        pair = [sizes[i] + sizes[i + 1] for i in range(len(sizes) - 1)]
        i = pair.index(min(pair))
        sizes[i] = pair[i]
        groups[i] += groups[i + 1]
        del sizes[i + 1], groups[i + 1]
    Nk = np.array(sizes)
    Nj_k = np.array(groups)
    j_ind_k = np.concatenate([np.repeat(np.arange(g), Nj[o:o + g])
                              for g, o in zip(Nj_k, np.concatenate(([0], np.cumsum(Nj_k)[:-1])))]).astype(np.int32)
    return Nk, Nj_k, j_ind_k
