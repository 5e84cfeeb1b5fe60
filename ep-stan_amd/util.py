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

DEVICE = 0


def invert_normal_params(A, b=None, out_A=None, out_b=None, cho_form=False):
    """Invert moment parameters into natural parameters or vice versa.

    (S, m) -> (Q, r) and back: returns (A^-1, A^-1 b).  `out_A`/`out_b` may be
    None (new arrays), an ndarray, or the string 'in-place'.  With
    `cho_form=True`, `A` holds the UPPER Cholesky factor of the real matrix.
    Raises LinAlgError if A is not positive definite (util.py:82-86).
    """
    lib = _lib.load()
    if not isinstance(out_A, np.ndarray) and out_A == 'in-place':
        out_A = A
    elif out_A is None:
        out_A = A.copy(order='F')
    else:
        np.copyto(out_A, A)
    if not out_A.flags['FARRAY']:
        # C-order -> F-order by transposing (symmetric; util.py:95-99)
        out_A = out_A.T
        if not out_A.flags['FARRAY'] and out_A.shape[0] > 1:
            raise ValueError('Provided array A is inappropriate')
    if out_A.dtype != np.float64:
        raise ValueError('Provided array A has to be float64')
    if b is not None:
        if not isinstance(out_b, np.ndarray) and out_b == 'in-place':
            out_b = b
        elif out_b is None:
            out_b = b.copy()
        else:
            np.copyto(out_b, b)
    else:
        out_b = None
    d = out_A.shape[0]
    info = np.zeros(1, dtype=np.int32)
    if cho_form and not out_A.flags['C_CONTIGUOUS'] and d > 1:
        # the kernel reads the upper factor; a lower triangle of junk is ignored
        pass
    bb = None
    if out_b is not None:
        bb = out_b if (out_b.flags['C_CONTIGUOUS'] and out_b.dtype == np.float64) \
            else np.ascontiguousarray(out_b, dtype=np.float64)
    check(lib.epx_invert_normal_params(DEVICE, d, 1, dptr(out_A), dptr(bb), 1 if cho_form else 0,
                                       info.ctypes.data_as(_lib.c_int32_p)))
    if info[0]:
        raise LinAlgError('matrix is not positive definite')
    if out_b is not None and bb is not out_b:
        np.copyto(out_b, bb)
    return out_A, out_b


def olse(S, n, P=None, out=None):
    """Optimal linear shrinkage estimator of the precision matrix (Bodnar,
    Gupta, Parolya, arXiv:1308.0931) from a sample covariance `S` of `n`
    draws, shrinking towards `P` (None: the naive I/d prior)."""
    lib = _lib.load()
    if not isinstance(out, np.ndarray) and out == 'in-place':
        out = S
    elif out is None:
        out = S.copy(order='F')
    else:
        np.copyto(out, S)
    if not out.flags['FARRAY']:
        out = out.T
        if not out.flags['FARRAY']:
            raise ValueError('Provided array should be in F-order')
    d = out.shape[0]
    Pa = None
    if P is not None:
        Pa = np.require(P, dtype=np.float64, requirements=['F', 'A'])
    info = np.zeros(1, dtype=np.int32)
    check(lib.epx_olse(DEVICE, d, 1, dptr(out), int(n), dptr(Pa), info.ctypes.data_as(_lib.c_int32_p)))
    if info[0]:
        raise LinAlgError('matrix is not positive definite')
    return out


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
