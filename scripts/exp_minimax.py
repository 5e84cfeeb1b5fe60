"""Coefficients of the degree-N polynomial that replaces the degree-13 Taylor form of exp(r) on |r| <= ln2/2 in
csrc/epx_device.h (exp_d and its copies): the Chebyshev interpolant of exp on [-h, h], h = 0.3466 (near-minimax), in
the monomial basis, computed in 60-digit arithmetic; then the error of the DOUBLE-PRECISION Horner evaluation with the
rounded coefficients against the exact exponential on a dense grid.   python3 scripts/exp_minimax.py [degree]"""
import sys
import mpmath as mp
import numpy as np

mp.mp.dps = 60
N = int(sys.argv[1]) if len(sys.argv) > 1 else 11
h = mp.mpf('0.3466')                    # ln2/2 = 0.34657..., plus the slack of the two-constant range reduction
nodes = [h * mp.cos(mp.pi * (2 * j + 1) / (2 * (N + 1))) for j in range(N + 1)]
# Remez would be the last step; the Chebyshev interpolant is within a few per cent of the minimax error here
A = mp.matrix(N + 1, N + 1)
b = mp.matrix(N + 1, 1)
for j, x in enumerate(nodes):
    for i in range(N + 1):
        A[j, i] = x ** i
    b[j] = mp.e ** x
c = mp.lu_solve(A, b)
coef = [float(c[i]) for i in range(N + 1)]
print('degree', N)
for i in range(N, -1, -1):
    print('  c[%2d] = %.17g   (Taylor 1/%d! = %.17g)' % (i, coef[i], i, float(1 / mp.factorial(i))))
# approximation error of the exact-coefficient polynomial, and the error of the double Horner with rounded coefficients
xs = np.linspace(-float(h), float(h), 200001)
worst_apx, worst_dbl = 0.0, 0.0
for x in xs[::50]:
    xm = mp.mpf(float(x))
    ex = mp.e ** xm
    p = mp.mpf(0)
    for i in range(N, -1, -1):
        p = p * xm + c[i]
    worst_apx = max(worst_apx, abs(float((p - ex) / ex)))
p = np.full_like(xs, coef[N])
for i in range(N - 1, -1, -1):
    p = p * xs + coef[i]              # (numpy: mul + add, not fused -- an upper bound for the fused device form)
ref = np.array([float(mp.e ** mp.mpf(float(x))) for x in xs[::20]])
worst_dbl = np.max(np.abs(p[::20] - ref) / ref)
pt = np.full_like(xs, 1.6059043836821613e-10)
for cc in (2.08767569878681e-09, 2.505210838544172e-08, 2.755731922398589e-07, 2.7557319223985893e-06, 2.48015873015873e-05,
           1.984126984126984e-04, 1.388888888888889e-03, 8.333333333333333e-03, 4.1666666666666664e-02, 1.6666666666666666e-01, 0.5, 1.0, 1.0):
    pt = pt * xs + cc
print('approximation error of the exact polynomial: %.3g relative' % worst_apx)
print('double-precision Horner with the rounded coefficients: %.3g relative (the degree-13 Taylor form, same evaluation: %.3g)'
      % (worst_dbl, np.max(np.abs(pt[::20] - ref) / ref)))
