"""The exponent polynomial of f16x3.h's GELU: h(a) = 0.5 erfc(a / sqrt 2) ~= 2^Q(a), Q of degree 6, weighted minimax fit
(Lawson iterations) of log2 h on [0, 10] with weight a h(a) ln 2 -- the factor by which an error of Q enters
GELU(x) = max(x, 0) - |x| h(|x|) -- and the fp32 accuracy of the result against the Abramowitz-Stegun form it replaced.

    python tools/exp/gelu_fit.py
"""
import numpy as np
from numpy.polynomial import chebyshev as C, polynomial as P
from scipy.special import erf, erfc


def fit(X=10.0, n=6, iters=300):
    x = np.linspace(0, X, 40001)
    h = 0.5 * erfc(x / np.sqrt(2))
    target, w = np.log2(h), x * h * np.log(2) + 1e-30
    V, lw = C.chebvander(2 * x / X - 1, n), np.ones_like(x)
    for _ in range(iters):
        coef, *_ = np.linalg.lstsq(V * (w * lw)[:, None], target * w * lw, rcond=None)
        err = np.abs((V @ coef - target) * w)
        lw = lw * (0.3 + err / err.max())
        lw /= lw.mean()
    return C.Chebyshev(coef, domain=[0, X]).convert(kind=P.Polynomial, domain=[-1, 1], window=[-1, 1]).coef, err.max()


def fma(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def gelu_exp2(x, c):
    x = x.astype(np.float32)
    ax = np.abs(x)
    a = np.minimum(ax, np.float32(12))
    q = np.full_like(a, c[6])
    for k in range(5, -1, -1):
        q = fma(q, a, np.full_like(a, c[k]))
    e = np.exp2(q.astype(np.float64)).astype(np.float32)  # the hardware's v_exp_f32 is good to 1 ulp
    return fma(-ax, e, np.maximum(x, np.float32(0)))


def gelu_as(x):
    x = x.astype(np.float32)
    ax = np.abs(x)
    t = (1 / fma(ax, np.full_like(x, np.float32(0.3275911 * 0.70710678118654752440)), np.ones_like(x)).astype(np.float64)).astype(np.float32)
    p = fma(np.full_like(x, np.float32(0.5 * 1.061405429)), t, np.full_like(x, np.float32(0.5 * -1.453152027)))
    for k in (0.5 * 1.421413741, 0.5 * -0.284496736, 0.5 * 0.254829592):
        p = fma(p, t, np.full_like(x, np.float32(k)))
    e = np.exp2(((x * x) * np.float32(-0.72134752044448170368)).astype(np.float64)).astype(np.float32)
    return fma(-ax, ((p * t).astype(np.float32) * e).astype(np.float32), np.maximum(x, np.float32(0)))


if __name__ == "__main__":
    np.set_printoptions(precision=17)
    coef, err = fit()
    print("coefficients (ascending powers):", coef, "\nmax |GELU error| in exact arithmetic: %.2e" % err)
    xs = np.concatenate([np.linspace(-14, 14, 2000001), np.random.default_rng(0).normal(0, 1.5, 2000000)]).astype(np.float32)
    ref = 0.5 * xs.astype(np.float64) * (1 + erf(xs.astype(np.float64) / np.sqrt(2)))
    for name, g in (("2^Q form, fp32", gelu_exp2(xs, coef.astype(np.float32))), ("Abramowitz-Stegun 7.1.26, fp32", gelu_as(xs))):
        e = np.abs(g.astype(np.float64) - ref)
        print("%-32s max %.3e  rms %.2e" % (name, e.max(), np.sqrt((e ** 2).mean())))
