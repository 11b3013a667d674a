"""Near-minimax polynomial coefficients (Chebyshev interpolation, 60-digit arithmetic) for the SO(3) series of the kernels:
    atan(u) / u        as a polynomial in y = -u^2,  u^2 <= 1/16   (so3_log: rotations below ~28 degrees)
    cos(sqrt x), sin(sqrt x) / sqrt x   as polynomials in y = -x,  x <= 1/4   (so3_exp: rotations below 1 rad)
Prints the coefficients (highest power first, as the Horner loops take them), rounded to double, and the maximum
relative error of the rounded polynomial over a dense grid.   python tools/series_coefficients.py
"""
from decimal import Decimal as D, getcontext
import math

getcontext().prec = 60
PI = D("3.14159265358979323846264338327950288419716939937510582097494459")


def dcos(x):
    x = D(x); s, t, k = D(0), D(1), 0
    while abs(t) > D(10) ** -58:
        s += t; k += 1; t = -t * x * x / ((2 * k - 1) * (2 * k))
    return s


def series(fn_terms, y):           # sum c_k y^k
    s, p = D(0), D(1)
    for c in fn_terms:
        s += c * p; p *= y
    return s


def atan_over_u(y):                # y = -u^2
    return series([D(1) / (2 * k + 1) for k in range(80)], y)


def cos_sqrt(y):                   # y = -x
    c, out = D(1), []
    for k in range(40):
        out.append(c); c = c / ((2 * k + 1) * (2 * k + 2))
    return series(out, y)


def sinc_sqrt(y):
    c, out = D(1), []
    for k in range(40):
        out.append(c); c = c / ((2 * k + 2) * (2 * k + 3))
    return series(out, y)


def cheb_fit(f, lo, hi, n):
    """Monomial coefficients (in y) of the degree-n interpolant of f at the Chebyshev nodes of [lo, hi]."""
    nodes = [(lo + hi) / 2 + (hi - lo) / 2 * dcos(PI * (2 * i + 1) / (2 * (n + 1))) for i in range(n + 1)]
    vals = [f(t) for t in nodes]
    # Newton divided differences, then expansion to monomials (all in 60 digits)
    coef = list(vals)
    for j in range(1, n + 1):
        for i in range(n, j - 1, -1):
            coef[i] = (coef[i] - coef[i - 1]) / (nodes[i] - nodes[i - j])
    poly = [D(0)] * (n + 1)
    for i in range(n, -1, -1):      # poly = poly * (y - nodes[i]) + coef[i]
        new = [D(0)] * (n + 1)
        for k in range(n):
            new[k + 1] += poly[k]
            new[k] -= nodes[i] * poly[k]
        new[0] += coef[i]
        poly = new
    return poly


def report(name, f, lo, hi, n):
    poly = cheb_fit(f, D(lo), D(hi), n)
    dbl = [float(c) for c in poly]
    worst = D(0)
    for i in range(2001):
        y = D(lo) + (D(hi) - D(lo)) * i / 2000
        p = D(0)
        for c in reversed(dbl):
            p = p * y + D(c)
        worst = max(worst, abs(p / f(y) - 1))
    print(f"{name}: degree {n} in y on [{lo}, {hi}], max relative error of the rounded polynomial {float(worst):.2e}")
    print("   " + ", ".join(float(c).hex() for c in reversed(dbl)))
    print("   " + ", ".join(repr(float(c)) for c in reversed(dbl)))


if __name__ == "__main__":
    for n in (8, 9, 10):
        report("atan(u)/u", atan_over_u, "-0.0625", "0", n)
    for n in (5, 6):
        report("cos sqrt x", cos_sqrt, "-0.25", "0", n)
        report("sin sqrt x / sqrt x", sinc_sqrt, "-0.25", "0", n)
