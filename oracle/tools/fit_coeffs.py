#!/usr/bin/env python3
"""Fit the polynomial kernels of the rm_math spec (DESIGN.md "rm_math").

Test-infrastructure tooling: produces the fp32 coefficient tables that BOTH the
CPU oracle (oracle/rm_math.h) and the HIP kernel (raymarcher_amd/csrc/rm_math.hip.h)
hard-code.  Weighted least-squares on Chebyshev nodes in float64, coefficients rounded
to float32, max error measured on a dense grid (float64 evaluation of the rounded
coefficients, so only approximation error, not fp32 evaluation rounding).
Run:  python oracle/tools/fit_coeffs.py
"""
import numpy as np

def cheb_nodes(a, b, n):
    k = np.arange(n)
    x = np.cos(np.pi * (k + 0.5) / n)
    return 0.5 * (a + b) + 0.5 * (b - a) * x

def lsq_fit(basis_fn, target_fn, weight_fn, a, b, ncoef, iters=40):
    """min max |w*(sum c_i basis_i - target)| by iteratively re-weighted LSQ (Lawson)."""
    x = cheb_nodes(a, b, 4000)
    A = np.stack([basis_fn(x, i) for i in range(ncoef)], axis=1)
    t = target_fn(x)
    w = weight_fn(x)
    lw = np.ones_like(x)
    c = None
    for _ in range(iters):
        W = (w * np.sqrt(lw))[:, None]
        c, *_ = np.linalg.lstsq(A * W, t * W[:, 0], rcond=None)
        err = np.abs(w * (A @ c - t))
        lw = lw * (err / err.max() + 1e-3)
        lw /= lw.sum()
    return c

def report(name, c32, approx_fn, exact_fn, a, b, rel=True):
    x = np.linspace(a, b, 2000001)
    x = x[x != 0]
    e = approx_fn(x, c32.astype(np.float64)) - exact_fn(x)
    if rel:
        e = e / np.abs(exact_fn(x))
    print(f"// {name}: max {'rel' if rel else 'abs'} approx err = {np.abs(e).max():.3e}")
    print(f"static const float {name}[] = {{" + ", ".join(f"{v:.9e}f" for v in c32) + "};")
    print("//   hex: " + " ".join(hex(int(np.float32(v).view(np.uint32))) for v in c32))

def main():
    q = np.pi / 4
    # sin(r) = r + r^3 * S(z), z=r^2  on |r|<=pi/4
    for n in (3, 4):
        c = lsq_fit(lambda x, i: x ** (3 + 2 * i), lambda x: np.sin(x) - x,
                    lambda x: 1 / np.abs(np.sin(x)), 1e-4, q * 1.02, n)
        c32 = c.astype(np.float32)
        report(f"RM_SIN_C{n}", c32,
               lambda x, cc: x + sum(cc[i] * x ** (3 + 2 * i) for i in range(len(cc))), np.sin, -q, q)
    # cos(r) = 1 - z/2 + z^2 * C(z)
    for n in (3, 4):
        c = lsq_fit(lambda x, i: x ** (4 + 2 * i), lambda x: np.cos(x) - 1 + x * x / 2,
                    lambda x: 1 / np.abs(np.cos(x)), 1e-4, q * 1.02, n)
        c32 = c.astype(np.float32)
        report(f"RM_COS_C{n}", c32,
               lambda x, cc: 1 - x * x / 2 + sum(cc[i] * x ** (4 + 2 * i) for i in range(len(cc))), np.cos, -q, q)
    # asin(x) = x + x*z*P(z), z = x^2 in [0, 0.25]
    for n in (5, 6):
        c = lsq_fit(lambda x, i: x ** (3 + 2 * i), lambda x: np.arcsin(x) - x,
                    lambda x: 1 / np.abs(np.arcsin(x)), 1e-4, 0.5 * 1.01, n)
        c32 = c.astype(np.float32)
        report(f"RM_ASIN_C{n}", c32,
               lambda x, cc: x + sum(cc[i] * x ** (3 + 2 * i) for i in range(len(cc))), np.arcsin, -0.5, 0.5)
    # acos(x) = sqrt(1-x) * P(x), x in [0,1)  (Abramowitz & Stegun 4.4.46 form; P analytic on [0,1], P(1) = sqrt(2))
    acr = lambda x: np.arccos(np.minimum(x, 1.0)) / np.sqrt(np.maximum(1.0 - x, 1e-300))
    for n in (7, 8, 9):
        c = lsq_fit(lambda x, i: x ** i, acr, lambda x: 1 / np.abs(acr(x)), 0.0, 1.0 - 1e-9, n, iters=80)
        c32 = c.astype(np.float32)
        report(f"RM_ACOS_C{n}", c32, lambda x, cc: sum(cc[i] * x ** i for i in range(len(cc))), acr, 0.0, 1.0 - 1e-9)
    # atan(t) = t + t*s*P(s), s=t^2, t in [0,1]
    for n in (8, 9, 10):
        c = lsq_fit(lambda x, i: x ** (3 + 2 * i), lambda x: np.arctan(x) - x,
                    lambda x: 1 / np.abs(np.arctan(x)), 1e-4, 1.0, n)
        c32 = c.astype(np.float32)
        report(f"RM_ATAN_C{n}", c32,
               lambda x, cc: x + sum(cc[i] * x ** (3 + 2 * i) for i in range(len(cc))), np.arctan, 1e-6, 1.0)
    # log2(1+f) = f * L(f), f in [sqrt(.5)-1, sqrt(2)-1]
    lo, hi = np.sqrt(0.5) - 1, np.sqrt(2.0) - 1
    for n in (8, 9, 10):
        c = lsq_fit(lambda x, i: x ** (1 + i), lambda x: np.log2(1 + x),
                    lambda x: 1 / np.abs(np.log2(1 + x)), lo, hi, n)
        c32 = c.astype(np.float32)
        report(f"RM_LOG2_C{n}", c32,
               lambda x, cc: sum(cc[i] * x ** (1 + i) for i in range(len(cc))), lambda x: np.log2(1 + x), lo, hi)
    # 2^f = 1 + f*E(f), f in [-0.5, 0.5]
    for n in (5, 6, 7):
        c = lsq_fit(lambda x, i: x ** (1 + i), lambda x: np.exp2(x) - 1,
                    lambda x: 1 / np.exp2(x), -0.5, 0.5, n)
        c32 = c.astype(np.float32)
        report(f"RM_EXP2_C{n}", c32,
               lambda x, cc: 1 + sum(cc[i] * x ** (1 + i) for i in range(len(cc))), np.exp2, -0.5, 0.5)

if __name__ == "__main__":
    main()
