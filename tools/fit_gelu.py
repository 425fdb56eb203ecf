#!/usr/bin/env python3
"""Coefficients of the polynomial erf-GELU used where the result is rounded to bf16 (csrc/common.h: gelu_bf16*).

    xc = clamp(x, -4, 4), u = xc^2, t = 0.5 + xc * R(u)  ~  Phi(x),   GELU(x) = x * t,   GELU'(x) = t + xc * phi(xc)
R is fitted (Lawson-reweighted least squares towards minimax) to erf(x / sqrt 2) / (2 x) on (0, 4] with R(16) * 4 = 0.5 EXACTLY, so
that beyond the clamp t is exactly 0 / 1.  Error weight: the GELU error x^2 |dR| relative to max(|GELU(x)|, |GELU(-x)| floored at
2^-3).  Prints C initialisers and the error of an fp32 Horner evaluation over [-12, 12].
"""
import numpy as np
from scipy.special import erf

C_CLAMP, DEG = 4.0, 6      # round 4: degree 6 (one fma less; 5.8e-4 / 2.3e-4 instead of 4.3e-4 / 1.7e-4, still a quarter of a bf16 ulp)


def gelu(x):
    return 0.5 * x * (1 + erf(x / np.sqrt(2)))


def fit(c=C_CLAMP, d=DEG, iters=200):
    n = 6000
    x = np.cos(np.pi * (np.arange(n) + 0.5) / n) * 0.5 * c + 0.5 * c
    u = x * x
    target = erf(x / np.sqrt(2)) / (2 * x)
    w = np.maximum(x * x / np.abs(gelu(x)), x * x / np.maximum(np.abs(gelu(-x)), 0.125))
    # R(u) = 0.5/c + (u - c^2) Q(u), Q of degree d-1 in v = u / c^2
    base = 0.5 / c
    V = np.vander(u / (c * c), d, increasing=True) * (u - c * c)[:, None]
    lw = np.ones(n)
    for _ in range(iters):
        W = w * lw
        q = np.linalg.lstsq(V * W[:, None], (target - base) * W, rcond=None)[0]
        err = np.abs(V @ q + base - target) * w
        lw *= 0.5 + err / err.max()
        lw /= lw.mean()
    qu = q / (c * c) ** np.arange(d)                      # Q in powers of u
    r = np.zeros(d + 1)
    r[0] = base
    r[1:] += qu
    r[:-1] -= c * c * qu
    return r, err.max()


def horner32(r, xs):
    xc = np.clip(xs, -C_CLAMP, C_CLAMP).astype(np.float32)
    uc = (xc * xc).astype(np.float32)
    R = np.full_like(xs, np.float32(r[-1]))
    for k in range(len(r) - 2, -1, -1):
        R = (R * uc + np.float32(r[k])).astype(np.float32)
    t = (xc * R + np.float32(0.5)).astype(np.float32)
    y = (xs * t).astype(np.float32)
    e = np.exp2((uc * np.float32(-0.72134752)).astype(np.float32)).astype(np.float32)
    dy = ((xc * e).astype(np.float32) * np.float32(0.3989422804) + t).astype(np.float32)
    return y, dy, t


def exact_at_clamp(r):
    """Nudge r[0] until the fp32 Horner chain gives R(16) = 0.125 EXACTLY: then t = fma(+-4, 0.125, 0.5) is exactly 1 / 0 at and
    beyond the clamp (otherwise GELU(x) = x * 7e-7 for very negative x).  The shift is ~2e-7, far below the fit error."""
    r = np.array(r, np.float64)
    for _ in range(50):
        R = np.float32(r[-1])
        for k in range(len(r) - 2, -1, -1):
            R = np.float32(np.float32(R * np.float32(16.0)) + np.float32(r[k])) if False else np.float32(np.float64(R) * 16.0 + np.float64(np.float32(r[k])))
        if R == np.float32(0.125):
            return r
        r[0] += 0.125 - float(R)
    raise RuntimeError("no exact fp32 fixed point")


if __name__ == "__main__":
    r, e = fit()
    r = exact_at_clamp(r)
    print("weighted fit error", e)
    print("static constexpr float GELU_R[%d] = {%s};" % (len(r), ", ".join("%.10ef" % v for v in r)))
    xs = np.linspace(-12, 12, 1200001).astype(np.float32)
    y, dy, t = horner32(r, xs)
    x64 = xs.astype(np.float64)
    ref = gelu(x64)
    dref = 0.5 * (1 + erf(x64 / np.sqrt(2))) + x64 * np.exp(-0.5 * x64 * x64) / np.sqrt(2 * np.pi)
    err = np.abs(y - ref)
    print("max |err| / max(|gelu|, 2^-3):", (err / np.maximum(np.abs(ref), 0.125)).max())
    print("max |err| / |gelu| for x > 0:", (err / np.maximum(np.abs(ref), 1e-30))[xs > 1e-3].max())
    print("max |err| for x < 0:", err[xs < 0].max(), "at", xs[xs < 0][np.argmax(err[xs < 0])])
    print("max abs err of GELU':", np.abs(dy - dref).max(), "at", xs[np.argmax(np.abs(dy - dref))])
    print("t outside the clamp:", t[0], t[-1])
