#!/usr/bin/env python3
"""The GELU of the tail kernel (lightglue_amd/csrc/lg_tail.hip: gelu_fast2), its fit and its fp32 error.

    GELU(u) = 0.5 u (1 + erf(u / sqrt 2))                                   (reference: nn.GELU(), lightglue.py:152-157)
            = 0.5 u + 0.5 |u| (1 - erfc(|u| / sqrt 2)),      erfc(a / sqrt 2) = exp2(-a Q(a))

Q = weighted minimax polynomial fit of -log2(erfc(a / sqrt 2)) / a on [0, 7]; the weight is the sensitivity of the GELU to Q
(0.5 ln2 a^2 erfc).  Degree 7 is the lowest degree whose leading coefficient comes out positive with a fit error below fp32
round-off (degree 5 / 6 / 8 have a NEGATIVE leading coefficient: a Q(a) eventually turns negative, exp2 of a large positive
number -> inf for large |u|).  With Q >= Q(0) = 1.151 for every a >= 0 the exponent only ever runs to -inf (erfc -> 0).

    tools/fit_gelu.py            fit, print the coefficients, evaluate the kernel's fp32 instruction sequence against float64
                                 (also for the Abramowitz-Stegun 7.1.26 form the kernel used in rounds 2-3)
tests/test_gelu_fit.py pins KERNEL_COEFFS to the source and bounds the error."""
import numpy as np
from scipy.special import erf, erfc

SQ = np.sqrt(0.5)
f32 = np.float32
# c0 .. c7 of Q as they stand (negated) in gelu_fast2
KERNEL_COEFFS = [1.1511110067367554, 0.4591621458530426, 0.052627623081207275, -0.00724543584510684, 0.00027208542451262474,
                 0.0001314696710323915, -2.8056274459231645e-05, 1.902015583254979e-06]


def fit(deg=7, a_max=7.0, n=20001, iters=60):
    a = np.linspace(1e-4, a_max, n)
    e = erfc(a * SQ)
    g = -np.log2(e) / a
    w = 0.5 * np.log(2) * a * a * e
    V = np.vander(a, deg + 1, increasing=True)
    ww = w.copy()
    for _ in range(iters):   # Lawson iteration towards the weighted minimax fit
        c, *_ = np.linalg.lstsq(V * ww[:, None], g * ww, rcond=None)
        err = np.abs((V @ c - g) * w)
        ww = ww * (0.5 + err / err.max())
        ww /= ww.max() / w.max()
    gelu_err = np.abs(0.5 * a * np.exp2(-(a * (V @ c))) - 0.5 * a * e)   # the GELU's error with Q evaluated in float64
    return c, float(gelu_err.max())


def gelu_kernel_fp32(u, c=KERNEL_COEFFS):
    """The instruction sequence of gelu_fast2, one rounding per instruction (fma = one rounding)."""
    u = np.asarray(u, f32)
    a = np.abs(u)
    p = np.full_like(a, f32(-c[-1]))
    for k in range(len(c) - 2, -1, -1):
        p = (p.astype(np.float64) * a + f32(-c[k])).astype(f32)
    arg = (p * a).astype(f32)
    e = np.exp2(arg.astype(np.float64)).astype(f32)                 # v_exp_f32 (1 ulp)
    wgt = (e.astype(np.float64) * -0.5 + 0.5).astype(f32)
    half_u = (u * f32(0.5)).astype(f32)
    return (a.astype(np.float64) * wgt + half_u).astype(f32)


def gelu_as_fp32(u):
    """Rounds 2-3: erf by Abramowitz-Stegun 7.1.26 (degree 8 refit), with a reciprocal."""
    u = np.asarray(u, f32)
    x = (u * f32(0.70710678118654752440)).astype(f32)
    ax = np.abs(x)
    tt = (1.0 / (ax.astype(np.float64) * f32(0.3275911) + 1.0).astype(f32).astype(np.float64)).astype(f32)
    cs = [-0.0779742014, 0.151737503, 0.39572154, -0.574341196, 0.810336914, -0.151473053, 0.270560832, 0.175431661]
    p = (tt.astype(np.float64) * f32(cs[0]) + f32(cs[1])).astype(f32)
    for ck in cs[2:]:
        p = (p.astype(np.float64) * tt + f32(ck)).astype(f32)
    p = (p * tt).astype(f32)
    e = np.exp2(((ax * ax).astype(f32) * f32(-1.44269504088896340736)).astype(f32).astype(np.float64)).astype(f32)
    er = (1.0 - p.astype(np.float64) * e).astype(f32)
    half_u = (u * f32(0.5)).astype(f32)
    return (half_u.astype(np.float64) * np.copysign(er, x) + half_u).astype(f32)


def gelu_exact(u):
    u = np.asarray(u, np.float64)
    return 0.5 * u * (1.0 + erf(u * SQ))


def error_report(fn, u):
    ref = gelu_exact(u)
    err = np.abs(fn(u).astype(np.float64) - ref)
    ulp = np.spacing(np.maximum(np.abs(ref), 1e-30).astype(f32)).astype(np.float64)
    small = np.abs(u) < 2
    return {"max_abs": float(err.max()), "at": float(u[err.argmax()]), "max_abs_below_2": float(err[small].max()),
            "rms": float(np.sqrt((err ** 2).mean())), "max_in_ulp_or_6e-8": float((err / np.maximum(ulp, 6e-8)).max())}


def main():
    for deg in (5, 6, 7, 8):
        c, e = fit(deg)
        aa = np.linspace(0.0, 5000.0, 500001)
        print(f"degree {deg}: GELU fit error (Q in float64) {e:.2e}, leading coefficient {c[-1]:+.3e}, min Q on [0, 5000] {np.polyval(c[::-1], aa).min():.4g}")
        if deg == 7:
            print("  coefficients (fp32):", [float(f32(x)) for x in c])
            print("  max |fit - kernel| coefficient difference:", float(np.abs(np.array([float(f32(x)) for x in c]) - np.array(KERNEL_COEFFS)).max()))
    u = np.concatenate([np.linspace(-10, 10, 2000001), np.random.default_rng(0).normal(0, 3, 1000000)]).astype(f32)
    u = u[np.abs(u) < 10]
    print("kernel form (exp2(-a Q(a))), fp32:", error_report(gelu_kernel_fp32, u))
    print("rounds 2-3 form (A&S 7.1.26), fp32:", error_report(gelu_as_fp32, u))
    big = np.array([20, 50, 100, 300, 1e3, 1e4, 1e6, 3e38], dtype=f32)
    print("large |u|:", gelu_kernel_fp32(big), gelu_kernel_fp32(-big))


if __name__ == "__main__":
    main()
