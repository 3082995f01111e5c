"""The tail kernel's GELU (csrc/lg_tail.hip: gelu_fast2) against nn.GELU()'s erf form (ref lightglue.py:152-157), on the CPU:
the coefficients in the source are the ones tools/fit_gelu.py produces, the fp32 instruction sequence stays inside fp32 round-off
of the exact function, and the exponent polynomial cannot turn positive for any input."""
import re
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))
import fit_gelu  # noqa: E402


def source_coefficients():
    src = (ROOT / "lightglue_amd" / "csrc" / "lg_tail.hip").read_text()
    body = src[src.index("f32x2 gelu_fast2(f32x2 u)"):]
    body = body[:body.index("const f32x2 arg = p * a;")]
    nums = [float(x) for x in re.findall(r"(-?\d+\.\d+(?:e[-+]?\d+)?)f", body)]
    assert len(nums) == 8, nums
    return [-x for x in nums[::-1]]          # the source holds -Q, highest degree first (Horner)


def test_kernel_coefficients_are_the_fitted_ones():
    got = source_coefficients()
    np.testing.assert_allclose(got, fit_gelu.KERNEL_COEFFS, rtol=0, atol=0)
    fitted, fit_err = fit_gelu.fit(7, n=4001, iters=40)
    assert fit_err < 2e-8
    np.testing.assert_allclose(np.float32(fitted), np.float32(got), rtol=2e-3, atol=2e-7)   # (a coarser grid than the script's: same polynomial to its last digits)


def test_fp32_sequence_is_within_round_off_of_the_erf_form():
    u = np.concatenate([np.linspace(-10, 10, 400001), np.random.default_rng(0).normal(0, 3, 200000)]).astype(np.float32)
    u = u[np.abs(u) < 10]
    rep = fit_gelu.error_report(fit_gelu.gelu_kernel_fp32, u)
    assert rep["max_abs"] < 4e-7 and rep["max_abs_below_2"] < 1.5e-7 and rep["max_in_ulp_or_6e-8"] < 2.0, rep


def test_exponent_never_turns_positive():
    c = fit_gelu.KERNEL_COEFFS
    a = np.concatenate([np.linspace(0, 100, 100001), np.logspace(2, 38, 2000)])
    assert np.polyval(c[::-1], a).min() >= c[0] - 1e-6            # Q >= Q(0) > 0: a Q(a) grows monotonically
    big = np.array([30, 1e3, 1e6, 3e38], dtype=np.float32)
    with np.errstate(over="ignore"):
        np.testing.assert_array_equal(fit_gelu.gelu_kernel_fp32(big), big)
        np.testing.assert_array_equal(fit_gelu.gelu_kernel_fp32(-big), np.zeros(4, np.float32))
