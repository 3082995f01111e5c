#!/usr/bin/env python3
"""Error of ONE 512-long product under the operand schemes discussed in DESIGN.md §1 / §7.1 (numpy emulation, no GPU), relative to
sum |x||w| per output: split-bf16 x3 (the round 1-2 default), f16 hi planes with exact / fp6 / fp8 cross terms, one and two f16 products.
usage: tools/study_product_error.py"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from oracle import lightglue_oracle as O

sys.argv = [sys.argv[0]]
src = (ROOT / "tools" / "study_fp8_cross.py").read_text().split("class Ctx8")[0]   # the MX rounding emulators only
ns = {"__file__": str(ROOT / "tools" / "study_fp8_cross.py")}
exec(src, ns)
q6, q8 = ns["q_e2m3_mx"], ns["q_e4m3_mx"]

rng = np.random.default_rng(0)
R, K, N = 256, 512, 512
x = (rng.standard_normal((R, K)) * rng.choice([1, 4], size=(R, K))).astype(np.float32)   # residual-stream-like spread
w = (rng.standard_normal((K, N)) * 0.06).astype(np.float32)
f = lambda a: np.asarray(a, np.float64)
ref = f(x) @ f(w)
mag = np.abs(f(x)) @ np.abs(f(w))


def rep(name, y):
    e = np.abs(y - ref) / mag
    print(f"{name:44s} max {e.max():.3e}   rms {np.sqrt(np.mean(e * e)):.3e}")


xh, wh = O.round_bf16(x), O.round_bf16(w)
xl, wl = O.round_bf16(x - xh), O.round_bf16(w - wh)
rep("split-bf16, 3 products (round 1-2 default)", f(xh) @ f(wh) + f(xh) @ f(wl) + f(xl) @ f(wh))
xh, wh = O.round_fp16(x), O.round_fp16(w)
xl, wl = f(x) - f(xh), f(w) - f(wh)
rep("f16 hi planes + exact cross terms", f(xh) @ f(wh) + f(xh) @ wl + xl @ f(wh))
rep("f16 hi planes + fp6 e2m3 cross terms", f(xh) @ f(wh) + q6(f(xh), -1) @ q6(wl, -2) + q6(xl, -1) @ q6(f(wh), -2))
rep("f16 hi planes + fp8 e4m3 cross terms", f(xh) @ f(wh) + q8(f(xh), -1) @ q8(wl, -2) + q8(xl, -1) @ q8(f(wh), -2))
rep("f16 x f16, one product", f(xh) @ f(wh))
rep("f16 activations x split-f16 weights (q/k/v)", f(xh) @ (f(wh) + f(O.round_fp16(wl.astype(np.float32)))))
