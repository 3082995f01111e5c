#!/usr/bin/env python3
"""CPU emulation study (round 6): the two CROSS terms of a split-f16 product on the int8 matrix pipe.
    x w  ~  xh wh  (f16 x f16, fp32 accumulate: one 16x16x32 f16 MFMA per 32 k)
          + [ Q(xh) Q'(wl) + Q'(xl) Q(wh) ]  (int8 x int8, int32 accumulate: two 16x16x64 i8 MFMAs per 64 k = ONE f16-MFMA time per 32 k)
with per-row fixed-point scales: Q(v) = round(v / s 127) for s = max |v| over the contraction axis, Q'(lo) on the scale s 2^-11 (|lo| <= 2^-11 |hi| elementwise).
Matrix time 2 instead of 3 per product, and int8 MACs cost a fraction of an f16 FMA's energy — the one lever left on a power-limited chip (LAB_NOTES round 6) —
IF ~18 operand bits (instead of 22) still hold the 1e-3 bar.  Prints max / rms |dscore| and index flips against the reference's fixtures.
usage: tools/study_int8_cross_terms.py <fixture> [...]"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests")); sys.path.insert(0, str(ROOT / "tools"))
from conftest import load_golden, oracle_conf_for  # noqa: E402
import make_golden  # noqa: E402
from oracle import lightglue_oracle as O  # noqa: E402

f16 = lambda a: a.astype(np.float16).astype(np.float32)


def q8(v, axis, shift=0.0):
    """fixed point, 8 bits incl. sign, one scale per vector along `axis` (2^-shift of the hi plane's scale for a lo plane) -> dequantised value"""
    return v  # placeholder, replaced below


def mm_f16_i8(a, b, lo_bits=11):
    a = np.asarray(a, np.float32); b = np.asarray(b, np.float32)
    ah, bh = f16(a), f16(b)
    al, bl = a - ah, b - bh
    sa = np.maximum(np.abs(ah).max(-1, keepdims=True), 1e-30)         # per row of a (over k)
    sb = np.maximum(np.abs(bh).max(-2, keepdims=True), 1e-30)         # per column of b (over k)
    qa = np.clip(np.rint(ah / sa * 127.0), -127, 127) * (sa / 127.0)
    qb = np.clip(np.rint(bh / sb * 127.0), -127, 127) * (sb / 127.0)
    sla, slb = sa * 2.0 ** -lo_bits, sb * 2.0 ** -lo_bits
    qal = np.clip(np.rint(al / sla * 127.0), -127, 127) * (sla / 127.0)
    qbl = np.clip(np.rint(bl / slb * 127.0), -127, 127) * (slb / 127.0)
    return np.matmul(ah, bh) + np.matmul(qa, qbl) + np.matmul(qal, qb)


class Ctx(O._Ctx):
    def __init__(self, where_i8):
        super().__init__(np.float32, O.DEFAULT_PRECISION_QUANT)
        self.where_i8 = where_i8

    def mm(self, a, b, where="lin"):
        key = "lin" if where.startswith("lin") else where
        if key in self.where_i8 or where in self.where_i8:
            return mm_f16_i8(np.ascontiguousarray(a), np.ascontiguousarray(b))
        return super().mm(a, b, where)


modes = {"default f16x3": None, "linear layers": {"lin"}, "q k^T": {"attn_qk"}, "P V": {"attn_pv"}, "linear + q k^T + P V": {"lin", "attn_qk", "attn_pv"},
         "everything (incl. final proj + similarity)": {"lin", "attn_qk", "attn_pv", "final"}}
print("| fixture | " + " | ".join(f"{m}: flips / max / rms" for m in modes) + " |")
print("|---|" + "---|" * len(modes))
orig_make_ctx = O.make_ctx
for name in sys.argv[1:]:
    meta, gold = load_golden(name)
    case = meta["case"]
    sd, data = make_golden.case_inputs(case)
    cells = []
    for m, where in modes.items():
        O.make_ctx = (lambda dtype, quant, backend="numpy", _w=where: Ctx(_w)) if where is not None else orig_make_ctx
        try:
            out = O.forward(sd, oracle_conf_for(case), data, quant=O.DEFAULT_PRECISION_QUANT)
        finally:
            O.make_ctx = orig_make_ctx
        d = np.abs(np.asarray(out["matching_scores0"]) - gold["matching_scores0"]).ravel()
        flips = int((np.asarray(out["matches0"]) != gold["matches0"]).sum())
        same = (np.asarray(out["matches0"]) == gold["matches0"]).ravel()
        cells.append(f"{flips} / {d[same].max():.2e} / {np.sqrt(np.mean(d[same] ** 2)):.2e}")
    print(f"| {name} | " + " | ".join(cells) + " |", flush=True)
