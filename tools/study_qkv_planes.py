#!/usr/bin/env python3
"""CPU study (oracle + operand-rounding emulation, no GPU): how few weight bits do the q/k/v projections need?

The fused projection (lg_proj_body.h, PREC_QKV_F16W2) multiplies ONE f16 activation plane by split-f16 weights
(w_hi x + w_lo x: two MFMAs and two 1 KB weight fragments per product) and its MFMA loop is bound by the L2 -> VGPR weight
stream, not by the matrix pipe (62 B/clk/CU wanted, ~50 delivered).  Every byte of the lo plane that can go
shortens it.  Candidates, all with f16 activations:
  v1      value columns (Wqkv rows 3i+2, to_v) with w_hi only (one product), q/k unchanged
  lo8     the lo plane truncated to its top byte (f16 -> e5m2: sign, 5 exponent bits, 2 mantissa bits; expanded in registers
          with one v_perm_b32 per two elements) for all of q/k/v
  lo8+v1  both
usage: tools/study_qkv_planes.py [N] [seeds]        env STUDY_SEED0"""
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from oracle import lightglue_oracle as O
from lightglue_amd import synthetic as synth


def trunc_e5m2(x16_as_f32):
    """Keep the top byte of the f16 encoding (truncation toward zero, what dropping the low byte does)."""
    h = np.asarray(x16_as_f32, np.float32).astype(np.float16)
    bits = h.view(np.uint16) & np.uint16(0xFF00)
    return bits.view(np.float16).astype(np.float32)


def round_e5m2(x16_as_f32):
    """Round-to-nearest-even to the top byte (the host can do this when it packs)."""
    h = np.asarray(x16_as_f32, np.float32).astype(np.float16)
    b = h.view(np.uint16).astype(np.uint32)
    lsb = (b >> 8) & 1
    b = (b + 0x7F + lsb) & 0xFF00
    return b.astype(np.uint16).view(np.float16).astype(np.float32)


class CtxQKV(O._Ctx):
    mode = ""   # "", "v1", "lo8", "lo8+v1", "lo8r", "lo8r+v1"

    def _weights(self, w, is_v):
        """w [out, in] fp32 -> the effective weight the kernel multiplies with."""
        hi = O.round_fp16(w)
        lo = O.round_fp16(w - hi)
        if "lo8r" in self.mode:
            lo = round_e5m2(lo)
        elif "lo8" in self.mode:
            lo = trunc_e5m2(lo)
        eff = hi + lo
        if "v1" in self.mode:
            eff = np.where(is_v[:, None], hi, eff)
        return eff.astype(np.float32)

    def linear(self, x, w, b=None, where="lin"):
        if where == "lin_qkv" and self.mode:
            out = w.shape[0]
            if out == 768:                       # Wqkv: output column c = h*192 + d*3 + {q, k, v}  (ref :166)
                is_v = (np.arange(out) % 3) == 2
            elif self._next_is_v(w):
                is_v = np.ones(out, bool)
            else:
                is_v = np.zeros(out, bool)
            y = np.matmul(O.round_fp16(x), self._weights(w, is_v).T)
            return y + b if b is not None else y
        return super().linear(x, w, b, where)

    _v_ids = set()

    def _next_is_v(self, w):
        return id(w) in self._v_ids


def run(sd, conf, data, mode):
    CtxQKV.mode = mode
    CtxQKV._v_ids = {id(v) for k, v in sd.items() if k.endswith("to_v.weight")}
    orig = O._Ctx
    O._Ctx = CtxQKV
    try:
        return O.forward(sd, conf, data, quant=O.DEFAULT_PRECISION_QUANT)
    finally:
        O._Ctx = orig


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    seed0 = int(os.environ.get("STUDY_SEED0", "0"))
    conf = O.make_conf(depth_confidence=-1, width_confidence=-1)
    modes = ["", "v1", "lo8", "lo8r", "lo8+v1", "lo8r+v1"]
    res = {m: [] for m in modes}
    for seed in range(seed0, seed0 + seeds):
        sd = synth.make_state_dict(seed, recipe="A")
        data = synth.make_batch(100 + seed, 1, n, n)
        ref = O.forward(sd, conf, data)
        for m in modes:
            out = run(sd, conf, data, m)
            d = np.abs(out["matching_scores0"] - ref["matching_scores0"]).ravel()
            flips = int((out["matches0"] != ref["matches0"]).sum())
            res[m].append((flips, float(d.max()), float(np.sqrt(np.mean(d * d)))))
        print("seed", seed, {m or "default": tuple(round(v, 7) for v in res[m][-1]) for m in modes}, flush=True)
    print(f"\nN = M = {n}, {seeds} seeds, vs the fp32 oracle: index flips / max |dscore| / rms dscore (mean over seeds)")
    for m in modes:
        r = res[m]
        print(f"  {m or 'default (f16 x split-f16, 2 products)':40s} flips {sum(x[0] for x in r):3d}   max {max(x[1] for x in r):.2e}   rms {np.mean([x[2] for x in r]):.2e}")


if __name__ == "__main__":
    main()
