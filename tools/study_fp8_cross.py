#!/usr/bin/env python3
"""CPU study (oracle + operand-rounding emulation, no GPU): could the two CROSS terms of a split-bf16 product run on the fp8 matrix
path?   x w = hi(x) hi(w) + hi(x) lo(w) + lo(x) hi(w)  with hi / lo bf16 is what the HIP path issues for the FFN chain (3 bf16
MFMAs).  gfx950's v_mfma_scale_f32_16x16x128_f8f6f4 runs fp8 at twice the bf16 rate with a shared power-of-two scale per 32-element
block (MX): the two cross terms as ONE fp8 instruction each would make the product cost 2 bf16-equivalents instead of 3 — if
rounding hi(x), hi(w) (and the lo planes) to e4m3 inside the cross terms keeps the match scores within the 1e-3 bar.
Emulation: per contraction class, product = bf16hi(a) bf16hi(b) + q8(hi a) q8(lo b) + q8(lo a) q8(hi b), q8 = e4m3 with MX block
scaling along K.   usage: tools/study_fp8_cross.py [N] [seeds]"""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from oracle import lightglue_oracle as O
from lightglue_amd import synthetic as synth


def q_e4m3_mx(x, axis):
    """Round to fp8 e4m3 with one power-of-two scale per 32-element block along `axis` (MX): block max -> (224, 448]."""
    x = np.moveaxis(np.asarray(x, np.float64), axis, -1)
    k = x.shape[-1]; pad = (-k) % 32
    xp = np.pad(x, [(0, 0)] * (x.ndim - 1) + [(0, pad)]).reshape(*x.shape[:-1], -1, 32)
    amax = np.abs(xp).max(-1, keepdims=True)
    scale = 2.0 ** np.ceil(np.log2(np.where(amax > 0, amax, 1.0) / 448.0))   # smallest power of two with amax / scale <= 448 (no clipping)
    v = xp / scale
    mag = np.abs(v)
    ex = np.floor(np.log2(np.where(mag > 0, mag, 1.0)))
    ex = np.maximum(ex, -6)                             # subnormals: fixed step 2^-9
    step = 2.0 ** (ex - 3)
    q = np.clip(np.round(v / step) * step, -448, 448)
    out = (q * scale).reshape(*x.shape[:-1], -1)[..., :k]
    return np.moveaxis(out, -1, axis)


def q_e2m3_mx(x, axis):
    """Round to fp6 e2m3 (max 7.5, subnormal step 0.125) with one power-of-two scale per 32-element block (MX)."""
    x = np.moveaxis(np.asarray(x, np.float64), axis, -1)
    k = x.shape[-1]; pad = (-k) % 32
    xp = np.pad(x, [(0, 0)] * (x.ndim - 1) + [(0, pad)]).reshape(*x.shape[:-1], -1, 32)
    amax = np.abs(xp).max(-1, keepdims=True)
    scale = 2.0 ** np.ceil(np.log2(np.where(amax > 0, amax, 1.0) / 7.5))     # smallest power of two with amax / scale <= 7.5
    v = xp / scale
    mag = np.abs(v)
    ex = np.maximum(np.floor(np.log2(np.where(mag > 0, mag, 1.0))), 0)     # below 1.0: subnormal, step 2^-3
    step = 2.0 ** (ex - 3)
    q = np.clip(np.round(v / step) * step, -7.5, 7.5)
    out = (q * scale).reshape(*x.shape[:-1], -1)[..., :k]
    return np.moveaxis(out, -1, axis)


class Ctx8(O._Ctx):
    fp8_classes = ()
    fmt = "fp8"       # cross-term format: "fp8" (e4m3) | "fp6" (e2m3, twice the fp8 rate on gfx950)
    hi = "bf16"       # precision of the hi planes (the main product runs on the 16-bit matrix path): "bf16" | "fp16"

    def mm(self, a, b, where="lin"):
        if where in self.fp8_classes:
            rnd = O.round_bf16 if self.hi == "bf16" else O.round_fp16
            a64, b64 = np.asarray(a, np.float64), np.asarray(b, np.float64)
            ha, hb = rnd(a.astype(np.float32)).astype(np.float64), rnd(b.astype(np.float32)).astype(np.float64)
            la, lb = a64 - ha, b64 - hb                      # (rounded to fp8 below)
            q = q_e2m3_mx if self.fmt == "fp6" else q_e4m3_mx
            y = ha @ hb + q(ha, -1) @ q(lb, -2) + q(la, -1) @ q(hb, -2)
            return y.astype(self.dtype)
        return super().mm(a, b, where)

    # ---- "folded" mode: ffn.0 as the HIP tail kernel computes it — out_proj folded into the ctx half of ONE 512-long
    # contraction  h = x W1x^T + ctx (W1m Wo)^T + (b1 + W1m bo)  — so that the two halves can get different operand formats
    folded = False

    def linear(self, x, w, b=None, where="lin"):
        if self.folded and where == "lin_out":
            y = (np.asarray(x, np.float64) @ np.asarray(w, np.float64).T + np.asarray(b, np.float64)).astype(self.dtype)
            self._stash = getattr(self, "_stash", []) + [(y, x, w, b)]
            return y
        if self.folded and where == "lin_ffn0":
            d = x.shape[-1] // 2
            msg = x[..., d:]
            hit = [i for i, e in enumerate(self._stash) if e[0].shape == msg.shape and np.array_equal(e[0], msg)]
            _, ctx_in, wo, bo = self._stash.pop(hit[0])
            w1x, w1m = np.asarray(w[:, :d], np.float64), np.asarray(w[:, d:], np.float64)
            wf = (w1m @ np.asarray(wo, np.float64)).astype(np.float32)            # the host folds in double and stores fp32 before splitting
            bf = np.asarray(b, np.float64) + w1m @ np.asarray(bo, np.float64)
            y = self.mm(x[..., :d], w1x.astype(np.float32).T, "fold_x").astype(np.float64) + self.mm(ctx_in, wf.T, "fold_ctx").astype(np.float64) + bf
            return y.astype(self.dtype)
        return super().linear(x, w, b, where)


def run(sd, conf, data, classes, hi="bf16"):
    Ctx8.fp8_classes = classes; Ctx8.hi = hi.split("+")[0]; Ctx8.fmt = "fp6" if "+fp6" in hi else "fp8"; Ctx8.folded = "+folded" in hi
    orig = O._Ctx
    O._Ctx = Ctx8
    quant = O.DEFAULT_PRECISION_QUANT
    if "+f16base" in hi:   # every OTHER split product on f16 planes too (3 f16 MFMAs instead of 3 bf16 ones: same cost, ~100x smaller product error)
        quant = {**quant, "lin": "fp16x2", "final": "fp16x2"}
    try:
        return O.forward(sd, conf, data, quant=quant)
    finally:
        O._Ctx = orig


n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
conf = O.make_conf(depth_confidence=-1, width_confidence=-1)
ALL = ("lin_ffn0", "lin_ffn3", "lin_out")
rows = [("default precision (split-bf16 x3; q/k/v f16 x2)", (), "bf16"),
        ("bf16 hi planes: ffn.0 cross terms in fp8", ("lin_ffn0",), "bf16"), ("bf16 hi planes: ffn.0 + ffn.3 + out_proj", ALL, "bf16"),
        ("f16 hi planes: ffn.0 cross terms in fp8", ("lin_ffn0",), "fp16"), ("f16 hi planes: ffn.3 cross terms in fp8", ("lin_ffn3",), "fp16"),
        ("f16 hi planes: out_proj cross terms in fp8", ("lin_out",), "fp16"), ("f16 hi planes: ffn.0 + ffn.3 + out_proj", ALL, "fp16"),
        ("f16 hi planes: the same + final_proj / similarity", ALL + ("final",), "fp16"),
        ("f16 hi planes, cross terms in fp6 e2m3: ffn.0 + ffn.3 + out_proj", ALL, "fp16+fp6"),
        ("f16 hi planes, cross terms in fp6 e2m3: out_proj", ("lin_out",), "fp16+fp6"),
        ("folded ffn.0 (as the HIP kernel), default operands", (), "fp16+fp6+folded"),
        ("folded: ctx half in f16 + fp6 cross terms", ("fold_ctx",), "fp16+fp6+folded"),
        ("folded: ctx half + ffn.3", ("fold_ctx", "lin_ffn3"), "fp16+fp6+folded"),
        ("folded: ctx half + x half", ("fold_ctx", "fold_x"), "fp16+fp6+folded"),
        ("folded: ctx half + x half + ffn.3", ("fold_ctx", "fold_x", "lin_ffn3"), "fp16+fp6+folded"),
        ("f16base: all split products on f16 planes (f16x3), no fp6", (), "fp16+fp6+folded+f16base"),
        ("f16base + ctx half fp6", ("fold_ctx",), "fp16+fp6+folded+f16base"),
        ("f16base + ctx half + ffn.3 fp6", ("fold_ctx", "lin_ffn3"), "fp16+fp6+folded+f16base"),
        ("f16base + ctx half + x half fp6", ("fold_ctx", "fold_x"), "fp16+fp6+folded+f16base"),
        ("f16base + ctx half + x half + ffn.3 fp6", ("fold_ctx", "fold_x", "lin_ffn3"), "fp16+fp6+folded+f16base"),
        ("f16base + the same + final_proj / similarity fp6", ("fold_ctx", "fold_x", "lin_ffn3", "final"), "fp16+fp6+folded+f16base")]
import os
if os.environ.get("STUDY_ROWS"):   # e.g. STUDY_ROWS="default,f16 hi planes: ffn.0 + ffn.3" keeps the rows whose name contains one of the keys
    keys = os.environ["STUDY_ROWS"].split(",")
    rows = [r for r in rows if any(k in r[0] for k in keys)]
seed0 = int(os.environ.get("STUDY_SEED0", "0"))
res = {name: [] for name, _, _ in rows}
for seed in range(seed0, seed0 + seeds):
    sd = synth.make_state_dict(seed, recipe="A")
    data = synth.make_batch(100 + seed, 1, n, n)
    ref = O.forward(sd, conf, data)
    for name, classes, hi in rows:
        out = run(sd, conf, data, classes, hi)
        flips = int((out["matches0"] != ref["matches0"]).sum())
        d = np.abs(out["matching_scores0"] - ref["matching_scores0"]).ravel()
        res[name].append((flips, float(d.max()), float(np.sqrt(np.mean(d * d)))))
    print("seed", seed, {k: v[-1] for k, v in res.items()}, flush=True)
print(f"\nN = M = {n}, {seeds} seeds, vs the fp32 oracle: index flips / max |dscore|")
for name, _, _ in rows:
    print(f"  {name:70s} flips {sum(r[0] for r in res[name]):3d}   max |dscore| {max(r[1] for r in res[name]):.2e}   rms (mean over seeds) {np.mean([r[2] for r in res[name]]):.2e}")
