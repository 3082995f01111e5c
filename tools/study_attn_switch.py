#!/usr/bin/env python3
"""Study for the data-dependent attention precision switch: for families of seeded weights that interpolate between
diffuse (recipe A) and sharp (recipes D / E) attention, print per forward
  * the Cauchy-Schwarz logit bound the device can compute cheaply, L = max_i |q_i| max_j |k_j| / sqrt(64), maximum over all 18 attention
    calls and heads (and the true max |logit| and the median per-row logit spread next to it), and
  * the final-score error of the HYBRID arithmetic (everything split-f16 except the attention contractions, whose q / k / v / P operands
    are ONE f16 plane) against the fp32 evaluation of the same weights, next to the default arithmetic's error.
CPU emulation only (oracle operand rounding); no GPU.   usage: tools/study_attn_switch.py [n=512]"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from lightglue_amd import synthetic as synth  # noqa: E402
from oracle import lightglue_oracle as O  # noqa: E402

HYBRID = {"lin": "fp16x2", "attn": "fp16", "final": "fp16x2"}


def scale_qk(sd, f, n_layers=9):
    sd = {k: v.copy() for k, v in sd.items()}
    for i in range(n_layers):
        s, c = f"transformers.{i}.self_attn.", f"transformers.{i}.cross_attn."
        w = sd[s + "Wqkv.weight"].reshape(4, 64, 3, -1); b = sd[s + "Wqkv.bias"].reshape(4, 64, 3)
        w[:, :, 0:2] *= np.float32(f); b[:, :, 0:2] *= np.float32(f)
        sd[c + "to_qk.weight"] *= np.float32(f); sd[c + "to_qk.bias"] *= np.float32(f)
    return sd


def stats(sd, data, conf):
    tr = {"_full_layers": tuple(range(9))}
    i0, i1 = data["image0"], data["image1"]
    O.forward_pair(sd, conf, i0["keypoints"][0], i1["keypoints"][0], i0["descriptors"][0], i1["descriptors"][0], i0["image_size"][0], i1["image_size"][0], trace=tr, backend="torch")
    Lb, Lt, sp = 0.0, 0.0, []
    def one(q, k, s):
        nonlocal Lb, Lt
        q = np.asarray(q, np.float64); k = np.asarray(k, np.float64)
        Lb = max(Lb, float((np.linalg.norm(q, axis=-1).max(-1) * np.linalg.norm(k, axis=-1).max(-1)).max() * s))
        lg = np.einsum("hnd,hmd->hnm", q, k) * s
        Lt = max(Lt, float(np.abs(lg).max())); sp.append(float(np.median(lg.max(-1) - lg.min(-1))))
    for i in range(9):
        for t in ("self0", "self1"):
            one(tr[f"l{i}_{t}_q"], tr[f"l{i}_{t}_k"], 0.125)
        one(tr[f"l{i}_cross_qk0"], tr[f"l{i}_cross_qk1"], 0.125); one(tr[f"l{i}_cross_qk1"], tr[f"l{i}_cross_qk0"], 0.125)
    return Lb, Lt, max(sp)


def err(sd, data, conf, quant, ref):
    out = O.forward(sd, conf, data, quant=quant)
    d = np.abs(np.asarray(out["matching_scores0"], np.float64) - ref["matching_scores0"])
    return float(d.max()), int((np.asarray(out["matches0"]) != ref["matches0"]).sum())


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    conf = O.make_conf(depth_confidence=-1, width_confidence=-1)
    fams = [("A", None, 1, (1, 2, 4, 8, 16)), ("D", synth.RECIPE_D_DATA, 601, (1, 0.5, 0.3, 0.2, 0.1)), ("E", synth.RECIPE_E_DATA, 601, (1, 0.5, 0.3, 0.2, 0.1))]
    print("| recipe | q/k scale | bound L | true max abs logit | max median row spread | default: max dscore / flips | hybrid: max dscore / flips | scores > 0.5 |")
    print("|---|---|---|---|---|---|---|---|")
    for rec, dkw, dseed, fs in fams:
        base = synth.make_state_dict(0, recipe=rec)
        data = synth.make_batch(dseed, 1, n, n, **(dkw or {}))
        for f in fs:
            sd = scale_qk(base, f)
            ref = O.forward(sd, conf, data)
            ref = {k: np.asarray(v) for k, v in ref.items() if k in ("matching_scores0", "matches0")}
            Lb, Lt, sp = stats(sd, data, conf)
            e_def = err(sd, data, conf, O.DEFAULT_PRECISION_QUANT, ref)
            e_hyb = err(sd, data, conf, HYBRID, ref)
            print(f"| {rec} | {f} | {Lb:.1f} | {Lt:.1f} | {sp:.1f} | {e_def[0]:.1e} / {e_def[1]} | {e_hyb[0]:.1e} / {e_hyb[1]} | {(ref['matching_scores0'] > 0.5).mean():.2f} |", flush=True)


if __name__ == "__main__":
    main()
