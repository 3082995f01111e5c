#!/usr/bin/env python3
"""Calibrate the per-block scales of synthetic recipe E (trained-model statistics + confident matches) and print its statistics.

Walks the 9 layers of ONE seeded pair with the oracle's block functions (CPU fp32).  For every attention call the q / k scale is set
so that the median per-row logit spread (max - min, base e) is `--spread` (quadratic in the scale: one measurement per block); for
every block the ffn.3 scale is set so that the residual rms follows 3.5 -> 27 geometrically over the layers (the block's update is
linear in that scale: a quadratic equation).  Prints the four tables for lightglue_amd/synthetic.py and, with --check, the statistics
of the recipe as it is defined there (also for recipe D, for comparison).

    python tools/calibrate_recipe.py            # calibrate on weights seed 0, pair seed 601, N = M = 512
    python tools/calibrate_recipe.py --check E  # statistics of the committed tables
"""
import argparse
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from lightglue_amd import synthetic as synth  # noqa: E402
from oracle import lightglue_oracle as O  # noqa: E402


def spread(q, k):
    lg = np.einsum("hnd,hmd->hnm", np.asarray(q, np.float64), np.asarray(k, np.float64)) / 8.0
    return float(np.median(lg.max(-1) - lg.min(-1)))


def rms(*xs):
    return float(np.sqrt(np.mean(np.concatenate([np.asarray(x, np.float64).ravel() for x in xs]) ** 2)))


def solve_f(xs, ds, target):
    """f such that rms(x + f d) = target over the concatenation of the (x, d) pairs."""
    a = sum(float((np.asarray(d, np.float64) ** 2).sum()) for d in ds)
    b = sum(float((np.asarray(x, np.float64) * np.asarray(d, np.float64)).sum()) for x, d in zip(xs, ds))
    c = sum(float((np.asarray(x, np.float64) ** 2).sum()) for x in xs) - target ** 2 * sum(x.size for x in xs)
    return (-b + np.sqrt(max(b * b - a * c, 0.0))) / a


def inputs(sd, data):
    ctx = O.make_ctx(np.float32, None, "torch")
    p = {k: np.asarray(v, np.float32) for k, v in sd.items()}
    k0 = O.normalize_keypoints(data["image0"]["keypoints"][0], data["image0"]["image_size"][0])
    k1 = O.normalize_keypoints(data["image1"]["keypoints"][0], data["image1"]["image_size"][0])
    cos0, sin0 = O.posenc(ctx, p["posenc.Wr.weight"], k0)
    cos1, sin1 = O.posenc(ctx, p["posenc.Wr.weight"], k1)
    return ctx, p, data["image0"]["descriptors"][0], data["image1"]["descriptors"][0], (cos0, sin0), (cos1, sin1)


def calibrate(wseed, dseed, n, target_spread, rms0=3.5, rms1=27.0):
    L = 9
    qs, qc, fs, fc = [1.0] * L, [1.0] * L, [1.0] * L, [1.0] * L
    data = synth.make_batch(dseed, 1, n, n, **synth.RECIPE_E_DATA)
    x0 = x1 = None
    for i in range(L):
        t_self = rms0 * (rms1 / rms0) ** ((i - 0.5) / (L - 1)) if i else rms0 * 0.8
        t_cross = rms0 * (rms1 / rms0) ** (i / (L - 1))
        for which in ("self_qk", "self_f3", "cross_qk", "cross_f3"):
            sd = synth.make_state_dict(wseed, recipe=None)
            synth._apply_recipe_e(sd, wseed, L, adaptive=False, scales=(qs, qc, fs, fc))
            ctx, p, d0, d1, r0, r1 = inputs(sd, data)
            if i == 0:
                x0, x1 = d0, d1
            tr = {}
            if which.startswith("self"):
                y0 = O.self_block(ctx, p, i, x0, *r0, 4, tr, "a_")
                y1 = O.self_block(ctx, p, i, x1, *r1, 4, tr, "b_")
                if which == "self_qk":
                    s = 0.5 * (spread(tr["a_q"], tr["a_k"]) + spread(tr["b_q"], tr["b_k"]))
                    qs[i] *= float(np.sqrt(target_spread / s))
                else:
                    fs[i] *= float(solve_f([x0, x1], [y0 - x0, y1 - x1], t_self))
            else:
                xs0 = O.self_block(ctx, p, i, x0, *r0, 4)
                xs1 = O.self_block(ctx, p, i, x1, *r1, 4)
                y0, y1 = O.cross_block(ctx, p, i, xs0, xs1, 4, tr, "c_")
                if which == "cross_qk":
                    s = 0.5 * (spread(tr["c_qk0"], tr["c_qk1"]) + spread(tr["c_qk1"], tr["c_qk0"]))
                    qc[i] *= float(np.sqrt(target_spread / s))
                else:
                    fc[i] *= float(solve_f([xs0, xs1], [y0 - xs0, y1 - xs1], t_cross))
        # advance with the calibrated layer
        sd = synth.make_state_dict(wseed, recipe=None)
        synth._apply_recipe_e(sd, wseed, L, adaptive=False, scales=(qs, qc, fs, fc))
        ctx, p, d0, d1, r0, r1 = inputs(sd, data)
        xs0, xs1 = O.self_block(ctx, p, i, x0, *r0, 4), O.self_block(ctx, p, i, x1, *r1, 4)
        x0, x1 = O.cross_block(ctx, p, i, xs0, xs1, 4)
        print(f"layer {i}: qk self {qs[i]:.4f} cross {qc[i]:.4f}  f3 self {fs[i]:.3f} cross {fc[i]:.3f}  rms {rms(x0, x1):.2f}", flush=True)
    fmt = lambda v: "(" + ", ".join(f"{x:.4g}" for x in v) + ")"
    print(f"_E_QK_SELF = {fmt(qs)}\n_E_QK_CROSS = {fmt(qc)}\n_E_F3_SELF = {fmt(fs)}\n_E_F3_CROSS = {fmt(fc)}")
    return qs, qc, fs, fc


def check(recipe, wseed, dseed, n, m=None):
    m = m or n
    sd = synth.make_state_dict(wseed, recipe=recipe)
    kw = synth.RECIPE_E_DATA if recipe.startswith("E") else synth.RECIPE_D_DATA
    data = synth.make_batch(dseed, 1, n, m, **kw)
    conf = O.make_conf(depth_confidence=-1, width_confidence=-1)
    tr = {"_full_layers": tuple(range(9))}
    r = O.forward_pair(sd, conf, data["image0"]["keypoints"][0], data["image1"]["keypoints"][0], data["image0"]["descriptors"][0], data["image1"]["descriptors"][0],
                       data["image0"]["image_size"][0], data["image1"]["image_size"][0], trace=tr, backend="torch")
    for i in range(9):
        x = np.asarray(tr[f"desc0_l{i}"], np.float64)
        print(f"  layer {i}: logit spread self {spread(tr[f'l{i}_self0_q'], tr[f'l{i}_self0_k']):5.1f} cross {spread(tr[f'l{i}_cross_qk0'], tr[f'l{i}_cross_qk1']):5.1f}"
              f"   residual rms {rms(x):6.2f}  around the per-image mean {rms(x - x.mean(0, keepdims=True)):6.2f}")
    m0, s0 = np.asarray(r["matches0"]), np.asarray(r["matching_scores0"])
    perm = data_perm(dseed, n, m, kw)
    truth = np.full(n, -1); truth[perm[perm >= 0]] = np.nonzero(perm >= 0)[0]
    print(f"  recipe {recipe} weights seed {wseed} pair seed {dseed} {n}x{m}: matched {int((m0 > -1).sum())} / {n}, scores > 0.5: {int((s0 > 0.5).sum())}, "
          f"matched to the planted copy: {int(((m0 == truth) & (m0 > -1)).sum())}")


def data_perm(dseed, n, m, kw):
    return synth.make_pair(dseed, n, m, **kw)["perm"]


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", default=None, help="recipe name: print the statistics of the committed recipe instead of calibrating")
    ap.add_argument("--wseed", type=int, default=0)
    ap.add_argument("--dseed", type=int, default=601)
    ap.add_argument("--n", type=int, default=512)
    ap.add_argument("--m", type=int, default=None)
    ap.add_argument("--spread", type=float, default=25.0)
    a = ap.parse_args()
    import torch
    torch.set_num_threads(8)
    if a.check:
        check(a.check, a.wseed, a.dseed, a.n, a.m)
    else:
        calibrate(a.wseed, a.dseed, a.n, a.spread)
