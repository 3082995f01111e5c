#!/usr/bin/env python3
"""Seeded random walk over the CONSTRUCTOR space of the matcher against the pinned oracle (oracle/, torch backend) on the GPU: input_dim 64 ... 256 (with / without
input_proj), add_scale_ori, n_layers 1 ... 9, filter_threshold, early stop and point pruning separately and together (recipe-C weights: mixed stop depths), pruning
thresholds below / above the keypoint counts, image_size present or absent (bounding-box normalisation), batch 1 ... 3, keypoint counts 1 ... 700 per image.
The fixtures and the seed sweeps hold the data axis at the default constructor; this holds the constructor axis.  Bar = the product's: scores within 1e-3, indices equal
up to flips the oracle's own decision boundaries explain (tests/conftest.py), stop layers and prune counters equal.

usage: fuzz_configs.py [--cases 40] [--seed 0] [--checkpoint trained.pth]      (exit code 1 on the first failing case, which is printed with its full recipe)"""
import argparse
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import gpu_util
from conftest import SCORE_TOL, assert_parity_with_explained_flips
from lightglue_amd import synthetic as synth
from oracle import lightglue_oracle as O


def draw(rng):
    adaptive = int(rng.integers(0, 4))          # 0 none, 1 early stop only, 2 pruning only, 3 both
    n_layers = int(rng.choice([1, 2, 3, 5, 9]))
    case = dict(dim=int(rng.choice([64, 128, 192, 256])), sift=bool(rng.integers(0, 2)), n_layers=n_layers, adaptive=adaptive,
                filter_threshold=float(rng.choice([0.0, 0.1, 0.3])), B=int(rng.integers(1, 4)), n=int(rng.integers(1, 701)), m=int(rng.integers(1, 701)),
                drop_size=bool(rng.integers(0, 3) == 0), wseed=int(rng.integers(0, 1000)), dseed=int(rng.integers(0, 100000)),
                prune_th=int(rng.choice([-1, 64, 300, 2000])), depth_confidence=float(rng.choice([0.95, 0.9, 0.99])), width_confidence=float(rng.choice([0.99, 0.95])))
    return case


CHECKPOINT = None      # --checkpoint: a trained state dict (256-d, 9 layers) instead of the seeded recipes — the walk then covers shapes, thresholds and adaptivity modes on ITS heads


def run_case(case):
    recipe = "C" if case["adaptive"] else "A"
    if CHECKPOINT is not None:
        case.update(dim=256, sift=False, n_layers=9)
    kw = dict(n_layers=case["n_layers"], filter_threshold=case["filter_threshold"],
              depth_confidence=case["depth_confidence"] if case["adaptive"] in (1, 3) else -1,
              width_confidence=case["width_confidence"] if case["adaptive"] in (2, 3) else -1)
    if case["dim"] != 256:
        kw["input_dim"] = case["dim"]
    if case["sift"]:
        kw["add_scale_ori"] = True
    sd = CHECKPOINT if CHECKPOINT is not None else synth.make_state_dict(case["wseed"], input_dim=case["dim"], add_scale_ori=case["sift"], n_layers=case["n_layers"], recipe=recipe)
    data = synth.make_batch(case["dseed"], case["B"], case["n"], case["m"], case["dim"], add_scale_ori=case["sift"])
    if case["drop_size"]:
        for img in ("image0", "image1"):
            data[img].pop("image_size")
    okw = dict(kw, pruning_min_kpts=case["prune_th"])
    ref = O.forward(sd, O.make_conf(**okw), data, backend="torch")
    model = gpu_util.make_model(sd, "f16x3", pruning_min_kpts=case["prune_th"], **kw)
    out = model(gpu_util.to_torch(data))
    gold = {k: np.asarray(ref[k]) for k in ("matches0", "matches1", "matching_scores0", "matching_scores1")}
    pcase = {"conf": kw, "prune_th": case["prune_th"], "n": case["n"], "m": case["m"], "B": case["B"], "dim": case["dim"]}
    flips = assert_parity_with_explained_flips(out, gold, pcase, sd, data, score_tol=SCORE_TOL)
    stop = out["stop"] if torch.is_tensor(out["stop"]) else torch.tensor([out["stop"]])
    np.testing.assert_array_equal(stop.cpu().numpy().reshape(-1), np.asarray(ref["stop"]).reshape(-1))
    for k in ("prune0", "prune1"):
        np.testing.assert_array_equal(out[k].cpu().numpy(), np.asarray(ref[k]))
        assert str(out[k].dtype).endswith("int64" if kw["width_confidence"] > 0 else "float32"), (k, out[k].dtype)
    err = max(float(np.abs(out[f"matching_scores{s}"].cpu().numpy() - gold[f"matching_scores{s}"])[out[f"matches{s}"].cpu().numpy() == gold[f"matches{s}"]].max(initial=0.0)) for s in (0, 1))
    return flips, err, int((gold["matches0"] >= 0).sum()), np.asarray(ref["stop"]).reshape(-1).tolist()


def run(cases=40, seed=0, verbose=True):
    rng = np.random.default_rng(seed)
    torch.set_num_threads(8)
    worst, total_flips = 0.0, 0
    for i in range(cases):
        case = draw(rng)
        try:
            flips, err, nm, stop = run_case(case)
        except Exception:
            print(f"FAILED case {i}: {case}", flush=True)
            raise
        worst = max(worst, err); total_flips += sum(flips)
        if verbose:
            print(f"case {i:3d} dim {case['dim']:3d}{' +so' if case['sift'] else '    '} L={case['n_layers']} adaptive={case['adaptive']} th={case['filter_threshold']} prune_th={case['prune_th']:5d} "
                  f"B={case['B']} {case['n']:3d}x{case['m']:3d}{' bbox' if case['drop_size'] else '     '}: matches {nm:4d} stop {stop} explained flips {flips} max|dscore| {err:.1e}", flush=True)
    print(f"{cases} cases: all inside the bar; explained flips {total_flips}, max |dscore| {worst:.2e}", flush=True)
    return worst


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--checkpoint", default=None, help="state-dict file (module-tree names; tools/train_synthetic_checkpoint.py): its weights in every case")
    a = ap.parse_args()
    if a.checkpoint:
        CHECKPOINT = {k: v.float().numpy() for k, v in torch.load(a.checkpoint, map_location="cpu").items() if torch.is_tensor(v)}
    run(a.cases, a.seed)
