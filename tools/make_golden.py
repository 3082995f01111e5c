#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the UNMODIFIED reference matcher in this container.

The reference module is loaded standalone from /root/reference/lightglue/lightglue.py (its package
__init__ needs torchvision/kornia/cv2 which are absent) on CPU in fp32.  Weights and
inputs are regenerated from seeds by oracle/synth.py (numpy only), so a fixture stores just the case
description, a digest of the weights, and the reference outputs.

    python tools/make_golden.py [name ...]  # rewrites every fixture (or the named ones)
The fixtures are committed; the GPU box never needs /root/reference.
"""
from __future__ import annotations

import importlib.util
import json
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from lightglue_amd import synthetic as synth  # noqa: E402

REF = Path("/root/reference/lightglue/lightglue.py")

# name -> case description.  `conf` are LightGlue kwargs; `prune_th` sets the reference's class dict
# LightGlue.pruning_keypoint_thresholds (all device keys) like benchmark.py:178-181 does.
CASES = {
    "nonadaptive_512": dict(recipe="A", wseed=0, dseed=1, B=1, n=512, m=512, dim=256, conf=dict(depth_confidence=-1, width_confidence=-1)),
    "nonadaptive_b3_256x192": dict(recipe="A", wseed=0, dseed=11, B=3, n=256, m=192, dim=256, conf=dict(depth_confidence=-1, width_confidence=-1)),
    "nonadaptive_bbox_300x200": dict(recipe="A", wseed=0, dseed=21, B=1, n=300, m=200, dim=256, no_size=True, conf=dict(depth_confidence=-1, width_confidence=-1)),
    "disk128_256x320": dict(recipe="A", wseed=3, dseed=31, B=1, n=256, m=320, dim=128, conf=dict(depth_confidence=-1, width_confidence=-1, input_dim=128)),
    "sift_scale_ori_200x180": dict(recipe="A", wseed=4, dseed=41, B=1, n=200, m=180, dim=128, conf=dict(depth_confidence=-1, width_confidence=-1, input_dim=128, add_scale_ori=True)),
    "mutualnn_filter0_512": dict(recipe=None, wseed=0, dseed=51, B=1, n=512, m=512, dim=256, conf=dict(depth_confidence=-1, width_confidence=-1, filter_threshold=0.0)),
    "adaptive_1024": dict(recipe="B", wseed=0, dseed=61, B=1, n=1024, m=1024, dim=256, prune_th=-1, conf=dict()),
    "adaptive_asym_800x300": dict(recipe="B", wseed=0, dseed=71, B=1, n=800, m=300, dim=256, prune_th=-1, conf=dict()),
    "adaptive_prune_th512_700x400": dict(recipe="B", wseed=0, dseed=81, B=1, n=700, m=400, dim=256, prune_th=512, conf=dict()),
    "stop_only_600": dict(recipe="B", wseed=0, dseed=91, B=1, n=600, m=600, dim=256, prune_th=-1, conf=dict(width_confidence=-1)),
    "prune_only_600": dict(recipe="B", wseed=0, dseed=101, B=1, n=600, m=600, dim=256, prune_th=-1, conf=dict(depth_confidence=-1)),
    "empty_0x50": dict(recipe="A", wseed=0, dseed=111, B=1, n=0, m=50, dim=256, conf=dict(depth_confidence=-1, width_confidence=-1)),
    # fewer keypoints than one 16-row tile on either side (tile masking extremes; filter threshold 0 so that the handful of scores is compared as matches)
    "tiny_1x17": dict(recipe="A", wseed=0, dseed=121, B=1, n=1, m=17, dim=256, conf=dict(depth_confidence=-1, width_confidence=-1, filter_threshold=0.0)),
    "tiny_5x3_b2": dict(recipe="A", wseed=0, dseed=131, B=2, n=5, m=3, dim=256, conf=dict(depth_confidence=-1, width_confidence=-1, filter_threshold=0.0)),
    "tiny_adaptive_9x30": dict(recipe="B", wseed=0, dseed=141, B=1, n=9, m=30, dim=256, prune_th=-1, conf=dict(filter_threshold=0.0)),
    # ---- BASELINE.json configs at their own shapes (round 2; VERDICT r01 "next" item 1)
    # cfg #2: the first 4 pairs of bench.py's own batch (weights seed 0 recipe A, pair seeds 1..4)
    "nonadaptive_1024_b4": dict(recipe="A", wseed=0, dseed=1, B=4, n=1024, m=1024, dim=256, conf=dict(depth_confidence=-1, width_confidence=-1)),
    # cfg #3: N=M=2048 adaptive, with the reference's flash threshold (ref :339-344) and with pruning from layer 0
    "adaptive_2048_th1536": dict(recipe="B", wseed=0, dseed=201, B=1, n=2048, m=2048, dim=256, prune_th=1536, conf=dict()),
    "adaptive_2048_thm1": dict(recipe="B", wseed=0, dseed=202, B=1, n=2048, m=2048, dim=256, prune_th=-1, conf=dict()),
    # cfg #4: DISK 128-d, N=M=4096
    "disk128_4096": dict(recipe="A", wseed=3, dseed=301, B=1, n=4096, m=4096, dim=128, conf=dict(depth_confidence=-1, width_confidence=-1, input_dim=128)),
    # cfg #5: ALIKED 128-d, 2048 x 512, pruning on with the 1536 threshold (only image0 can be pruned, ref :551/:559)
    "aliked128_2048x512_prune1536": dict(recipe="C", wseed=2, dseed=401, B=2, n=2048, m=512, dim=128, prune_th=1536, conf=dict(input_dim=128)),
    # cfg #3 with recipe C: the two pairs stop at DIFFERENT depths (3 and 9 layers) and are pruned at several layers
    "adaptive_2048_th1536_mixed_depth": dict(recipe="C", wseed=0, dseed=201, B=2, n=2048, m=2048, dim=256, prune_th=1536, conf=dict()),
    # a18: the compiled path's semantics without torch.compile (ref :512-520, :529): inputs inside the static range are
    # padded with ones and masked, pruning off, early stop on; above the range nothing changes
    "static_lengths_256_512_in_range": dict(recipe="B", wseed=0, dseed=501, B=1, n=300, m=420, dim=256, prune_th=-1, static_lengths=[256, 512], conf=dict()),
    "static_lengths_256_512_above_range": dict(recipe="B", wseed=0, dseed=502, B=1, n=600, m=500, dim=256, prune_th=256, static_lengths=[256, 512], conf=dict()),
    # ---- recipe D (round 3; VERDICT r02 "next" item 5): trained-model statistics — attention logit spread 25 per row, LayerNorm
    # gains in [0.5, 4], residual rms growing to ~27, descriptor norms in [0.5, 3]; "DC" adds the adaptive-path edits
    "trained_stats_512": dict(recipe="D", data="D", wseed=0, dseed=601, B=1, n=512, m=512, dim=256, conf=dict(depth_confidence=-1, width_confidence=-1)),
    "trained_stats_1024_filter0": dict(recipe="D", data="D", wseed=0, dseed=611, B=1, n=1024, m=1024, dim=256, conf=dict(depth_confidence=-1, width_confidence=-1, filter_threshold=0.0)),
    "trained_stats_2048": dict(recipe="D", data="D", wseed=0, dseed=621, B=1, n=2048, m=2048, dim=256, conf=dict(depth_confidence=-1, width_confidence=-1)),
    "trained_stats_adaptive_1024": dict(recipe="DC", data="D", wseed=0, dseed=631, B=2, n=1024, m=1024, dim=256, prune_th=-1, conf=dict()),
    "trained_stats_adaptive_2048_th1536": dict(recipe="DC", data="D", wseed=0, dseed=641, B=2, n=2048, m=2048, dim=256, prune_th=1536, conf=dict()),
    # other weight seeds (the per-layer scales were calibrated on seed 0; these land at logit spreads of 20 - 30) and an asymmetric pair
    "trained_stats_1024_w1": dict(recipe="D", data="D", wseed=1, dseed=651, B=1, n=1024, m=1024, dim=256, conf=dict(depth_confidence=-1, width_confidence=-1, filter_threshold=0.0)),
    # (default filter threshold: at threshold 0 this asymmetric pair has mutual-nearest-neighbour ties among scores of 1e-12, which
    # even the exact-fp32 GPU mode resolves differently from torch's summation order — 3 of 1500 — a property of the fixture, not a signal)
    "trained_stats_1500x700_w2": dict(recipe="D", data="D", wseed=2, dseed=661, B=1, n=1500, m=700, dim=256, conf=dict(depth_confidence=-1, width_confidence=-1)),
    # the bench workload's shape: pairs 0..7 of `bench.py --recipe D` (pair seeds 1..8) — a batch that fills the chip, so the 64-row
    # tail workgroups and the 128-row attention workgroups run (the single-pair cases above take the small-grid shapes)
    "trained_stats_1024_b8": dict(recipe="D", data="D", wseed=0, dseed=1, B=8, n=1024, m=1024, dim=256, conf=dict(depth_confidence=-1, width_confidence=-1)),
    # 128-d descriptors through input_proj (DISK / ALIKED) and the SIFT-style scale / orientation encoding under the same statistics
    "trained_stats_disk128_1024x800": dict(recipe="D", data="D", wseed=3, dseed=681, B=1, n=1024, m=800, dim=128, conf=dict(depth_confidence=-1, width_confidence=-1, input_dim=128)),
    "trained_stats_sift_700x600": dict(recipe="D", data="D", wseed=4, dseed=691, B=1, n=700, m=600, dim=128, conf=dict(depth_confidence=-1, width_confidence=-1, input_dim=128, add_scale_ori=True)),
    # cfg #4's shape under the same statistics: 4096 keys per softmax row
    "trained_stats_disk128_4096": dict(recipe="D", data="D", wseed=3, dseed=701, B=1, n=4096, m=4096, dim=128, conf=dict(depth_confidence=-1, width_confidence=-1, input_dim=128)),
    # ---- recipe E (round 4; VERDICT r03 "next" item 4): D's statistics AND a confident-match regime — the reference matches > 50 % of
    # the keypoints with scores > 0.5 (recipe D: <= 12 %), so the 1e-3 score bar is exercised on scores near 1, not near 0
    "trained_stats_confident_512": dict(recipe="E", data="E", wseed=0, dseed=601, B=1, n=512, m=512, dim=256, conf=dict(depth_confidence=-1, width_confidence=-1)),
    "trained_stats_confident_1024_b8": dict(recipe="E", data="E", wseed=0, dseed=1, B=8, n=1024, m=1024, dim=256, conf=dict(depth_confidence=-1, width_confidence=-1)),
    "trained_stats_confident_2048x512": dict(recipe="E", data="E", wseed=0, dseed=711, B=1, n=2048, m=512, dim=256, conf=dict(depth_confidence=-1, width_confidence=-1)),
    "trained_stats_confident_1024_w1_filter0": dict(recipe="E", data="E", wseed=1, dseed=721, B=1, n=1024, m=1024, dim=256, conf=dict(depth_confidence=-1, width_confidence=-1, filter_threshold=0.0)),
    "trained_stats_confident_adaptive_1024": dict(recipe="EC", data="E", wseed=0, dseed=731, B=2, n=1024, m=1024, dim=256, prune_th=-1, conf=dict()),
    # descriptor-scale sweep on the confident fixture: descriptors x0.1 / x10 / x30 (x1 = trained_stats_confident_512)
    "trained_stats_confident_512_x0p1": dict(recipe="E", data="E", desc_scale=0.1, wseed=0, dseed=601, B=1, n=512, m=512, dim=256, conf=dict(depth_confidence=-1, width_confidence=-1)),
    "trained_stats_confident_512_x10": dict(recipe="E", data="E", desc_scale=10.0, wseed=0, dseed=601, B=1, n=512, m=512, dim=256, conf=dict(depth_confidence=-1, width_confidence=-1)),
    "trained_stats_confident_512_x30": dict(recipe="E", data="E", desc_scale=30.0, wseed=0, dseed=601, B=1, n=512, m=512, dim=256, conf=dict(depth_confidence=-1, width_confidence=-1)),
}


def load_reference():
    spec = importlib.util.spec_from_file_location("lg_ref", str(REF))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def case_inputs(case: dict):
    """Weights (numpy state dict) and the input dict of a case; shared with the tests."""
    conf = case["conf"]
    sd = synth.make_state_dict(case["wseed"], input_dim=conf.get("input_dim", 256), add_scale_ori=conf.get("add_scale_ori", False), recipe=case["recipe"])
    if case["n"] == 0 or case["m"] == 0:
        rng = np.random.Generator(np.random.PCG64(case["dseed"]))
        def img(k):
            d = rng.standard_normal((case["B"], k, case["dim"])).astype(np.float32)
            return {"keypoints": (rng.random((case["B"], k, 2)) * 500).astype(np.float32), "descriptors": d,
                    "image_size": np.tile(np.array([[1024.0, 768.0]], np.float32), (case["B"], 1))}
        data = {"image0": img(case["n"]), "image1": img(case["m"])}
    else:
        data_kw = dict({"D": synth.RECIPE_D_DATA, "E": synth.RECIPE_E_DATA}.get(case.get("data"), {}))
        data = synth.make_batch(case["dseed"], case["B"], case["n"], case["m"], case["dim"], add_scale_ori=conf.get("add_scale_ori", False), **data_kw)
        if "desc_scale" in case:  # descriptor-scale sweep: the input domain the score bar is claimed on
            for k in ("image0", "image1"):
                data[k]["descriptors"] = (data[k]["descriptors"] * np.float32(case["desc_scale"])).astype(np.float32)
    if case.get("no_size"):
        for k in ("image0", "image1"):
            data[k].pop("image_size")
    return sd, data


def run_reference(lg, case: dict):
    sd, data = case_inputs(case)
    torch.set_grad_enabled(False)
    torch.manual_seed(0)
    saved = dict(lg.LightGlue.pruning_keypoint_thresholds)
    if "prune_th" in case:
        for k in lg.LightGlue.pruning_keypoint_thresholds:
            lg.LightGlue.pruning_keypoint_thresholds[k] = case["prune_th"]
    try:
        model = lg.LightGlue(features=None, **case["conf"]).eval()
        if case.get("static_lengths"):
            model.static_lengths = list(case["static_lengths"])   # what compile() sets (ref :454), without torch.compile
        res = model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
        assert not res.unexpected_keys and set(res.missing_keys) <= {"confidence_thresholds"}, res
        outs = []
        for b in range(case["B"]):  # B = 1 calls: the reference's adaptive path is B=1-only, and batched == per-pair otherwise
            td = {k: {kk: torch.from_numpy(vv[b:b + 1]) for kk, vv in v.items()} for k, v in data.items()}
            outs.append(model(td))
    finally:
        lg.LightGlue.pruning_keypoint_thresholds.update(saved)
    out = {
        "matches0": np.stack([o["matches0"][0].numpy() for o in outs]).astype(np.int64),
        "matches1": np.stack([o["matches1"][0].numpy() for o in outs]).astype(np.int64),
        "matching_scores0": np.stack([o["matching_scores0"][0].numpy() for o in outs]).astype(np.float32),
        "matching_scores1": np.stack([o["matching_scores1"][0].numpy() for o in outs]).astype(np.float32),
        "stop": np.array([int(o["stop"]) for o in outs], np.int64),
        "prune0": np.stack([o["prune0"][0].numpy() for o in outs]).astype(np.float32),
        "prune1": np.stack([o["prune1"][0].numpy() for o in outs]).astype(np.float32),
        "n_matches": np.array([o["matches"][0].shape[0] for o in outs], np.int64),
    }
    return sd, out


def main():
    lg = load_reference()
    out_dir = ROOT / "tests" / "golden"
    out_dir.mkdir(parents=True, exist_ok=True)
    only = set(sys.argv[1:])
    for name, case in CASES.items():
        if only and name not in only:
            continue
        sd, out = run_reference(lg, case)
        meta = dict(case=case, weights_sha256=synth.state_dict_digest(sd), torch=torch.__version__, numpy=np.__version__,
                    reference="cvg/LightGlue lightglue/lightglue.py (CPU fp32, loaded standalone)")
        np.savez_compressed(out_dir / f"{name}.npz", meta=json.dumps(meta), **out)
        hist0 = np.bincount(out["prune0"].astype(np.int64).ravel(), minlength=10).tolist() if out["prune0"].size else []
        conf_frac = float((out["matching_scores0"] > 0.5).mean()) if out["matching_scores0"].size else 0.0
        print(f"{name:32s} matches {out['n_matches'].tolist()} stop {out['stop'].tolist()} prune0 hist {hist0} scores0 > 0.5: {conf_frac:.3f}")


if __name__ == "__main__":
    main()
