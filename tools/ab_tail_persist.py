#!/usr/bin/env python3
"""Persistent fused tail (engine option tail_persist, lg_tail.hip PERSIST) against the one-tile-per-workgroup form: bit identity of every output on full and
ragged fixed-depth batches (incl. NaN-poisoned padding and tile counts that are not multiples of the grid), then whole-step timing at cfg #2 / cfg #4 with
the per-class kernel times, alternating the two forms inside one process (same box, same clock state).

The kernel form lives in tools/experiments/tail_persist.patch (apply to lightglue_amd/csrc, `make`): measured +-0 and not adopted (LAB_NOTES.md round 6, call pa - pc).

usage: ab_tail_persist.py [--check-only] [--rounds 3] [--steps 40]"""
import argparse
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import gpu_util
from lightglue_amd import synthetic as synth

KEYS = ("matches0", "matches1", "matching_scores0", "matching_scores1")


def check():
    fixed = dict(depth_confidence=-1, width_confidence=-1)
    cases = [("A", (1, 32, 1024, 1024), {}, 256),
             ("D", (61, 8, 2048, 2048), dict(recipe_d_data=True), 256),
             ("A", (3, 9, 2048, 1920), {}, 256),                                   # 558 tiles on 256 workgroups: 3 / 2 tiles per workgroup
             ("A", (4, 24, 1100, 700), dict(nums=True, poison=True), 256),        # ragged: dead tiles between live ones, NaN padding
             ("A", (5, 6, 1024, 1024), {}, 40),                                   # few workgroups, many tiles each (5 / 4)
             ("A", (6, 16, 1024, 1024), dict(log_assignment=True), 97)]           # a grid that is no multiple of 8
    for recipe, (seed, B, n, m), opt, wgs in cases:
        sd = synth.make_state_dict(0, recipe=recipe)
        batch = synth.make_batch(seed, B, n, m, **(synth.RECIPE_D_DATA if opt.get("recipe_d_data") else {}))
        nums = None
        if opt.get("nums"):
            rng = np.random.default_rng(seed)
            nums = (rng.integers(0, n + 1, B), rng.integers(0, m + 1, B))
            nums[0][0], nums[1][0] = n, m
            nums[0][1] = 0
            for img, cnt in zip(("image0", "image1"), nums):
                for b, c in enumerate(cnt):
                    batch[img]["keypoints"][b, c:] = np.nan
                    batch[img]["descriptors"][b, c:] = np.nan
        data = gpu_util.to_torch(batch)
        if nums is not None:
            data["image0"]["num_keypoints"] = torch.as_tensor(nums[0], dtype=torch.int32, device="cuda")
            data["image1"]["num_keypoints"] = torch.as_tensor(nums[1], dtype=torch.int32, device="cuda")
        model = gpu_util.make_model(sd, "f16x3", **fixed)
        model.check_finite = False
        model.return_log_assignment = bool(opt.get("log_assignment"))
        off = model(data)
        model.set_option("tail_persist", wgs)
        on = [model(data) for _ in range(3)]
        model.set_option("tail_persist", 0)
        again = model(data)
        keys = KEYS + (("log_assignment",) if opt.get("log_assignment") else ())
        for k in keys:
            for r, o in enumerate(on):
                same = torch.equal(off[k], o[k]) if not torch.is_floating_point(off[k]) else torch.equal(torch.nan_to_num(off[k], nan=-7.0), torch.nan_to_num(o[k], nan=-7.0))
                assert same, (recipe, B, n, m, wgs, k, r, (off[k] != o[k]).sum().item())
            assert torch.equal(torch.nan_to_num(off[k].float(), nan=-7.0), torch.nan_to_num(again[k].float(), nan=-7.0)), (k, "second plain forward")
        nm = int((off["matches0"] >= 0).sum())
        print(f"check recipe {recipe} B={B} {n}x{m} wgs={wgs}{' ragged' if nums is not None else ''}: bit-identical ({nm} matches)", flush=True)


def timing(rounds, steps, configs):
    for cfg in configs:
        if cfg == 2:
            B, n, dim, kw = 32, 1024, 256, {}
        else:
            B, n, dim, kw = 32, 4096, 128, dict(input_dim=128)
        sd = synth.make_state_dict(0, recipe="A", input_dim=dim)
        model = gpu_util.make_model(sd, "f16x3", depth_confidence=-1, width_confidence=-1, **kw)
        data = gpu_util.to_torch(synth.make_batch(1, B, n, n, dim))
        model.reserve(B, n, n, "cuda")
        for _ in range(3):
            model(data)
        for r in range(rounds):
            for persist, mode in ((0, 0), (1, 0), (1, 1), (1, 2)):
                model.set_option("tail_persist", persist)
                model.set_option("tail_persist_mode", mode)
                pend = None
                for _ in range(5):
                    prev, pend = pend, model.forward_deferred(data)
                    if prev is not None:
                        prev.result()
                pend.result(); torch.cuda.synchronize()
                model.profile(True, "cuda")
                t0 = time.perf_counter(); pend = None
                for _ in range(steps):
                    prev, pend = pend, model.forward_deferred(data)
                    if prev is not None:
                        prev.result()
                pend.result(); torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                prof = {k: round(v[0] / steps, 3) for k, v in model.profile_read("cuda").items() if v[1] > 0}
                model.profile(False, "cuda")
                top = {k: prof[k] for k in ("fused_tail", "attn_self", "attn_cross") if k in prof}
                print(f"cfg{cfg} round {r} tail_persist={persist} mode={mode}: {B * steps / dt:8.1f} pairs/s  {1e3 * dt / steps:7.3f} ms/step  {top}", flush=True)
        model.set_option("tail_persist", 0)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--check-only", action="store_true")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--configs", default="2,4")
    a = ap.parse_args()
    check()
    if not a.check_only:
        timing(a.rounds, a.steps, [int(c) for c in a.configs.split(",")])
