#!/usr/bin/env python3
"""Latency / throughput sweep over the number of keypoints, with the reference benchmark's command line and timing
methodology (reference `benchmark.py`: 10 warm-up forwards, `--repeat` timed forwards, one device synchronisation per
forward) — on SYNTHETIC keypoints / descriptors, because the reference's image pairs and the SuperPoint conv stack
are outside this package.  "easy" = image1 is a permuted, lightly jittered copy of image0 (recipe A
weights: hundreds of matches); "difficult" = half of image1's points are unrelated and the rest carry heavy descriptor
noise.

    python tools/benchmark.py --num_keypoints 512 1024 2048 4096 --measure throughput
    python tools/benchmark.py --no_prune_thresholds --batch 32
"""
import argparse
import sys
from collections import defaultdict
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from lightglue_amd import InflightMatcher, LightGlue, synthetic  # noqa: E402

torch.set_grad_enabled(False)


def measure(matcher, data, r=100):
    """Mean / std of the forward time in ms: events around each forward, one synchronisation per repetition."""
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(10):
        matcher(data)
    times = np.zeros(r)
    for rep in range(r):
        start.record()
        matcher(data)
        end.record()
        torch.cuda.synchronize()
        times[rep] = start.elapsed_time(end)
    return {"mean": float(times.mean()), "std": float(times.std())}


def measure_lanes(matcher, data, lanes: int, r=100):
    """Throughput form with `lanes` forwards in flight (lightglue_amd.InflightMatcher: one engine + one HIP stream per lane; every result is the full
    output dict of forward): wall time per forward in ms over r forwards.  The reference's synchronous forward has no counterpart; a stream of single
    pairs is where it pays (a B = 1 forward fills a fraction of the chip)."""
    import time
    fly = InflightMatcher(matcher, lanes)
    for _ in fly.map(data for _ in range(10)):
        pass
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in fly.map(data for _ in range(r)):
        pass
    torch.cuda.synchronize()
    return {"mean": (time.perf_counter() - t0) / r * 1e3, "std": 0.0}


def make_pair(kind: str, batch: int, n: int, device):
    data = synthetic.make_batch(11, batch, n, n)
    if kind == "difficult":
        rng = np.random.Generator(np.random.PCG64(5))
        d1 = data["image1"]["descriptors"]
        noise = rng.standard_normal(d1.shape).astype(np.float32)
        d1 = d1 + 0.06 * noise   # unit-norm descriptors have elements of ~1/16: noise of the signal's own magnitude
        half = n // 2
        d1[:, half:] = rng.standard_normal(d1[:, half:].shape).astype(np.float32)
        d1 /= np.linalg.norm(d1, axis=-1, keepdims=True)
        data["image1"]["descriptors"] = d1.astype(np.float32)
        data["image1"]["keypoints"][:, half:] = rng.uniform(0, 1, data["image1"]["keypoints"][:, half:].shape).astype(np.float32) * np.array([1024, 768], np.float32)
    return {k: {kk: torch.from_numpy(np.ascontiguousarray(vv)).to(device) for kk, vv in v.items()} for k, v in data.items()}


def print_as_table(rows: dict, title: str, columns):
    head = f"{title:30} " + " ".join(f"{c:>8}" for c in columns)
    print("\n" + head + "\n" + "-" * len(head))
    for name, vals in rows.items():
        print(f"{name:30}", " ".join(f"{v:>8.1f}" for v in vals))


def main():
    ap = argparse.ArgumentParser(description="Benchmark sweep for lightglue_amd (reference benchmark.py command line)")
    ap.add_argument("--device", choices=["auto", "cuda"], default="auto")
    ap.add_argument("--compile", action="store_true", help="also run with LightGlue.compile() (static lengths: pruning off, as in the reference)")
    ap.add_argument("--no_flash", action="store_true", help="accepted for compatibility; the attention kernel is always the hand-written flash-style one")
    ap.add_argument("--no_prune_thresholds", action="store_true", help="disable pruning thresholds (i.e. always do pruning)")
    ap.add_argument("--measure", default="time", choices=["time", "log-time", "throughput"])
    ap.add_argument("--repeat", "--r", type=int, default=100)
    ap.add_argument("--num_keypoints", nargs="+", type=int, default=[256, 512, 1024, 2048, 4096])
    ap.add_argument("--batch", type=int, default=1, help="image pairs per forward (extension; the reference benchmark is B = 1)")
    ap.add_argument("--precision", default="f16x3", choices=["f16x3", "bf16", "fp16", "fp32"])
    ap.add_argument("--lanes", type=int, default=1, help="forwards in flight (extension, lightglue_amd.InflightMatcher); > 1 reports wall time per forward / pairs per second of the pipelined stream "
                    "instead of the latency of a synchronous forward")
    args = ap.parse_args()
    if not torch.cuda.is_available():
        raise SystemExit("lightglue_amd needs an MI355X (ROCm device type 'cuda'); there is no CPU path to benchmark.")
    device = torch.device("cuda")
    print("Running benchmark on device:", torch.cuda.get_device_name(0))

    configs = {"LightGlue-full": dict(depth_confidence=-1, width_confidence=-1), "LightGlue-adaptive": {}}
    if args.compile:
        configs.update({k + "-compile": v for k, v in list(configs.items())})
    sd = {k: torch.from_numpy(v) for k, v in synthetic.make_state_dict(0, recipe="B").items()}
    results = {kind: defaultdict(list) for kind in ("easy", "difficult")}
    for name, conf in configs.items():
        print("Run benchmark for:", name)
        extra = dict(pruning_min_kpts=-1) if args.no_prune_thresholds else {}
        matcher = LightGlue(features=None, precision=args.precision, **conf, **extra).eval()
        matcher.load_state_dict(sd, strict=False)
        if name.endswith("compile"):
            matcher.compile()
        for kind in results:
            for n in args.num_keypoints:
                pair = make_pair(kind, args.batch, n, device)
                ms = (measure_lanes(matcher, pair, args.lanes, r=args.repeat) if args.lanes > 1 else measure(matcher, pair, r=args.repeat))["mean"]
                results[kind][name].append(1000.0 * args.batch / ms if args.measure == "throughput" else ms)
        del matcher
    for kind, rows in results.items():
        print_as_table(rows, f"{kind} [{'pairs/s' if args.measure == 'throughput' else 'ms'}]", args.num_keypoints)


if __name__ == "__main__":
    main()
