#!/usr/bin/env python3
"""A/B on ONE box: the bench batch (cfg #2: 32 pairs of N = M = 1024) as ONE forward on one stream vs S sub-batches of 32 / S pairs on S
streams (one engine per stream).  Question: do the ramp / drain ends of the ~37 dependent launches of a forward (4 rounds of
workgroups each) fill with the other stream's kernels?  Same pipelining as bench.py (forward_deferred, results taken one step later).
usage: ab_streams.py [--pairs 32] [--kpts 1024] [--steps 20] [--warmup 5] [--streams 1,2,4] [--rounds 2]"""
import argparse
import json
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from lightglue_amd import LightGlue, synthetic  # noqa: E402


def build(sd, data_np, lo, hi, dev):
    model = LightGlue(features=None, depth_confidence=-1, width_confidence=-1).eval()
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    data = {k: {kk: torch.from_numpy(vv[lo:hi]).to(dev) for kk, vv in v.items()} for k, v in data_np.items()}
    model.reserve(hi - lo, data["image0"]["keypoints"].shape[1], data["image1"]["keypoints"].shape[1], dev)
    return model, data


def run(parts, streams, steps, warmup, dev):
    pending = [None] * len(parts)

    def step():
        outs = []
        for i, ((model, data), st) in enumerate(zip(parts, streams)):
            with torch.cuda.stream(st):
                prev, pending[i] = pending[i], model.forward_deferred(data)
            if prev is not None:
                outs.append(prev.result())
        return outs

    def drain():
        outs = []
        for i in range(len(parts)):
            if pending[i] is not None:
                outs.append(pending[i].result()); pending[i] = None
        return outs

    for _ in range(warmup):
        step()
    drain()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    last = drain()
    torch.cuda.synchronize(dev)
    return time.perf_counter() - t0, last


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=32)
    ap.add_argument("--kpts", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--streams", default="1,2,4")
    ap.add_argument("--rounds", type=int, default=2)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    B, n = args.pairs, args.kpts
    sd = synthetic.make_state_dict(0, recipe="A")
    data_np = synthetic.make_batch(1, B, n, n)
    ref = None
    for rnd in range(args.rounds):
        for S in [int(s) for s in args.streams.split(",")]:
            per = B // S
            parts = [build(sd, data_np, i * per, (i + 1) * per, dev) for i in range(S)]
            streams = [torch.cuda.current_stream(dev)] if S == 1 else [torch.cuda.Stream(dev) for _ in range(S)]
            dt, last = run(parts, streams, args.steps, args.warmup, dev)
            m0 = torch.cat([o["matches0"] for o in last]).cpu()
            if ref is None:
                ref = m0
            print(json.dumps({"round": rnd, "streams": S, "pairs_per_stream": per, "pairs_per_s": round(B * args.steps / dt, 1), "ms_per_step": round(1e3 * dt / args.steps, 3),
                              "matches_identical_to_first_run": bool((m0 == ref).all())}), flush=True)
            del parts


if __name__ == "__main__":
    main()
