#!/usr/bin/env python3
"""A/B on ONE box: K steps of a BASELINE config with F WHOLE batches in flight — step i runs on stream i % F with its own engine (one `LightGlue`
instance per stream, same weights), its result is taken F steps later (`forward_deferred`).  Question: at the adaptive configs most pairs of a
batch stop early and the late layers' launches fill a fraction of the chip (cfg #3': 4 of 16 pairs alive after layer 3) — does a second batch's
early layers fill it?  (tools/ab_streams.py asked the sub-batch form of this at cfg #2 in round 3: nothing.)
usage: ab_inflight.py [--configs 2,3,5] [--inflight 1,2,3] [--steps 24] [--warmup 6] [--rounds 2]"""
import argparse
import json
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402  (CONFIGS)
from lightglue_amd import LightGlue, synthetic  # noqa: E402


def build(sd, data_np, cfg, dev):
    kw = {} if cfg["adaptive"] else dict(depth_confidence=-1, width_confidence=-1)
    model = LightGlue(features=None, input_dim=cfg["dim"], **kw).eval()
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    model.track_inplace_weight_edits = False
    data = {k: {kk: torch.from_numpy(vv).to(dev) for kk, vv in v.items()} for k, v in data_np.items()}
    model.reserve(cfg["pairs"], cfg["n"], cfg["m"], dev)
    return model, data


def run(parts, streams, steps, warmup, dev):
    F = len(parts)
    pending = [None] * F
    last = [None]

    def step(i):
        k = i % F
        model, data = parts[k]
        with torch.cuda.stream(streams[k]):
            prev, pending[k] = pending[k], model.forward_deferred(data)
        if prev is not None:
            last[0] = prev.result()

    def drain():
        for k in range(F):
            if pending[k] is not None:
                last[0] = pending[k].result(); pending[k] = None

    for i in range(warmup):
        step(i)
    drain()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    drain()
    torch.cuda.synchronize(dev)
    return time.perf_counter() - t0, last[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="2,3,5", help="BASELINE config numbers; or shapes as B:N[:adaptive], e.g. 1:512 or 1:2048:a (config 1 and the B = 1 serving case)")
    ap.add_argument("--inflight", default="1,2,3")
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--rounds", type=int, default=2)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    for c in args.configs.split(","):
        if ":" in c:
            f = c.split(":")
            ad = len(f) > 2 and f[2].startswith("a")
            cfg = dict(pairs=int(f[0]), n=int(f[1]), m=int(f[1]), dim=256, recipe="C" if ad else "A", wseed=0, adaptive=ad)
        else:
            c = int(c); cfg = bench.CONFIGS[c]
        sd = synthetic.make_state_dict(cfg["wseed"], input_dim=cfg["dim"], recipe=cfg["recipe"])
        data_np = synthetic.make_batch(1, cfg["pairs"], cfg["n"], cfg["m"], dim=cfg["dim"])
        ref = None
        for rnd in range(args.rounds):
            for F in [int(s) for s in args.inflight.split(",")]:
                parts = [build(sd, data_np, cfg, dev) for _ in range(F)]
                streams = [torch.cuda.current_stream(dev)] if F == 1 else [torch.cuda.Stream(dev) for _ in range(F)]
                dt, last = run(parts, streams, args.steps, args.warmup, dev)
                m0 = last["matches0"].cpu()
                if ref is None:
                    ref = m0
                print(json.dumps({"config": c, "round": rnd, "batches_in_flight": F, "pairs_per_s": round(cfg["pairs"] * args.steps / dt, 1), "ms_per_step": round(1e3 * dt / args.steps, 3),
                                  "matches_identical_to_first_run": bool((m0 == ref).all())}), flush=True)
                del parts


if __name__ == "__main__":
    main()
