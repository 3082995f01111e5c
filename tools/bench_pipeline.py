#!/usr/bin/env python3
"""Image pairs in, matches out: the SuperPoint extractor feeding the matcher on one MI355X — B pairs of grey images per step, both images of every
pair through ONE extractor forward (2B images, top-k keypoints, ragged counts carried as `num_keypoints`), then ONE matcher forward on the B pairs.  Random-init weights
(seeded; the released checkpoints are network-only) and random images: the numbers are the pipeline's cost, not its matching quality.  Reports ms per step and image
pairs/s for the extractor alone, the matcher alone on the extractor's output, and both.

usage: bench_pipeline.py [--pairs 8] [--kpts 1024] [--sizes 480x640,768x1024] [--steps 20] [--conv-precision fp32|f16x3]"""
import argparse
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tools")); sys.path.insert(0, str(ROOT / "tests"))
import gpu_util
import make_golden_superpoint as G
from lightglue_amd import SuperPoint
from lightglue_amd import synthetic as synth


def timed(fn, steps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=8)
    ap.add_argument("--kpts", type=int, default=1024)
    ap.add_argument("--sizes", default="480x640,768x1024")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--conv-precision", default="fp32", choices=["fp32", "f16x3"], help="the extractor's convolution arithmetic (lightglue_amd.SuperPoint conv_precision)")
    a = ap.parse_args()
    ext = SuperPoint(weights=G.encoder_state_dict(0), max_num_keypoints=a.kpts, conv_precision=a.conv_precision).cuda().eval()
    matcher = gpu_util.make_model(synth.make_state_dict(0, recipe="A"), "f16x3", depth_confidence=-1, width_confidence=-1)
    B = a.pairs
    for size in a.sizes.split(","):
        h, w = (int(x) for x in size.split("x"))
        img = torch.rand(2 * B, 1, h, w, device="cuda")
        wh = torch.tensor([[w, h]], dtype=torch.float32, device="cuda").expand(B, 2).contiguous()

        def extract():
            f = ext({"image": img})
            halves = []
            for s in (slice(0, B), slice(B, 2 * B)):
                halves.append({"keypoints": f["keypoints"][s], "descriptors": f["descriptors"][s], "num_keypoints": f["num_keypoints"][s], "image_size": wh})
            return {"image0": halves[0], "image1": halves[1]}

        t_e, data = timed(extract, a.steps)
        t_m, out = timed(lambda: matcher(data), a.steps)
        t_b, out = timed(lambda: matcher(extract()), a.steps)
        # the two stages on their own HIP streams: the extractor of step i + 1 runs beside the matcher of step i (forward_deferred: no host synchronisation in between)
        s_ext, s_mat = torch.cuda.Stream(), torch.cuda.Stream()
        def overlapped(steps):
            pend, feats, ev = None, None, None
            for _ in range(steps + 1):
                with torch.cuda.stream(s_ext):
                    nxt = extract() if _ < steps else None
                    ev_n = torch.cuda.Event(); ev_n.record(s_ext)
                if feats is not None:
                    with torch.cuda.stream(s_mat):
                        s_mat.wait_event(ev)
                        prev, pend = pend, matcher.forward_deferred(feats)
                    if prev is not None:
                        prev.result()
                feats, ev = nxt, ev_n
            return pend.result()
        overlapped(3); torch.cuda.synchronize(); t0 = time.perf_counter(); out2 = overlapped(a.steps); torch.cuda.synchronize(); t_o = (time.perf_counter() - t0) / a.steps
        same = all(torch.equal(x, y) for x, y in zip(out["matches"], out2["matches"]))
        n = data["image0"]["num_keypoints"].float().mean().item()
        print(f"[{a.conv_precision}] {h}x{w}, {B} pairs per step, {n:.0f} keypoints per image on average (cap {a.kpts}): extractor {t_e * 1e3:6.2f} ms ({2 * B / t_e:6.0f} images/s), "
              f"matcher {t_m * 1e3:6.2f} ms ({B / t_m:6.0f} pairs/s), images -> matches {t_b * 1e3:6.2f} ms = {B / t_b:6.0f} image pairs/s; the two stages on two streams {t_o * 1e3:6.2f} ms = {B / t_o:6.0f} image pairs/s (same matches: {same}); matches per pair {sum(len(x) for x in out['matches']) / B:.0f}", flush=True)


if __name__ == "__main__":
    main()
