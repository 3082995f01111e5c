#!/usr/bin/env python3
"""Generate tests/golden/superpoint_head_*.npz with the reference's OWN `sample_descriptors`
(/root/reference/lightglue/superpoint.py:80-95) on CPU.

superpoint.py imports kornia and the package's utils (cv2, torchvision-free but kornia-dependent), none of which
exist in this container, so the module is executed standalone with inert stand-ins for those imports; only the
pure-torch helper functions are used.  Inputs are regenerated from seeds (`head_inputs`), the fixtures store the
reference outputs only.
    python tools/make_golden_superpoint.py
"""
from __future__ import annotations

import sys
import types
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
REF = Path("/root/reference/lightglue/superpoint.py")

# name -> (seed, B, h, w, N)   [h, w = descriptor-map size, i.e. image / 8]
CASES = {
    "superpoint_head_b2_60x80_n300": (0, 2, 60, 80, 300),
    "superpoint_head_b1_96x128_n1024": (1, 1, 96, 128, 1024),
    "superpoint_head_b3_17x23_n64": (2, 3, 17, 23, 64),
}


def head_inputs(seed: int, b: int, h: int, w: int, n: int, s: int = 8):
    """Raw dense descriptor map [b,256,h,w] and keypoints [b,n,2] (integer pixel coordinates like SuperPoint's
    detector gives, incl. image corners/borders, plus a few sub-pixel and slightly out-of-image ones)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    dense = rng.standard_normal((b, 256, h, w)).astype(np.float32) * rng.uniform(0.2, 3.0, (b, 1, h, w)).astype(np.float32)
    kp = np.stack([rng.integers(0, w * s, (b, n)), rng.integers(0, h * s, (b, n))], -1).astype(np.float32)
    kp[:, 0] = (0, 0); kp[:, 1] = (w * s - 1, h * s - 1); kp[:, 2] = (w * s - 1, 0); kp[:, 3] = (0, h * s - 1)
    kp[:, 4:12] += rng.uniform(-0.5, 0.5, (b, 8, 2)).astype(np.float32)
    kp[:, 12] = (-3.0, 5.0); kp[:, 13] = (w * s + 2.5, h * s + 1.0)     # outside: zero-padding branch
    return dense, kp


def load_reference_functions():
    stub = types.ModuleType("kornia"); color = types.ModuleType("kornia.color"); color.rgb_to_grayscale = None
    pkg = types.ModuleType("lgref"); pkg.__path__ = []
    utils = types.ModuleType("lgref.utils"); utils.Extractor = torch.nn.Module
    saved = {k: sys.modules.get(k) for k in ("kornia", "kornia.color", "lgref", "lgref.utils")}
    sys.modules.update({"kornia": stub, "kornia.color": color, "lgref": pkg, "lgref.utils": utils})
    try:
        src = REF.read_text()
        mod = types.ModuleType("lgref.superpoint"); mod.__package__ = "lgref"
        exec(compile(src, str(REF), "exec"), mod.__dict__)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return mod


# keypoint extraction: name -> (seed, B, H, W, max_num_keypoints)
DETECT_CASES = {
    "superpoint_detect_b2_120x160": (0, 2, 120, 160, None),
    "superpoint_detect_b1_240x320_top300": (1, 1, 240, 320, 300),
    "superpoint_detect_b2_67x91_top50": (2, 2, 67, 91, 50),
}


def score_map(seed: int, b: int, h: int, w: int):
    """A keypoint score map shaped like SuperPoint's (softmax probabilities, mostly tiny, smooth blobs + isolated peaks
    + exact ties on plateaus, which are what simple_nms's equality tests are sensitive to)."""
    rng = np.random.Generator(np.random.PCG64(1000 + seed))
    s = rng.random((b, h, w), dtype=np.float32) ** 8 * 0.3
    yy, xx = np.mgrid[0:h, 0:w]
    for i in range(b):
        for _ in range(40):
            cy, cx, a = rng.integers(0, h), rng.integers(0, w), rng.uniform(0.05, 0.9)
            s[i] += (a * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / rng.uniform(2.0, 30.0))).astype(np.float32)
        for _ in range(12):   # plateaus: equal neighbouring values
            cy, cx = rng.integers(0, h - 3), rng.integers(0, w - 3)
            s[i, cy:cy + rng.integers(1, 4), cx:cx + rng.integers(1, 4)] = np.float32(rng.uniform(0.3, 0.8))
    return np.clip(s, 0, 1).astype(np.float32)


def reference_detect(ref, scores: torch.Tensor, nms_radius=4, remove_borders=4, detection_threshold=0.0005, max_num_keypoints=None):
    """The reference's own simple_nms / top_k_keypoints around a restatement of the few tensor statements of
    SuperPoint.forward in between (superpoint.py:186-218), which cannot be called without the conv stack."""
    b = scores.shape[0]
    scores = ref.simple_nms(scores, nms_radius)                                       # ref :186
    if remove_borders:                                                                # ref :189-194
        pad = remove_borders
        scores[:, :pad] = -1; scores[:, :, :pad] = -1; scores[:, -pad:] = -1; scores[:, :, -pad:] = -1
    best_kp = torch.where(scores > detection_threshold)                               # ref :197
    sc = scores[best_kp]
    keypoints = [torch.stack(best_kp[1:3], dim=-1)[best_kp[0] == i] for i in range(b)]
    sc = [sc[best_kp[0] == i] for i in range(b)]
    if max_num_keypoints is not None:                                                 # ref :207-215
        keypoints, sc = list(zip(*[ref.top_k_keypoints(k, s, max_num_keypoints) for k, s in zip(keypoints, sc)]))
    keypoints = [torch.flip(k, [1]).float() for k in keypoints]                       # ref :218
    return keypoints, list(sc)


def main():
    ref = load_reference_functions()
    out_dir = ROOT / "tests" / "golden"
    for name, (seed, b, h, w, topk) in DETECT_CASES.items():
        smap = score_map(seed, b, h, w)
        with torch.no_grad():
            nms = ref.simple_nms(torch.from_numpy(smap.copy()), 4).numpy()
            kps, scs = reference_detect(ref, torch.from_numpy(smap.copy()), max_num_keypoints=topk)
        arrays = {"case": np.array([seed, b, h, w, -1 if topk is None else topk]), "nms_nonzero": np.packbits(nms != 0)}
        for i in range(b):
            arrays[f"kp{i}"] = kps[i].numpy(); arrays[f"sc{i}"] = scs[i].numpy()
        np.savez_compressed(out_dir / f"{name}.npz", **arrays)
        print(name, [len(k) for k in kps])
    for name, (seed, b, h, w, n) in CASES.items():
        dense, kp = head_inputs(seed, b, h, w, n)
        with torch.no_grad():
            d = torch.nn.functional.normalize(torch.from_numpy(dense), p=2, dim=1)                    # ref :218
            per_image = [ref.sample_descriptors(torch.from_numpy(kp[i:i + 1].copy()), d[i:i + 1], 8)[0] for i in range(b)]  # ref :221-224
            desc = torch.stack(per_image, 0).transpose(-1, -2).contiguous()                           # ref :228
            sampled_only = ref.sample_descriptors(torch.from_numpy(kp.copy()), torch.from_numpy(dense), 8)  # raw map, batched
        extra = {"sampled_unnormalized_map": sampled_only.numpy()} if n <= 64 else {}
        np.savez_compressed(out_dir / f"{name}.npz", case=np.array([seed, b, h, w, n]), descriptors=desc.numpy(), **extra)
        print(name, desc.shape, float(desc.norm(dim=-1).mean()))


if __name__ == "__main__":
    main()
