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


# full extractor (conv stack included): name -> (weight seed, image seed, B, H, W, max_num_keypoints)
ENCODE_CASES = {
    "superpoint_full_b1_120x160": (0, 10, 1, 120, 160, None),
    "superpoint_full_b2_64x96_top50": (1, 11, 2, 64, 96, 50),
    # height / width NOT multiples of 8: floor max-pooling, score map cropped to (H // 8 * 8, W // 8 * 8) — what the
    # reference's Extractor.extract produces for most real images after its resize (e.g. 1024 x 683)
    "superpoint_full_b1_75x109_top40": (2, 12, 1, 75, 109, 40),
}


def encoder_state_dict(seed: int):
    """Seeded weights with the reference's names / shapes (ref superpoint.py:127-141): nn.Conv2d default init ranges, the
    keypoint logits scaled up so that the softmax is far from uniform (peaky score maps like the trained network's)."""
    g = torch.Generator().manual_seed(seed)
    shapes = [("conv1a", 64, 1, 3), ("conv1b", 64, 64, 3), ("conv2a", 64, 64, 3), ("conv2b", 64, 64, 3), ("conv3a", 128, 64, 3),
              ("conv3b", 128, 128, 3), ("conv4a", 128, 128, 3), ("conv4b", 128, 128, 3), ("convPa", 256, 128, 3), ("convPb", 65, 256, 1),
              ("convDa", 256, 128, 3), ("convDb", 256, 256, 1)]
    sd = {}
    for name, co, ci, k in shapes:
        bound = 1.0 / np.sqrt(ci * k * k)
        gain = 3.0 if name != "convPb" else 1.5       # keep activations alive through 10 ReLU layers; spread the logits
        sd[name + ".weight"] = (torch.rand((co, ci, k, k), generator=g) * 2 - 1) * bound * gain
        sd[name + ".bias"] = (torch.rand((co,), generator=g) * 2 - 1) * bound
    return sd


def encoder_image(seed: int, b: int, h: int, w: int):
    rng = np.random.Generator(np.random.PCG64(seed))
    img = rng.random((b, 1, h, w), dtype=np.float32)
    yy, xx = np.mgrid[0:h, 0:w]
    for i in range(b):
        for _ in range(25):   # blobs and edges so that the response is structured
            cy, cx = rng.integers(0, h), rng.integers(0, w)
            img[i, 0] += (rng.uniform(0.3, 1.0) * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / rng.uniform(4.0, 60.0))).astype(np.float32)
    return np.clip(img, 0, 1.5).astype(np.float32)


def load_reference_superpoint(sd, **conf):
    """The reference SuperPoint CLASS itself (superpoint.py:98-232), unmodified: its module is executed with stand-ins for the
    absent kornia import and the package's Extractor base (whose __init__ only merges the conf, utils.py:127-131), and
    torch.hub.load_state_dict_from_url — the network download of ref :143-144 — returns the seeded state dict."""
    from types import SimpleNamespace

    class Extractor(torch.nn.Module):
        def __init__(self, **c):
            super().__init__()
            self.conf = SimpleNamespace(**{**self.default_conf, **c})

    stub = types.ModuleType("kornia"); color = types.ModuleType("kornia.color"); color.rgb_to_grayscale = None
    pkg = types.ModuleType("lgref"); pkg.__path__ = []
    utils = types.ModuleType("lgref.utils"); utils.Extractor = Extractor
    saved = {k: sys.modules.get(k) for k in ("kornia", "kornia.color", "lgref", "lgref.utils")}
    sys.modules.update({"kornia": stub, "kornia.color": color, "lgref": pkg, "lgref.utils": utils})
    orig = torch.hub.load_state_dict_from_url
    torch.hub.load_state_dict_from_url = lambda url, *a, **k: sd
    try:
        mod = types.ModuleType("lgref.superpoint"); mod.__package__ = "lgref"
        exec(compile(REF.read_text(), str(REF), "exec"), mod.__dict__)
        model = mod.SuperPoint(**conf).eval()
    finally:
        torch.hub.load_state_dict_from_url = orig
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return model


def reference_encode(model, image: torch.Tensor):
    """scores (before NMS) and the raw convDb map of the reference forward: the statements of superpoint.py:159-184 / :213-214
    executed on the reference module's own layers."""
    x = model.relu(model.conv1a(image)); x = model.relu(model.conv1b(x)); x = model.pool(x)
    x = model.relu(model.conv2a(x)); x = model.relu(model.conv2b(x)); x = model.pool(x)
    x = model.relu(model.conv3a(x)); x = model.relu(model.conv3b(x)); x = model.pool(x)
    x = model.relu(model.conv4a(x)); x = model.relu(model.conv4b(x))
    sc = model.convPb(model.relu(model.convPa(x)))
    sc = torch.nn.functional.softmax(sc, 1)[:, :-1]
    b, _, h, w = sc.shape
    sc = sc.permute(0, 2, 3, 1).reshape(b, h, w, 8, 8).permute(0, 1, 3, 2, 4).reshape(b, h * 8, w * 8)
    return sc, model.convDb(model.relu(model.convDa(x)))


def main():
    ref = load_reference_functions()
    out_dir0 = ROOT / "tests" / "golden"
    only = set(sys.argv[1:])
    for name, (wseed, iseed, b, h, w, topk) in ENCODE_CASES.items():
        if only and name not in only:
            continue
        sd = encoder_state_dict(wseed)
        model = load_reference_superpoint(sd, max_num_keypoints=topk)
        img = torch.from_numpy(encoder_image(iseed, b, h, w))
        with torch.no_grad():
            sc, dense = reference_encode(model, img)
            arrays = {"case": np.array([wseed, iseed, b, h, w, -1 if topk is None else topk]), "scores": sc.numpy().astype(np.float32),
                      "dense_digest": np.array([float(dense.double().abs().mean()), float(dense.double().pow(2).mean())]),
                      "dense_sample": dense[:, ::16, ::3, ::5].numpy().astype(np.float32)}
            if topk is not None or b == 1:   # the reference's forward stacks per-image results: equal counts needed
                out = model({"image": img})
                arrays.update(keypoints=out["keypoints"].numpy(), keypoint_scores=out["keypoint_scores"].numpy(), descriptors=out["descriptors"].numpy())
        np.savez_compressed(out_dir0 / f"{name}.npz", **arrays)
        print(name, "scores", tuple(sc.shape), "max", float(sc.max()), "keypoints", None if "keypoints" not in arrays else arrays["keypoints"].shape)
    out_dir = ROOT / "tests" / "golden"
    if only:
        return
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
