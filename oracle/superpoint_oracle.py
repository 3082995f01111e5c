"""TEST INFRASTRUCTURE ONLY — numpy restatement of the reference's SuperPoint descriptor head.

Only tests/ may import this file; the product path (lightglue_amd/superpoint_head.py -> lg_superpoint.hip) never
does.  Follows lightglue/superpoint.py:80-95 (`sample_descriptors`) and :216-228 (descriptor tail of
`SuperPoint.forward`); torch ops are restated from their documented semantics: `F.grid_sample(mode="bilinear",
padding_mode="zeros", align_corners=True)` and `F.normalize(p=2, eps=1e-12)`.
Parity is pinned by tests/golden/superpoint_head_*.npz, produced by running the reference's own function
(tools/make_golden_superpoint.py)."""
from __future__ import annotations

import numpy as np


def l2_normalize(x: np.ndarray, axis: int) -> np.ndarray:
    """F.normalize(p=2, dim=axis): x / max(||x||_2, 1e-12)."""
    nrm = np.sqrt((x.astype(np.float32) ** 2).sum(axis=axis, keepdims=True, dtype=np.float32))
    return (x / np.maximum(nrm, np.float32(1e-12))).astype(np.float32)


def grid_sample_bilinear_align(descriptors: np.ndarray, grid: np.ndarray) -> np.ndarray:
    """descriptors [c, h, w], grid [N, 2] in [-1, 1] (x, y) -> [c, N]; align_corners=True, zero padding."""
    c, h, w = descriptors.shape
    ix = (grid[:, 0] + np.float32(1)) / np.float32(2) * np.float32(w - 1)
    iy = (grid[:, 1] + np.float32(1)) / np.float32(2) * np.float32(h - 1)
    x0, y0 = np.floor(ix), np.floor(iy)
    tx, ty = (ix - x0).astype(np.float32), (iy - y0).astype(np.float32)
    out = np.zeros((c, grid.shape[0]), np.float32)
    for dy, dx, wgt in ((0, 0, (1 - tx) * (1 - ty)), (0, 1, tx * (1 - ty)), (1, 0, (1 - tx) * ty), (1, 1, tx * ty)):
        x, y = (x0 + dx).astype(np.int64), (y0 + dy).astype(np.int64)
        ok = (x >= 0) & (x < w) & (y >= 0) & (y < h)
        vals = descriptors[:, np.clip(y, 0, h - 1), np.clip(x, 0, w - 1)]
        out += vals * (wgt * ok).astype(np.float32)[None]
    return out


def sample_descriptors(keypoints: np.ndarray, descriptors: np.ndarray, s: int = 8) -> np.ndarray:
    """ref superpoint.py:80-95.  keypoints [b, N, 2] pixel (x, y); descriptors [b, c, h, w] -> [b, c, N]."""
    b, c, h, w = descriptors.shape
    k = keypoints.astype(np.float32) - np.float32(s / 2) + np.float32(0.5)                       # ref :83
    k = k / np.array([w * s - s / 2 - 0.5, h * s - s / 2 - 0.5], np.float32)[None]               # ref :84-88
    k = k * np.float32(2) - np.float32(1)                                                        # ref :89
    out = np.stack([grid_sample_bilinear_align(descriptors[i].astype(np.float32), k[i]) for i in range(b)])  # ref :91
    return l2_normalize(out, 1)                                                                  # ref :92-94


def descriptor_head(keypoints: np.ndarray, dense: np.ndarray, s: int = 8) -> np.ndarray:
    """ref superpoint.py:216-228: normalise the dense map over channels, sample, -> [b, N, c]."""
    return sample_descriptors(keypoints, l2_normalize(dense, 1), s).transpose(0, 2, 1).copy()


# --------------------------------------------------------------------------- keypoint extraction
def max_pool(x: np.ndarray, r: int) -> np.ndarray:
    """F.max_pool2d(kernel 2r+1, stride 1, padding r) on [H, W] (padding value -inf)."""
    h, w = x.shape
    p = np.full((h + 2 * r, w + 2 * r), -np.inf, dtype=x.dtype)
    p[r:r + h, r:r + w] = x
    out = np.full_like(x, -np.inf)
    for dy in range(2 * r + 1):
        for dx in range(2 * r + 1):
            out = np.maximum(out, p[dy:dy + h, dx:dx + w])
    return out


def simple_nms(scores: np.ndarray, nms_radius: int) -> np.ndarray:
    """ref superpoint.py:52-70 for one [H, W] map."""
    zeros = np.zeros_like(scores)
    max_mask = scores == max_pool(scores, nms_radius)                                   # ref :62
    for _ in range(2):                                                                  # ref :63
        supp_mask = max_pool(max_mask.astype(np.float32), nms_radius) > 0               # ref :64
        supp_scores = np.where(supp_mask, zeros, scores)                                # ref :65
        new_max_mask = supp_scores == max_pool(supp_scores, nms_radius)                 # ref :66
        max_mask = max_mask | (new_max_mask & (~supp_mask))                             # ref :67
    return np.where(max_mask, scores, zeros)                                            # ref :68


def detect_keypoints(scores: np.ndarray, nms_radius=4, remove_borders=4, detection_threshold=0.0005, max_num_keypoints=None):
    """ref superpoint.py:186-218 for one image: returns (keypoints [K, 2] float (x, y), scores [K])."""
    s = simple_nms(scores.astype(np.float32), nms_radius)
    if remove_borders:                                                                  # ref :189-194
        pad = remove_borders
        s = s.copy()
        s[:pad] = -1; s[:, :pad] = -1; s[-pad:] = -1; s[:, -pad:] = -1
    ys, xs = np.where(s > np.float32(detection_threshold))                              # ref :197 (row-major)
    sc = s[ys, xs]
    if max_num_keypoints is not None and max_num_keypoints < len(sc):                   # ref :73-77
        order = np.lexsort((np.arange(len(sc)), -sc.astype(np.float64)))[:max_num_keypoints]   # score desc, index asc on ties
        ys, xs, sc = ys[order], xs[order], sc[order]
    return np.stack([xs, ys], -1).astype(np.float32), sc                                # ref :218
