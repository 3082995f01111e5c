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
