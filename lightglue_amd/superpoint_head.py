"""SuperPoint descriptor head on the MI355X: the step between the SuperPoint conv stack and
the matcher.  Mirrors the reference's `sample_descriptors` (lightglue/superpoint.py:80-95) and the descriptor tail
of `SuperPoint.forward` (:216-228); both run in `lightglue_amd/csrc/lg_superpoint.hip` through
`lg_sp_sample_descriptors` (include/lightglue_amd.h).  The conv stack, NMS and top-k selection stay out of scope.
No CPU fallback: CPU tensors raise."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _cabi


def _run(keypoints: torch.Tensor, dense: torch.Tensor, s: int, normalize_dense: bool,
         num_keypoints: Optional[torch.Tensor]) -> torch.Tensor:
    if dense.device.type != "cuda":
        raise RuntimeError("lightglue_amd.superpoint_head runs on MI355X (ROCm device type 'cuda') only; there is no "
                           f"CPU fallback. Got a descriptor map on {dense.device}.")
    b, c, h, w = dense.shape
    assert keypoints.shape[0] == b and keypoints.shape[-1] == 2, "keypoints must be [B, N, 2]"
    device = dense.device
    f32 = lambda t: t.detach().to(device=device, dtype=torch.float32).contiguous()
    dense, keypoints = f32(dense), f32(keypoints)
    n = keypoints.shape[1]
    num = None
    if num_keypoints is not None:
        num = torch.as_tensor(num_keypoints).to(device=device, dtype=torch.int32).contiguous()
        assert num.shape == (b,)
    out = torch.empty((b, n, c), device=device, dtype=torch.float32)
    work = torch.empty((b, h, w, c), device=device, dtype=torch.float32)
    ptr = lambda t: None if t is None or t.numel() == 0 else t.data_ptr()
    with torch.cuda.device(device):
        stream = torch.cuda.current_stream(device).cuda_stream
        _cabi.check(_cabi.load().lg_sp_sample_descriptors(
            ptr(dense), b, c, h, w, ptr(keypoints), ptr(num), n, int(s), int(normalize_dense), ptr(work), ptr(out),
            C.c_void_p(stream)))
    return out


def sample_descriptors(keypoints: torch.Tensor, descriptors: torch.Tensor, s: int = 8) -> torch.Tensor:
    """Same contract as the reference function (superpoint.py:80-95): `keypoints [b, N, 2]` pixel (x, y),
    `descriptors [b, c, h, w]` -> L2-normalised bilinear samples `[b, c, N]` (a transposed view of the kernel's
    matcher-ready `[b, N, c]` output).  Unlike the reference, `keypoints` is not modified in place."""
    return _run(keypoints, descriptors, s, False, None).transpose(1, 2)


def descriptor_head(keypoints: torch.Tensor, dense_descriptors: torch.Tensor, s: int = 8,
                    num_keypoints: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Descriptor tail of SuperPoint.forward (superpoint.py:216-228) for a whole (ragged) batch: dense L2
    normalisation over channels, sampling at the keypoints, final normalisation, `[B, N, 256]` layout.
    `dense_descriptors` is the raw `convDb` output `[B, 256, H/8, W/8]`; rows >= num_keypoints[b] come back zero."""
    return _run(keypoints, dense_descriptors, s, True, num_keypoints)


def detect_keypoints(scores: torch.Tensor, nms_radius: int = 4, remove_borders: int = 4, detection_threshold: float = 0.0005,
                     max_num_keypoints: Optional[int] = None, capacity: Optional[int] = None):
    """Keypoint extraction of SuperPoint.forward (superpoint.py:186-218) on the dense score map `scores [B, H, W]`
    (after softmax / depth-to-space, :176-184): simple_nms, border removal, threshold, optional top-k.

    Returns `(keypoints [B, C, 2] float (x, y), keypoint_scores [B, C], num_keypoints [B] int32)` — a ragged batch in
    the form `LightGlue.forward` / `descriptor_head` take (`num_keypoints`); rows >= num_keypoints[b] are undefined.
    C = `capacity` (default: max_num_keypoints, or 1/8 of the pixels when it is None).  Without top-k, raises if an
    image has more detections than `capacity` rows (the internal candidate buffer always holds every pixel)."""
    if scores.device.type != "cuda":
        raise RuntimeError("lightglue_amd.superpoint_head runs on MI355X (ROCm device type 'cuda') only; there is no "
                           f"CPU fallback. Got scores on {scores.device}.")
    if max_num_keypoints is not None and max_num_keypoints <= 0:
        raise ValueError("max_num_keypoints must be positive or None")   # ref superpoint.py:143-144
    b, h, w = scores.shape
    device = scores.device
    scores = scores.detach().to(dtype=torch.float32).contiguous()
    k = int(max_num_keypoints) if max_num_keypoints is not None else 0
    cap = int(capacity) if capacity is not None else (k if k > 0 else max(1, h * w // 8))
    maxc = h * w   # candidates before top-k: worst case every pixel (8 bytes each, twice the score map)
    lib = _cabi.load()
    nbytes = lib.lg_sp_detect_workspace_bytes(b, h, w, maxc)
    work = torch.empty((nbytes,), device=device, dtype=torch.uint8)
    kpts = torch.empty((b, cap, 2), device=device, dtype=torch.float32)
    kscores = torch.empty((b, cap), device=device, dtype=torch.float32)
    counts = torch.empty((b,), device=device, dtype=torch.int32)
    totals = torch.empty((b,), device=device, dtype=torch.int32)
    with torch.cuda.device(device):
        stream = torch.cuda.current_stream(device).cuda_stream
        _cabi.check(lib.lg_sp_detect(scores.data_ptr(), b, h, w, int(nms_radius), int(remove_borders), float(detection_threshold), k, cap,
                                     maxc, work.data_ptr(), nbytes, kpts.data_ptr(), kscores.data_ptr(), counts.data_ptr(),
                                     totals.data_ptr(), C.c_void_p(stream)))
    if k == 0:
        worst = int(totals.max().item())
        if worst > cap:
            raise RuntimeError(f"{worst} detections in one image exceed `capacity` = {cap} output rows; pass a larger one")
    return kpts, kscores, counts
