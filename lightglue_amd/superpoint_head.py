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
