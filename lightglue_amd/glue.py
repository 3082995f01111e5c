"""Extractor -> matcher plumbing: behaviour of the reference's `rbd`, `batch_to_device` and
`match_pair` helpers (reference `lightglue/utils.py:55-69, 150-165`), written for this package.  Image IO, resizing
and extractor classes are out of scope; `match_pair` accepts any object with the reference's
`extract(image, **preprocess) -> dict` contract (`lightglue/utils.py:136-147`)."""
from __future__ import annotations

from typing import Any, Callable, Dict

import torch


def _apply_to_tensors(obj: Any, fn: Callable[[torch.Tensor], torch.Tensor]) -> Any:
    """Rebuild nested dict / list / tuple containers with `fn` applied to every tensor leaf."""
    if torch.is_tensor(obj):
        return fn(obj)
    if isinstance(obj, dict):
        return {key: _apply_to_tensors(val, fn) for key, val in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [_apply_to_tensors(val, fn) for val in obj]
    return obj  # str, int, None, ...


def batch_to_device(batch: Dict[str, Any], device: str = "cpu", non_blocking: bool = True) -> Dict[str, Any]:
    """Detached copy of every tensor in `batch` on `device`."""
    return _apply_to_tensors(batch, lambda t: t.detach().to(device=device, non_blocking=non_blocking))


def rbd(data: Dict[str, Any]) -> Dict[str, Any]:
    """Strip the leading batch dimension: tensors and per-batch lists/tuples yield their first item, scalars
    (e.g. `stop`) pass through."""
    out = {}
    for key, val in data.items():
        indexable = torch.is_tensor(val) or isinstance(val, (list, tuple))
        out[key] = val[0] if indexable else val
    return out


def match_pair(extractor, matcher, image0: torch.Tensor, image1: torch.Tensor, device: str = "cuda", **preprocess):
    """Extract both images, match them, and return (feats0, feats1, matches01) without batch dimension on `device`."""
    feats = [extractor.extract(img, **preprocess) for img in (image0, image1)]
    matches01 = matcher({"image0": feats[0], "image1": feats[1]})
    feats0, feats1, matches01 = (batch_to_device(rbd(d), device) for d in (feats[0], feats[1], matches01))
    return feats0, feats1, matches01
