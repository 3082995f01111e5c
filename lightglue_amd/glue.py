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


def prefetch_to_device(batches, device="cuda", depth: int = 2):
    """Host batches -> device batches, up to `depth` of them AHEAD of the consumer, copied on a separate HIP stream (no reference counterpart: the reference's
    callers do `batch_to_device` on the compute stream, `lightglue/utils.py:63-69`).  For a matcher fed from host memory — features arriving over the network or from
    an extractor on another device — the host-to-device copy of batch i + 1 then runs under the forward of batch i instead of in front of it (cfg #2: 68 MB per batch,
    about 1.2 ms over PCIe against a 7.4 ms forward; measured by `bench.py`'s `pcie_inclusive` block).  Tensors that are not in pinned memory are staged through a pinned
    copy first (a pageable source would make the copy synchronous); pass pinned tensors to avoid that extra host pass.  Every yielded batch is ordered behind its copy on
    the stream that is current when it is consumed, and its device memory is handed to that stream (`record_stream`), so nothing else is needed before `matcher(batch)`."""
    import collections
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("prefetch_to_device copies to an MI355X (ROCm device type 'cuda')")
    if depth < 1:
        raise ValueError("depth must be >= 1")
    copy_stream = torch.cuda.Stream(device)
    queue = collections.deque()

    def stage(batch):
        def to_dev(t):
            if t.is_cuda:
                return t
            src = t.detach()
            if not src.is_pinned():
                src = src.contiguous().pin_memory()
            return src.to(device, non_blocking=True), src
        keep = []
        def conv(t):
            r = to_dev(t)
            if isinstance(r, tuple):
                keep.append(r[1])
                return r[0]
            return r
        with torch.cuda.stream(copy_stream):
            dev = _apply_to_tensors(batch, conv)
            done = torch.cuda.Event()
            done.record(copy_stream)
        return dev, done, keep           # `keep`: the pinned sources stay alive until the copy has been waited for

    def release(item):
        dev, done, _keep = item
        cur = torch.cuda.current_stream(device)
        cur.wait_event(done)
        _apply_to_tensors(dev, lambda t: (t.record_stream(cur), t)[1] if t.is_cuda else t)
        return dev

    for batch in batches:
        queue.append(stage(batch))
        if len(queue) > depth:
            yield release(queue.popleft())
    while queue:
        yield release(queue.popleft())


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


def extracted_to_image_frame(feats: Dict[str, Any], original_hw, scales: torch.Tensor) -> Dict[str, Any]:
    """The last two steps of the reference's `Extractor.extract` (`lightglue/utils.py:142-147`) for an extractor that
    ran on a RESIZED image: keypoints go back to the original image's pixel frame, `(k + 0.5) / scales - 0.5`, and
    `image_size` = original (width, height) is attached — the two fields the matcher's keypoint normalisation reads
    (`lightglue.py:32-43`).  `original_hw` = (H, W) of the image before resizing, `scales` = resized / original per axis
    (x, y) as the reference's ImagePreprocessor returns it."""
    kp = feats["keypoints"]
    scales = torch.as_tensor(scales, dtype=kp.dtype, device=kp.device)
    out = dict(feats)
    out["keypoints"] = (kp + 0.5) / scales[None] - 0.5
    h, w = int(original_hw[0]), int(original_hw[1])
    out["image_size"] = torch.tensor([[w, h]], dtype=kp.dtype, device=kp.device)
    return out


_PER_KEYPOINT = ("keypoints", "descriptors", "scales", "oris", "keypoint_scores")


def collate_features(feats: list) -> Dict[str, Any]:
    """Stack per-image feature dicts with DIFFERENT keypoint counts into one ragged batch for `LightGlue.forward`.

    Each item follows the extractor contract (reference `lightglue/utils.py:136-147`), with or without the leading
    batch dimension of 1: `keypoints [N_i,2]`, `descriptors [N_i,D]`, optional `scales`/`oris`/`keypoint_scores
    [N_i]` and `image_size [2]`.  Per-keypoint tensors are zero-padded to max N_i and `num_keypoints [B]` carries
    the true counts; the engine never reads the padding (the reference instead pads with ones and masks every
    attention call, `lightglue.py:46-55, 512-520`, and only inside `compile()`d B=1 calls)."""
    items = []
    for f in feats:
        kp = f["keypoints"]
        items.append(rbd(f) if kp.dim() == 3 else f)
    counts = [int(f["keypoints"].shape[0]) for f in items]
    nmax = max(counts) if counts else 0
    out: Dict[str, Any] = {}
    for key in _PER_KEYPOINT:
        if not all(key in f for f in items):
            continue
        first = items[0][key]
        buf = first.new_zeros((len(items), nmax) + tuple(first.shape[1:]))
        for b, f in enumerate(items):
            buf[b, : counts[b]] = f[key]
        out[key] = buf
    if all("image_size" in f for f in items):
        out["image_size"] = torch.stack([torch.as_tensor(f["image_size"], dtype=torch.float32).reshape(2) for f in items]).to(out["keypoints"].device)
    out["num_keypoints"] = torch.tensor(counts, dtype=torch.int32, device=out["keypoints"].device)
    return out


def match_batch(matcher, feats0: list, feats1: list) -> list:
    """Match B independent image pairs with ragged keypoint counts in ONE forward; returns one `rbd`-style result
    dict per pair, trimmed to that pair's own keypoint counts (what B separate `matcher(...)` calls would give)."""
    assert len(feats0) == len(feats1)
    batch0, batch1 = collate_features(feats0), collate_features(feats1)
    res = matcher({"image0": batch0, "image1": batch1})
    n0, n1 = batch0["num_keypoints"].tolist(), batch1["num_keypoints"].tolist()
    out = []
    for b in range(len(feats0)):
        stop = res["stop"]
        out.append({
            "matches0": res["matches0"][b, : n0[b]], "matches1": res["matches1"][b, : n1[b]],
            "matching_scores0": res["matching_scores0"][b, : n0[b]], "matching_scores1": res["matching_scores1"][b, : n1[b]],
            "matches": res["matches"][b], "scores": res["scores"][b],
            "prune0": res["prune0"][b, : n0[b]], "prune1": res["prune1"][b, : n1[b]],
            "stop": int(stop[b]) if torch.is_tensor(stop) else stop,
        })
    return out


def cm_prune(prune):
    """RGBA colour per keypoint from the matcher's ``prune0`` / ``prune1`` output (the layer at which a point was
    dropped; points alive to the end carry the maximum) — the consumer the reference ships for that output
    (``lightglue/viz2d.py:33-39`` with its ``cm_BlRdGn`` ramp, :22-31): survivors blue, the others red (dropped after
    layer 1) -> yellow -> green (layer 10).  Pure numpy: returns ``[N, 4]`` floats in [0, 1] for any plotting backend."""
    import numpy as np
    p = np.asarray(prune.detach().cpu() if isinstance(prune, torch.Tensor) else prune, dtype=np.float64)
    t = np.where(p == p.max(), -1.0, (p - 1.0) / 9.0)              # -1 marks the survivors
    up = 2.0 * np.clip(t, 0.0, 1.0)[..., None]                       # red (0) -> green (1)
    ramp = up * np.array([0.0, 1.0, 0.0, 1.0]) + (2.0 - up) * np.array([1.0, 0.0, 0.0, 1.0])
    dn = -2.0 * np.clip(t, -1.0, 0.0)[..., None]                     # red (0) -> blue (-1)
    blue = dn * np.array([0.0, 0.1, 1.0, 1.0]) + (2.0 - dn) * np.array([1.0, 0.0, 0.0, 1.0])
    return np.clip(np.where(t[..., None] < 0.0, blue, ramp), 0.0, 1.0)
