"""Extractor -> matcher glue of the reference (`lightglue/utils.py:55-69, 150-165`), SURVEY.md §8 f2.

Only the tensor plumbing is mirrored (`batch_to_device`, `rbd`, `match_pair`); image IO, resizing and the
`Extractor` base class stay out of scope (feature extractors are not part of the hot path).  `match_pair` takes any
extractor object with the reference's `extract(image, **preprocess) -> dict` contract (`lightglue/utils.py:136-147`).
"""
from __future__ import annotations

import collections.abc as collections
from typing import Callable

import torch


def map_tensor(input_, func: Callable):
    """ref utils.py:41-52"""
    string_classes = (str, bytes)
    if isinstance(input_, string_classes):
        return input_
    elif isinstance(input_, collections.Mapping):
        return {k: map_tensor(sample, func) for k, sample in input_.items()}
    elif isinstance(input_, collections.Sequence):
        return [map_tensor(sample, func) for sample in input_]
    elif isinstance(input_, torch.Tensor):
        return func(input_)
    else:
        return input_


def batch_to_device(batch: dict, device: str = "cpu", non_blocking: bool = True):
    """Move batch (dict) to device (ref utils.py:55-61)."""
    return map_tensor(batch, lambda tensor: tensor.to(device=device, non_blocking=non_blocking).detach())


def rbd(data: dict) -> dict:
    """Remove batch dimension from elements in data (ref utils.py:64-69)."""
    return {k: v[0] if isinstance(v, (torch.Tensor, list, tuple)) else v for k, v in data.items()}


def match_pair(extractor, matcher, image0: torch.Tensor, image1: torch.Tensor, device: str = "cuda", **preprocess):
    """Match a pair of images (image0, image1) with an extractor and a matcher (ref utils.py:150-165)."""
    feats0 = extractor.extract(image0, **preprocess)
    feats1 = extractor.extract(image1, **preprocess)
    matches01 = matcher({"image0": feats0, "image1": feats1})
    data = [feats0, feats1, matches01]
    feats0, feats1, matches01 = [batch_to_device(rbd(x), device) for x in data]  # remove batch dim and move
    return feats0, feats1, matches01
