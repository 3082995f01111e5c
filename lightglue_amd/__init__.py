"""lightglue_amd — MI355X-native (gfx950) LightGlue matcher forward path.

Drop-in for the hot path of cvg/LightGlue's ``lightglue.LightGlue`` (reference
``lightglue/__init__.py:4``): same constructor, same ``forward({'image0','image1'})`` dict API, all
arithmetic in hand-written HIP kernels behind the C ABI of ``include/lightglue_amd.h``.
SURVEY.md §8 f3: the SuperPoint extractor (conv stack, keypoint extraction, descriptor head) runs on the same library
(``lightglue_amd.SuperPoint``).  Other extractors, image I/O and visualisation are out of scope.
"""
from .lightglue import LightGlue  # noqa: F401
from .superpoint import SuperPoint  # noqa: F401
from .parallel import PairShardedMatcher, shard_range  # noqa: F401
from .inflight import InflightMatcher  # noqa: F401
from .glue import batch_to_device, cm_prune, collate_features, extracted_to_image_frame, match_batch, match_pair, prefetch_to_device, rbd  # noqa: F401

__all__ = ["LightGlue", "SuperPoint", "PairShardedMatcher", "InflightMatcher", "shard_range", "match_pair", "match_batch", "collate_features", "extracted_to_image_frame", "rbd", "cm_prune",
           "batch_to_device", "prefetch_to_device"]
__version__ = "0.2.0"
