#!/usr/bin/env python3
"""Time the SuperPoint descriptor head (lg_superpoint.hip) against its HBM roofline.
Workload: B images of 1024x768 (descriptor map 96x128x256) with N keypoints each.
Algorithmic bytes: dense pass 2*4*256*h*w per image, sampling 5 KB per keypoint."""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from lightglue_amd import superpoint_head as H  # noqa: E402

B, h, w, N = (int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (8, 96, 128, 2048)))
g = torch.Generator(device="cuda").manual_seed(0)
dense = torch.randn(B, 256, h, w, device="cuda", generator=g)
kp = torch.rand(B, N, 2, device="cuda", generator=g) * torch.tensor([w * 8.0, h * 8.0], device="cuda")
for _ in range(5):
    H.descriptor_head(kp, dense)
torch.cuda.synchronize()
reps = 50
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    H.descriptor_head(kp, dense)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
bytes_alg = B * (2 * 4 * 256 * h * w + N * 5 * 1024)
print(json.dumps({"workload": f"B={B} map={h}x{w}x256 N={N}", "ms": ms, "images_per_s": B / ms * 1e3,
                  "algorithmic_GB": bytes_alg / 1e9, "achieved_GBps": bytes_alg / ms / 1e6, "frac_of_8TBps": bytes_alg / ms / 1e6 / 8000}))

# keypoint extraction on the matching score maps (8 x 768 x 1024), top-2048
smap = torch.rand(B, h * 8, w * 8, device="cuda", generator=g) ** 6
for _ in range(3):
    H.detect_keypoints(smap, max_num_keypoints=2048)
torch.cuda.synchronize()
e0.record()
for _ in range(20):
    H.detect_keypoints(smap, max_num_keypoints=2048)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(json.dumps({"workload": f"detect_keypoints B={B} map={h * 8}x{w * 8} top-2048", "ms": ms, "images_per_s": B / ms * 1e3,
                  "score_map_GB": B * h * w * 64 * 4 / 1e9}))
