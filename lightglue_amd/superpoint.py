"""SuperPoint on the MI355X: the extractor that produces the matcher's `(B, N, 256)` inputs.

Same module tree / parameter names as the reference class (`lightglue/superpoint.py:98-145`: conv1a ... convDb), so the
released `superpoint_v1.pth` loads with `load_state_dict` unchanged, and the same `forward({"image": ...})` contract
(`:147-232`).  Everything runs in `lightglue_amd/csrc/`: the conv stack as exact-fp32 MFMA implicit GEMMs
(`lg_sp_encoder.hip`, `lg_sp_encode`), keypoint extraction (`lg_sp_detect`: NMS, borders, threshold, top-k) and the
descriptor head (`lg_sp_sample_descriptors`).  No CPU fallback.  Image IO / resizing (`ImagePreprocessor`, kornia) stay out of
scope: `extract()` takes an already sized image; `glue.extracted_to_image_frame` maps keypoints back if the caller resized."""
from __future__ import annotations

import ctypes as C
from types import SimpleNamespace
from typing import Optional

import torch
from torch import nn

from . import _cabi
from .superpoint_head import descriptor_head, detect_keypoints

_LAYERS = ("conv1a", "conv1b", "conv2a", "conv2b", "conv3a", "conv3b", "conv4a", "conv4b", "convPa", "convPb", "convDa", "convDb")


class SuperPoint(nn.Module):
    default_conf = {"descriptor_dim": 256, "nms_radius": 4, "max_num_keypoints": None, "detection_threshold": 0.0005,
                    "remove_borders": 4,   # ref :108-114
                    # extension: "fp32" = exact f32 MFMA convolutions (default); "f16x3" = split-f16 operands, three f16 MFMAs per product, fp32 accumulation (22 operand bits:
                    # below an fp32 convolution's summation-order noise, cf. the reference's own GPU default of TF32 convolutions) at several times the f32 MFMA rate
                    "conv_precision": "fp32"}
    required_data_keys = ["image"]

    def __init__(self, weights: Optional[dict] = None, **conf):
        """`weights`: a state dict with the reference's names (e.g. torch.load('superpoint_v1.pth')); None = PyTorch's default
        init (the released file is network-only, ref :143-144, and there is no network here)."""
        super().__init__()
        self.conf = SimpleNamespace(**{**self.default_conf, **conf})
        c1, c2, c3, c4, c5 = 64, 64, 128, 128, 256
        self.conv1a = nn.Conv2d(1, c1, 3, 1, 1); self.conv1b = nn.Conv2d(c1, c1, 3, 1, 1)
        self.conv2a = nn.Conv2d(c1, c2, 3, 1, 1); self.conv2b = nn.Conv2d(c2, c2, 3, 1, 1)
        self.conv3a = nn.Conv2d(c2, c3, 3, 1, 1); self.conv3b = nn.Conv2d(c3, c3, 3, 1, 1)
        self.conv4a = nn.Conv2d(c3, c4, 3, 1, 1); self.conv4b = nn.Conv2d(c4, c4, 3, 1, 1)
        self.convPa = nn.Conv2d(c4, c5, 3, 1, 1); self.convPb = nn.Conv2d(c5, 65, 1, 1, 0)
        self.convDa = nn.Conv2d(c4, c5, 3, 1, 1); self.convDb = nn.Conv2d(c5, self.conf.descriptor_dim, 1, 1, 0)
        if self.conf.descriptor_dim != 256:
            raise ValueError("lightglue_amd builds descriptor_dim = 256 only")
        if self.conf.conv_precision not in ("fp32", "f16x3"):
            raise ValueError("conv_precision must be 'fp32' or 'f16x3'")
        if self.conf.max_num_keypoints is not None and self.conf.max_num_keypoints <= 0:
            raise ValueError("max_num_keypoints must be positive or None")   # ref :146-147
        if weights is not None:
            self.load_state_dict(weights)
        self._packed = None   # (signature, [24 device tensors])

    # ------------------------------------------------------------------ weights -> kernel layout
    def _params(self, device):
        split = self.conf.conv_precision == "f16x3"
        sig = (str(device), split) + tuple((p._version, p.data_ptr()) for p in self.parameters())
        if self._packed is not None and self._packed[0] == sig:
            return self._packed[1]
        lib = _cabi.load()
        out = []
        with torch.cuda.device(device):
            stream = torch.cuda.current_stream(device).cuda_stream
            for name in _LAYERS:
                conv = getattr(self, name)
                w = conv.weight.detach().to(device=device, dtype=torch.float32).contiguous()
                cout, cin, k, _ = w.shape
                dst = torch.empty(w.numel(), device=device, dtype=torch.float32)
                pack = lib.lg_sp_pack_conv_weight_split if (split and name != "conv1a") else lib.lg_sp_pack_conv_weight   # (the split form fills the same bytes: two f16 planes)
                _cabi.check(pack(w.data_ptr(), cout, cin, k, dst.data_ptr(), C.c_void_p(stream)))
                out += [dst, conv.bias.detach().to(device=device, dtype=torch.float32).contiguous()]
            torch.cuda.current_stream(device).synchronize()   # `w` temporaries may be freed after this
        self._packed = (sig, out)
        return out

    # ------------------------------------------------------------------ conv stack
    @torch.no_grad()
    def encode(self, image: torch.Tensor):
        """image [B, 1, H, W] (or [B, 3, H, W]: converted like kornia's rgb_to_grayscale, ref :155-156) ->
        (scores [B, H // 8 * 8, W // 8 * 8], dense raw descriptors [B, 256, H // 8, W // 8]) — ref :159-184 and :213-214.  Any
        H, W >= 8: the three 2 x 2 max-pools floor like the reference's nn.MaxPool2d, so the score map is cropped to whole 8 x 8 cells."""
        if image.device.type != "cuda":
            raise RuntimeError("lightglue_amd.SuperPoint runs on MI355X (ROCm device type 'cuda') only; there is no CPU fallback. "
                               f"Got an image on {image.device}.")
        if image.shape[1] == 3:
            r, g, b = image[:, 0:1], image[:, 1:2], image[:, 2:3]
            image = 0.299 * r + 0.587 * g + 0.114 * b
        assert image.dim() == 4 and image.shape[1] == 1, "image must be [B, 1|3, H, W]"
        device = image.device
        image = image.detach().to(dtype=torch.float32).contiguous()
        bsz, _, h, w = image.shape
        assert h >= 8 and w >= 8, "image must be at least 8 x 8"
        lib = _cabi.load()
        params = self._params(device)
        arr = (C.c_void_p * 24)(*[t.data_ptr() for t in params])
        nbytes = lib.lg_sp_encode_workspace_bytes(bsz, h, w)
        work = torch.empty((nbytes,), device=device, dtype=torch.uint8)
        scores = torch.empty((bsz, h // 8 * 8, w // 8 * 8), device=device, dtype=torch.float32)
        dense = torch.empty((bsz, 256, h // 8, w // 8), device=device, dtype=torch.float32)
        with torch.cuda.device(device):
            stream = torch.cuda.current_stream(device).cuda_stream
            encode = lib.lg_sp_encode_split if self.conf.conv_precision == "f16x3" else lib.lg_sp_encode
            _cabi.check(encode(image.data_ptr(), bsz, h, w, arr, work.data_ptr(), nbytes, scores.data_ptr(), dense.data_ptr(), C.c_void_p(stream)))
        return scores, dense

    # ------------------------------------------------------------------ the reference's forward
    @torch.no_grad()
    def forward(self, data: dict) -> dict:
        """ref :147-232.  Returns keypoints [B, N, 2] (x, y), keypoint_scores [B, N], descriptors [B, N, 256] and — extension for
        ragged batches — num_keypoints [B] (rows beyond an image's count are padding; with max_num_keypoints and enough
        detections every image has exactly that many, as the reference's torch.stack requires)."""
        for key in self.required_data_keys:
            assert key in data, f"Missing key {key} in data"
        c = self.conf
        scores, dense = self.encode(data["image"])
        kpts, kscores, counts = detect_keypoints(scores, c.nms_radius, c.remove_borders, c.detection_threshold, c.max_num_keypoints)
        nmax = int(counts.max().item()) if counts.numel() else 0
        kpts, kscores = kpts[:, :nmax].contiguous(), kscores[:, :nmax].contiguous()
        if bool((counts < nmax).any()):   # ragged batch: rows beyond an image's count are padding — zero them (descriptor_head does the same)
            live = torch.arange(nmax, device=counts.device)[None, :] < counts[:, None]
            kpts = kpts * live[..., None]; kscores = kscores * live
        desc = descriptor_head(kpts, dense, 8, counts)
        return {"keypoints": kpts, "keypoint_scores": kscores, "descriptors": desc, "num_keypoints": counts}   # counts: consumed by LightGlue.forward

    @torch.no_grad()
    def extract(self, img: torch.Tensor, **conf) -> dict:
        """ref utils.py:136-147 WITHOUT the resize step: the reference's Extractor.extract first resizes the long side to 1024
        (ImagePreprocessor, kornia — out of scope here); this method uses the image at its own size, so on the same file it
        detects at a different scale than the reference unless the caller resizes first.  scales = 1, the keypoints already live
        in the image's pixel frame; `image_size` = (w, h) is attached for the matcher."""
        if img.dim() == 3:
            img = img[None]
        assert img.dim() == 4 and img.shape[0] == 1
        feats = self.forward({"image": img})
        h, w = img.shape[-2:]
        feats["image_size"] = torch.tensor([[w, h]], dtype=torch.float32, device=img.device)
        return feats
