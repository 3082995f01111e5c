"""Host-side mirror of the reference matcher interface, backed by the HIP engine.

``LightGlue(features=..., **conf).forward({'image0': ..., 'image1': ...})`` keeps the constructor
kwargs, ``state_dict`` names/shapes, input dict and output dict of the reference class
(``lightglue/lightglue.py:321`` ``LightGlue``; ctor :376-437, forward :456-481, output :619-629),
but the module tree here only HOLDS parameters: all arithmetic of the forward path runs in
hand-written gfx950 kernels reached through the C ABI (``include/lightglue_amd.h`` via
``_cabi.py``).  There is no PyTorch / CPU fallback — ``forward`` on a non-GPU tensor, or without the
built library, raises.

Extensions over the reference (documented in DESIGN.md):
  * batches with adaptive depth/width work for B > 1 (every pair stops / prunes independently;
    the reference only defines this for B = 1); ``stop`` is an ``int`` for B = 1 and an
    int64 tensor [B] for B > 1;
  * ``precision`` conf key: "f16x3" (default: every contraction — linear layers, q k^T, P V, similarity — on split-f16
    operands, 3 MFMAs per product; holds index parity / the 1e-3 score bar with the fp32 reference, also for sharp
    attention), "fp32" (exact f32 MFMAs, the parity anchor), "bf16" / "fp16" (one MFMA per product: fast, outside the bar);
  * ``attention_precision`` conf key: None (= ``precision``) or "fp16" together with precision "f16x3": q / k / v as ONE f16
    plane (what the reference's GPU path feeds its fp16 SDPA, ref :119) — ~25 % faster, within the bar only while attention
    is diffuse (it fails the trained-statistics fixtures by 20x, tests/test_gpu_parity.py);
  * ``pruning_min_kpts`` conf key overrides the device-keyed class dict (ref :339-344).
"""
from __future__ import annotations

import ctypes as C
import warnings
from pathlib import Path
from types import SimpleNamespace
from typing import Optional

import numpy as np
import torch
from torch import nn

from . import _cabi


class _Holder(nn.Module):
    """Parameter container; never called."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("lightglue_amd parameter containers are not callable; use LightGlue.forward")


def _ffn(d: int) -> nn.Sequential:
    # names ffn.0 / ffn.1 / ffn.3 as in ref :152-157 (index 2 is the parameter-free GELU)
    return nn.Sequential(nn.Linear(2 * d, 2 * d), nn.LayerNorm(2 * d, elementwise_affine=True), nn.GELU(), nn.Linear(2 * d, d))


def _self_block(d: int) -> nn.Module:  # ref :141-157
    m = _Holder()
    m.Wqkv = nn.Linear(d, 3 * d)
    m.out_proj = nn.Linear(d, d)
    m.ffn = _ffn(d)
    return m


def _cross_block(d: int) -> nn.Module:  # ref :176-192
    m = _Holder()
    m.to_qk = nn.Linear(d, d)
    m.to_v = nn.Linear(d, d)
    m.to_out = nn.Linear(d, d)
    m.ffn = _ffn(d)
    return m


class DeferredMatches:
    """Handle of a forward whose outputs are on their way (LightGlue.forward_deferred)."""

    def __init__(self, done, host_sizes, assemble, buffers=()):
        self._done, self._host, self._assemble, self._out = done, host_sizes, assemble, None
        self.buffers = buffers     # the allocations every output tensor is a view of (InflightMatcher hands THEM to the consumer's stream: 3 - 4 record_stream calls, not one per output)

    def result(self) -> dict:
        if self._out is None:
            self._done.synchronize()
            self._out = self._assemble(self._host.tolist())
            self._assemble = None
        return self._out


class LightGlue(nn.Module):
    default_conf = {
        "name": "lightglue",
        "input_dim": 256,
        "descriptor_dim": 256,
        "add_scale_ori": False,
        "n_layers": 9,
        "num_heads": 4,
        "flash": True,  # kept for interface parity; attention is always the fused HIP kernel
        "mp": False,  # kept for interface parity; see `precision`
        "depth_confidence": 0.95,  # early stopping, disable with -1
        "width_confidence": 0.99,  # point pruning, disable with -1
        "filter_threshold": 0.1,
        "weights": None,
        # ---- lightglue_amd extensions
        "precision": "f16x3",
        "attention_precision": None,
        "pruning_min_kpts": None,  # None -> pruning_keypoint_thresholds (ref :339-344, :658-662)
    }

    # ref :339-344 (values tuned by the reference for RTX 30xx; kept for behavioural parity)
    pruning_keypoint_thresholds = {"cpu": -1, "mps": -1, "cuda": 1024, "flash": 1536}

    required_data_keys = ["image0", "image1"]

    version = "v0.1_arxiv"
    url = "https://github.com/cvg/LightGlue/releases/download/{}/{}_lightglue.pth"

    features = {
        "superpoint": {"weights": "superpoint_lightglue", "input_dim": 256},
        "disk": {"weights": "disk_lightglue", "input_dim": 128},
        "aliked": {"weights": "aliked_lightglue", "input_dim": 128},
        "sift": {"weights": "sift_lightglue", "input_dim": 128, "add_scale_ori": True},
        "doghardnet": {"weights": "doghardnet_lightglue", "input_dim": 128, "add_scale_ori": True},
    }

    def __init__(self, features="superpoint", **conf) -> None:
        super().__init__()
        self.conf = conf = SimpleNamespace(**{**self.default_conf, **conf})
        if features is not None:
            if features not in self.features:
                raise ValueError(f"Unsupported features: {features} not in {{{','.join(self.features)}}}")
            for k, v in self.features[features].items():
                setattr(conf, k, v)
        if conf.precision not in _cabi.LG_PREC:
            raise ValueError(f"precision must be one of {sorted(_cabi.LG_PREC)}")
        if conf.attention_precision not in (None, conf.precision) and not (conf.precision == "f16x3" and conf.attention_precision == "fp16"):
            raise ValueError("attention_precision must be None (= precision), or 'fp16' together with precision 'f16x3'")
        if conf.descriptor_dim != 256 or conf.num_heads != 4:
            raise ValueError("lightglue_amd builds descriptor_dim=256, num_heads=4 (head_dim 64) only")

        d, L = conf.descriptor_dim, conf.n_layers
        self.input_proj = nn.Linear(conf.input_dim, d, bias=True) if conf.input_dim != d else nn.Identity()
        self.posenc = _Holder()
        self.posenc.Wr = nn.Linear(2 + 2 * int(conf.add_scale_ori), d // conf.num_heads // 2, bias=False)
        nn.init.normal_(self.posenc.Wr.weight.data, mean=0, std=1.0)  # ref :74 (gamma = 1)
        layers = []
        for _ in range(L):
            layer = _Holder()
            layer.self_attn = _self_block(d)
            layer.cross_attn = _cross_block(d)
            layers.append(layer)
        self.transformers = nn.ModuleList(layers)
        assign = []
        for _ in range(L):
            a = _Holder()
            a.matchability = nn.Linear(d, 1, bias=True)
            a.final_proj = nn.Linear(d, d, bias=True)
            assign.append(a)
        self.log_assignment = nn.ModuleList(assign)
        toks = []
        for _ in range(L - 1):
            t = _Holder()
            t.token = nn.Sequential(nn.Linear(d, 1), nn.Sigmoid())
            toks.append(t)
        self.token_confidence = nn.ModuleList(toks)
        self.register_buffer("confidence_thresholds", torch.Tensor([self.confidence_threshold(i) for i in range(L)]))

        state_dict = None
        if features is not None:
            state_dict = self._find_pretrained(features, conf.weights)
        elif conf.weights is not None:
            path = Path(__file__).parent / "weights" / f"{conf.weights}.pth"
            state_dict = torch.load(str(path), map_location="cpu")
        if state_dict:
            self.load_state_dict(self.rename_legacy_keys(state_dict, L), strict=False)

        self.static_lengths = None
        # extension: also return the full [B, M+1, N+1] log-assignment incl. dustbins (ref :265-277)
        self.return_log_assignment = False
        # extension: range guard (include/lightglue_amd.h LG_FLAG_CHECK_FINITE).  The split-f16 arithmetic needs |x| < 65504 for the residual stream
        # and q / k / v (like the reference's own fp16 mode); with the guard on, a forward whose values leave that range raises instead of returning
        # inf / NaN scores.  "first" (default, round 6): on for the FIRST forward after the weights changed (load_state_dict, .to(), in-place edits) —
        # a checkpoint / input scale that does not fit the operand range shows on the first batch —, off afterwards; True: every forward; False: never
        # (one compare per value in the tail and projection epilogues: not measurable, LAB_NOTES round 5 call h).
        self.check_finite = "first"
        self._guard_pending = True
        self.track_inplace_weight_edits = True     # see _sync_weights
        self._engine = None  # (handle, device_index, config signature)
        self._weights_sig = None
        self._plist = None
        self.requires_grad_(False)

    # ------------------------------------------------------------------ weights
    @staticmethod
    def rename_legacy_keys(state_dict: dict, n_layers: int) -> dict:
        """Released checkpoints name the blocks `self_attn.{i}.*` / `cross_attn.{i}.*`; the module tree uses
        `transformers.{i}.self_attn.*` / `.cross_attn.*` (ref :427-434).  Already-renamed keys pass through."""
        out = {}
        for key, val in state_dict.items():
            for i in range(n_layers):
                for kind in ("self_attn", "cross_attn"):
                    old = f"{kind}.{i}."
                    if key.startswith(old):
                        key = f"transformers.{i}.{kind}." + key[len(old):]
            out[key] = val
        return out

    def _find_pretrained(self, features: str, weights: str):
        """The reference downloads `{features}_lightglue.pth` (ref :416-421).  Look for the same file
        where torch.hub would have cached it or in ./weights; download only if the network allows."""
        fname = f"{weights}_{self.version.replace('.', '-')}.pth"
        candidates = [Path(torch.hub.get_dir()) / "checkpoints" / fname, Path(__file__).parent / "weights" / fname,
                      Path(__file__).parent / "weights" / f"{weights}.pth"]
        for c in candidates:
            if c.exists():
                return torch.load(str(c), map_location="cpu")
        try:
            return torch.hub.load_state_dict_from_url(self.url.format(self.version, features), file_name=fname, map_location="cpu")
        except Exception as exc:  # offline
            raise RuntimeError(
                f"pretrained weights '{fname}' not found in {[str(c.parent) for c in candidates]} and download failed "
                f"({exc}); place the file there or construct LightGlue(features=None, input_dim=...) and load_state_dict()."
            ) from exc

    def confidence_threshold(self, layer_index: int) -> float:
        """ref :631-634"""
        threshold = 0.8 + 0.1 * np.exp(-4.0 * layer_index / self.conf.n_layers)
        return np.clip(threshold, 0, 1)

    def pruning_min_kpts(self, device: torch.device) -> int:
        """ref :658-662.  ROCm devices report type 'cuda'; the fused attention kernel plays the role of
        the reference's flash path."""
        if self.conf.pruning_min_kpts is not None:
            return int(self.conf.pruning_min_kpts)
        if self.conf.flash and device.type == "cuda":
            return self.pruning_keypoint_thresholds["flash"]
        return self.pruning_keypoint_thresholds[device.type]

    def compile(self, mode="reduce-overhead", static_lengths=[256, 512, 768, 1024, 1280, 1536]):
        """Interface parity with ref :439-454.  Nothing is traced here: the HIP kernels take ragged
        lengths natively, so no padding happens.  As in the reference (ref :529), point pruning is
        disabled for inputs that fall inside the static-length range."""
        if self.conf.width_confidence != -1:
            warnings.warn("Point pruning is partially disabled for compiled forward.", stacklevel=2)
        self.static_lengths = list(static_lengths)

    # ------------------------------------------------------------------ engine plumbing
    def _config_sig(self, device: torch.device):
        c = self.conf
        return (device.index if device.index is not None else torch.cuda.current_device(), c.input_dim, c.n_layers, bool(c.add_scale_ori),
                float(c.depth_confidence), float(c.width_confidence), float(c.filter_threshold), self.pruning_min_kpts(device), c.precision, c.attention_precision)

    def _get_engine(self, device: torch.device):
        lib = _cabi.load()
        sig = self._config_sig(device)
        if self._engine is not None and self._engine[1] == sig:
            return self._engine[0]
        self._drop_engine()
        c = self.conf
        cfg = _cabi.LgConfig(c.input_dim, c.descriptor_dim, c.n_layers, c.num_heads, int(bool(c.add_scale_ori)),
                             float(c.depth_confidence), float(c.width_confidence), float(c.filter_threshold),
                             int(sig[7]), _cabi.LG_PREC[c.precision], -1 if c.attention_precision is None else _cabi.LG_PREC[c.attention_precision])
        handle = C.c_void_p()
        with torch.cuda.device(device):
            _cabi.check(lib.lg_engine_create(C.byref(cfg), C.byref(handle)))
        self._engine = (handle, sig)
        self._weights_sig = None
        return handle

    def _drop_engine(self):
        eng = self.__dict__.get("_engine")
        if eng is not None:
            self.__dict__["_engine"] = None   # not via nn.Module.__setattr__: this also runs at interpreter shutdown
            try:
                _cabi.load().lg_engine_destroy(eng[0])
            except Exception:  # pragma: no cover
                pass

    def __del__(self):  # pragma: no cover
        self._drop_engine()

    def _sync_weights(self, handle, device):
        """Re-pack and upload whenever any parameter changed (load_state_dict, .to(), in-place edit)."""
        # cheap change detector (runs every forward): in-place edits and load_state_dict bump `_version`,
        # .to()/.cuda() move the storage.  Walking the 251 parameters costs 10-30 us of host time per forward (3 % of a B = 1, N = 512 forward):
        # `track_inplace_weight_edits = False` skips the walk — load_state_dict / .to() / refresh_weights() still trigger the re-pack
        if not self.track_inplace_weight_edits and self._weights_sig is not None:
            return
        plist = self._plist
        if plist is None:
            plist = self._plist = list(self.parameters())
        ver = 0
        for p in plist:
            ver += p._version
        sig = (ver, plist[0].data_ptr(), plist[-1].data_ptr(), len(plist))
        if sig == self._weights_sig:
            return   # (edits through `.data` bypass the version counter: call refresh_weights() after those)
        lib = _cabi.load()
        keep = []
        for name, p in self.state_dict().items():
            if name == "confidence_thresholds":
                continue
            arr = p.detach().to("cpu", torch.float32).contiguous().numpy()
            keep.append(arr)
            shape = (C.c_int64 * arr.ndim)(*arr.shape)
            _cabi.check(lib.lg_engine_set_weight(handle, name.encode(), arr.ctypes.data_as(C.c_void_p), shape, arr.ndim))
        with torch.cuda.device(device):
            _cabi.check(lib.lg_engine_finalize_weights(handle))
        self._weights_sig = sig
        self._guard_pending = True      # check_finite == "first": the next forward runs with the range guard

    def refresh_weights(self):
        """Force a re-pack + upload of the parameters at the next forward (needed after edits the version counters do
        not see, e.g. ``p.data.copy_()`` or module surgery)."""
        self._weights_sig = None
        self._plist = None

    def _apply(self, fn, *a, **kw):                      # .to() / .cuda() / .half() ...
        out = super()._apply(fn, *a, **kw)
        self.refresh_weights()
        return out

    def load_state_dict(self, *a, **kw):                 # incl. assign=True, which replaces the Parameter objects
        out = super().load_state_dict(*a, **kw)
        self.refresh_weights()
        return out

    def __getstate__(self):                              # copy.deepcopy / pickle / torch.save(model): the engine handle
        st = dict(self.__dict__)                         # is process-local; the copy creates its own lazily
        st["_engine"] = None; st["_plist"] = None; st["_weights_sig"] = None
        return st

    def reserve(self, batch: int, n0: int, n1: int, device=None):
        """Pre-size the engine workspace (avoids a synchronising re-allocation inside forward)."""
        device = torch.device(device if device is not None else "cuda")
        h = self._get_engine(device)
        self._sync_weights(h, device)
        with torch.cuda.device(device):
            _cabi.check(_cabi.load().lg_engine_reserve(h, batch, n0, n1))

    # ------------------------------------------------------------------ forward
    def will_prune(self, m: int, n: int) -> bool:
        """Whether a forward on [B, m] x [B, n] keypoints runs with point pruning (ref :529-531: width_confidence > 0 and not the static-length mode) —
        i.e. whether prune0 / prune1 come back as int64 counters or as the float fill (ref :616-617)."""
        do_compile = bool(self.static_lengths) and max(m, n) <= max(self.static_lengths)
        return self.conf.width_confidence > 0 and not do_compile

    def _carve_plan(self, b, m, n, kmax, pruning):
        """Sizes and 4-element-aligned offsets of the pieces of the three output allocations (int32 / fp32 / int64), cached per shape."""
        key = (b, m, n, pruning)
        cache = self.__dict__.setdefault("_plans", {})
        plan = cache.get(key)
        if plan is None:
            def carve(sizes):
                off = [0]
                for x in sizes:
                    off.append(off[-1] + ((x + 3) & ~3))
                return off
            isz = [b * m, b * n, b * kmax * 2, b * m if pruning else 0, b * n if pruning else 0, 3 * b]
            fsz = [b * m, b * n, b * kmax, 0 if pruning else b * m, 0 if pruning else b * n]
            lsz = [b * m, b * n, b * kmax * 2, b, b * m if pruning else 0, b * n if pruning else 0]
            if len(cache) > 256:
                cache.clear()
            plan = cache[key] = (isz, carve(isz), fsz, carve(fsz), lsz, carve(lsz))
        return plan

    def forward_raw(self, data: dict, wire: Optional[torch.Tensor] = None) -> dict:
        """The forward without output widening, ragged lists or the host synchronisation: int32 `matches0/1` [B, M|N], fp32
        `matching_scores0/1`, int32 `stop` and `status` [B] — views of the engine's own output buffers, valid on the current stream.
        `wire`: optional int32 [pairs >= B][>= 3M + 3N + 2] buffer; the engine's last kernel then also packs one row per pair,
        [matches0 | score0 bits | matches1 | score1 bits | stop | status | prune0 | prune1] — the send buffer of `PairShardedMatcher`
        (`lg_unpack_wire` rebuilds the output dict's tensors on the receiving side)."""
        return self.forward(data, _raw=True if wire is None else wire)

    @staticmethod
    def _raise_on_status(status) -> None:
        bad = [(i, int(c)) for i, c in enumerate(status) if int(c) != _cabi.LG_OK]
        if bad:
            what = {_cabi.LG_ERR_RANGE: "values outside the f16 operand range (|x| >= 65504, inf or NaN; LG_ERR_RANGE)",
                    _cabi.LG_ERR_DEVICE: "an internal device-side wait expired (LG_ERR_DEVICE)"}
            raise _cabi.LightGlueAmdError("; ".join(f"pair {i}: {what.get(c, c)}" for i, c in bad[:8]))

    def forward_deferred(self, data: dict) -> "DeferredMatches":
        """The forward enqueued WITHOUT its host synchronisation: `.result()` of the returned handle waits for this forward
        only (an event), then builds the same dict as `forward`.  A caller that issues forward i + 1 before it asks for the
        result of forward i keeps the GPU busy while Python assembles the ragged lists (bench.py does; the reference's
        synchronous `forward` cannot)."""
        return self.forward(data, _defer=True)

    @torch.no_grad()
    def forward(self, data: dict, _raw: bool = False, _defer: bool = False) -> dict:
        """Match keypoints and descriptors between two images (same dict contract as ref :456-481).

        Input (dict):  image0/image1: {keypoints [B,N,2], descriptors [B,N,D], image_size [B,2] (optional),
                                       scales/oris [B,N] iff add_scale_ori,
                                       num_keypoints [B] int (optional, extension: ragged batch — pair b uses only
                                       its first num_keypoints[b] rows and behaves exactly like a separate B=1
                                       call on them; padding rows come back as -1 / 0, see glue.collate_features)}
        Output (dict): matches0 [B,M] int64, matching_scores0 [B,M], matches1 [B,N], matching_scores1 [B,N],
                       matches List[[S_i,2]], scores List[[S_i]], stop, prune0 [B,M], prune1 [B,N]
        """
        for key in self.required_data_keys:
            assert key in data, f"Missing key {key} in data"
        d0, d1 = data["image0"], data["image1"]
        kpts0, kpts1 = d0["keypoints"], d1["keypoints"]
        device = kpts0.device
        if device.type != "cuda":
            raise RuntimeError("lightglue_amd runs on MI355X (ROCm device type 'cuda') only; there is no CPU fallback. "
                               f"Got keypoints on {device}.")
        b, m, _ = kpts0.shape
        b, n, _ = kpts1.shape
        conf = self.conf
        def f32(t):   # only the data pointer is handed on: tensors that already qualify pass through untouched
            if t.dtype is torch.float32 and t.device == device and t.is_contiguous():
                return t
            return t.detach().to(device=device, dtype=torch.float32).contiguous()

        k0, k1 = f32(kpts0), f32(kpts1)
        desc0, desc1 = f32(d0["descriptors"]), f32(d1["descriptors"])
        assert desc0.shape[-1] == conf.input_dim
        assert desc1.shape[-1] == conf.input_dim
        size0, size1 = d0.get("image_size"), d1.get("image_size")

        def as_size(s):
            if s is None:
                return None
            if not isinstance(s, torch.Tensor):
                s = torch.tensor(s, dtype=torch.float32)
            s = f32(s).reshape(-1, 2)
            assert s.shape[0] in (1, b), f"image_size must have shape [2], [1, 2] or [{b}, 2]"
            return s if s.shape[0] == b else s.expand(b, 2).contiguous()   # the reference broadcasts (ref :35-42)

        size0, size1 = as_size(size0), as_size(size1)
        extra = [None] * 4
        if conf.add_scale_ori:
            extra = [f32(d0["scales"]), f32(d0["oris"]), f32(d1["scales"]), f32(d1["oris"])]

        def as_count(c, nmax):
            if c is None:
                return None
            c = torch.as_tensor(c).detach().to(device=device, dtype=torch.int32).contiguous()
            assert c.shape == (b,), f"num_keypoints must have shape [{b}]"
            return c.clamp(0, nmax)

        num0, num1 = as_count(d0.get("num_keypoints"), m), as_count(d1.get("num_keypoints"), n)
        ragged = num0 is not None or num1 is not None

        do_early_stop = conf.depth_confidence > 0
        do_point_pruning = self.will_prune(m, n)

        # ---- outputs: ONE int32, ONE int64 and ONE fp32 allocation, carved into the tensors of the C ABI (16-byte aligned pieces).  The engine's
        # last kernel writes the reference's dtypes itself (int64 indices / stop / prune counters, float prune0/1 without pruning, ref :616-629;
        # round-5 extension of lg_forward_io): no framework kernel runs between or behind the engine's launches.
        kmax = min(m, n)
        plan = self._carve_plan(b, m, n, kmax, do_point_pruning)      # offsets of the pieces (cached per shape: the host path of a B = 1 forward is ~0.3 ms, and this was a tenth of it)
        isz, ioff, fsz, foff, lsz, loff = plan
        ibuf = torch.empty((ioff[-1],), device=device, dtype=torch.int32)

        def ipiece(buf, off, sz, k, *shape):
            return buf[off[k]: off[k] + sz[k]].view(shape)
        m0, m1, mlist = ipiece(ibuf, ioff, isz, 0, b, m), ipiece(ibuf, ioff, isz, 1, b, n), ipiece(ibuf, ioff, isz, 2, b, kmax, 2)
        prune0_i32 = ipiece(ibuf, ioff, isz, 3, b, m) if do_point_pruning else None
        prune1_i32 = ipiece(ibuf, ioff, isz, 4, b, n) if do_point_pruning else None
        stop_nm = ipiece(ibuf, ioff, isz, 5, 3, b)              # [0] = stop, [1] = n_matches, [2] = status (LG_OK / LG_ERR_RANGE / LG_ERR_DEVICE)
        fbuf = torch.empty((foff[-1],), device=device, dtype=torch.float32)
        ms0, ms1 = ipiece(fbuf, foff, fsz, 0, b, m), ipiece(fbuf, foff, fsz, 1, b, n)
        mscore_list = ipiece(fbuf, foff, fsz, 2, b, kmax)
        raw_mode = _raw is not False
        m0_64 = m1_64 = mlist64 = stop64 = prune0 = prune1 = lbuf = None
        if not raw_mode:
            lbuf = torch.empty((loff[-1],), device=device, dtype=torch.int64)
            m0_64, m1_64 = ipiece(lbuf, loff, lsz, 0, b, m), ipiece(lbuf, loff, lsz, 1, b, n)
            mlist64, stop64 = ipiece(lbuf, loff, lsz, 2, b, kmax, 2), ipiece(lbuf, loff, lsz, 3, b)
            if do_point_pruning:
                prune0, prune1 = ipiece(lbuf, loff, lsz, 4, b, m), ipiece(lbuf, loff, lsz, 5, b, n)
            else:   # ref :616-617 (padding rows of a ragged batch: 0)
                prune0, prune1 = ipiece(fbuf, foff, fsz, 3, b, m), ipiece(fbuf, foff, fsz, 4, b, n)
        wire = _raw if torch.is_tensor(_raw) else None          # PairShardedMatcher: [pairs >= b][>= 3m + 3n + 2] int32 rows the engine packs itself

        log_assignment = None
        if self.return_log_assignment and m > 0 and n > 0:
            log_assignment = torch.empty((b, m + 1, n + 1), device=device, dtype=torch.float32)
        handle = self._get_engine(device)
        self._sync_weights(handle, device)
        ptr = lambda t: None if t is None or t.numel() == 0 else t.data_ptr()
        guard = self.check_finite is True or (self.check_finite == "first" and self._guard_pending)
        self._guard_pending = False
        if wire is not None:   # the engine's last kernel writes b rows of 3m + 3n + 2 int32 into it: a short or mistyped buffer would be an out-of-bounds device write
            if not (wire.dtype is torch.int32 and wire.device == device and wire.dim() == 2 and wire.shape[0] >= b
                    and wire.shape[1] >= _cabi.wire_width(m, n) and (wire.shape[1] == 1 or wire.stride(1) == 1)):
                raise ValueError(f"wire must be an int32 [>= {b}, >= {_cabi.wire_width(m, n)}] tensor on {device} with unit column stride, got "
                                 f"{wire.dtype} {tuple(wire.shape)} stride {tuple(wire.stride())} on {wire.device}")
        flags = _cabi.LG_FLAG_EXT | (0 if do_point_pruning or conf.width_confidence <= 0 else _cabi.LG_FLAG_NO_PRUNING) | (_cabi.LG_FLAG_CHECK_FINITE if guard else 0)
        io = _cabi.LgForwardIO(
            b, m, n, flags,
            ptr(k0), ptr(k1), ptr(desc0), ptr(desc1), ptr(size0), ptr(size1),
            ptr(extra[0]), ptr(extra[1]), ptr(extra[2]), ptr(extra[3]),
            ptr(m0), ptr(m1), ptr(ms0), ptr(ms1), stop_nm[0].data_ptr(), ptr(prune0_i32), ptr(prune1_i32),
            ptr(mlist), ptr(mscore_list), stop_nm[1].data_ptr(), ptr(num0), ptr(num1), ptr(log_assignment),
            ptr(m0_64), ptr(m1_64), ptr(mlist64), ptr(stop64),
            ptr(prune0) if (do_point_pruning and not raw_mode) else None, ptr(prune1) if (do_point_pruning and not raw_mode) else None,
            None if (do_point_pruning or raw_mode) else ptr(prune0), None if (do_point_pruning or raw_mode) else ptr(prune1),
            ptr(wire), 0 if wire is None else wire.stride(0), stop_nm[2].data_ptr())
        with torch.cuda.device(device):
            cur_stream = torch.cuda.current_stream(device)
            _cabi.check(_cabi.load().lg_engine_forward(handle, C.byref(io), C.c_void_p(cur_stream.cuda_stream)))

        if getattr(self, "_debug_step", -1) >= 0:  # test tap: the pipeline stopped early, outputs are not written
            torch.cuda.synchronize(device)
            return None
        if raw_mode:
            return {"matches0": m0, "matches1": m1, "matching_scores0": ms0, "matching_scores1": ms1, "stop": stop_nm[0], "status": stop_nm[2],
                    "pruning": do_point_pruning}   # "pruning": the wire row's prune block holds int counters (else the float fill's bits)
        # ---- output assembly (ref :593-629): the engine has written every fixed-shape tensor; only the ragged lists need the host.

        def assemble(host):   # host = [[stop per pair], [matches per pair], [status per pair]]
            self._raise_on_status(host[2])
            counts = host[1]
            matches = [row[:c] for row, c in zip(mlist64.unbind(0), counts)]
            mscores = [row[:c] for row, c in zip(mscore_list.unbind(0), counts)]
            stop_out = int(host[0][0]) if b == 1 else stop64
            extra_out = {} if log_assignment is None else {"log_assignment": log_assignment}
            return {
                **extra_out,
                "matches0": m0_64,
                "matches1": m1_64,
                "matching_scores0": ms0,
                "matching_scores1": ms1,
                "stop": stop_out,
                "matches": matches,
                "scores": mscores,
                "prune0": prune0,
                "prune1": prune1,
            }

        if _defer:   # the sizes travel to pinned host memory behind an event; nothing waits here
            hbuf = torch.empty((3, b), dtype=torch.int32, pin_memory=True)
            hbuf.copy_(stop_nm, non_blocking=True)
            done = torch.cuda.Event()
            done.record(cur_stream)
            return DeferredMatches(done, hbuf, assemble, tuple(t for t in (ibuf, fbuf, lbuf, log_assignment) if t is not None))
        return assemble(stop_nm.tolist())  # THE host sync of the forward: the ragged lists need their sizes (and B = 1 its `stop`)

    def set_option(self, key: str, value: int, device="cuda"):
        """Engine options (include/lightglue_amd.h lg_engine_set_option), e.g. ("fused_tail", 0)."""
        h = self._get_engine(torch.device(device))
        _cabi.check(_cabi.load().lg_engine_set_option(h, key.encode(), int(value)))

    # ------------------------------------------------------------------ per-kernel timing (HIP events)
    def profile(self, enable: bool, device="cuda", only: str = None):
        """HIP-event timing per kernel class on the launch stream; `only` = one class name (see profile_read) to
        bracket just that class and leave every other launch untouched."""
        lib = _cabi.load()
        h = self._get_engine(torch.device(device))
        cls = -1
        if only is not None:
            names = [lib.lg_profile_class_name(i).decode() for i in range(lib.lg_profile_num_classes())]
            cls = names.index(only)
        _cabi.check(lib.lg_engine_set_option(h, b"profile_only", cls))
        _cabi.check(lib.lg_engine_profile_enable(h, int(bool(enable))))

    def profile_read(self, device="cuda") -> dict:
        """{kernel class: (total ms, launch sites)} accumulated since the last read."""
        h = self._get_engine(torch.device(device))
        lib = _cabi.load()
        n = lib.lg_profile_num_classes()
        ms, cnt = (C.c_double * n)(), (C.c_int64 * n)()
        _cabi.check(lib.lg_engine_profile_read(h, ms, cnt, n))
        return {lib.lg_profile_class_name(i).decode(): (ms[i], cnt[i]) for i in range(n)}

    # ------------------------------------------------------------------ test / profiling taps
    def debug_stop_after(self, step: int, device="cuda"):
        h = self._get_engine(torch.device(device))
        _cabi.check(_cabi.load().lg_engine_debug_stop_after(h, int(step)))
        self._debug_step = int(step)

    def debug_read(self, name: str, dtype=np.float32, device="cuda") -> np.ndarray:
        h = self._get_engine(torch.device(device))
        lib = _cabi.load()
        nbytes = C.c_int64()
        _cabi.check(lib.lg_engine_debug_read(h, name.encode(), None, 0, C.byref(nbytes)))
        out = np.empty(nbytes.value // np.dtype(dtype).itemsize, dtype=dtype)
        _cabi.check(lib.lg_engine_debug_read(h, name.encode(), out.ctypes.data_as(C.c_void_p), out.nbytes, C.byref(nbytes)))
        return out

    def debug_caps(self, device="cuda"):
        h = self._get_engine(torch.device(device))
        c0, c1 = C.c_int32(), C.c_int32()
        _cabi.check(_cabi.load().lg_engine_debug_caps(h, C.byref(c0), C.byref(c1)))
        return c0.value, c1.value
