"""CPU oracle for the LightGlue matcher forward path.  TEST INFRASTRUCTURE ONLY.

This file is a plain-numpy restatement of the algorithm in the reference file
``lightglue/lightglue.py`` (cvg/LightGlue).  It exists so that the HIP path in
``lightglue_amd`` can be checked for parity; it is NOT part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it.  The product path never routes through it.

Pinning status: the reference ships no tests / golden vectors, so
the oracle is pinned against *outputs of the reference itself*:
``tools/make_golden.py`` imports the unmodified reference module in the build
container, runs it on seeded inputs/weights and stores the results under
``tests/golden/``; ``tests/test_oracle_golden.py`` replays them through this
file (and, when ``/root/reference`` is present, compares live as well).

Every function cites the reference lines it follows (``ref :A-B`` means
``lightglue/lightglue.py`` lines A..B).

``dtype`` may be ``np.float32`` (the reference's CPU arithmetic) or
``np.float64`` (a "truth" run used to decide which of two fp32 results is the
closer one when they disagree on a near-tie).  ``quant`` rounds both operands of
a contraction to bf16 / fp16 / split-bf16 before an exact product/sum; it emulates
a 16-bit-operand / fp32-accumulate matrix core and is used only to derive the
tolerances written in the tests (see ``_Ctx``).
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import Dict, Optional

import numpy as np

try:  # exact erf for nn.GELU() (approximate="none")
    from scipy.special import erf as _erf
except Exception:  # pragma: no cover - scipy is in the image; keep a slow fallback
    _erf = np.vectorize(math.erf, otypes=[np.float64])


# --------------------------------------------------------------------------- conf
# ref :322-335
DEFAULT_CONF = {
    "name": "lightglue",
    "input_dim": 256,
    "descriptor_dim": 256,
    "add_scale_ori": False,
    "n_layers": 9,
    "num_heads": 4,
    "flash": True,
    "mp": False,
    "depth_confidence": 0.95,
    "width_confidence": 0.99,
    "filter_threshold": 0.1,
    "weights": None,
    # not in the reference's dict: the reference keys this on device type through the
    # class dict ``pruning_keypoint_thresholds`` (ref :339-344, :658-662).  The oracle
    # takes the resolved number.  -1 == the reference's CPU behaviour (always prune).
    "pruning_min_kpts": -1,
}


def make_conf(**kw) -> SimpleNamespace:
    unknown = set(kw) - set(DEFAULT_CONF)
    if unknown:
        raise KeyError(f"unknown conf keys {sorted(unknown)}")
    return SimpleNamespace(**{**DEFAULT_CONF, **kw})


def confidence_threshold(layer_index: int, n_layers: int) -> float:
    """ref :631-634"""
    threshold = 0.8 + 0.1 * np.exp(-4.0 * layer_index / n_layers)
    return float(np.clip(threshold, 0, 1))


def confidence_thresholds_f32(n_layers: int) -> np.ndarray:
    """The registered buffer is ``torch.Tensor([...])`` == float32 (ref :408-413)."""
    return np.array([confidence_threshold(i, n_layers) for i in range(n_layers)], dtype=np.float32)


# --------------------------------------------------------------------------- helpers
def round_bf16(x: np.ndarray) -> np.ndarray:
    """Round-to-nearest-even to bfloat16, returned in the input's float type."""
    x32 = np.ascontiguousarray(x, dtype=np.float32)
    u = x32.view(np.uint32)
    rounded = (u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)
    return rounded.view(np.float32).astype(x.dtype, copy=False)


def round_fp16(x: np.ndarray) -> np.ndarray:
    return x.astype(np.float16).astype(x.dtype)


def round_bf16x2(x: np.ndarray) -> np.ndarray:
    """hi + lo with both bf16 (the "split-bf16" operand used by a 3-product MFMA scheme)."""
    hi = round_bf16(x)
    return hi + round_bf16(x - hi)


def round_fp16x2(x: np.ndarray) -> np.ndarray:
    """hi + lo with both fp16 (split-f16 weights of the q/k/v projections in the HIP path's default precision)."""
    hi = round_fp16(x)
    return hi + round_fp16(x - hi)


_ROUNDERS = {None: lambda a: a, "fp32": lambda a: a, "bf16": round_bf16, "fp16": round_fp16, "bf16x2": round_bf16x2, "fp16x2": round_fp16x2}

# operand rounding of the HIP path's precisions, for emulation studies (tools/emulated_margins.py, tests):
#   DEFAULT_PRECISION_QUANT  "f16x3": every contraction — linear layers, q k^T, P V, final projection + similarity — on split-f16
#                            operands (hi + lo, 22 bits), fp32 accumulation
#   FAST_ATTENTION_QUANT     "f16x3" with attention_precision "fp16" (the opt-in; the round-2 default's arithmetic on f16 planes): the
#                            q/k/v projections take ONE f16 plane of x (split-f16 weights), q / k / v / P are one f16 plane each
DEFAULT_PRECISION_QUANT = {"lin": "fp16x2", "attn": "fp16x2", "final": "fp16x2"}
FAST_ATTENTION_QUANT = {"lin": "fp16x2", "attn": "fp16", "final": "fp16x2", "lin_qkv": ("fp16", "fp16x2")}


class _Ctx:
    """``quant``: None | mode | {"lin": mode, "attn"|"attn_qk"/"attn_pv": mode, "final": mode} with
    mode in None/"fp32"/"bf16"/"fp16"/"bf16x2"/"fp16x2" or an (activation, weight) tuple: operand rounding per contraction class (linear
    layers, attention QK^T / PV, final projection + similarity)."""

    def __init__(self, dtype, quant):
        self.dtype = dtype
        if not isinstance(quant, dict):
            quant = {"lin": quant, "attn": quant, "final": quant}
        quant = dict(quant)
        if "attn" in quant:  # shorthand for both attention contractions
            quant.setdefault("attn_qk", quant["attn"])
            quant.setdefault("attn_pv", quant["attn"])
        self.quant = quant

    def mm(self, a, b, where="lin"):
        """a @ b with optional operand rounding; accumulation in ``dtype``."""
        mode = self.quant.get(where, self.quant.get("lin"))   # lin_* tags fall back to "lin"
        if isinstance(mode, tuple):   # (activation rounding, weight rounding) for asymmetric-operand studies
            a, b = _ROUNDERS[mode[0]](a), _ROUNDERS[mode[1]](b)
        else:
            a, b = _ROUNDERS[mode](a), _ROUNDERS[mode](b)
        # BLAS needs a unit stride in one of the last two axes; the q/k/v views of the interleaved Wqkv
        # output (ref :166-167) have stride 3 and would fall back to numpy's slow generic loop
        if a.ndim >= 2 and a.strides[-1] != a.itemsize and a.strides[-2] != a.itemsize:
            a = np.ascontiguousarray(a)
        if b.ndim >= 2 and b.strides[-1] != b.itemsize and b.strides[-2] != b.itemsize:
            b = np.ascontiguousarray(b)
        return np.matmul(a, b)

    def linear(self, x, w, b=None, where="lin"):
        """nn.Linear: x @ w.T + b"""
        y = self.mm(x, np.swapaxes(w, -1, -2), where)
        if b is not None:
            y = y + b
        return y

    # the normalisation / activation passes of the path; numpy here, torch's CPU kernels in _TorchCtx
    def softmax(self, x):
        return _softmax(x, -1)

    def log_softmax(self, x):
        return _log_softmax(x, -1)

    def layernorm(self, x, g, b):
        return _layernorm(x, g, b)

    def gelu(self, x):
        return _gelu(x)

    def sdpa(self, q, k, v):
        """softmax(q k^T / sqrt(d)) v  (ref :127-130)"""
        s = q.shape[-1] ** -0.5
        sim = self.mm(q, np.swapaxes(k, -1, -2), "attn_qk") * s
        return self.mm(self.softmax(sim), v, "attn_pv")

    def scaled(self, x, s):
        return x * s

    def rotary(self, t, cos, sin):
        return apply_rotary(t, cos, sin)


class _TorchCtx(_Ctx):
    """Same restatement, but every heavy pass runs on torch's CPU kernels (the library the reference itself computes with: ATen
    matmul / softmax / layer_norm / gelu and the fused fp32 SDPA of ref :127-130) instead of numpy's single-core loops.  fp32 only,
    no operand rounding.  Used for the timed ``cpu_baseline`` leg of bench.py (numpy's elementwise passes made the plain port
    1.5-6.5x slower than the reference on the same cores, profiles/r03_cpu_reference.md) and pinned against the same golden
    fixtures as the numpy form (tests/test_oracle_golden.py)."""

    def __init__(self, dtype, quant):
        assert dtype == np.float32 and quant in (None, "fp32"), "the torch-kernel backend is fp32 / unquantised only"
        super().__init__(dtype, None)
        import torch
        import torch.nn.functional as F
        self.torch, self.F = torch, F

    def _t(self, a):
        a = np.asarray(a)
        if not a.flags.writeable or any(st < 0 for st in a.strides):
            a = np.array(a)
        return self.torch.from_numpy(a)

    def mm(self, a, b, where="lin"):
        with self.torch.no_grad():
            return self.torch.matmul(self._t(a), self._t(b)).numpy()

    def linear(self, x, w, b=None, where="lin"):
        with self.torch.no_grad():
            return self.F.linear(self._t(x), self._t(w), None if b is None else self._t(b)).numpy()

    def softmax(self, x):
        with self.torch.no_grad():   # transposed views arrive here (ref :221 makes the same contiguous copy first)
            return self.torch.softmax(self._t(x).contiguous(), -1).numpy()

    def log_softmax(self, x):
        with self.torch.no_grad():
            return self.torch.log_softmax(self._t(x).contiguous(), -1).numpy()

    def layernorm(self, x, g, b):
        with self.torch.no_grad():
            return self.F.layer_norm(self._t(x), (x.shape[-1],), self._t(g), self._t(b), 1e-5).numpy()

    def gelu(self, x):
        with self.torch.no_grad():
            return self.F.gelu(self._t(x)).numpy()

    def sdpa(self, q, k, v):
        with self.torch.no_grad():   # ref :127-130: the fused fp32 kernel, contiguous operands (:124)
            q, k, v = (self._t(np.ascontiguousarray(t))[None] for t in (q, k, v))
            return self.F.scaled_dot_product_attention(q, k, v)[0].numpy()

    def scaled(self, x, s):
        with self.torch.no_grad():
            return (self._t(x) * s).numpy()

    def rotary(self, t, cos, sin):
        """ref :58-65, same pairing as apply_rotary"""
        with self.torch.no_grad():
            t, c, sn = self._t(t), self._t(cos), self._t(sin)
            te, to = t[..., 0::2], t[..., 1::2]
            out = self.torch.empty(t.shape, dtype=t.dtype)
            out[..., 0::2] = te * c - to * sn
            out[..., 1::2] = to * c + te * sn
            return out.numpy()


def make_ctx(dtype, quant, backend="numpy"):
    if backend == "numpy":
        return _Ctx(dtype, quant)
    if backend == "torch":
        return _TorchCtx(dtype, quant)
    raise ValueError(f"unknown oracle backend {backend!r}")


def _sigmoid(x):
    # numerically stable; matches torch.sigmoid to rounding
    out = np.empty_like(x)
    pos = x >= 0
    out[pos] = 1.0 / (1.0 + np.exp(-x[pos]))
    e = np.exp(x[~pos])
    out[~pos] = e / (1.0 + e)
    return out


def _logsigmoid(x):
    # F.logsigmoid(x) = -softplus(-x) = min(x,0) - log1p(exp(-|x|))
    return np.minimum(x, 0) - np.log1p(np.exp(-np.abs(x)))


def _softmax(x, axis=-1):
    m = np.max(x, axis=axis, keepdims=True)
    e = np.exp(x - m)
    return e / np.sum(e, axis=axis, keepdims=True)


def _log_softmax(x, axis=-1):
    m = np.max(x, axis=axis, keepdims=True)
    s = x - m
    return s - np.log(np.sum(np.exp(s), axis=axis, keepdims=True))


def _layernorm(x, g, b, eps=1e-5):
    """nn.LayerNorm(512, elementwise_affine=True): biased variance, eps inside sqrt."""
    mu = np.mean(x, axis=-1, keepdims=True)
    xc = x - mu
    var = np.mean(xc * xc, axis=-1, keepdims=True)
    return xc / np.sqrt(var + eps) * g + b


def _gelu(x):
    """nn.GELU() default = exact erf form."""
    return (0.5 * x * (1.0 + _erf(x / math.sqrt(2.0)))).astype(x.dtype, copy=False)


# --------------------------------------------------------------------------- pieces
def normalize_keypoints(kpts: np.ndarray, size: Optional[np.ndarray]) -> np.ndarray:
    """ref :32-43.  kpts [n,2]; size [2] (w,h) or None -> 1 + max - min bbox."""
    if size is None:
        size = 1 + kpts.max(-2) - kpts.min(-2)
    size = np.asarray(size, dtype=kpts.dtype)
    shift = size / 2
    scale = size.max(-1) / 2
    return (kpts - shift[..., None, :]) / scale[..., None, None]


def posenc(ctx: _Ctx, Wr: np.ndarray, kpts: np.ndarray):
    """ref :76-81 without the repeat_interleave: returns cos,sin of shape [n, 32].
    The 64-wide tensor the reference builds holds each of these twice, adjacent."""
    projected = kpts @ Wr.T  # Wr [32, 2|4]; tiny contraction, kept unquantised (fp32 in ours too)
    return np.cos(projected), np.sin(projected)


def apply_rotary(t: np.ndarray, cos: np.ndarray, sin: np.ndarray) -> np.ndarray:
    """ref :58-65.  t [H, n, 64]; pairs are ADJACENT elements (2j, 2j+1), freq j shared by heads.
    out[2j] = t[2j] c_j - t[2j+1] s_j ; out[2j+1] = t[2j+1] c_j + t[2j] s_j"""
    te, to = t[..., 0::2], t[..., 1::2]
    out = np.empty_like(t)
    out[..., 0::2] = te * cos - to * sin
    out[..., 1::2] = to * cos + te * sin
    return out


def attention(ctx: _Ctx, q, k, v):
    """ref :113-130 (CPU branch = fp32 SDPA, scale 1/sqrt(64)).  q [H,nq,64], k,v [H,nk,64]."""
    if q.shape[-2] == 0 or k.shape[-2] == 0:
        return np.zeros((*q.shape[:-1], v.shape[-1]), dtype=q.dtype)
    return ctx.sdpa(q, k, v)


def _ffn(ctx: _Ctx, p: Dict[str, np.ndarray], prefix: str, x, msg, tr=None, tag=""):
    """ref :152-157 / :187-192 applied as x + ffn(cat[x,msg]) (ref :172, :228-229)."""
    h = ctx.linear(np.concatenate([x, msg], -1), p[prefix + "ffn.0.weight"], p[prefix + "ffn.0.bias"], "lin_ffn0")
    if tr is not None:
        tr[tag + "h1"] = h
    h = ctx.layernorm(h, p[prefix + "ffn.1.weight"], p[prefix + "ffn.1.bias"])
    h = ctx.gelu(h)
    if tr is not None:
        tr[tag + "g"] = h
    return x + ctx.linear(h, p[prefix + "ffn.3.weight"], p[prefix + "ffn.3.bias"], "lin_ffn3")


def self_block(ctx: _Ctx, p, i: int, x, cos, sin, heads: int, tr=None, tag=""):
    """ref :159-172.  x [n,256].  ``tr`` (dict) receives the intermediates under ``tag``-prefixed keys."""
    pre = f"transformers.{i}.self_attn."
    n = x.shape[0]
    qkv = ctx.linear(x, p[pre + "Wqkv.weight"], p[pre + "Wqkv.bias"], "lin_qkv")  # [n,768]
    qkv = qkv.reshape(n, heads, -1, 3).transpose(1, 0, 2, 3)  # unflatten(-1,(H,-1,3)).transpose(1,2)
    q, k, v = qkv[..., 0], qkv[..., 1], qkv[..., 2]  # [H,n,64]
    q = ctx.rotary(q, cos, sin)
    k = ctx.rotary(k, cos, sin)
    context = attention(ctx, q, k, v)  # [H,n,64]
    message = ctx.linear(context.transpose(1, 0, 2).reshape(n, -1), p[pre + "out_proj.weight"], p[pre + "out_proj.bias"], "lin_out")
    if tr is not None:
        tr[tag + "q"], tr[tag + "k"], tr[tag + "v"] = q, k, v
        tr[tag + "ctx"], tr[tag + "msg"] = context.transpose(1, 0, 2).reshape(n, -1), message
    return _ffn(ctx, p, pre, x, message, tr, tag)


def cross_block(ctx: _Ctx, p, i: int, x0, x1, heads: int, tr=None, tag=""):
    """ref :201-230, CPU branch (:216-223): one shared sim, softmax along both axes."""
    pre = f"transformers.{i}.cross_attn."
    def heads_of(t):
        return t.reshape(t.shape[0], heads, -1).transpose(1, 0, 2)
    qk0 = heads_of(ctx.linear(x0, p[pre + "to_qk.weight"], p[pre + "to_qk.bias"], "lin_qkv"))
    qk1 = heads_of(ctx.linear(x1, p[pre + "to_qk.weight"], p[pre + "to_qk.bias"], "lin_qkv"))
    v0 = heads_of(ctx.linear(x0, p[pre + "to_v.weight"], p[pre + "to_v.bias"], "lin_qkv"))
    v1 = heads_of(ctx.linear(x1, p[pre + "to_v.weight"], p[pre + "to_v.bias"], "lin_qkv"))
    if tr is not None:
        tr[tag + "qk0"], tr[tag + "qk1"], tr[tag + "v0"], tr[tag + "v1"] = qk0, qk1, v0, v1
    if x0.shape[0] == 0 or x1.shape[0] == 0:
        m0 = np.zeros_like(qk0)
        m1 = np.zeros_like(qk1)
    else:
        scale = (qk0.shape[-1] ** -0.5) ** 0.5
        qk0, qk1 = ctx.scaled(qk0, scale), ctx.scaled(qk1, scale)
        sim = ctx.mm(qk0, np.swapaxes(qk1, -1, -2), "attn_qk")  # [H,n0,n1]
        attn01 = ctx.softmax(sim)
        attn10 = ctx.softmax(np.swapaxes(sim, -1, -2))  # [H,n1,n0]
        m0 = ctx.mm(attn01, v1, "attn_pv")
        m1 = ctx.mm(attn10, v0, "attn_pv")
    def merge(t):
        return t.transpose(1, 0, 2).reshape(t.shape[1], -1)
    if tr is not None:
        tr[tag + "ctx0"], tr[tag + "ctx1"] = merge(m0), merge(m1)
    m0 = ctx.linear(merge(m0), p[pre + "to_out.weight"], p[pre + "to_out.bias"], "lin_out")
    m1 = ctx.linear(merge(m1), p[pre + "to_out.weight"], p[pre + "to_out.bias"], "lin_out")
    if tr is not None:
        tr[tag + "msg0"], tr[tag + "msg1"] = m0, m1
    return (_ffn(ctx, p, pre, x0, m0, tr, tag + "i0_"), _ffn(ctx, p, pre, x1, m1, tr, tag + "i1_"))


def token_confidence(ctx: _Ctx, p, i: int, x):
    """ref :89-94: sigmoid(Linear(256->1)(x)).  GEMV kept unquantised (fp32 in ours too)."""
    w, b = p[f"token_confidence.{i}.token.0.weight"], p[f"token_confidence.{i}.token.0.bias"]
    return _sigmoid((x @ w.T + b)[..., 0])


def matchability_logit(p, i: int, x):
    """ref :293-294 / :298-299 (pre-sigmoid).  Uses the PRE-projection descriptors."""
    w, b = p[f"log_assignment.{i}.matchability.weight"], p[f"log_assignment.{i}.matchability.bias"]
    return (x @ w.T + b)[..., 0]


def log_assignment(ctx: _Ctx, p, i: int, x0, x1):
    """ref :287-296 + :265-277.  Returns scores [m+1, n+1] (dustbins included) and sim."""
    pre = f"log_assignment.{i}."
    d = x0.shape[-1]
    md0 = ctx.linear(x0, p[pre + "final_proj.weight"], p[pre + "final_proj.bias"], "final") / d**0.25
    md1 = ctx.linear(x1, p[pre + "final_proj.weight"], p[pre + "final_proj.bias"], "final") / d**0.25
    sim = ctx.mm(md0, md1.T, "final")
    z0 = matchability_logit(p, i, x0)[:, None]
    z1 = matchability_logit(p, i, x1)[:, None]
    m, n = sim.shape
    certainties = _logsigmoid(z0) + _logsigmoid(z1).T
    scores0 = ctx.log_softmax(sim)
    scores1 = ctx.log_softmax(sim.T).T
    scores = np.zeros((m + 1, n + 1), dtype=sim.dtype)
    scores[:m, :n] = scores0 + scores1 + certainties
    scores[:-1, -1] = _logsigmoid(-z0[:, 0])
    scores[-1, :-1] = _logsigmoid(-z1[:, 0])
    return scores, sim


def filter_matches(scores: np.ndarray, th: float):
    """ref :302-318 on one pair.  np.argmax == torch.max first-index tie-break."""
    inner = scores[:-1, :-1]
    m, n = inner.shape
    if m == 0 or n == 0:
        return (np.full(m, -1, np.int64), np.full(n, -1, np.int64),
                np.zeros(m, scores.dtype), np.zeros(n, scores.dtype))
    m0 = inner.argmax(1)
    m1 = inner.argmax(0)
    max0 = inner.max(1)
    mutual0 = np.arange(m) == m1[m0]
    mutual1 = np.arange(n) == m0[m1]
    max0_exp = np.exp(max0)
    mscores0 = np.where(mutual0, max0_exp, 0).astype(scores.dtype)
    mscores1 = np.where(mutual1, mscores0[m1], 0).astype(scores.dtype)
    valid0 = mutual0 & (mscores0 > th)
    valid1 = mutual1 & valid0[m1]
    m0 = np.where(valid0, m0, -1).astype(np.int64)
    m1 = np.where(valid1, m1, -1).astype(np.int64)
    return m0, m1, mscores0, mscores1


# --------------------------------------------------------------------------- forward
def forward_pair(params: Dict[str, np.ndarray], conf: SimpleNamespace,
                 kpts0, kpts1, desc0, desc1, size0=None, size1=None,
                 scales0=None, oris0=None, scales1=None, oris1=None,
                 dtype=np.float32, quant: Optional[str] = None, trace: Optional[dict] = None, backend: str = "numpy"):
    """ref :483-629 for ONE pair (the reference's adaptive path is only defined for B=1,
    SURVEY.md §0).  Inputs: kpts [n,2] pixels, desc [n,D_in].  Returns the reference's dict
    with the batch dimension dropped.  ``trace`` (optional dict) receives per-layer
    descriptors for localising a mismatch.  ``backend``: "numpy" (the checker) or "torch" (the same restatement on torch's CPU
    kernels, fp32 only: the timed cpu_baseline leg)."""
    ctx = make_ctx(dtype, quant, backend)
    p = {k: np.asarray(v, dtype=dtype) for k, v in params.items()}
    L, H = conf.n_layers, conf.num_heads
    thr = confidence_thresholds_f32(L).astype(dtype)
    m, n = kpts0.shape[0], kpts1.shape[0]
    assert desc0.shape[-1] == conf.input_dim and desc1.shape[-1] == conf.input_dim  # ref :505-506

    k0 = normalize_keypoints(np.asarray(kpts0, dtype), None if size0 is None else np.asarray(size0, dtype)) if m else np.zeros((0, 2), dtype)
    k1 = normalize_keypoints(np.asarray(kpts1, dtype), None if size1 is None else np.asarray(size1, dtype)) if n else np.zeros((0, 2), dtype)
    if conf.add_scale_ori:  # ref :495-501
        k0 = np.concatenate([k0, np.asarray(scales0, dtype)[:, None], np.asarray(oris0, dtype)[:, None]], -1)
        k1 = np.concatenate([k1, np.asarray(scales1, dtype)[:, None], np.asarray(oris1, dtype)[:, None]], -1)
    x0 = np.asarray(desc0, dtype)
    x1 = np.asarray(desc1, dtype)
    if conf.input_dim != conf.descriptor_dim:  # ref :388-391, :521-522
        x0 = ctx.linear(x0, p["input_proj.weight"], p["input_proj.bias"])
        x1 = ctx.linear(x1, p["input_proj.weight"], p["input_proj.bias"])
    cos0, sin0 = posenc(ctx, p["posenc.Wr.weight"], k0)  # ref :524-525
    cos1, sin1 = posenc(ctx, p["posenc.Wr.weight"], k1)
    if trace is not None:
        trace["x0_in"], trace["x1_in"] = x0.copy(), x1.copy()
        trace["cos0"], trace["sin0"], trace["cos1"], trace["sin1"] = cos0, sin0, cos1, sin1

    do_early_stop = conf.depth_confidence > 0  # ref :528
    do_point_pruning = conf.width_confidence > 0  # ref :529 (no compile path here)
    pruning_th = conf.pruning_min_kpts  # ref :530
    if do_point_pruning:  # ref :531-536
        ind0, ind1 = np.arange(m), np.arange(n)
        prune0, prune1 = np.ones(m, np.int64), np.ones(n, np.int64)
    token0 = token1 = None
    i = 0
    for i in range(L):  # ref :538
        if x0.shape[0] == 0 or x1.shape[0] == 0:  # ref :539-540
            break
        full = trace is not None and i in trace.get("_full_layers", ())
        x0 = self_block(ctx, p, i, x0, cos0, sin0, H, trace if full else None, f"l{i}_self0_")  # ref :251
        x1 = self_block(ctx, p, i, x1, cos1, sin1, H, trace if full else None, f"l{i}_self1_")  # ref :252
        if full:
            trace[f"l{i}_xs0"], trace[f"l{i}_xs1"] = x0.copy(), x1.copy()
        x0, x1 = cross_block(ctx, p, i, x0, x1, H, trace if full else None, f"l{i}_cross_")  # ref :253
        if trace is not None:
            trace[f"desc0_l{i}"], trace[f"desc1_l{i}"] = x0.copy(), x1.copy()
        if i == L - 1:  # ref :544-545
            continue
        if do_early_stop:  # ref :547-550
            token0, token1 = token_confidence(ctx, p, i, x0), token_confidence(ctx, p, i, x1)
            conf_all = np.concatenate([token0, token1], -1)
            # ref :653-656: float32 sum of a 0/1 mask divided by the ORIGINAL point count
            ratio = 1.0 - np.float32((conf_all < thr[i]).astype(np.float32).sum()) / np.float32(m + n)
            if ratio > conf.depth_confidence:
                break
        if do_point_pruning and x0.shape[0] > pruning_th:  # ref :551-558
            s0 = _sigmoid(matchability_logit(p, i, x0))
            keep = s0 > (1 - conf.width_confidence)  # ref :640
            if token0 is not None:
                keep |= token0 <= thr[i]  # ref :641-642
            keep0 = np.where(keep)[0]
            ind0, x0, cos0, sin0 = ind0[keep0], x0[keep0], cos0[keep0], sin0[keep0]
            prune0[ind0] += 1
        if do_point_pruning and x1.shape[0] > pruning_th:  # ref :559-566
            s1 = _sigmoid(matchability_logit(p, i, x1))
            keep = s1 > (1 - conf.width_confidence)
            if token1 is not None:
                keep |= token1 <= thr[i]
            keep1 = np.where(keep)[0]
            ind1, x1, cos1, sin1 = ind1[keep1], x1[keep1], cos1[keep1], sin1[keep1]
            prune1[ind1] += 1

    if x0.shape[0] == 0 or x1.shape[0] == 0:  # ref :568-588
        out = {
            "matches0": np.full(m, -1, np.int64), "matches1": np.full(n, -1, np.int64),
            "matching_scores0": np.zeros(m, dtype), "matching_scores1": np.zeros(n, dtype),
            "stop": i + 1, "matches": np.zeros((0, 2), np.int64), "scores": np.zeros((0,), dtype),
        }
        if not do_point_pruning:
            prune0 = np.ones(m, dtype) * L
            prune1 = np.ones(n, dtype) * L
        out["prune0"], out["prune1"] = prune0, prune1
        return out

    scores, _ = log_assignment(ctx, p, i, x0, x1)  # ref :591
    m0, m1, ms0, ms1 = filter_matches(scores, conf.filter_threshold)  # ref :592
    valid = m0 > -1  # ref :595-602
    mi0 = np.where(valid)[0]
    mi1 = m0[valid]
    if do_point_pruning:
        mi0, mi1 = ind0[mi0], ind1[mi1]
    matches = np.stack([mi0, mi1], -1).astype(np.int64)
    mscores = ms0[valid]
    if do_point_pruning:  # ref :605-614
        m0_ = np.full(m, -1, np.int64)
        m1_ = np.full(n, -1, np.int64)
        m0_[ind0] = np.where(m0 == -1, -1, ind1[np.clip(m0, 0, None)])
        m1_[ind1] = np.where(m1 == -1, -1, ind0[np.clip(m1, 0, None)])
        ms0_ = np.zeros(m, dtype)
        ms1_ = np.zeros(n, dtype)
        ms0_[ind0] = ms0
        ms1_[ind1] = ms1
        m0, m1, ms0, ms1 = m0_, m1_, ms0_, ms1_
    else:  # ref :616-617
        prune0 = np.ones(m, dtype) * L
        prune1 = np.ones(n, dtype) * L
    if trace is not None:
        trace["scores_full"] = scores
        trace["ind0"], trace["ind1"] = (ind0, ind1) if do_point_pruning else (np.arange(m), np.arange(n))
    return {
        "matches0": m0, "matches1": m1, "matching_scores0": ms0, "matching_scores1": ms1,
        "stop": i + 1, "matches": matches, "scores": mscores, "prune0": prune0, "prune1": prune1,
    }


def forward(params, conf, data: dict, dtype=np.float32, quant=None, backend: str = "numpy"):
    """Batched wrapper: a Python loop of B=1 calls (every pair prunes/stops
    independently).  data = {"image0": {"keypoints" [B,N,2], "descriptors" [B,N,D],
    optional "image_size" [B,2], "scales","oris" [B,N]}, "image1": {...}}.
    Returns stacked arrays for the fixed-shape outputs and lists for the ragged ones;
    ``stop`` is a list of ints (one per pair)."""
    for key in ("image0", "image1"):  # ref :484-485
        assert key in data, f"Missing key {key} in data"
    d0, d1 = data["image0"], data["image1"]
    B = d0["keypoints"].shape[0]
    outs = []
    for b in range(B):
        def g(d, k):
            v = d.get(k)
            return None if v is None else np.asarray(v)[b]
        outs.append(forward_pair(
            params, conf, g(d0, "keypoints"), g(d1, "keypoints"), g(d0, "descriptors"), g(d1, "descriptors"),
            g(d0, "image_size"), g(d1, "image_size"), g(d0, "scales"), g(d0, "oris"), g(d1, "scales"), g(d1, "oris"),
            dtype=dtype, quant=quant, backend=backend))
    res = {k: np.stack([o[k] for o in outs]) for k in
           ("matches0", "matches1", "matching_scores0", "matching_scores1", "prune0", "prune1")}
    res["stop"] = [o["stop"] for o in outs]
    res["matches"] = [o["matches"] for o in outs]
    res["scores"] = [o["scores"] for o in outs]
    return res
