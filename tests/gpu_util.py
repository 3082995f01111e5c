"""Helpers for the GPU parity tests: run the HIP path through the public class, tap intermediate
buffers through the C-ABI debug entry points and line them up with the oracle's trace."""
import numpy as np
import torch

from lightglue_amd import LightGlue
from oracle import lightglue_oracle as O


def to_torch(data, device="cuda"):
    return {k: {kk: torch.from_numpy(np.ascontiguousarray(vv)).to(device) for kk, vv in v.items()} for k, v in data.items()}


QK_PRESCALE = 0.42466090014400953   # lg_proj_body.h: q and k leave the projections times sqrt(log2(e) / sqrt(64))


def make_model(sd, precision, **conf):
    """precision: "fp32" | "bf16" | "fp16" | "f16x3" (default: split attention) | "f16x3/fp16" (f16x3 linear layers with the
    single-plane f16 attention, the fast opt-in)."""
    precision, _, attn = precision.partition("/")
    model = LightGlue(features=None, precision=precision, attention_precision=attn or None, **conf).eval()
    res = model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    assert not res.unexpected_keys and set(res.missing_keys) <= {"confidence_thresholds"}, res
    return model


def read_attn_buf(model, name):
    """q / k / v^T as fp32 in the oracle's units: the split attention's hi + lo planes are summed, q and k are un-scaled."""
    kind = model.conf.attention_precision or model.conf.precision
    if kind == "fp32":
        buf = model.debug_read(name, np.float32)
    elif kind == "fp16":
        buf = model.debug_read(name, np.float16).astype(np.float32)
    elif kind == "f16x3":
        planes = model.debug_read(name, np.float16).astype(np.float32)
        buf = planes[: planes.size // 2] + planes[planes.size // 2:]
    else:
        buf = (model.debug_read(name, np.uint16).astype(np.uint32) << 16).view(np.float32)
    return buf / np.float32(QK_PRESCALE) if name in ("Q", "K") else buf


class Rows:
    """global row index of (pair, image, r) for the engine's row space"""

    def __init__(self, B, n0, n1, cap0, cap1):
        self.B, self.n0, self.n1, self.c0, self.c1 = B, n0, n1, cap0, cap1
        self.R = B * (cap0 + cap1)

    def sl(self, pair, image):
        base = pair * (self.c0 + self.c1) + image * self.c0
        return slice(base, base + (self.n1 if image else self.n0))


def err(a, b):
    """max abs error and max abs error relative to the reference's rms"""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    d = np.abs(a - b).max() if a.size else 0.0
    rms = np.sqrt((b * b).mean()) if b.size else 1.0
    return float(d), float(d / (rms + 1e-30))


def stage_errors(sd, data, precision, conf_kw, layer=0, fused=False):
    """Run the pipeline up to every step of `layer` and compare the step's output buffer with the
    oracle.  Returns {stage name: (max abs err, err / rms)}."""
    B, n0 = data["image0"]["keypoints"].shape[:2]
    n1 = data["image1"]["keypoints"].shape[1]
    conf = O.make_conf(**{**conf_kw, "pruning_min_kpts": conf_kw.get("pruning_min_kpts", -1) if conf_kw.get("pruning_min_kpts") is not None else -1})
    traces = []
    for b in range(B):
        tr = {"_full_layers": (layer,)}
        g = lambda d, k: (None if d.get(k) is None else np.asarray(d[k])[b])
        d0, d1 = data["image0"], data["image1"]
        O.forward_pair(sd, conf, g(d0, "keypoints"), g(d1, "keypoints"), g(d0, "descriptors"), g(d1, "descriptors"),
                       g(d0, "image_size"), g(d1, "image_size"), g(d0, "scales"), g(d0, "oris"), g(d1, "scales"), g(d1, "oris"), trace=tr)
        traces.append(tr)
    model = make_model(sd, precision, **conf_kw)
    model.set_option("fused_tail", int(fused))   # unfused: every intermediate buffer of the chain exists
    tdata = to_torch(data)
    res = {}
    L = layer
    base = 1 + 12 * L

    def run_to(step):
        model.debug_stop_after(step)
        model(tdata)
        c0, c1 = model.debug_caps()
        return Rows(B, n0, n1, c0, c1)

    def cmp_rows(name, buf, width, key_fn):
        worst = (0.0, 0.0)
        for b in range(B):
            for im in (0, 1):
                got = buf.reshape(rows.R, width)[rows.sl(b, im)]
                e = err(got, key_fn(traces[b], im))
                worst = max(worst, e)
        res[name] = worst

    def cmp_heads(name, buf, transposed, key_fn):
        worst = (0.0, 0.0)
        for b in range(B):
            for im in (0, 1):
                ref = key_fn(traces[b], im)  # [H, n, 64]
                if transposed:
                    got = buf.reshape(4, 64, rows.R)[:, :, rows.sl(b, im)].transpose(0, 2, 1)
                else:
                    got = buf.reshape(4, rows.R, 64)[:, rows.sl(b, im), :]
                worst = max(worst, err(got, ref))
        res[name] = worst

    if L == 0:
        rows = run_to(0)
        cmp_rows("prep.X", model.debug_read("X"), 256, lambda t, im: t[f"x{im}_in"])
        cmp_rows("prep.COS", model.debug_read("COS"), 32, lambda t, im: t[f"cos{im}"])
        cmp_rows("prep.SIN", model.debug_read("SIN"), 32, lambda t, im: t[f"sin{im}"])
    rows = run_to(base + 0)
    cmp_heads("self.q(rope)", read_attn_buf(model, "Q"), False, lambda t, im: t[f"l{L}_self{im}_q"])
    cmp_heads("self.k(rope)", read_attn_buf(model, "K"), False, lambda t, im: t[f"l{L}_self{im}_k"])
    cmp_heads("self.v^T", read_attn_buf(model, "VT"), True, lambda t, im: t[f"l{L}_self{im}_v"])
    rows = run_to(base + 1)
    cmp_rows("self.attn_ctx", model.debug_read("CTX"), 256, lambda t, im: t[f"l{L}_self{im}_ctx"])
    if not fused:
        rows = run_to(base + 2)
        cmp_rows("self.out_proj", model.debug_read("MSG"), 256, lambda t, im: t[f"l{L}_self{im}_msg"])
        rows = run_to(base + 3)
        cmp_rows("self.ffn0", model.debug_read("H1"), 512, lambda t, im: t[f"l{L}_self{im}_h1"])
        rows = run_to(base + 4)
        cmp_rows("self.ln_gelu", model.debug_read("G"), 512, lambda t, im: t[f"l{L}_self{im}_g"])
    rows = run_to(base + 5)
    cmp_rows("self.x_out", model.debug_read("X"), 256, lambda t, im: t[f"l{L}_xs{im}"])
    rows = run_to(base + 6)
    cmp_heads("cross.qk", read_attn_buf(model, "Q"), False, lambda t, im: t[f"l{L}_cross_qk{im}"])
    cmp_heads("cross.v^T", read_attn_buf(model, "VT"), True, lambda t, im: t[f"l{L}_cross_v{im}"])
    rows = run_to(base + 7)
    cmp_rows("cross.attn_ctx", model.debug_read("CTX"), 256, lambda t, im: t[f"l{L}_cross_ctx{im}"])
    if not fused:
        rows = run_to(base + 8)
        cmp_rows("cross.to_out", model.debug_read("MSG"), 256, lambda t, im: t[f"l{L}_cross_msg{im}"])
        rows = run_to(base + 9)
        cmp_rows("cross.ffn0", model.debug_read("H1"), 512, lambda t, im: t[f"l{L}_cross_i{im}_h1"])
        rows = run_to(base + 10)
        cmp_rows("cross.ln_gelu", model.debug_read("G"), 512, lambda t, im: t[f"l{L}_cross_i{im}_g"])
    rows = run_to(base + 11)
    cmp_rows("cross.x_out", model.debug_read("X"), 256, lambda t, im: t[f"desc{im}_l{L}"])
    model.debug_stop_after(-1)
    return res
