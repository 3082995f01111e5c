"""GPU tests of the round-5 additions: the extension outputs of lg_forward_io (reference dtypes written by the engine, packed wire row,
per-pair status), the range guard, lg_unpack_wire, and parity BEYOND the fixed fixtures — weight / data seeds drawn on the fly, checked
against the pinned oracle (torch-kernel backend of oracle/lightglue_oracle.py, pinned by tests/test_oracle_golden.py)."""
import ctypes as C

import numpy as np
import pytest
import torch

import gpu_util
from conftest import SCORE_TOL, assert_parity_with_explained_flips, require_gpu
from lightglue_amd import _cabi
from lightglue_amd import synthetic as synth
from oracle import lightglue_oracle as O

pytestmark = pytest.mark.gpu

DATA_KW = {"A": {}, "D": synth.RECIPE_D_DATA, "E": synth.RECIPE_E_DATA}


@pytest.mark.parametrize("recipe", ["A", "D", "E"])
@pytest.mark.parametrize("seed", range(10))
def test_seed_sweep_parity(recipe, seed):
    """Default precision at N = M = 1024 on weights AND data no fixture holds: scores within 1e-3 of the oracle, every
    index flip traced to an oracle-side decision boundary (conftest.explain_mismatches), stop layer identical."""
    require_gpu()
    torch.set_num_threads(8)   # the oracle's torch backend: 8 threads per pytest worker (the box has 256 logical cores; the default oversubscribes under -n 4)
    wseed, dseed = 100 + seed, 7000 + 31 * seed + {"A": 0, "D": 1, "E": 2}[recipe]
    sd = synth.make_state_dict(wseed, recipe=recipe)
    data = synth.make_batch(dseed, 1, 1024, 1024, **DATA_KW[recipe])
    conf_kw = dict(depth_confidence=-1, width_confidence=-1)
    ref = O.forward(sd, O.make_conf(**conf_kw), data, backend="torch")
    gold = {k: np.asarray(ref[k]) for k in ("matches0", "matches1", "matching_scores0", "matching_scores1")}
    model = gpu_util.make_model(sd, "f16x3", **conf_kw)
    out = model(gpu_util.to_torch(data))
    torch.cuda.synchronize()
    case = {"conf": conf_kw, "recipe": recipe, "wseed": wseed, "dseed": dseed, "n": 1024, "m": 1024, "B": 1, "dim": 256}
    flips = assert_parity_with_explained_flips(out, gold, case, sd, data, score_tol=SCORE_TOL)
    assert sum(flips) <= 2, f"{flips} index flips (each explained by an oracle-side boundary, but more than seed noise allows)"
    assert out["stop"] == 9


def _forward_with_guard(scale):
    sd = synth.make_state_dict(0, recipe="A")
    data = synth.make_batch(11, 2, 256, 192)
    data["image0"]["descriptors"][1] *= np.float32(scale)          # pair 1 only
    model = gpu_util.make_model(sd, "f16x3", depth_confidence=-1, width_confidence=-1)
    model.check_finite = True
    return model, gpu_util.to_torch(data)


def test_range_guard_flags_the_overflowing_pair_only():
    """LG_FLAG_CHECK_FINITE: descriptors of pair 1 scaled far outside the f16 operand range -> that pair's status is LG_ERR_RANGE, pair 0's
    LG_OK (forward_raw exposes the status array; forward raises)."""
    require_gpu()
    model, td = _forward_with_guard(1e6)
    raw = model.forward_raw(td)
    torch.cuda.synchronize()
    assert raw["status"].cpu().tolist() == [_cabi.LG_OK, _cabi.LG_ERR_RANGE]
    with pytest.raises(_cabi.LightGlueAmdError, match="pair 1"):
        model(td)
    # in range: nothing flagged, and the guard does not change a single output bit
    model, td = _forward_with_guard(1.0)
    a = model(td)
    model.check_finite = False
    b = model(td)
    for k in ("matches0", "matches1", "matching_scores0", "matching_scores1"):
        assert torch.equal(a[k], b[k])


def test_nan_input_is_flagged():
    require_gpu()
    model, td = _forward_with_guard(1.0)
    td["image1"]["descriptors"][0, 5, 7] = float("nan")
    raw = model.forward_raw(td)
    torch.cuda.synchronize()
    assert raw["status"].cpu().tolist() == [_cabi.LG_ERR_RANGE, _cabi.LG_OK]


@pytest.mark.parametrize("adaptive", [False, True])
def test_extension_outputs_equal_the_int32_ones(adaptive):
    """The engine's last kernel writes the reference's dtypes itself: they must be the widened int32 outputs bit for bit, the float prune fill
    n_layers / 0 (ragged padding), and the wire row the packed concatenation lg_unpack_wire inverts."""
    require_gpu()
    sd = synth.make_state_dict(3, recipe="C" if adaptive else "A")
    kw = dict(depth_confidence=0.95, width_confidence=0.99, pruning_min_kpts=64) if adaptive else dict(depth_confidence=-1, width_confidence=-1)
    data = synth.make_batch(21, 3, 200, 168)
    td = gpu_util.to_torch(data)
    td["image0"]["num_keypoints"] = torch.tensor([200, 150, 0], device="cuda")
    td["image1"]["num_keypoints"] = torch.tensor([168, 168, 100], device="cuda")
    model = gpu_util.make_model(sd, "f16x3", **kw)
    out = model(td)
    B, m, n = 3, 200, 168
    W = _cabi.wire_width(m, n)                                   # 3m + 3n + 2 (LG_WIRE_WIDTH)
    wire = torch.full((B + 1, W + 4), -7, dtype=torch.int32, device="cuda")   # one spare row, four spare columns: must stay untouched
    raw = model.forward_raw(td, wire=wire)
    torch.cuda.synchronize()
    assert torch.equal(out["matches0"], raw["matches0"].long()) and torch.equal(out["matches1"], raw["matches1"].long())
    assert out["matches0"].dtype == torch.int64 and out["stop"].dtype == torch.int64
    assert torch.equal(out["stop"], raw["stop"].long())
    assert raw["pruning"] == adaptive
    if adaptive:
        assert out["prune0"].dtype == torch.int64 and out["prune1"].dtype == torch.int64
        assert (out["prune0"][2] == 0).all() and (out["prune0"][1, 150:] == 0).all() and (out["prune0"][0] >= 1).all()
    else:
        assert out["prune0"].dtype == torch.float32
        live0 = torch.arange(m, device="cuda")[None] < td["image0"]["num_keypoints"][:, None]
        assert torch.equal(out["prune0"], live0.float() * 9) and (out["prune1"][2, :100] == 9).all() and (out["prune1"][2, 100:] == 0).all()
    w = wire[:B].cpu()
    wp = 2 * m + 2 * n + 2
    assert (wire[B] == -7).all() and (wire[:B, W:] == -7).all()
    assert torch.equal(w[:, :m], raw["matches0"].cpu()) and torch.equal(w[:, 2 * m:2 * m + n], raw["matches1"].cpu())
    assert torch.equal(w[:, m:2 * m].contiguous().view(torch.float32), raw["matching_scores0"].cpu())
    assert torch.equal(w[:, 2 * m + n:2 * m + 2 * n].contiguous().view(torch.float32), raw["matching_scores1"].cpu())
    assert torch.equal(w[:, wp - 2], raw["stop"].cpu()) and torch.equal(w[:, wp - 1], raw["status"].cpu())
    if adaptive:
        assert torch.equal(w[:, wp:wp + m].long(), out["prune0"].cpu()) and torch.equal(w[:, wp + m:wp + m + n].long(), out["prune1"].cpu())
    else:
        assert torch.equal(w[:, wp:wp + m].contiguous().view(torch.float32), out["prune0"].cpu())
    # a wire buffer that is too narrow / of the wrong type is refused before the engine sees it
    with pytest.raises(ValueError):
        model.forward_raw(td, wire=torch.zeros((B, W - 1), dtype=torch.int32, device="cuda"))
    with pytest.raises(ValueError):
        model.forward_raw(td, wire=torch.zeros((B, W), dtype=torch.int64, device="cuda"))
    # lg_unpack_wire with a row permutation and a skipped row: the reference's dtypes, the sorted match list, the host block
    order = torch.tensor([2, -1, 0, 1], dtype=torch.int32, device="cuda")
    rows = torch.cat([wire[2:3], wire[B:B + 1], wire[0:1], wire[1:2]]).contiguous()
    new = lambda shape, dt: torch.full(shape, -3, dtype=dt, device="cuda")
    kmax = min(m, n)
    pdt = torch.int64 if adaptive else torch.float32
    o = {"m0": new((B, m), torch.int64), "s0": new((B, m), torch.float32), "m1": new((B, n), torch.int64), "s1": new((B, n), torch.float32), "stop": new((B,), torch.int64),
         "p0": new((B, m), pdt), "p1": new((B, n), pdt), "ml": new((B, kmax, 2), torch.int64), "ms": new((B, kmax), torch.float32), "info": new((3, B), torch.int32)}
    dp = lambda t: t.data_ptr()
    io = _cabi.LgUnpackIO(dp(rows), rows.stride(0), 4, m, n, int(adaptive), B, dp(order), dp(o["m0"]), dp(o["m1"]), dp(o["stop"]), dp(o["s0"]), dp(o["s1"]),
                          dp(o["p0"]) if adaptive else None, dp(o["p1"]) if adaptive else None, None if adaptive else dp(o["p0"]), None if adaptive else dp(o["p1"]),
                          dp(o["ml"]), dp(o["ms"]), dp(o["info"]))
    _cabi.check(_cabi.load().lg_unpack_wire(C.byref(io), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    assert torch.equal(o["m0"], out["matches0"]) and torch.equal(o["m1"], out["matches1"]) and torch.equal(o["stop"], out["stop"])
    assert torch.equal(o["s0"], out["matching_scores0"]) and torch.equal(o["s1"], out["matching_scores1"])
    assert torch.equal(o["p0"], out["prune0"]) and torch.equal(o["p1"], out["prune1"])
    info = o["info"].cpu()
    assert torch.equal(info[0].long(), out["stop"].cpu()) and (info[2] == 0).all()
    for b in range(B):
        c = int(info[1, b])
        assert c == out["matches"][b].shape[0]
        assert torch.equal(o["ml"][b, :c], out["matches"][b]) and torch.equal(o["ms"][b, :c], out["scores"][b])
    # an empty gather is a no-op, not an error
    io0 = _cabi.LgUnpackIO(None, 0, 0, m, n, 0, 0, *([None] * 13))
    _cabi.check(_cabi.load().lg_unpack_wire(C.byref(io0), None))


def test_no_framework_kernels_between_engine_launches():
    """with the extension outputs the forward launches no ATen kernel behind (or between) the engine's own: the torch profiler
    sees only memcpy / memset activity and kernels whose names live in namespace lg."""
    require_gpu()
    sd = synth.make_state_dict(0, recipe="A")
    model = gpu_util.make_model(sd, "f16x3", depth_confidence=-1, width_confidence=-1)
    td = gpu_util.to_torch(synth.make_batch(5, 4, 512, 512))
    model(td); torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        model(td); torch.cuda.synchronize()
    kernels = [e.name for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA and "Memcpy" not in e.name and "Memset" not in e.name]
    foreign = [k for k in kernels if any(tag in k for tag in ("at::", "at_cuda", "c10::", "rocprim", "hipcub", "elementwise", "Elementwise"))]
    ours = [k for k in kernels if "lg::" in k or "anonymous namespace" in k]
    assert len(ours) >= 20 and not foreign, f"framework kernels on the product path: {sorted(set(foreign))}"


@pytest.mark.parametrize("seed", range(6))
def test_seed_sweep_adaptive_parity(seed):
    """The adaptive path (early stop + point pruning on the device: decide / compact / un-prune) on weights and data no fixture holds: recipe C
    (pairs stop at mixed depths and prune), N = 1100 / M = 900 with the pruning threshold lowered to 512 so that both images prune.  Against the
    oracle: same stop layer, same prune counters, indices / scores under the usual bar."""
    require_gpu()
    torch.set_num_threads(8)
    wseed, dseed = 200 + seed, 9100 + 17 * seed
    sd = synth.make_state_dict(wseed, recipe="C")
    data = synth.make_batch(dseed, 1, 1100, 900)
    conf_kw = dict(pruning_min_kpts=512)
    ref = O.forward(sd, O.make_conf(**conf_kw), data, backend="torch")
    gold = {k: np.asarray(ref[k]) for k in ("matches0", "matches1", "matching_scores0", "matching_scores1", "prune0", "prune1")}
    model = gpu_util.make_model(sd, "f16x3", **conf_kw)
    out = model(gpu_util.to_torch(data))
    torch.cuda.synchronize()
    assert int(out["stop"]) == int(ref["stop"][0])
    np.testing.assert_array_equal(out["prune0"].cpu().numpy(), gold["prune0"])
    np.testing.assert_array_equal(out["prune1"].cpu().numpy(), gold["prune1"])
    case = {"conf": conf_kw, "prune_th": 512, "recipe": "C", "wseed": wseed, "dseed": dseed, "n": 1100, "m": 900, "B": 1, "dim": 256}
    flips = assert_parity_with_explained_flips(out, gold, case, sd, data, score_tol=SCORE_TOL)
    assert sum(flips) <= 2, flips
