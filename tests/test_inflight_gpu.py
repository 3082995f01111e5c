"""InflightMatcher (several whole batches in flight on one GPU, one engine + one HIP stream per lane): every result must equal the plain
synchronous ``forward`` of the same batch bit for bit — key set, dtypes, ragged lists, stop — whatever the depth, with adaptive depth / width,
with batches of different shapes dealt to the same lane one after another, and a range-guard failure must surface from ``result()``."""
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

import gpu_util
from conftest import require_gpu
from lightglue_amd import InflightMatcher
from lightglue_amd import synthetic as synth
from lightglue_amd._cabi import LightGlueAmdError

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _same(res, plain):
    assert set(res) == set(plain), set(res) ^ set(plain)
    for k in ("matches0", "matches1", "matching_scores0", "matching_scores1", "prune0", "prune1"):
        assert res[k].dtype == plain[k].dtype and torch.equal(res[k], plain[k]), k
    if torch.is_tensor(plain["stop"]):
        assert torch.equal(res["stop"], plain["stop"])
    else:
        assert isinstance(res["stop"], int) and res["stop"] == plain["stop"]
    assert len(res["matches"]) == len(plain["matches"])
    for a, b, c, d in zip(res["matches"], plain["matches"], res["scores"], plain["scores"]):
        assert a.dtype == b.dtype and torch.equal(a, b) and torch.equal(c, d)


@pytest.mark.parametrize("depth", [1, 2, 3])
@pytest.mark.parametrize("adaptive", [False, True])
def test_results_equal_plain_forward(depth, adaptive):
    require_gpu()
    sd = synth.make_state_dict(0, recipe="C" if adaptive else "A")
    kw = dict(pruning_min_kpts=64) if adaptive else dict(depth_confidence=-1, width_confidence=-1)
    model = gpu_util.make_model(sd, "f16x3", **kw)
    shapes = [(3, 300, 260), (1, 512, 512), (4, 130, 200), (2, 640, 64), (3, 300, 260), (1, 96, 700), (5, 257, 255)]   # lanes see different shapes back to back
    batches = [gpu_util.to_torch(synth.make_batch(40 + i, b, n, m)) for i, (b, n, m) in enumerate(shapes)]
    plain = [model(d) for d in batches]
    if adaptive:
        assert any((int(p["stop"]) if not torch.is_tensor(p["stop"]) else int(p["stop"].min())) < 9 for p in plain), "the adaptive cases should include an early stop"
    lanes = InflightMatcher(model, depth)
    for _ in range(2):                                   # second pass: every lane re-used after a drain
        handles = [lanes.submit(d) for d in batches]     # all submitted before the first result is asked for
        assert {h.lane for h in handles} == set(range(depth))
        for h, p in zip(handles, plain):
            _same(h.result(), p)
    for r, p in zip(lanes.map(batches), plain):          # the bounded form: at most `depth` in flight
        _same(r, p)


def test_inputs_produced_on_the_callers_stream_are_waited_for():
    """submit() orders the lane's stream behind the caller's current stream: inputs still being written there when submit() returns must be seen complete."""
    require_gpu()
    model = gpu_util.make_model(synth.make_state_dict(0, recipe="A"), "f16x3", depth_confidence=-1, width_confidence=-1)
    base = gpu_util.to_torch(synth.make_batch(9, 2, 384, 384))
    plain = model(base)
    lanes = InflightMatcher(model, 2)
    spin = torch.empty(4096, 4096, device="cuda")
    for _ in range(3):
        data = {k: {kk: torch.zeros_like(vv) for kk, vv in v.items()} for k, v in base.items()}
        for _ in range(4):
            spin = spin @ spin * 0.0 + 1.0               # keeps the caller's stream busy ...
        for k, v in base.items():
            for kk, vv in v.items():
                data[k][kk].copy_(vv)                    # ... in front of the copies that fill the inputs
        _same(lanes.submit(data).result(), plain)


def test_range_guard_failure_surfaces_from_result():
    require_gpu()
    model = gpu_util.make_model(synth.make_state_dict(0, recipe="A"), "f16x3", depth_confidence=-1, width_confidence=-1)
    model.check_finite = True
    good = gpu_util.to_torch(synth.make_batch(5, 2, 256, 256))
    bad = gpu_util.to_torch(synth.make_batch(5, 2, 256, 256))
    bad["image0"]["descriptors"][1, 7, 3] = float("nan")
    plain = model(good)                                  # (before the lanes are busy: lane 0 IS this model, and an engine runs one forward at a time)
    lanes = InflightMatcher(model, 2)
    for lane in lanes._lanes:
        lane.check_finite = True
    h0, h1, h2 = lanes.submit(good), lanes.submit(bad), lanes.submit(good)
    _same(h0.result(), plain)
    with pytest.raises(LightGlueAmdError, match="pair 1"):
        h1.result()
    _same(h2.result(), plain)


def test_bench_inflight_line():
    """`bench.py --config 3 --inflight 2` prints one contract line that names the lanes; parity of its step output is the tests above."""
    require_gpu()
    import json
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--config", "3", "--inflight", "2", "--steps", "6", "--warmup", "3", "--no-cpu-baseline", "--no-gather-probe", "--no-calibration"],
                         capture_output=True, text=True, timeout=600, cwd=str(ROOT))
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["config"]["batches_in_flight"] == 2 and "InflightMatcher" in line["config"]["host_pipelining"]
    assert line["value"] > 0 and line["steps"] == 6


@pytest.mark.parametrize("depth", [1, 2, 4])
def test_prefetch_to_device_feeds_the_same_batches(depth):
    """lightglue_amd.prefetch_to_device: host batches (pinned and pageable, different shapes in one stream of batches) arrive on the device in order, `depth` ahead on a
    copy stream; a forward on each yielded batch equals the forward on a plain `.cuda()` copy bit for bit — also when the consumer runs on a stream of its own and
    through InflightMatcher lanes."""
    require_gpu()
    from lightglue_amd import prefetch_to_device
    sd = synth.make_state_dict(0, recipe="A")
    model = gpu_util.make_model(sd, "f16x3", depth_confidence=-1, width_confidence=-1)
    shapes = [(2, 300, 333), (1, 129, 64), (3, 512, 40), (2, 300, 333), (1, 5, 700), (2, 64, 64)]
    host = []
    for i, (B, n, m) in enumerate(shapes):
        b = synth.make_batch(40 + i, B, n, m)
        t = {k: {kk: torch.from_numpy(np.ascontiguousarray(vv)) for kk, vv in v.items()} for k, v in b.items()}
        if i % 2 == 0:
            t = {k: {kk: vv.pin_memory() for kk, vv in v.items()} for k, v in t.items()}
        host.append(t)
    plain = [model({k: {kk: vv.cuda() for kk, vv in v.items()} for k, v in h.items()}) for h in host]
    got = [model(d) for d in prefetch_to_device(iter(host), "cuda", depth)]
    assert len(got) == len(plain)
    for g, p in zip(got, plain):
        _same(g, p)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        got = [model(d) for d in prefetch_to_device(iter(host), "cuda", depth)]
    side.synchronize()
    for g, p in zip(got, plain):
        _same(g, p)
    lanes = InflightMatcher(model, 2)
    for g, p in zip(lanes.map(prefetch_to_device(iter(host), "cuda", depth)), plain):
        _same(g, p)
    assert list(prefetch_to_device(iter([]), "cuda", depth)) == []
