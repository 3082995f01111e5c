"""GPU parity: the HIP path (through the public class -> ctypes -> C ABI) against the golden vectors
of the real reference and against the oracle.  Bar (BASELINE.json): indices bit-identical, scores
within 1e-3.  "fp32" (exact f32 MFMA) must match everywhere; the default "f16x3" (every contraction on
split-f16 operands, attention included) must match except where the oracle sits within tolerance of a
decision boundary, proven per element by conftest.explain_mismatches.  The fast opt-in ("f16x3/fp16":
single-plane f16 attention) holds the same bar on the diffuse-attention fixtures and has its measured
envelope asserted on the trained-statistics (recipe D) ones."""
import json
from pathlib import Path

import numpy as np
import pytest
import torch

import gpu_util
import make_golden
from conftest import SCORE_TOL, assert_parity_with_explained_flips, explain_mismatches, golden_names, load_golden, oracle_conf_for, require_gpu
from lightglue_amd import synthetic as synth
from oracle import lightglue_oracle as O

pytestmark = pytest.mark.gpu


def score_bar(name):
    """The 1e-3 score bar is claimed on the reference's input domain: unit-norm descriptors (what SuperPoint / DISK / ALIKED emit, times the
    norm spread [0.5, 3] of the trained-statistics recipes) up to a common factor of 10.  At x30 (descriptor norms up to 90: the first
    attention's base-2 logits reach 1e4, where ONE fp32 ulp of a logit is 1e-3) the network is ill-conditioned in fp32 itself: the exact-fp32
    GPU mode differs from the reference by 1.6e-3, the default precision by 1.8e-3 (round 4, profiles/r04_fixture_errors.md; the reference's own
    output is 4.4e-4 from a float64 evaluation there) — zero index flips in both.  The x30 fixture documents that envelope: 3e-3."""
    return 3e-3 if name.endswith("_x30") else SCORE_TOL


def run_case(name, precision):
    meta, gold = load_golden(name)
    case = meta["case"]
    sd, data = make_golden.case_inputs(case)
    kw = dict(case["conf"])
    if "prune_th" in case:
        kw["pruning_min_kpts"] = case["prune_th"]
    model = gpu_util.make_model(sd, precision, **kw)
    if case.get("static_lengths"):
        model.static_lengths = list(case["static_lengths"])   # what compile() sets (ref :454)
    out = model(gpu_util.to_torch(data))
    torch.cuda.synchronize()
    return case, sd, data, gold, out


RECORDED = json.loads((Path(__file__).resolve().parent / "golden" / "recorded_score_errors.json").read_text())["max_abs_dscore"]


def assert_recorded_envelope(name, precision, out, gold):
    """the shared tolerances (1e-3, 3e-3 on *_x30) are loose where a fixture's measured error is 50x smaller, so a real regression in the
    attention / GELU paths would pass them.  Every fixture also asserts its OWN recorded error (tools/record_score_errors.py on the GPU, both modes,
    tests/golden/recorded_score_errors.json): max |matching_scores0 - reference| <= 2 x recorded + 2e-5.  The factor is what a legitimate change of
    summation order has moved single fixtures by (the round-4 GELU form: 7.1e-5 -> 1.2e-4 on trained_stats_1500x700_w2); a precision regression is 10-100x."""
    same = out["matches0"].cpu().numpy() == gold["matches0"]
    d = np.abs(out["matching_scores0"].cpu().numpy() - gold["matching_scores0"])
    err = float(d[same].max()) if same.any() else 0.0
    assert err <= 2.0 * RECORDED[name][precision] + 2e-5, f"{name} {precision}: max |dscore| {err:.3e} vs recorded {RECORDED[name][precision]:.3e}"


@pytest.mark.parametrize("name", golden_names())
def test_fp32_mode_matches_golden_exactly(name):
    require_gpu()
    case, sd, data, gold, out = run_case(name, "fp32")
    np.testing.assert_array_equal(out["matches0"].cpu().numpy(), gold["matches0"])
    np.testing.assert_array_equal(out["matches1"].cpu().numpy(), gold["matches1"])
    assert_recorded_envelope(name, "fp32", out, gold)
    # fp32 summation order differs from torch's; on the recipe-D fixtures (residual rms 27, sharp softmax rows) that alone moves
    # a score by up to 2.2e-4 (the numpy oracle itself: 1.4e-4) — still 4x inside the 1e-3 bar
    # On the confident-match fixtures (recipe E) the fp32 floor is wider still: the oracle's own float64 evaluation is up to 7.1e-4 from
    # the reference (tests/test_oracle_golden.py:oracle_score_atol), so the bar itself (1e-3) is the assertion there
    atol = score_bar(name) if name.startswith("trained_stats_confident") else 5e-4 if name.startswith("trained_stats") else 2e-4
    np.testing.assert_allclose(out["matching_scores0"].cpu().numpy(), gold["matching_scores0"], atol=atol, rtol=0)
    np.testing.assert_allclose(out["matching_scores1"].cpu().numpy(), gold["matching_scores1"], atol=atol, rtol=0)
    stop = out["stop"] if not torch.is_tensor(out["stop"]) else out["stop"].cpu().tolist()
    assert np.atleast_1d(stop).tolist() == gold["stop"].tolist()
    np.testing.assert_array_equal(out["prune0"].cpu().numpy().astype(np.float32), gold["prune0"])
    np.testing.assert_array_equal(out["prune1"].cpu().numpy().astype(np.float32), gold["prune1"])
    # output dict contract (ref :619-629)
    assert out["matches0"].dtype == torch.int64 and out["matching_scores0"].dtype == torch.float32
    assert [int(x.shape[0]) for x in out["matches"]] == gold["n_matches"].tolist()
    for b, ml in enumerate(out["matches"]):
        ml = ml.cpu().numpy()
        assert ml.dtype == np.int64 and (np.diff(ml[:, 0]) > 0).all()          # sorted by index0 (ref :596)
        np.testing.assert_array_equal(ml[:, 1], gold["matches0"][b][ml[:, 0]])
    pruning = case["conf"].get("width_confidence", 0.99) > 0
    if case.get("static_lengths") and max(case["n"], case["m"]) <= max(case["static_lengths"]):
        pruning = False                                                        # ref :529
    assert out["prune0"].dtype == (torch.int64 if pruning else torch.float32)  # ref :535 vs :616


@pytest.mark.parametrize("name", golden_names())
def test_default_precision_parity(name):
    """The default precision (f16x3, split attention) against the reference's fixtures — recipe D included — on BOTH image
    sides: scores within 1e-3, every index flip traced to an oracle-side decision boundary within 2 x tolerance in score space;
    stop layers and both prune counters identical."""
    require_gpu()
    case, sd, data, gold, out = run_case(name, "f16x3")
    flips = assert_parity_with_explained_flips(out, gold, case, sd, data, score_tol=score_bar(name))
    assert flips == (0, 0), f"index mismatches in the default precision: {flips}"   # round 3: not a single flip on any fixture
    assert_recorded_envelope(name, "f16x3", out, gold)
    stop = out["stop"] if not torch.is_tensor(out["stop"]) else out["stop"].cpu().tolist()
    assert np.atleast_1d(stop).tolist() == gold["stop"].tolist()
    np.testing.assert_array_equal(out["prune0"].cpu().numpy().astype(np.float32), gold["prune0"])
    np.testing.assert_array_equal(out["prune1"].cpu().numpy().astype(np.float32), gold["prune1"])


@pytest.mark.parametrize("name", golden_names())
def test_fast_attention_opt_in(name):
    """precision f16x3 with attention_precision fp16 (one f16 plane for q / k / v, two MFMAs per product in their projections —
    the round-2 default's arithmetic, ~25 % faster).  On the diffuse-attention fixtures (recipes A - C) it holds the full bar.
    On the trained-statistics fixtures it does NOT: an f16 logit of magnitude 30 is off by 1e-2.  The measured envelope is
    asserted so that the limit of the opt-in is documented: scores within 5e-2 where the index agrees, at most 1 % index flips."""
    require_gpu()
    case, sd, data, gold, out = run_case(name, "f16x3/fp16")
    if not name.startswith("trained_stats"):
        assert_parity_with_explained_flips(out, gold, case, sd, data)
        stop = out["stop"] if not torch.is_tensor(out["stop"]) else out["stop"].cpu().tolist()
        assert np.atleast_1d(stop).tolist() == gold["stop"].tolist()
        return
    # the confident-match fixtures (recipe E) put scores where d score / d logit is largest: the same operand error shows as up to 1.3e-1
    # in the emulation (profiles/r04_attn_switch_study.md) — asserted as an envelope of 2.5e-1 / 2 % flips, i.e. "do not use this mode here"
    # (measured, round 4: 7.4e-2 ... 1.6e-1 at descriptor scale 1; at x10 / x30 the scores are not usable at all — 0.4 / 1.0 — only the flip rate is asserted)
    confident = name.startswith("trained_stats_confident")
    scaled_up = name.endswith("_x10") or name.endswith("_x30")
    for side in (0, 1):
        m = out[f"matches{side}"].cpu().numpy(); sc = out[f"matching_scores{side}"].cpu().numpy()
        same = m == gold[f"matches{side}"]
        assert (~same).mean() <= (0.02 if confident else 0.01)
        if not scaled_up:
            assert np.abs(sc - gold[f"matching_scores{side}"])[same].max(initial=0.0) <= (2.5e-1 if confident else 5e-2)


def test_fp16_mode_envelope_on_the_pruning_config():
    """BASELINE cfg #5 names fp16.  precision='fp16' (single f16 MFMA per product) does NOT hold the 1e-3 bar — the measured
    envelope on the cfg #5 fixture is asserted here so that it is documented, not hidden: scores within 3e-2, at most 1 % index
    flips, stop layers identical.  (The parity-holding mode for this config is f16x3, covered above.)"""
    require_gpu()
    case, sd, data, gold, out = run_case("aliked128_2048x512_prune1536", "fp16")
    m0 = out["matches0"].cpu().numpy()
    assert (m0 != gold["matches0"]).mean() <= 0.01
    both = (m0 == gold["matches0"])
    assert np.abs(out["matching_scores0"].cpu().numpy() - gold["matching_scores0"])[both].max(initial=0.0) <= 3e-2
    stop = out["stop"] if not torch.is_tensor(out["stop"]) else out["stop"].cpu().tolist()
    assert np.atleast_1d(stop).tolist() == gold["stop"].tolist()


def test_all_points_pruned_ends_the_pair_like_the_reference():
    """An image whose points are ALL pruned at layer i: the reference leaves its loop at the next iteration's empty guard
    (ref :539-540) -> stop = i + 2, empty matches, prune counters untouched after that layer.  Later layers must not
    replay the compaction of the layer that emptied the image."""
    require_gpu()
    sd = synth.make_state_dict(0, recipe="B")
    sd = {k: v.copy() for k, v in sd.items()}
    for i in range(9):
        sd[f"log_assignment.{i}.matchability.bias"] -= np.float32(200.0)
    kw = dict(depth_confidence=-1, pruning_min_kpts=-1)
    for B in (1, 3):
        data = synth.make_batch(5, B, 300, 260)
        ref = O.forward(sd, O.make_conf(**kw), data)
        assert ref["stop"] == [2] * B and all(len(m) == 0 for m in ref["matches"])
        model = gpu_util.make_model(sd, "fp32", **kw)
        out = model(gpu_util.to_torch(data))
        stop = [int(out["stop"])] if B == 1 else out["stop"].cpu().tolist()
        assert stop == ref["stop"]
        np.testing.assert_array_equal(out["prune0"].cpu().numpy(), ref["prune0"])
        np.testing.assert_array_equal(out["prune1"].cpu().numpy(), ref["prune1"])
        assert (out["matches0"] == -1).all() and (out["matches1"] == -1).all() and all(x.shape[0] == 0 for x in out["matches"])


def test_image_size_is_broadcast_over_the_batch():
    """ref :35-42 broadcasts a [1, 2] (or [2]) image_size over B; the engine reads B rows."""
    require_gpu()
    sd = synth.make_state_dict(0, recipe="A")
    model = gpu_util.make_model(sd, "fp32", depth_confidence=-1, width_confidence=-1)
    t = gpu_util.to_torch(synth.make_batch(31, 3, 200, 180))
    full = model(t)
    for shape in ((1, 2), (2,)):
        t2 = {k: dict(v) for k, v in t.items()}
        for k in t2:
            t2[k]["image_size"] = t[k]["image_size"][0].reshape(shape)
        out = model(t2)
        assert torch.equal(out["matches0"], full["matches0"]) and torch.equal(out["matching_scores0"], full["matching_scores0"])


def test_nan_descriptors_do_not_fault_and_do_not_leak_into_other_pairs():
    """NaN in one pair's descriptors: the reference returns garbage for that pair but does not crash; here the argmax
    sentinel of an all-NaN row must not be used as an index, and the other pairs must be untouched."""
    require_gpu()
    sd = synth.make_state_dict(0, recipe="A")
    model = gpu_util.make_model(sd, "f16x3", depth_confidence=-1, width_confidence=-1)
    t = gpu_util.to_torch(synth.make_batch(41, 3, 300, 280))
    clean = model(t)
    t["image0"]["descriptors"][1] = float("nan")
    out = model(t)
    torch.cuda.synchronize()
    for b in (0, 2):
        assert torch.equal(out["matches0"][b], clean["matches0"][b]) and torch.equal(out["matching_scores0"][b], clean["matching_scores0"][b])
    assert int(out["matches"][1].shape[0]) == 0 and (out["matches0"][1] == -1).all()


def test_legacy_named_checkpoint_from_the_weights_directory():
    """SURVEY §8 f1 end to end on the GPU: a released-style .pth (self_attn.{i}.* / cross_attn.{i}.* keys, ref :427-434) in the
    package's weights/ directory, loaded through LightGlue(features=None, weights=...) (ref :422-425), must reproduce the golden."""
    require_gpu()
    import os
    from pathlib import Path
    import lightglue_amd
    meta, gold = load_golden("nonadaptive_512")
    sd, data = make_golden.case_inputs(meta["case"])
    legacy = {}
    for k, v in sd.items():
        for i in range(9):
            k = k.replace(f"transformers.{i}.self_attn", f"self_attn.{i}").replace(f"transformers.{i}.cross_attn", f"cross_attn.{i}")
        legacy[k] = torch.from_numpy(v)
    assert any(k.startswith("cross_attn.8.") for k in legacy)
    wdir = Path(lightglue_amd.__file__).parent / "weights"
    wdir.mkdir(exist_ok=True)
    name = f"_test_legacy_{os.getpid()}"
    path = wdir / f"{name}.pth"
    torch.save(legacy, str(path))
    try:
        model = lightglue_amd.LightGlue(features=None, weights=name, precision="fp32", **meta["case"]["conf"]).eval()
    finally:
        path.unlink()
        if not any(wdir.iterdir()):
            wdir.rmdir()
    out = model(gpu_util.to_torch(data))
    np.testing.assert_array_equal(out["matches0"].cpu().numpy(), gold["matches0"])
    np.testing.assert_array_equal(out["matches1"].cpu().numpy(), gold["matches1"])
    np.testing.assert_allclose(out["matching_scores0"].cpu().numpy(), gold["matching_scores0"], atol=2e-4, rtol=0)


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("precision,tol", [("fp32", 2e-5), ("f16x3", 5e-4), ("f16x3/fp16", 5e-4), ("fp16", 2e-2), ("bf16", 1e-1)])
def test_pipeline_stages_layer0(precision, tol, fused):
    """Every kernel of layer 0 against the oracle's intermediate tensors (err relative to rms); once with
    the unfused per-op kernels (every intermediate is observable) and once with the fused block tail."""
    require_gpu()
    sd = synth.make_state_dict(0, recipe="A")
    data = synth.make_batch(7, 2, 200, 160)
    res = gpu_util.stage_errors(sd, data, precision, dict(depth_confidence=-1, width_confidence=-1), fused=fused)
    bad = {k: v for k, v in res.items() if not (v[1] <= (tol if not k.startswith(("self.q", "self.k", "self.v", "cross.qk", "cross.v")) else max(tol, {"fp32": 2e-5, "f16x3": 5e-5, "f16x3/fp16": 3e-3, "fp16": 5e-3, "bf16": 5e-2}[precision])))}   # q / k / v in the attention's operand precision
    assert not bad, bad


def test_unfused_path_matches_golden():
    """The per-op kernels (fused_tail = 0) stay a valid product configuration."""
    require_gpu()
    meta, gold = load_golden("nonadaptive_512")
    sd, data = make_golden.case_inputs(meta["case"])
    model = gpu_util.make_model(sd, "fp32", **meta["case"]["conf"])
    model.set_option("fused_tail", 0)
    out = model(gpu_util.to_torch(data))
    np.testing.assert_array_equal(out["matches0"].cpu().numpy(), gold["matches0"])
    np.testing.assert_allclose(out["matching_scores0"].cpu().numpy(), gold["matching_scores0"], atol=2e-4, rtol=0)


def test_adaptive_batch_against_oracle_loop():
    """Batch extension (every pair stops / prunes on its own) against the oracle's loop of B = 1 calls, mixed sizes,
    fp32 mode: indices, stop layers and prune counters must be identical pair by pair."""
    require_gpu()
    sd = synth.make_state_dict(0, recipe="B")
    data = synth.make_batch(321, 3, 420, 380)
    conf = O.make_conf(pruning_min_kpts=-1)
    ref = O.forward(sd, conf, data)
    model = gpu_util.make_model(sd, "fp32", pruning_min_kpts=-1)
    out = model(gpu_util.to_torch(data))
    np.testing.assert_array_equal(out["matches0"].cpu().numpy(), ref["matches0"])
    np.testing.assert_array_equal(out["matches1"].cpu().numpy(), ref["matches1"])
    assert out["stop"].cpu().tolist() == ref["stop"]
    np.testing.assert_array_equal(out["prune0"].cpu().numpy(), ref["prune0"])
    np.testing.assert_array_equal(out["prune1"].cpu().numpy(), ref["prune1"])
    np.testing.assert_allclose(out["matching_scores0"].cpu().numpy(), ref["matching_scores0"], atol=2e-4, rtol=0)
    assert len(set(ref["stop"])) >= 1


def test_three_layer_model_and_no_image_size():
    """n_layers != 9 and bounding-box normalisation (no image_size, ref :35-36) against the oracle."""
    require_gpu()
    sd = synth.make_state_dict(11, recipe="A", n_layers=3)
    data = synth.make_batch(77, 2, 150, 170)
    for k in ("image0", "image1"):
        data[k].pop("image_size")
    conf = O.make_conf(n_layers=3, depth_confidence=-1, width_confidence=-1)
    ref = O.forward(sd, conf, data)
    model = gpu_util.make_model(sd, "fp32", n_layers=3, depth_confidence=-1, width_confidence=-1)
    out = model(gpu_util.to_torch(data))
    np.testing.assert_array_equal(out["matches0"].cpu().numpy(), ref["matches0"])
    np.testing.assert_allclose(out["matching_scores0"].cpu().numpy(), ref["matching_scores0"], atol=2e-4, rtol=0)
    assert (out["prune0"] == 3).all()


def _ragged_reference(sd, conf, data, counts0, counts1):
    """The contract of a ragged batch: pair b == a separate B = 1 oracle call on its first counts[b] rows."""
    refs = []
    d0, d1 = data["image0"], data["image1"]
    size = lambda d, b: np.asarray(d["image_size"])[b] if "image_size" in d else None
    for b, (c0, c1) in enumerate(zip(counts0, counts1)):
        refs.append(O.forward_pair(sd, conf, d0["keypoints"][b][:c0], d1["keypoints"][b][:c1],
                                   d0["descriptors"][b][:c0], d1["descriptors"][b][:c1], size(d0, b), size(d1, b)))
    return refs


@pytest.mark.parametrize("mode", ["nonadaptive", "adaptive", "bbox"])
def test_ragged_batch_equals_per_pair_calls(mode):
    """num_keypoints per pair (incl. a 1-point image, an exact multiple of the tile size and an
    EMPTY image inside the batch) — every pair must equal its own B = 1 oracle call; padding rows give -1 / 0."""
    require_gpu()
    counts0, counts1 = [300, 128, 1, 0, 257], [200, 333, 5, 77, 256]
    nmax0, nmax1 = 320, 333
    adaptive = mode == "adaptive"
    sd = synth.make_state_dict(0, recipe="B" if adaptive else "A")
    data = synth.make_batch(99, len(counts0), nmax0, nmax1)
    kw = dict(pruning_min_kpts=64) if adaptive else dict(depth_confidence=-1, width_confidence=-1)
    if mode == "bbox":
        for k in ("image0", "image1"):
            data[k].pop("image_size")
    refs = _ragged_reference(sd, O.make_conf(**kw), data, counts0, counts1)
    tdata = gpu_util.to_torch(data)
    # poison the padding rows: the engine must never read them
    for k, counts in (("image0", counts0), ("image1", counts1)):
        for b, c in enumerate(counts):
            tdata[k]["keypoints"][b, c:] = float("nan")
            tdata[k]["descriptors"][b, c:] = float("nan")
        tdata[k]["num_keypoints"] = torch.tensor(counts, dtype=torch.int32)
    model = gpu_util.make_model(sd, "fp32", **kw)
    out = model(tdata)
    stops = out["stop"].cpu().tolist()
    for b, ref in enumerate(refs):
        c0, c1 = counts0[b], counts1[b]
        m0, m1 = out["matches0"][b].cpu().numpy(), out["matches1"][b].cpu().numpy()
        np.testing.assert_array_equal(m0[:c0], ref["matches0"]); np.testing.assert_array_equal(m1[:c1], ref["matches1"])
        assert (m0[c0:] == -1).all() and (m1[c1:] == -1).all()
        s0 = out["matching_scores0"][b].cpu().numpy()
        np.testing.assert_allclose(s0[:c0], ref["matching_scores0"], atol=2e-4, rtol=0)
        assert (s0[c0:] == 0).all()
        assert stops[b] == ref["stop"], (b, stops[b], ref["stop"])
        np.testing.assert_array_equal(out["prune0"][b].cpu().numpy()[:c0], ref["prune0"])
        np.testing.assert_array_equal(out["prune1"][b].cpu().numpy()[:c1], ref["prune1"])
        assert (out["prune0"][b].cpu().numpy()[c0:] == 0).all()
        np.testing.assert_array_equal(out["matches"][b].cpu().numpy(), ref["matches"])
    if adaptive:
        assert any(r["prune0"].min(initial=99) < r["stop"] for r in refs if len(r["prune0"])), "fixture never pruned"
    assert refs[3]["stop"] == 1 and stops[3] == 1   # empty image: ref :539-540


def test_match_batch_helper_trims_to_own_counts():
    require_gpu()
    sd = synth.make_state_dict(0, recipe="A")
    model = gpu_util.make_model(sd, "f16x3", depth_confidence=-1, width_confidence=-1)
    from lightglue_amd import match_batch
    feats0, feats1, singles = [], [], []
    for b, (c0, c1) in enumerate([(140, 90), (64, 200)]):
        d = gpu_util.to_torch(synth.make_batch(5 + b, 1, c0, c1))
        feats0.append(d["image0"]); feats1.append(d["image1"])
        singles.append(model(d))
    res = match_batch(model, feats0, feats1)
    for r, one in zip(res, singles):
        assert torch.equal(r["matches0"], one["matches0"][0]) and torch.equal(r["matches1"], one["matches1"][0])
        assert torch.allclose(r["matching_scores0"], one["matching_scores0"][0], atol=1e-5)
        assert torch.equal(r["matches"], one["matches"][0]) and r["stop"] == one["stop"]


@pytest.mark.parametrize("adaptive", [False, True])
def test_full_log_assignment_output(adaptive):
    """optional [B, M+1, N+1] log-assignment incl. dustbins (ref :265-277) in original index space;
    pruned keypoints hold -inf.  Checked entry by entry against the oracle's matrix."""
    require_gpu()
    sd = synth.make_state_dict(0, recipe="B" if adaptive else "A")
    kw = dict(pruning_min_kpts=64) if adaptive else dict(depth_confidence=-1, width_confidence=-1)
    data = synth.make_batch(7, 2, 333, 290)
    model = gpu_util.make_model(sd, "fp32", **kw)
    model.return_log_assignment = True
    out = model(gpu_util.to_torch(data))
    la = out["log_assignment"].cpu().numpy()
    assert la.shape == (2, 334, 291)
    conf = O.make_conf(**kw)
    for b in range(2):
        tr = {}
        d0, d1 = data["image0"], data["image1"]
        O.forward_pair(sd, conf, d0["keypoints"][b], d1["keypoints"][b], d0["descriptors"][b], d1["descriptors"][b],
                       d0["image_size"][b], d1["image_size"][b], trace=tr)
        full, i0, i1 = tr["scores_full"], tr["ind0"], tr["ind1"]
        if adaptive:
            assert len(i0) < 333 or len(i1) < 290, "fixture never pruned"
        r = np.concatenate([i0, [333]]); c = np.concatenate([i1, [290]])
        # entries reach -150: fp32 dot products / LSEs carry a RELATIVE error, the match-relevant ones (near 0) 2e-4 abs
        np.testing.assert_allclose(la[b][np.ix_(r, c)], full, atol=2e-4, rtol=1e-4)
        dead = np.ones((334, 291), bool); dead[np.ix_(r, c)] = False
        assert np.isneginf(la[b][dead]).all()
        assert la[b][333, 290] == 0.0


@pytest.mark.parametrize("precision,tol", [("fp32", 2e-4), ("f16x3", 2e-3), ("f16x3/fp16", 5e-2)])
def test_attention_deferred_rescale_with_sharp_logits(precision, tol):
    """The attention kernel rescales its running output only when a row's maximum grows by more than 2^8 (deferred
    rescale).  Random weights never exercise that branch after the first tile, so sharpen layer 0's logits (Wqkv x 60:
    base-2 logit spread ~7): the oracle's own q/k must then show BOTH rows whose maximum jumps past the threshold in a later
    64-key tile and rows that stay below it, and the attention context must still match the oracle."""
    require_gpu()
    sd = synth.make_state_dict(0, recipe="A")
    sd = {k: v.copy() for k, v in sd.items()}
    sd["transformers.0.self_attn.Wqkv.weight"] *= 60.0
    data = synth.make_batch(5, 1, 320, 280)
    conf_kw = dict(depth_confidence=-1, width_confidence=-1)
    tr = {"_full_layers": (0,)}
    d0, d1 = data["image0"], data["image1"]
    O.forward_pair(sd, O.make_conf(**conf_kw), d0["keypoints"][0], d1["keypoints"][0], d0["descriptors"][0], d1["descriptors"][0],
                   d0["image_size"][0], d1["image_size"][0], trace=tr)
    q, k = tr["l0_self0_q"], tr["l0_self0_k"]                       # [H, n, 64]
    logits2 = np.einsum("hqd,hkd->hqk", q, k) * 0.125 * 1.4426950408889634   # base-2 units, as the kernel sees them
    tile_max = np.stack([logits2[:, :, i:i + 64].max(-1) for i in range(0, logits2.shape[-1], 64)], -1)
    run = np.maximum.accumulate(tile_max, -1)
    jumps = tile_max[..., 1:] - run[..., :-1]
    assert (jumps > 8).any() and ((jumps > 0) & (jumps <= 8)).any(), "fixture must exercise both sides of the threshold"
    res = gpu_util.stage_errors(sd, data, precision, conf_kw, fused=True)
    assert res["self.attn_ctx"][1] <= tol, res["self.attn_ctx"]


@pytest.mark.parametrize("precision", ["f16x3", "f16x3/fp16", "bf16", "fp16"])
def test_fused_next_projection_is_bit_identical(precision):
    """The tail kernel runs the next block's q/k/v projection on the x tile it has just produced (engine option
    fused_next, default on): same arithmetic on the same fp32 values as the standalone projection kernel, so every
    output must be BIT-identical with the option off — non-adaptive (both block boundaries fused) and adaptive
    (only SelfBlock -> CrossBlock fused, rows move between layers)."""
    require_gpu()
    # third case: early stop on, pruning enabled but every image below the threshold (1536 by default) -> rows never move, the
    # cross -> self projection is fused speculatively across the stop decision
    for recipe, kw in (("A", dict(depth_confidence=-1, width_confidence=-1)), ("B", dict(pruning_min_kpts=64)), ("C", dict())):
        sd = synth.make_state_dict(0, recipe=recipe)
        data = gpu_util.to_torch(synth.make_batch(17, 3, 300, 333))
        model = gpu_util.make_model(sd, precision, **kw)
        fused = model(data)
        model.set_option("fused_next", 0)
        plain = model(data)
        for key in ("matches0", "matches1", "matching_scores0", "matching_scores1", "prune0", "prune1"):
            assert torch.equal(fused[key], plain[key]), (recipe, key)
        assert torch.equal(torch.as_tensor(fused["stop"]), torch.as_tensor(plain["stop"]))


@pytest.mark.parametrize("precision", ["f16x3", "fp32", "bf16"])
def test_fused_prep_is_bit_identical(precision):
    """The first SelfBlock projection launch also does the per-keypoint preparation (normalisation, Fourier rotary rows, index set, descriptor rows ->
    residual stream; engine option fused_prep, default on, input_dim == 256): the same expressions in the same order as prep_kernel + proj_kernel, so
    every output must be BIT-identical with the option off — image_size given and absent (bounding boxes), ragged counts incl. an empty image,
    scale / orientation inputs (4-D positional encoding), adaptive depth / width."""
    require_gpu()
    cases = [("A", dict(depth_confidence=-1, width_confidence=-1), (31, 3, 300, 333), dict()),
             ("A", dict(depth_confidence=-1, width_confidence=-1), (32, 2, 200, 129), dict(drop_size=True)),
             ("C", dict(pruning_min_kpts=64), (33, 3, 260, 200), dict(nums=([260, 77, 0], [200, 5, 130]))),
             ("A", dict(depth_confidence=-1, width_confidence=-1, add_scale_ori=True), (34, 2, 150, 170), dict(sift=True))]
    for recipe, kw, (seed, B, n, m), opt in cases:
        sd = synth.make_state_dict(0, recipe=recipe, add_scale_ori=bool(opt.get("sift")))
        data = gpu_util.to_torch(synth.make_batch(seed, B, n, m, add_scale_ori=bool(opt.get("sift"))))
        if opt.get("drop_size"):
            for k in ("image0", "image1"):
                data[k].pop("image_size")
        if "nums" in opt:
            data["image0"]["num_keypoints"] = torch.as_tensor(opt["nums"][0], dtype=torch.int32, device="cuda")
            data["image1"]["num_keypoints"] = torch.as_tensor(opt["nums"][1], dtype=torch.int32, device="cuda")
        model = gpu_util.make_model(sd, precision, **kw)
        on = model(data)
        model.set_option("fused_prep", 0)
        off = model(data)
        for key in ("matches0", "matches1", "matching_scores0", "matching_scores1", "prune0", "prune1"):
            assert torch.equal(on[key], off[key]), (recipe, key, opt)
        assert torch.equal(torch.as_tensor(on["stop"]), torch.as_tensor(off["stop"]))


@pytest.mark.parametrize("precision", ["f16x3", "f16x3/fp16", "bf16"])
def test_attention_rows_per_wave_variants_are_equivalent(precision):
    """Engine option attn_rows = 64 / 16 (four / one 16-row query tiles per wave; 256-row workgroup tiles are laid out
    per segment): same arithmetic per query row as the default 32-row kernel, so the outputs must be bit-identical — including
    capacities that are not multiples of 256 (tiles overhang their segment and, for the last one, the row space).  (The split
    attention has the 32- and 16-row shapes; 64 maps to 32 there.)"""
    require_gpu()
    for (n0, n1, recipe, kw) in ((300, 333, "A", dict(depth_confidence=-1, width_confidence=-1)), (130, 520, "B", dict(pruning_min_kpts=64)),
                                 (1024, 1024, "A", dict(depth_confidence=-1, width_confidence=-1)), (700, 900, "D", dict(depth_confidence=-1, width_confidence=-1))):
        sd = synth.make_state_dict(0, recipe=recipe)
        data = gpu_util.to_torch(synth.make_batch(23, 2, n0, n1, **(synth.RECIPE_D_DATA if recipe == "D" else {})))
        model = gpu_util.make_model(sd, precision, **kw)
        base = model(data)
        for rows in (64, 16):
            model.set_option("attn_rows", rows)
            other = model(data)
            for key in ("matches0", "matches1", "matching_scores0", "matching_scores1", "prune0", "prune1"):
                assert torch.equal(base[key], other[key]), (n0, n1, rows, key)
        model.set_option("attn_rows", 32)


@pytest.mark.parametrize("precision", ["f16x3", "f16x3/fp16", "bf16"])
def test_attention_lds_dma_kernel_is_bit_identical(precision):
    """The default attention kernel brings K / V^T tiles into LDS by DMA (option attn_dma, default on; two buffers, one
    barrier per tile); with the option off the register-staged kernel runs.  Same arithmetic in the same order, so all outputs
    must be BIT-identical: key counts that end mid-tile, cross attention between unequal sets, one-tile key sets, adaptive
    runs with compaction — and a workspace that a previous call left full of NaN (a partial tile's dead V^T columns are
    fixed up in LDS, they must never reach the MFMA).  The split attention (f16x3) is always the DMA kernel: for it this test
    is the NaN-poisoned-workspace check of its four-plane tile fix-up."""
    require_gpu()
    for (n0, n1, recipe, kw) in ((300, 333, "A", dict(depth_confidence=-1, width_confidence=-1)), (130, 520, "B", dict(pruning_min_kpts=64)),
                                 (1024, 1024, "A", dict(depth_confidence=-1, width_confidence=-1)), (40, 700, "C", dict())):
        sd = synth.make_state_dict(0, recipe=recipe)
        model = gpu_util.make_model(sd, precision, **kw)
        model.check_finite = False                         # the poisoning forward feeds NaN on purpose (the default guard would raise on a model's first forward)
        poison = gpu_util.to_torch(synth.make_batch(5, 2, max(n0, 384), max(n1, 384)))
        poison["image0"]["descriptors"][:] = float("nan"); poison["image1"]["descriptors"][:] = float("nan")
        model(poison)                                      # every row of Q / K / V^T now holds NaN
        data = gpu_util.to_torch(synth.make_batch(23, 2, n0, n1))
        dma = model(data)
        model.set_option("attn_dma", 0)
        model(poison)
        staged = model(data)
        for key in ("matches0", "matches1", "matching_scores0", "matching_scores1", "prune0", "prune1"):
            assert torch.equal(dma[key], staged[key]), (n0, n1, key)
        assert torch.isfinite(dma["matching_scores0"]).all() and (dma["matches0"] > -1).any(), (n0, n1)


@pytest.mark.parametrize("precision", ["f16x3", "f16x3/fp16", "bf16", "fp16"])
def test_tail_row_tile_shapes_are_bit_identical(precision):
    """The fused tail runs 64-row workgroups when they fill the chip and 32- / 16-row ones for small grids (engine option
    tail_row_tiles: 0 = by grid fill, 4 | 2 | 1 = forced).  The per-row arithmetic and its order do not depend on the shape, so
    every output must be BIT-identical across shapes — fused next projection included; ragged, adaptive and batched cases."""
    require_gpu()
    for (b, n0, n1, recipe, kw) in ((3, 300, 333, "A", dict(depth_confidence=-1, width_confidence=-1)), (2, 130, 520, "B", dict(pruning_min_kpts=64)),
                                    (1, 1024, 1024, "A", dict(depth_confidence=-1, width_confidence=-1)), (2, 40, 700, "C", dict()),
                                    (2, 640, 512, "D", dict(depth_confidence=-1, width_confidence=-1))):
        sd = synth.make_state_dict(0, recipe=recipe)
        model = gpu_util.make_model(sd, precision, **kw)
        data = gpu_util.to_torch(synth.make_batch(31, b, n0, n1, **(synth.RECIPE_D_DATA if recipe == "D" else {})))
        outs = {}
        for shape in (4, 2, 1, 0):
            model.set_option("tail_row_tiles", shape)
            outs[shape] = model(data)
        for shape in (2, 1, 0):
            for key in ("matches0", "matches1", "matching_scores0", "matching_scores1", "prune0", "prune1"):
                assert torch.equal(outs[4][key], outs[shape][key]), (b, n0, n1, shape, key)
            assert torch.equal(torch.as_tensor(outs[4]["stop"]), torch.as_tensor(outs[shape]["stop"]))


def test_deferred_forward_gives_the_same_dict():
    """forward_deferred: the host synchronisation of forward i is taken after forward i + 1 has been enqueued; the handles'
    results must equal the synchronous forwards' outputs (ragged lists included), in issue order and in reverse."""
    require_gpu()
    sd = synth.make_state_dict(0, recipe="C")
    model = gpu_util.make_model(sd, "f16x3")
    batches = [gpu_util.to_torch(synth.make_batch(41 + i, 2, 300 + 40 * i, 333)) for i in range(3)]
    want = [model(d) for d in batches]
    handles = [model.forward_deferred(d) for d in batches]
    got = [h.result() for h in reversed(handles)][::-1]
    for w, g in zip(want, got):
        for key in ("matches0", "matches1", "matching_scores0", "matching_scores1", "prune0", "prune1"):
            assert torch.equal(w[key], g[key]), key
        assert torch.equal(torch.as_tensor(w["stop"]), torch.as_tensor(g["stop"]))
        assert len(w["matches"]) == len(g["matches"]) and all(torch.equal(a, b) for a, b in zip(w["matches"], g["matches"]))
        assert all(torch.equal(a, b) for a, b in zip(w["scores"], g["scores"]))
    assert handles[0].result() is got[0]      # cached


def test_plain_bf16_mismatch_rate_is_reported_not_hidden():
    """precision='bf16' (single bf16 MFMA everywhere) is the fast mode; it does NOT hold the 1e-3 bar.
    Assert the documented envelope: <= 2 % index flips, |dscore| <= 0.5."""
    require_gpu()
    case, sd, data, gold, out = run_case("nonadaptive_512", "bf16")
    m0 = out["matches0"].cpu().numpy()
    flips = (m0 != gold["matches0"]).mean()
    assert flips <= 0.02, flips
    assert np.abs(out["matching_scores0"].cpu().numpy() - gold["matching_scores0"]).max() <= 0.5
