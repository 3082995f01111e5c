"""GPU tests of the round-6 additions: the call envelope (include/lightglue_amd.h LG_MAX_*), the range guard's new default ("first": on for the first forward after
the weights changed), and the largest shape inside the envelope (N = M = 8192) against the pinned oracle, fixed depth and adaptive."""
import numpy as np
import pytest
import torch

import gpu_util
from conftest import SCORE_TOL, assert_parity_with_explained_flips, require_gpu
from lightglue_amd import _cabi
from lightglue_amd import synthetic as synth
from oracle import lightglue_oracle as O

pytestmark = pytest.mark.gpu


def _empty_batch(B, n, m, dim=256):
    e = lambda *shape: torch.empty(shape, device="cuda", dtype=torch.float32)
    return {"image0": {"keypoints": e(B, n, 2), "descriptors": e(B, n, dim), "image_size": e(B, 2)},
            "image1": {"keypoints": e(B, m, 2), "descriptors": e(B, m, dim), "image_size": e(B, 2)}}


def test_calls_outside_the_envelope_are_refused():
    """B * (cap0 + cap1) beyond 2^21 rows (the compaction's 32-bit byte offsets and 4 GB buffer descriptors) used to be silent corruption;
    now the forward — and reserve — refuse it with a message, and the engine stays usable."""
    require_gpu()
    sd = synth.make_state_dict(0, recipe="C")
    model = gpu_util.make_model(sd, "f16x3")
    with pytest.raises(AssertionError, match="LG_MAX_ROWS"):
        model(_empty_batch(1100, 1024, 1024))            # 1100 x 2048 = 2.25 M rows
    with pytest.raises(AssertionError, match="LG_MAX_KEYPOINTS"):
        model(_empty_batch(1, 8320, 64))
    with pytest.raises(AssertionError, match="LG_MAX_SIM_ELEMS"):
        model.reserve(129, 4096, 4096)
    data = gpu_util.to_torch(synth.make_batch(3, 2, 200, 160))
    out = model(data)
    assert torch.isfinite(out["matching_scores0"]).all()


def test_range_guard_is_on_for_the_first_forward_after_a_weight_change():
    """check_finite == "first" (the default): an input scale / checkpoint that leaves the f16 operand range raises on the FIRST forward after load_state_dict
    (and after any later weight change), costs nothing afterwards, and changes no output bit."""
    require_gpu()
    sd = synth.make_state_dict(0, recipe="A")
    data = synth.make_batch(11, 2, 256, 192)
    bad = {k: {kk: vv.copy() for kk, vv in v.items()} for k, v in data.items()}
    bad["image0"]["descriptors"][1] *= np.float32(1e6)
    model = gpu_util.make_model(sd, "f16x3", depth_confidence=-1, width_confidence=-1)
    assert model.check_finite == "first"
    with pytest.raises(_cabi.LightGlueAmdError, match="pair 1"):
        model(gpu_util.to_torch(bad))
    model(gpu_util.to_torch(bad))                          # second forward: guard off (the documented default), no raise
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    with pytest.raises(_cabi.LightGlueAmdError, match="pair 1"):
        model(gpu_util.to_torch(bad))                      # armed again by the weight change
    a = gpu_util.make_model(sd, "f16x3", depth_confidence=-1, width_confidence=-1)(gpu_util.to_torch(data))     # guarded forward
    never = gpu_util.make_model(sd, "f16x3", depth_confidence=-1, width_confidence=-1)
    never.check_finite = False
    b = never(gpu_util.to_torch(data))
    for k in ("matches0", "matches1", "matching_scores0", "matching_scores1"):
        assert torch.equal(a[k], b[k])


@pytest.mark.parametrize("adaptive", [False, True])
def test_largest_shape_of_the_envelope_against_the_oracle(adaptive):
    """N = M = 8192 (LG_MAX_KEYPOINTS), B = 1: scores within the bar of the pinned oracle, index flips explained; adaptive: recipe C with both images pruning
    (64 chunks per segment through the in-place compaction), stop layer and prune counters identical."""
    require_gpu()
    torch.set_num_threads(8)
    recipe = "C" if adaptive else "A"
    sd = synth.make_state_dict(1, recipe=recipe)
    data = synth.make_batch(91, 1, 8192, 8192)
    conf_kw = dict() if adaptive else dict(depth_confidence=-1, width_confidence=-1)
    ref = O.forward(sd, O.make_conf(**conf_kw), data, backend="torch")
    model = gpu_util.make_model(sd, "f16x3", **conf_kw)
    out = model(gpu_util.to_torch(data))
    torch.cuda.synchronize()
    gold = {k: np.asarray(ref[k]) for k in ("matches0", "matches1", "matching_scores0", "matching_scores1")}
    case = {"conf": conf_kw, "recipe": recipe, "wseed": 1, "dseed": 91, "n": 8192, "m": 8192, "B": 1, "dim": 256}
    flips = assert_parity_with_explained_flips(out, gold, case, sd, data, score_tol=SCORE_TOL)
    assert sum(flips) <= 4, flips
    assert int(out["stop"]) == int(np.asarray(ref["stop"]).reshape(-1)[0])
    if adaptive:
        assert (np.asarray(ref["prune0"]) < int(out["stop"])).any(), "the fixture must actually prune"
        np.testing.assert_array_equal(out["prune0"].cpu().numpy(), np.asarray(ref["prune0"]))
        np.testing.assert_array_equal(out["prune1"].cpu().numpy(), np.asarray(ref["prune1"]))


def test_verify_pretrained_on_a_released_format_checkpoint(tmp_path):
    """tools/verify_pretrained.py runs offline on a recipe-D checkpoint written with the released checkpoints' key names and reproduces the
    committed reference fixture trained_stats_1024_b8 (inputs AND reference outputs replayed) — the one-command job for whoever holds the real weights."""
    require_gpu()
    import subprocess
    import sys
    from pathlib import Path
    from test_tools import _legacy_checkpoint
    root = Path(__file__).resolve().parents[1]
    ck = _legacy_checkpoint(tmp_path / "recipeD_legacy.pth")
    p = subprocess.run([sys.executable, str(root / "tools" / "verify_pretrained.py"), str(ck), "--sizes", "512", "--pairs", "1", "--fixture", "trained_stats_1024_b8"],
                       capture_output=True, text=True, timeout=1500, cwd=str(root))
    assert p.returncode == 0, (p.stdout[-3000:], p.stderr[-2000:])
    line = [ln for ln in p.stdout.splitlines() if "fixture trained_stats_1024_b8" in ln][0]
    assert "| 0 / 0 |" in line and line.rstrip().endswith("| ok |"), line
    assert "RESULT: inside the bar" in p.stdout


@pytest.mark.parametrize("precision", ["f16x3", "f16x3/fp16", "fp32"])
def test_gather_projection_equals_the_in_place_compaction(precision):
    """Adaptive width without a compaction launch (engine option adapt_gather, default on): the SelfBlock projection behind a pruning step gathers its rows
    through the decide kernel's inverse index map into a second buffer set.  The same fp32 rows reach the same tile positions as after the in-place
    compaction kernel, so every output — indices, scores, stop layers, prune counters — must be BIT-identical with the option off: both images pruning,
    only one image above the threshold, ragged counts with an empty image, pairs stopping at different layers (their rows stay in the buffer set they
    were in: per-pair buffer select of the final projection), pruning without early stop, and the full log-assignment side output."""
    require_gpu()
    cases = [("C", dict(pruning_min_kpts=64), (17, 3, 300, 333), {}),
             ("C", dict(), (201, 2, 2048, 2048), {}),                                   # cfg #3's shape: mixed stop depths, both images above 1536
             ("C", dict(pruning_min_kpts=256), (33, 3, 260, 700), dict(nums=([260, 77, 0], [700, 5, 130]))),
             ("B", dict(pruning_min_kpts=64, depth_confidence=-1), (41, 2, 500, 400), {}),   # pruning only: the last tail runs the final projection itself
             ("C", dict(pruning_min_kpts=512), (55, 2, 700, 400), dict(log_assignment=True)),
             ("D", dict(pruning_min_kpts=64), (61, 2, 640, 512), dict(recipe_d_data=True))]
    for recipe, kw, (seed, B, n, m), opt in cases:
        sd = synth.make_state_dict(0, recipe=recipe)
        data = gpu_util.to_torch(synth.make_batch(seed, B, n, m, **(synth.RECIPE_D_DATA if opt.get("recipe_d_data") else {})))
        if "nums" in opt:
            data["image0"]["num_keypoints"] = torch.as_tensor(opt["nums"][0], dtype=torch.int32, device="cuda")
            data["image1"]["num_keypoints"] = torch.as_tensor(opt["nums"][1], dtype=torch.int32, device="cuda")
        model = gpu_util.make_model(sd, precision, **kw)
        model.return_log_assignment = bool(opt.get("log_assignment"))
        on = model(data)
        again = model(data)                                  # the buffer sets flip inside a forward: a second forward must start from set 0 again
        model.set_option("adapt_gather", 0)
        off = model(data)
        keys = ("matches0", "matches1", "matching_scores0", "matching_scores1", "prune0", "prune1") + (("log_assignment",) if opt.get("log_assignment") else ())
        for key in keys:
            assert torch.equal(on[key], off[key]), (recipe, key, n, m)
            assert torch.equal(on[key], again[key]), (recipe, key, "second forward")
        assert torch.equal(torch.as_tensor(on["stop"]), torch.as_tensor(off["stop"]))
        if recipe == "C" and "nums" not in opt:
            assert (on["prune0"] < torch.as_tensor(on["stop"]).reshape(-1, 1).to(on["prune0"].device)).any(), "the case must actually prune"


def test_randomised_soak_of_the_gather_and_sharded_paths():
    """tools/stress_adaptive_paths.py, 14 seeded random cases (shapes, thresholds, ragged counts incl. empty images, recipes, precisions): gather projection ==
    in-place compaction, second forward == first, PairShardedMatcher's world-of-one step == forward() — all bit for bit (the 80-case run is in profiles/)."""
    require_gpu()
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    p = subprocess.run([sys.executable, str(root / "tools" / "stress_adaptive_paths.py"), "14", "7"], capture_output=True, text=True, timeout=900, cwd=str(root))
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-1500:]
    assert "all bit-identical" in p.stdout


@pytest.mark.parametrize("B,n,m,adaptive", [(1, 1, 8192, False), (1, 8192, 3, False), (2, 4100, 130, True), (96, 17, 40, False), (3, 129, 127, True)])
def test_extreme_aspect_shapes_against_the_oracle(B, n, m, adaptive):
    """Corners of the envelope the fixtures do not hold: one keypoint against 8 192, 8 192 against three (single-row / single-tile segments next to 64-tile ones),
    a segment one row past a tile boundary with pruning, many tiny pairs per batch, lengths straddling 128."""
    require_gpu()
    torch.set_num_threads(8)
    recipe = "C" if adaptive else "A"
    sd = synth.make_state_dict(2, recipe=recipe)
    data = synth.make_batch(300 + n + m, B, n, m)
    conf_kw = dict(pruning_min_kpts=64) if adaptive else dict(depth_confidence=-1, width_confidence=-1, filter_threshold=0.0 if min(n, m) < 8 else 0.1)
    ref = O.forward(sd, O.make_conf(**conf_kw), data, backend="torch")
    model = gpu_util.make_model(sd, "f16x3", **conf_kw)
    out = model(gpu_util.to_torch(data))
    gold = {k: np.asarray(ref[k]) for k in ("matches0", "matches1", "matching_scores0", "matching_scores1")}
    case = {"conf": conf_kw, "recipe": recipe, "wseed": 2, "dseed": 300 + n + m, "n": n, "m": m, "B": B, "dim": 256, **({"prune_th": 64} if adaptive else {})}
    flips = assert_parity_with_explained_flips(out, gold, case, sd, data, score_tol=SCORE_TOL)
    assert sum(flips) <= 2, flips
    stop = out["stop"] if torch.is_tensor(out["stop"]) else torch.tensor([out["stop"]])
    np.testing.assert_array_equal(stop.cpu().numpy().reshape(-1), np.asarray(ref["stop"]).reshape(-1))
    if adaptive:
        np.testing.assert_array_equal(out["prune0"].cpu().numpy(), np.asarray(ref["prune0"]))
        np.testing.assert_array_equal(out["prune1"].cpu().numpy(), np.asarray(ref["prune1"]))


def test_input_dtype_and_layout_variants_give_the_same_result():
    """The class accepts what the reference accepts (ref :483-506 takes any float tensors): float64 or non-contiguous inputs, an image_size given as a list, int64
    num_keypoints — all normalised to the engine's fp32 / int32 contiguous buffers on the way in; fp16 inputs match a forward on their fp32-widened values."""
    require_gpu()
    sd = synth.make_state_dict(0, recipe="A")
    model = gpu_util.make_model(sd, "f16x3", depth_confidence=-1, width_confidence=-1)
    data = gpu_util.to_torch(synth.make_batch(9, 2, 300, 280))
    base = model(data)
    keys = ("matches0", "matches1", "matching_scores0", "matching_scores1")
    same = lambda a: all(torch.equal(a[k], base[k]) for k in keys)
    f64 = {k: {kk: vv.double() for kk, vv in v.items()} for k, v in data.items()}
    assert same(model(f64))
    strided = {k: dict(v) for k, v in data.items()}
    for k in strided:   # [B, N, D] view of a [B, D, N] buffer: same values, permuted strides
        strided[k]["descriptors"] = data[k]["descriptors"].transpose(1, 2).contiguous().transpose(1, 2)
        assert not strided[k]["descriptors"].is_contiguous()
    assert same(model(strided))
    lists = {k: dict(v) for k, v in data.items()}
    for k in lists:
        lists[k]["image_size"] = data[k]["image_size"][0].tolist()          # one [w, h] for the whole batch (the batch shares it in this fixture)
    assert torch.equal(data["image0"]["image_size"][0], data["image0"]["image_size"][1])
    assert same(model(lists))
    counted = {k: dict(v) for k, v in data.items()}
    counted["image0"]["num_keypoints"] = torch.tensor([300, 300], dtype=torch.int64)      # on the CPU, int64: full counts == no counts
    counted["image1"]["num_keypoints"] = [280, 280]
    assert same(model(counted))
    half = {k: {kk: (vv.half() if kk == "descriptors" else vv) for kk, vv in v.items()} for k, v in data.items()}
    widened = {k: {kk: (vv.half().float() if kk == "descriptors" else vv) for kk, vv in v.items()} for k, v in data.items()}
    a, b = model(half), model(widened)
    assert all(torch.equal(a[k], b[k]) for k in keys)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model({k: {kk: vv.cpu() for kk, vv in v.items()} for k, v in data.items()})


def test_similarity_on_planes_equals_the_generic_kernel():
    """Split-f16 precision: the final projection stores f16 hi / lo planes and `sim_planes_kernel` (lg_sim.hip) multiplies them straight from an LDS-DMA ring
    (engine option sim_planes, default on) — the same split of the same fp32 values and the same MFMA sequence per output element as the generic
    `sim_kernel`, which splits fp32 rows per K stage.  Every output incl. the full log-assignment matrix must be BIT-identical with the option off: fixed depth
    (the last tail runs the final projection) and adaptive (its own launch, per-pair layer select), tile-edge and sub-tile sizes, more than one 256-row chunk
    of image-1 rows, ragged counts with an empty image, NaN-poisoned padding."""
    require_gpu()
    cases = [("A", dict(depth_confidence=-1, width_confidence=-1), (3, 2, 1024, 1024), {}),
             ("A", dict(depth_confidence=-1, width_confidence=-1), (4, 3, 130, 700), dict(log_assignment=True)),
             ("A", dict(depth_confidence=-1, width_confidence=-1), (5, 2, 5, 3), dict(log_assignment=True)),
             ("A", dict(depth_confidence=-1, width_confidence=-1), (6, 1, 257, 65), {}),
             ("A", dict(depth_confidence=-1, width_confidence=-1), (7, 1, 64, 1537), {}),
             ("A", dict(depth_confidence=-1, width_confidence=-1), (8, 3, 300, 520), dict(nums=([300, 0, 129], [520, 64, 1]), poison=True, log_assignment=True)),
             ("C", dict(pruning_min_kpts=64), (17, 3, 300, 333), dict(log_assignment=True)),
             ("C", dict(), (201, 2, 2048, 2048), {}),
             ("D", dict(depth_confidence=-1, width_confidence=-1), (61, 2, 640, 512), dict(recipe_d_data=True))]
    for recipe, kw, (seed, B, n, m), opt in cases:
        sd = synth.make_state_dict(0, recipe=recipe)
        batch = synth.make_batch(seed, B, n, m, **(synth.RECIPE_D_DATA if opt.get("recipe_d_data") else {}))
        if opt.get("poison"):
            for img, nums in zip(("image0", "image1"), opt["nums"]):
                for b, c in enumerate(nums):
                    batch[img]["keypoints"][b, c:] = np.nan
                    batch[img]["descriptors"][b, c:] = np.nan
        data = gpu_util.to_torch(batch)
        if "nums" in opt:
            data["image0"]["num_keypoints"] = torch.as_tensor(opt["nums"][0], dtype=torch.int32, device="cuda")
            data["image1"]["num_keypoints"] = torch.as_tensor(opt["nums"][1], dtype=torch.int32, device="cuda")
        model = gpu_util.make_model(sd, "f16x3", **kw)
        model.check_finite = False
        model.return_log_assignment = bool(opt.get("log_assignment"))
        on = model(data)
        model.set_option("sim_planes", 0)
        off = model(data)
        model.set_option("sim_planes", 1)
        again = model(data)
        for chunk in (64, 256, 512):                     # image-1 rows per workgroup (default: by grid fill): the arithmetic per element does not depend on it
            model.set_option("sim_chunk", chunk)
            forced = model(data)
            for key in ("matching_scores0", "matching_scores1", "matches0"):
                assert torch.equal(on[key], forced[key]), (recipe, key, n, m, chunk)
        model.set_option("sim_chunk", 0)
        keys = ("matches0", "matches1", "matching_scores0", "matching_scores1") + (("log_assignment",) if opt.get("log_assignment") else ())
        for key in keys:
            assert torch.equal(on[key], off[key]), (recipe, key, n, m)
            assert torch.equal(on[key], again[key]), (recipe, key, "second forward")
        assert torch.equal(torch.as_tensor(on["stop"]), torch.as_tensor(off["stop"]))
        assert (on["matches0"] >= 0).any() or n < 8, "the case should produce matches"


def test_constructor_space_walk_against_the_oracle():
    """tools/fuzz_configs.py: seeded random constructor arguments (input_dim 64 ... 256, add_scale_ori, n_layers 1 ... 9, filter_threshold, early stop / pruning alone and
    together, pruning thresholds around the keypoint counts, image_size absent) x random shapes against the pinned oracle — the axis the fixtures (default constructor)
    do not walk.  14 cases here (another seed than the 60 + 240 of profiles/r06q_fuzz_configs*.log)."""
    require_gpu()
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tools"))
    import fuzz_configs
    worst = fuzz_configs.run(cases=14, seed=7, verbose=False)
    assert worst <= SCORE_TOL


def test_synthetically_trained_checkpoint_when_present():
    """tools/train_synthetic_checkpoint.py trains the unmodified reference module on a synthetic matching task in the build container (47 MB: git-ignored under
    tests/golden/_local/, so this test skips on a tree without it).  Its attention is 2 - 4 x sharper than the calibrated stand-ins (per-row logit spread up to 116 against
    recipe D / E's 25).  tools/verify_pretrained.py must hold the bar on it in the default precision — synthetic keypoints and a pair of the task's own distribution, fixed
    depth and adaptive — with the oracle as the CPU side (pinned against the reference on this very checkpoint in the build container, profiles/r06tr_trained_cpu_pin.log)."""
    require_gpu()
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    ck = root / "tests" / "golden" / "_local" / "synthetic_trained_L9.pth"
    pair = ck.with_name("synthetic_task_pair_1024.npz")
    if not ck.exists():
        pytest.skip("no trained checkpoint in this tree (tools/train_synthetic_checkpoint.py, build container only)")
    p = subprocess.run([sys.executable, str(root / "tools" / "verify_pretrained.py"), str(ck)] + ([str(pair)] if pair.exists() else []) + ["--sizes", "512", "1024", "--pairs", "1"],
                       capture_output=True, text=True, timeout=1500, cwd=str(root))
    assert p.returncode == 0, (p.stdout[-3000:], p.stderr[-2000:])
    assert "RESULT: inside the bar" in p.stdout
    default = p.stdout.split("## precision f16x3, attention_precision")[0]
    rows = [ln for ln in default.splitlines() if ln.startswith("| ") and ("fixed depth" in ln or "adaptive" in ln)]
    assert len(rows) >= 4 and all("| 0 / 0 |" in ln and ln.rstrip().endswith("| ok |") for ln in rows), rows
    assert all(float(ln.split("|")[3]) <= 1e-4 for ln in rows), rows      # measured 7.9e-6: two orders inside the bar
