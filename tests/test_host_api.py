"""Host-side mirror of the reference interface (no GPU): constructor, conf, state-dict contract,
error behaviour, and the rule that there is no CPU fallback."""
import numpy as np
import pytest
import torch

from lightglue_amd import LightGlue
from oracle import lightglue_oracle as O
from lightglue_amd import synthetic as synth
import make_golden


def test_state_dict_names_and_shapes_match_reference_contract():
    # 252 tensors / 11 851 601 params for SuperPoint; +input_proj for 128-d
    m = LightGlue(features=None)
    sd = m.state_dict()
    spec = synth.state_dict_spec()
    assert [k for k in sd if k != "confidence_thresholds"] == [n for n, _, _ in spec] or \
        set(k for k in sd if k != "confidence_thresholds") == set(n for n, _, _ in spec)
    for n, shape, _ in spec:
        assert tuple(sd[n].shape) == tuple(shape), n
    assert sum(p.numel() for p in m.parameters()) == 11_851_601
    m128 = LightGlue(features=None, input_dim=128, add_scale_ori=True)
    assert tuple(m128.state_dict()["input_proj.weight"].shape) == (256, 128)
    assert tuple(m128.state_dict()["posenc.Wr.weight"].shape) == (32, 4)


@pytest.mark.skipif(not make_golden.REF.exists(), reason="reference tree not mounted")
def test_state_dict_interchangeable_with_reference_module():
    lg = make_golden.load_reference()
    ref = lg.LightGlue(features=None, input_dim=128)
    ours = LightGlue(features=None, input_dim=128)
    assert list(ref.state_dict().keys()) == list(ours.state_dict().keys())
    res = ours.load_state_dict(ref.state_dict(), strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    for k, v in ref.state_dict().items():
        assert torch.equal(v, ours.state_dict()[k])


def test_default_conf_and_thresholds():
    m = LightGlue(features=None)
    for k, v in dict(input_dim=256, descriptor_dim=256, add_scale_ori=False, n_layers=9, num_heads=4, flash=True, mp=False,
                     depth_confidence=0.95, width_confidence=0.99, filter_threshold=0.1, weights=None).items():
        assert getattr(m.conf, k) == v
    np.testing.assert_allclose(m.confidence_thresholds.numpy(), O.confidence_thresholds_f32(9), rtol=0, atol=0)
    assert m.pruning_min_kpts(torch.device("cuda")) == 1536
    assert m.pruning_min_kpts(torch.device("cpu")) == -1
    assert LightGlue(features=None, flash=False).pruning_min_kpts(torch.device("cuda")) == 1024
    assert LightGlue(features=None, pruning_min_kpts=7).pruning_min_kpts(torch.device("cuda")) == 7
    assert LightGlue.required_data_keys == ["image0", "image1"]


def test_errors_mirror_reference():
    with pytest.raises(ValueError):
        LightGlue(features="orb")
    m = LightGlue(features=None)
    with pytest.raises(AssertionError):
        m({"image0": {}})
    with pytest.raises(ValueError):
        LightGlue(features=None, precision="int8")


def test_feature_presets_select_dims():
    # released weights cannot be fetched offline: the presets must still resolve dims before loading
    for name, dim, so in [("superpoint", 256, False), ("disk", 128, False), ("aliked", 128, False), ("sift", 128, True), ("doghardnet", 128, True)]:
        assert LightGlue.features[name]["input_dim"] == dim
        assert bool(LightGlue.features[name].get("add_scale_ori", False)) == so
    with pytest.raises(RuntimeError, match="pretrained weights"):
        LightGlue(features="disk")


def test_no_cpu_fallback():
    m = LightGlue(features=None)
    d = {"keypoints": torch.rand(1, 8, 2), "descriptors": torch.rand(1, 8, 256)}
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m({"image0": d, "image1": d})


def test_product_code_never_imports_the_oracle():
    import pathlib
    pkg = pathlib.Path(__file__).resolve().parent.parent / "lightglue_amd"
    for f in list(pkg.glob("*.py")) + list((pkg / "csrc").glob("*")):
        if f.is_file() and f.suffix in (".py", ".hip", ".h"):
            assert "oracle" not in f.read_text().replace("no CPU / PyTorch fallback", ""), f


def test_utils_glue_mirrors_reference():
    """rbd / batch_to_device / match_pair (ref utils.py:55-69, 150-165) with a stub extractor and matcher."""
    from lightglue_amd import batch_to_device, match_pair, rbd

    class Ext:
        def extract(self, img, **kw):
            n = int(img.sum().item())
            return {"keypoints": torch.rand(1, n, 2), "descriptors": torch.rand(1, n, 256), "image_size": torch.tensor([[640.0, 480.0]])}

    def matcher(d):
        m = d["image0"]["keypoints"].shape[1]
        return {"matches0": torch.full((1, m), -1), "matches": [torch.zeros((0, 2), dtype=torch.long)], "stop": 9}

    f0, f1, m01 = match_pair(Ext(), matcher, torch.ones(5), torch.ones(7), device="cpu")
    assert f0["keypoints"].shape == (5, 2) and f1["descriptors"].shape == (7, 256)
    assert m01["matches0"].shape == (5,) and m01["matches"].shape == (0, 2) and m01["stop"] == 9
    d = rbd({"a": torch.zeros(1, 3), "b": [torch.ones(2)], "c": 4})
    assert d["a"].shape == (3,) and d["b"].shape == (2,) and d["c"] == 4
    moved = batch_to_device({"x": {"y": torch.zeros(2)}, "s": "name"}, "cpu")
    assert moved["s"] == "name" and moved["x"]["y"].device.type == "cpu"


def test_legacy_checkpoint_key_rename():
    """ref :427-434: released .pth files use self_attn.{i}/cross_attn.{i}; they must load unchanged."""
    m = LightGlue(features=None, n_layers=3)
    sd = m.state_dict()
    legacy = {}
    for k, v in sd.items():
        for i in range(3):
            k = k.replace(f"transformers.{i}.self_attn", f"self_attn.{i}").replace(f"transformers.{i}.cross_attn", f"cross_attn.{i}")
        legacy[k] = v + 1.0 if v.dtype.is_floating_point else v
    assert any(k.startswith("self_attn.2.") for k in legacy) and not any(k.startswith("transformers.") for k in legacy)
    renamed = LightGlue.rename_legacy_keys(legacy, 3)
    assert set(renamed) == set(sd)
    res = m.load_state_dict(renamed, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    assert torch.equal(m.state_dict()["transformers.2.cross_attn.to_qk.weight"], legacy["cross_attn.2.to_qk.weight"])
    assert LightGlue.rename_legacy_keys(renamed, 3).keys() == renamed.keys()   # idempotent


def test_collate_features_pads_and_counts():
    """glue.collate_features: ragged per-image features -> one padded batch + num_keypoints."""
    from lightglue_amd import collate_features
    g = torch.Generator().manual_seed(0)
    feats = [{"keypoints": torch.rand(n, 2, generator=g), "descriptors": torch.rand(n, 256, generator=g),
              "keypoint_scores": torch.rand(n, generator=g), "image_size": torch.tensor([640.0, 480.0])} for n in (5, 9, 0)]
    feats[1] = {k: v[None] for k, v in feats[1].items()}   # extractor output with batch dim 1 is accepted too
    batch = collate_features(feats)
    assert batch["keypoints"].shape == (3, 9, 2) and batch["descriptors"].shape == (3, 9, 256)
    assert batch["num_keypoints"].tolist() == [5, 9, 0] and batch["num_keypoints"].dtype == torch.int32
    assert batch["image_size"].shape == (3, 2)
    assert torch.equal(batch["keypoints"][0, :5], feats[0]["keypoints"]) and (batch["keypoints"][0, 5:] == 0).all()
    assert torch.equal(batch["descriptors"][1], feats[1]["descriptors"][0])


def test_extracted_to_image_frame_matches_reference_formula():
    """ref utils.py:142-147: keypoints back to the original frame + image_size (w, h)."""
    from lightglue_amd import extracted_to_image_frame
    kp = torch.tensor([[[0.0, 0.0], [99.5, 49.5], [10.0, 20.0]]])
    feats = {"keypoints": kp, "descriptors": torch.zeros(1, 3, 256)}
    out = extracted_to_image_frame(feats, (480, 640), torch.tensor([0.5, 0.25]))
    assert torch.allclose(out["keypoints"], (kp + 0.5) / torch.tensor([0.5, 0.25])[None] - 0.5)
    assert out["image_size"].tolist() == [[640.0, 480.0]] and out["descriptors"] is feats["descriptors"]
    assert torch.equal(feats["keypoints"], kp)   # input not mutated


def test_cm_prune_colours_the_prune_output_like_the_reference():
    """ref viz2d.py:33-39 (cm_prune) over :22-31 (cm_BlRdGn): survivors blue, dropped points red -> green by layer.
    Compared live against the reference function when /root/reference (and matplotlib, which its module imports) is there."""
    import numpy as np
    from lightglue_amd import cm_prune
    prune = torch.tensor([1, 2, 5, 9, 10, 10, 3])
    c = cm_prune(prune)
    assert c.shape == (7, 4) and np.all((c >= 0) & (c <= 1))
    assert np.allclose(c[4], [0.0, 0.2, 1.0, 1.0]) and np.allclose(c[5], c[4])            # alive to the end
    assert np.allclose(c[0], [1.0, 0.0, 0.0, 1.0])                                          # dropped after the first layer: red
    assert np.allclose(c[2], [1.0, 8.0 / 9.0, 0.0, 1.0]) and np.allclose(c[3], [2.0 / 9.0, 1.0, 0.0, 1.0])
    import importlib.util, pathlib
    ref = pathlib.Path("/root/reference/lightglue/viz2d.py")
    if ref.exists() and importlib.util.find_spec("matplotlib") is not None:
        spec = importlib.util.spec_from_file_location("ref_viz2d", ref); mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
        for p in (prune, torch.tensor([0, 1, 4, 4]), torch.tensor([3, 3, 3])):
            assert np.allclose(cm_prune(p), mod.cm_prune(p)), p


def test_deferred_handle_synchronises_once_and_caches():
    """DeferredMatches (LightGlue.forward_deferred): .result() waits for ITS event, assembles once, then returns the same dict."""
    from lightglue_amd.lightglue import DeferredMatches
    calls = []

    class Ev:
        def synchronize(self):
            calls.append("sync")

    h = DeferredMatches(Ev(), torch.tensor([[9, 9], [3, 1]], dtype=torch.int32), lambda host: (calls.append("assemble"), {"sizes": host})[1])
    out = h.result()
    assert out == {"sizes": [[9, 9], [3, 1]]} and h.result() is out and calls == ["sync", "assemble"]


def test_model_can_be_deep_copied_and_pickled():
    """The reference module can be copied / pickled; the process-local engine handle must not get in the way."""
    import copy, pickle
    m = LightGlue(features=None, n_layers=2)
    m._engine = ("not-a-real-handle", None)     # what a forward would have left behind
    m._plist = list(m.parameters()); m._weights_sig = (1, 2, 3, 4)
    c = copy.deepcopy(m)
    assert c._engine is None and c._plist is None and c._weights_sig is None
    assert torch.equal(c.state_dict()["transformers.1.self_attn.Wqkv.weight"], m.state_dict()["transformers.1.self_attn.Wqkv.weight"])
    r = pickle.loads(pickle.dumps(m))
    assert r._engine is None and r.conf.n_layers == 2
    m.__dict__["_engine"] = None                # do not hand the fake handle to lg_engine_destroy


def test_weight_changes_invalidate_the_packed_copy():
    m = LightGlue(features=None, n_layers=2)
    m._plist = list(m.parameters()); m._weights_sig = (0, 0, 0, 0)
    m.load_state_dict(m.state_dict())
    assert m._weights_sig is None and m._plist is None
    m._weights_sig = (0, 0, 0, 0)
    m.float()                                   # any _apply (.to / .cuda / .half)
    assert m._weights_sig is None
    m._weights_sig = (0, 0, 0, 0)
    m.refresh_weights()
    assert m._weights_sig is None



def test_bench_traffic_constants_cannot_go_stale_silently(tmp_path, monkeypatch):
    """bench.py reports PMC traffic only while profiles/pmc_traffic.json was measured on the kernel sources of this tree."""
    import json

    import bench
    key = "f16x3/f16x3/B32/N1024/fused_tail+next"
    rec = {"kernel_source_digest": bench.kernel_source_digest(), "bytes_per_launch": {key: 123.0}, "source": "test"}
    f = tmp_path / "pmc_traffic.json"
    f.write_text(json.dumps(rec))
    monkeypatch.setattr(bench, "PMC_TRAFFIC_FILE", f)
    assert bench.pmc_traffic(key) == (123.0, "test")
    assert bench.pmc_traffic("no/such/key")[0] is None
    rec["kernel_source_digest"] = "0" * 16
    f.write_text(json.dumps(rec))
    v, why = bench.pmc_traffic(key)
    assert v is None and "stale" in why
    monkeypatch.setattr(bench, "PMC_TRAFFIC_FILE", tmp_path / "missing.json")
    assert bench.pmc_traffic(key)[0] is None


def test_bench_spawns_its_own_ranks_without_a_launcher(monkeypatch):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment re-launches itself under torch.distributed.run with N ranks on a
    loopback port and returns the launcher's exit code (the GPU rehearsal is tests/test_parallel_gpu.py)."""
    import subprocess
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
    import bench
    seen = {}

    class Done:
        returncode = 7

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return Done()

    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    with pytest.raises(SystemExit) as exc:
        bench.main()
    assert exc.value.code == 7
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "127.0.0.1" in cmd
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_prefetch_to_device_refuses_a_cpu_target():
    from lightglue_amd import prefetch_to_device
    with pytest.raises(RuntimeError, match="MI355X"):
        next(prefetch_to_device(iter([{}]), "cpu"))
    with pytest.raises(ValueError):
        next(prefetch_to_device(iter([{}]), "cuda", depth=0))
