"""SuperPoint descriptor head: oracle vs the reference's own outputs (CPU), HIP kernels vs both
(GPU).  Tolerance: fp32 interpolation + normalisation, 5e-6 absolute on unit-norm descriptors."""
from pathlib import Path

import numpy as np
import pytest
import torch

import make_golden_superpoint as G
from conftest import require_gpu
from oracle import superpoint_oracle as SO

GOLD = Path(__file__).resolve().parent / "golden"
NAMES = sorted(G.CASES)
TOL = 5e-6   # ix ~ 100 has an fp32 ulp of 7.6e-6: the interpolation weights differ by that between evaluation orders


def load(name):
    z = np.load(GOLD / f"{name}.npz")
    seed, b, h, w, n = (int(v) for v in z["case"])
    dense, kp = G.head_inputs(seed, b, h, w, n)
    return z, dense, kp


@pytest.mark.parametrize("name", NAMES)
def test_oracle_matches_reference_fixture(name):
    z, dense, kp = load(name)
    np.testing.assert_allclose(SO.descriptor_head(kp, dense), z["descriptors"], atol=TOL, rtol=0)
    if "sampled_unnormalized_map" in z:
        np.testing.assert_allclose(SO.sample_descriptors(kp, dense), z["sampled_unnormalized_map"], atol=TOL, rtol=0)


def test_cpu_tensors_raise():
    from lightglue_amd import superpoint_head as H
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        H.descriptor_head(torch.zeros(1, 4, 2), torch.zeros(1, 256, 8, 8))


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_hip_descriptor_head_matches_reference_fixture(name):
    require_gpu()
    from lightglue_amd import superpoint_head as H
    z, dense, kp = load(name)
    kpt, dt = torch.from_numpy(kp).cuda(), torch.from_numpy(dense).cuda()
    out = H.descriptor_head(kpt, dt)
    assert out.shape == z["descriptors"].shape and out.is_contiguous()
    np.testing.assert_allclose(out.cpu().numpy(), z["descriptors"], atol=TOL, rtol=0)
    assert torch.equal(kpt.cpu(), torch.from_numpy(kp)), "inputs must not be modified"
    if "sampled_unnormalized_map" in z:   # the reference-signature function: [b, c, N], no dense normalisation
        np.testing.assert_allclose(H.sample_descriptors(kpt, dt).cpu().numpy(), z["sampled_unnormalized_map"], atol=TOL, rtol=0)


@pytest.mark.gpu
def test_hip_descriptor_head_ragged_and_odd_map():
    """Ragged counts (padding rows zero, never read) and a map whose h*w is not a multiple of the 32-location tile."""
    require_gpu()
    from lightglue_amd import superpoint_head as H
    dense, kp = G.head_inputs(5, 3, 13, 11, 40)
    counts = [40, 17, 0]
    ref = SO.descriptor_head(kp, dense)
    kpt = torch.from_numpy(kp).cuda()
    for b, c in enumerate(counts):
        kpt[b, c:] = float("nan")
    out = H.descriptor_head(kpt, torch.from_numpy(dense).cuda(), num_keypoints=torch.tensor(counts)).cpu().numpy()
    for b, c in enumerate(counts):
        np.testing.assert_allclose(out[b, :c], ref[b, :c], atol=TOL, rtol=0)
        assert (out[b, c:] == 0).all()


@pytest.mark.gpu
def test_head_feeds_matcher():
    """End to end: descriptor head output -> matcher, vs the oracle chain (descriptor oracle -> matcher oracle)."""
    require_gpu()
    import gpu_util
    from lightglue_amd import superpoint_head as H
    from lightglue_amd import synthetic as synth
    from oracle import lightglue_oracle as O
    dense, kp = G.head_inputs(9, 2, 48, 64, 256)
    kp = kp[:, 14:]                                      # drop the deliberately out-of-image points
    d = SO.descriptor_head(kp, dense)
    sd = synth.make_state_dict(0, recipe="A")
    conf = O.make_conf(depth_confidence=-1, width_confidence=-1)
    size = np.array([512.0, 384.0], np.float32)
    ref = O.forward_pair(sd, conf, kp[0], kp[1], d[0], d[1], size, size)
    model = gpu_util.make_model(sd, "fp32", depth_confidence=-1, width_confidence=-1)
    desc = H.descriptor_head(torch.from_numpy(kp).cuda(), torch.from_numpy(dense).cuda())
    out = model({"image0": {"keypoints": torch.from_numpy(kp[:1]).cuda(), "descriptors": desc[:1], "image_size": torch.from_numpy(size)[None].cuda()},
                 "image1": {"keypoints": torch.from_numpy(kp[1:]).cuda(), "descriptors": desc[1:], "image_size": torch.from_numpy(size)[None].cuda()}})
    np.testing.assert_array_equal(out["matches0"][0].cpu().numpy(), ref["matches0"])
    np.testing.assert_allclose(out["matching_scores0"][0].cpu().numpy(), ref["matching_scores0"], atol=2e-4, rtol=0)


# --------------------------------------------------------------------------- keypoint extraction (NMS / threshold / top-k)
DET_NAMES = sorted(G.DETECT_CASES)


def load_detect(name):
    z = np.load(GOLD / f"{name}.npz")
    seed, b, h, w, topk = (int(v) for v in z["case"])
    return z, G.score_map(seed, b, h, w), (None if topk < 0 else topk)


def assert_same_detections(kp, sc, ref_kp, ref_sc, smap_candidates=None):
    """Exact equality, except that torch.topk leaves the order (and, at the cut, the choice) among EQUAL scores open."""
    np.testing.assert_array_equal(sc, ref_sc)
    if smap_candidates is None:                       # no top-k: row-major order is defined
        np.testing.assert_array_equal(kp, ref_kp)
        return
    as_set = lambda a: {tuple(r) for r in a.tolist()}
    for v in np.unique(sc):
        mine, theirs = as_set(kp[sc == v]), as_set(ref_kp[ref_sc == v])
        if v == sc.min():                             # the tie group the cut may split: any candidates with that score
            cand_kp, cand_sc = smap_candidates
            assert mine <= as_set(cand_kp[cand_sc == v])
        else:
            assert mine == theirs


@pytest.mark.parametrize("name", DET_NAMES)
def test_detect_oracle_matches_reference_fixture(name):
    z, smap, topk = load_detect(name)
    nms_nonzero = np.unpackbits(z["nms_nonzero"])[: smap.size].reshape(smap.shape).astype(bool)
    for i in range(smap.shape[0]):
        np.testing.assert_array_equal(SO.simple_nms(smap[i], 4) != 0, nms_nonzero[i])
        kp, sc = SO.detect_keypoints(smap[i], max_num_keypoints=topk)
        cand = SO.detect_keypoints(smap[i]) if topk else None
        assert_same_detections(kp, sc, z[f"kp{i}"], z[f"sc{i}"], cand)


@pytest.mark.gpu
@pytest.mark.parametrize("name", DET_NAMES)
def test_hip_detect_matches_reference_fixture(name):
    require_gpu()
    from lightglue_amd import superpoint_head as H
    z, smap, topk = load_detect(name)
    kp, sc, num = H.detect_keypoints(torch.from_numpy(smap).cuda(), max_num_keypoints=topk)
    num = num.cpu().tolist()
    for i in range(smap.shape[0]):
        assert num[i] == len(z[f"sc{i}"])
        cand = SO.detect_keypoints(smap[i]) if topk else None
        assert_same_detections(kp[i, : num[i]].cpu().numpy(), sc[i, : num[i]].cpu().numpy(), z[f"kp{i}"], z[f"sc{i}"], cand)


@pytest.mark.gpu
def test_hip_detect_options_against_oracle():
    """Other radii / borders / thresholds, a map smaller than one tile, fewer detections than k (ref :74-75: unsorted)."""
    require_gpu()
    from lightglue_amd import superpoint_head as H
    for seed, (h, w), kw in ((3, (40, 50), dict(nms_radius=2, remove_borders=0, detection_threshold=0.01)),
                             (4, (9, 13), dict(nms_radius=4, remove_borders=2, detection_threshold=0.0005)),
                             (5, (100, 70), dict(nms_radius=3, remove_borders=4, detection_threshold=0.2, max_num_keypoints=4000)),
                             (6, (64, 128), dict(nms_radius=0, remove_borders=1, detection_threshold=0.5))):
        smap = G.score_map(seed, 2, h, w)
        kp, sc, num = H.detect_keypoints(torch.from_numpy(smap).cuda(), capacity=h * w, **kw)
        for i in range(2):
            rkp, rsc = SO.detect_keypoints(smap[i], **kw)
            n = int(num[i])
            assert n == len(rsc), (seed, i, n, len(rsc))
            np.testing.assert_array_equal(sc[i, :n].cpu().numpy(), rsc)
            np.testing.assert_array_equal(kp[i, :n].cpu().numpy(), rkp)


@pytest.mark.gpu
def test_detect_then_describe_then_match():
    """The whole post-CNN SuperPoint tail feeding the matcher as ONE ragged batch: detect -> descriptor head -> LightGlue."""
    require_gpu()
    import gpu_util
    from lightglue_amd import LightGlue, superpoint_head as H
    from lightglue_amd import synthetic as synth
    smap = G.score_map(7, 2, 96, 128)
    dense, _ = G.head_inputs(7, 2, 12, 16, 16)
    kp, sc, num = H.detect_keypoints(torch.from_numpy(smap).cuda(), max_num_keypoints=256)
    desc = H.descriptor_head(kp, torch.from_numpy(dense).cuda(), 8, num)
    assert (desc[0, int(num[0]):] == 0).all()
    model = gpu_util.make_model(synth.make_state_dict(0, recipe="A"), "f16x3", depth_confidence=-1, width_confidence=-1)
    size = torch.tensor([[128.0, 96.0]]).cuda()
    out = model({"image0": {"keypoints": kp[:1], "descriptors": desc[:1], "image_size": size, "num_keypoints": num[:1]},
                 "image1": {"keypoints": kp[1:], "descriptors": desc[1:], "image_size": size, "num_keypoints": num[1:]}})
    assert out["matches0"].shape == (1, kp.shape[1]) and (out["matches0"][0, int(num[0]):] == -1).all()
