"""SuperPoint conv stack + full extractor on the MI355X against fixtures produced by the REAL reference class
(tools/make_golden_superpoint.py: superpoint.py executed unmodified with seeded weights in place of the network download).

Bars: the conv stack is exact fp32 (f32 MFMA) — scores and the raw descriptor map agree with the reference's torch convolutions
to summation-order round-off.  The detector then thresholds / NMS-compares those scores, so a keypoint can legitimately differ
from the reference's only where two scores tie to within that round-off; the test allows at most 1 % such differences and
requires everything else (coordinates, scores, descriptors of the common keypoints) to match."""
from pathlib import Path

import numpy as np
import pytest
import torch

import make_golden_superpoint as G
from conftest import require_gpu

GOLD = Path(__file__).resolve().parent / "golden"
NAMES = sorted(G.ENCODE_CASES)


def _case(name):
    z = np.load(GOLD / f"{name}.npz")
    wseed, iseed, b, h, w, topk = (int(v) for v in z["case"])
    return z, G.encoder_state_dict(wseed), G.encoder_image(iseed, b, h, w), (None if topk < 0 else topk)


def test_state_dict_contract_and_no_cpu_fallback():
    """Parameter names / shapes of the reference class (superpoint.py:127-141): its state dict loads unchanged; CPU images raise."""
    from lightglue_amd import SuperPoint
    sd = G.encoder_state_dict(0)
    m = SuperPoint(weights=sd)
    assert list(m.state_dict().keys()) == list(sd.keys())
    assert all(tuple(m.state_dict()[k].shape) == tuple(v.shape) for k, v in sd.items())
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m({"image": torch.zeros(1, 1, 16, 16)})
    with pytest.raises(ValueError):
        SuperPoint(max_num_keypoints=0)      # ref :146-147
    with pytest.raises(AssertionError):
        m({})                                # ref :149-150


@pytest.mark.skipif(not G.REF.exists(), reason="reference tree not mounted")
def test_fixture_generator_runs_the_reference_class():
    """The generator's stand-ins leave the reference forward itself untouched: its module tree has the reference's layers and the
    seeded state dict was loaded by the reference's own constructor."""
    sd = G.encoder_state_dict(3)
    ref = G.load_reference_superpoint(sd, max_num_keypoints=7)
    assert type(ref).__name__ == "SuperPoint" and ref.conf.max_num_keypoints == 7 and ref.conf.nms_radius == 4
    assert torch.equal(ref.state_dict()["convPb.weight"], sd["convPb.weight"])


@pytest.mark.gpu
@pytest.mark.parametrize("conv_precision", ["fp32", "f16x3"])
@pytest.mark.parametrize("name", NAMES)
def test_conv_stack_matches_reference(name, conv_precision):
    """Both arithmetic forms of the conv stack against the reference's own fp32 CPU convolutions: exact f32 MFMAs (default) and the opt-in split-f16 form (22 operand bits,
    three f16 MFMAs per product) — held to the SAME tolerances."""
    require_gpu()
    from lightglue_amd import SuperPoint
    z, sd, img, topk = _case(name)
    model = SuperPoint(weights=sd, max_num_keypoints=topk, conv_precision=conv_precision).cuda().eval()
    scores, dense = model.encode(torch.from_numpy(img).cuda())
    # probabilities in [0, 1].  The split-f16 form carries 22 operand bits against fp32's 24: one of 19 200 entries of one fixture sits 3.7e-5 (relative) from the reference
    atol, rtol = (2e-6, 2e-5) if conv_precision == "fp32" else (1e-5, 1e-4)
    np.testing.assert_allclose(scores.cpu().numpy(), z["scores"], atol=atol, rtol=rtol)
    d = dense.double()
    np.testing.assert_allclose([float(d.abs().mean()), float(d.pow(2).mean())], z["dense_digest"], rtol=1e-5)
    np.testing.assert_allclose(dense[:, ::16, ::3, ::5].cpu().numpy(), z["dense_sample"], atol=5e-5, rtol=2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("conv_precision", ["fp32", "f16x3"])
@pytest.mark.parametrize("name", NAMES)
def test_full_extractor_matches_reference(name, conv_precision):
    require_gpu()
    from lightglue_amd import SuperPoint
    z, sd, img, topk = _case(name)
    model = SuperPoint(weights=sd, max_num_keypoints=topk, conv_precision=conv_precision).cuda().eval()
    out = model({"image": torch.from_numpy(img).cuda()})
    kp, sc, desc, cnt = (out[k].cpu().numpy() for k in ("keypoints", "keypoint_scores", "descriptors", "num_keypoints"))
    ref_kp, ref_sc, ref_desc = z["keypoints"], z["keypoint_scores"], z["descriptors"]
    assert kp.shape[0] == ref_kp.shape[0] and desc.shape[-1] == 256
    for b in range(ref_kp.shape[0]):
        n = int(cnt[b])
        got = {(int(x), int(y)): i for i, (x, y) in enumerate(kp[b, :n])}
        ref = {(int(x), int(y)): i for i, (x, y) in enumerate(ref_kp[b])}
        common = sorted(set(got) & set(ref))
        assert len(common) >= 0.99 * len(ref) and abs(n - len(ref)) <= max(1, len(ref) // 100), (n, len(ref), len(common))
        gi = np.array([got[c] for c in common]); ri = np.array([ref[c] for c in common])
        np.testing.assert_allclose(sc[b][gi], ref_sc[b][ri], atol=2e-6, rtol=2e-5)
        np.testing.assert_allclose(desc[b][gi], ref_desc[b][ri], atol=2e-5, rtol=0)
        if topk is None:   # row-major order of torch.where (ref :197): identical keypoint sets come in identical order
            if len(common) == len(ref) == n:
                np.testing.assert_array_equal(kp[b, :n], ref_kp[b])


@pytest.mark.gpu
def test_extractor_feeds_the_matcher():
    """extract() -> LightGlue.forward: the dict contract of utils.py:136-147 / lightglue.py:460-468 end to end on the device."""
    require_gpu()
    import gpu_util
    from lightglue_amd import SuperPoint, rbd
    from lightglue_amd import synthetic as synth
    ext = SuperPoint(weights=G.encoder_state_dict(0), max_num_keypoints=128).cuda().eval()
    img = torch.from_numpy(G.encoder_image(10, 1, 120, 160)).cuda()
    f0 = ext.extract(img[0])
    f1 = ext.extract(torch.roll(img[0], shifts=(3, 5), dims=(-2, -1)))
    assert f0["keypoints"].shape == (1, 128, 2) and f0["descriptors"].shape == (1, 128, 256) and f0["image_size"].tolist() == [[160.0, 120.0]]
    matcher = gpu_util.make_model(synth.make_state_dict(0, recipe="A"), "f16x3", depth_confidence=-1, width_confidence=-1)
    out = rbd(matcher({"image0": f0, "image1": f1}))
    assert out["matches0"].shape == (128,) and out["matches"].shape[1] == 2


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 333, 517), (1, 8, 8), (3, 40, 1000), (1, 1025, 31)])
def test_split_conv_stack_agrees_with_the_exact_form_at_odd_sizes(shape):
    """The LDS-staged split-f16 convolution (conv_precision "f16x3") against the exact-fp32 kernels on sizes the fixtures do not hold: widths / heights that are no
    multiples of the 32 x 8 pixel tile at every pyramid level, one-tile images, very wide and very tall ones, batch > 1 — halo staging, zero padding and the pooled /
    unpooled epilogues at every edge.  Tolerance = the split form's own round-off against fp32 (22 vs 24 operand bits through twelve layers)."""
    require_gpu()
    from lightglue_amd import SuperPoint
    B, H, W = shape
    sd = G.encoder_state_dict(5)
    img = torch.from_numpy(G.encoder_image(77, B, H, W)).cuda()
    exact = SuperPoint(weights=sd, conv_precision="fp32").cuda().eval()
    split = SuperPoint(weights=sd, conv_precision="f16x3").cuda().eval()
    s0, d0 = exact.encode(img)
    s1, d1 = split.encode(img)
    assert s0.shape == s1.shape and d0.shape == d1.shape
    np.testing.assert_allclose(s1.cpu().numpy(), s0.cpu().numpy(), atol=1e-5, rtol=2e-4)
    scale = float(d0.abs().max())
    assert float((d1 - d0).abs().max()) <= 2e-5 * max(scale, 1.0), (float((d1 - d0).abs().max()), scale)
    again = split.encode(img)
    assert torch.equal(again[0], s1) and torch.equal(again[1], d1)          # deterministic
