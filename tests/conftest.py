"""Shared test plumbing.  `-m "not gpu"` runs here on CPU; `-m gpu` needs an MI355X.

Tolerances (BASELINE.json north_star): match indices bit-identical, scores within 1e-3 (fp32).
An index may differ from the oracle ONLY where the oracle itself is within the score tolerance of a
decision boundary (filter threshold, or a top-2 log-score margin) — `explain_mismatches` proves that
per element from the oracle's full score matrix and the tests assert nothing else differs.
"""
import json
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))

SCORE_TOL = 1e-3  # north_star: scores within 1e-3


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_names():
    return sorted(p.stem for p in (ROOT / "tests" / "golden").glob("*.npz") if not p.stem.startswith("superpoint_"))


def load_golden(name):
    z = np.load(ROOT / "tests" / "golden" / f"{name}.npz", allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    return meta, {k: z[k] for k in z.files if k != "meta"}


def oracle_conf_for(case):
    """LightGlue kwargs of a golden case -> oracle conf (the oracle takes the resolved pruning threshold)."""
    from oracle import lightglue_oracle as O
    kw = dict(case["conf"])
    kw["pruning_min_kpts"] = case.get("prune_th", -1)  # reference ran on CPU: class dict 'cpu' = -1 unless overridden
    if case.get("static_lengths") and max(case["n"], case["m"]) <= max(case["static_lengths"]):
        kw["width_confidence"] = -1                    # ref :529: no point pruning inside the static-length range
    return O.make_conf(**kw)


def explain_mismatches(got_m0, got_s0, ref, score_tol=SCORE_TOL, filter_threshold=0.1, scores_full=None, ind0=None, ind1=None, side=0):
    """Return the number of UNEXPLAINED index mismatches on one image side of one pair.
    ref: oracle output dict (single pair).  scores_full: oracle log-assignment [m'+1, n'+1] in pruned index space with
    ind0/ind1 mapping to original indices.  A mismatch is explained only if the ORACLE ITSELF sits on a decision boundary
    within 2 x the score tolerance, measured in SCORE space (exp of the log-assignment, the unit of the 1e-3 bar):
      * the filter threshold: one side says -1 and the matching score is within tolerance of the threshold, or
      * an argmax near-tie: the best and second-best score of the keypoint's row — or of the column of the oracle's best
        partner, whose argmax decides mutuality — differ by at most 2 x tolerance."""
    key_m, key_s = ("matches0", "matching_scores0") if side == 0 else ("matches1", "matching_scores1")
    ref_m, ref_s = ref[key_m], ref[key_s]
    S = None
    if scores_full is not None:
        S = scores_full[:-1, :-1] if side == 0 else scores_full[:-1, :-1].T
        own, other = (ind0, ind1) if side == 0 else (ind1, ind0)
    diff = np.where(got_m0 != ref_m)[0]
    unexplained = 0

    def near_tie(vec):
        # a tie only matters where the winner can become a match at all: top score at or above the filter threshold
        if vec.size < 2:
            return False
        top2 = np.sort(vec)[-2:]
        return float(np.exp(top2[1])) > filter_threshold - score_tol and float(np.exp(top2[1]) - np.exp(top2[0])) <= 2 * score_tol

    for a in diff:
        near_thr = abs(float(ref_s[a]) - filter_threshold) <= score_tol or abs(float(got_s0[a]) - filter_threshold) <= score_tol
        if near_thr and (got_m0[a] == -1 or ref_m[a] == -1):
            continue
        if S is not None and own is not None and (own == a).any():
            pa = int(np.where(own == a)[0][0])
            row = S[pa]
            if near_tie(row) or near_tie(S[:, int(row.argmax())]):
                # the flipped keypoint's own score must still be the oracle's score of whatever it picked (0 when unmatched)
                if got_m0[a] == -1:
                    want = 0.0
                else:
                    pb = np.where(other == got_m0[a])[0]
                    want = float(np.exp(row[int(pb[0])])) if pb.size else None
                if want is not None and abs(float(got_s0[a]) - want) <= 2 * score_tol:
                    continue
        unexplained += 1
    return unexplained


def assert_parity_with_explained_flips(out, gold, case, sd, data, score_tol=SCORE_TOL):
    """The default-precision bar on BOTH image sides: scores within tolerance everywhere, indices identical except for flips
    that `explain_mismatches` traces to an oracle-side decision boundary.  Returns (flips0, flips1)."""
    from oracle import lightglue_oracle as O
    res = []
    conf = oracle_conf_for(case)
    traces = {}
    for side in (0, 1):
        m = out[f"matches{side}"].cpu().numpy(); s = out[f"matching_scores{side}"].cpu().numpy()
        gm, gs = gold[f"matches{side}"], gold[f"matching_scores{side}"]
        bad = np.abs(s - gs) > score_tol
        flips = m != gm
        # a score may differ by more than the tolerance only where the index legitimately flipped (the score then belongs
        # to another partner, or drops to 0 when mutuality is lost); explain_mismatches bounds the flipped keypoint's score too
        assert not (bad & ~flips).any(), f"side {side}: {int((bad & ~flips).sum())} scores off by more than {score_tol} without an index flip"
        for b in np.where(flips.any(axis=1))[0]:
            if b not in traces:
                tr = {}
                g = lambda d, k: None if d.get(k) is None else np.asarray(d[k])[b]
                d0, d1 = data["image0"], data["image1"]
                ref = O.forward_pair(sd, conf, g(d0, "keypoints"), g(d1, "keypoints"), g(d0, "descriptors"), g(d1, "descriptors"),
                                     g(d0, "image_size"), g(d1, "image_size"), g(d0, "scales"), g(d0, "oris"), g(d1, "scales"), g(d1, "oris"), trace=tr)
                traces[b] = (ref, tr)
            ref, tr = traces[b]
            un = explain_mismatches(m[b], s[b], ref, score_tol=score_tol, filter_threshold=conf.filter_threshold, scores_full=tr["scores_full"],
                                    ind0=tr["ind0"], ind1=tr["ind1"], side=side)
            assert un == 0, f"{un} unexplained index mismatches on side {side} of pair {b}"
        res.append(int(flips.sum()))
    return tuple(res)


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
