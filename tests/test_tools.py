"""The command-line tools of the package: argument parsing and their no-GPU behaviour (CPU suite); the runs themselves are in the GPU suite
(tests/test_gpu_round6.py)."""
import re
import subprocess
import sys
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parents[1]


def _run(*argv, timeout=900):
    return subprocess.run([sys.executable, *argv], capture_output=True, text=True, timeout=timeout, cwd=str(ROOT))


def test_benchmark_cli_parses_the_reference_command_line():
    """tools/benchmark.py keeps the reference benchmark.py's flags (/root/reference/benchmark.py:117-152): --help lists them, a bad choice is refused by
    argparse, and without a GPU the tool exits with a message instead of falling back to a CPU path."""
    p = _run("tools/benchmark.py", "--help")
    assert p.returncode == 0
    for flag in ("--device", "--compile", "--no_flash", "--no_prune_thresholds", "--measure", "--repeat", "--num_keypoints"):
        assert flag in p.stdout, flag
    bad = _run("tools/benchmark.py", "--measure", "joules")
    assert bad.returncode == 2 and "invalid choice" in bad.stderr
    if not torch.cuda.is_available():
        p = _run("tools/benchmark.py", "--num_keypoints", "256", "--repeat", "1")
        assert p.returncode != 0 and "no CPU path" in (p.stderr + p.stdout)


def _legacy_checkpoint(path, recipe="D"):
    """recipe-D weights written the way the released checkpoints name their tensors (`self_attn.{i}.*` / `cross_attn.{i}.*`, ref lightglue.py:427-434)"""
    from lightglue_amd import synthetic as synth
    out = {}
    for k, v in synth.make_state_dict(0, recipe=recipe).items():
        m = re.match(r"transformers\.(\d+)\.(self_attn|cross_attn)\.(.*)", k)
        out[f"{m.group(2)}.{m.group(1)}.{m.group(3)}" if m else k] = torch.from_numpy(v)
    torch.save(out, str(path))
    return path


def test_verify_pretrained_statistics_without_a_gpu(tmp_path):
    ck = _legacy_checkpoint(tmp_path / "legacy.pth")
    p = _run("tools/verify_pretrained.py", str(ck), "--cpu-only", "--sizes", "192", "--pairs", "1")
    assert p.returncode == 0, p.stderr[-2000:]
    assert "released (legacy) key names" in p.stdout and "'n_layers': 9" in p.stdout
    rows = [ln for ln in p.stdout.splitlines() if re.match(r"\| \d \|", ln)]
    assert len(rows) == 9                                                  # one statistics row per layer
    if not torch.cuda.is_available():
        p = _run("tools/verify_pretrained.py", str(ck), "--sizes", "192")
        assert p.returncode != 0 and "no GPU visible" in (p.stderr + p.stdout)
