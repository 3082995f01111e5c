"""The LDS-DMA wait discipline of the attention kernels, checked on the compiled gfx950 ISA (tools/check_isa.py; ADVICE r03: the DMA is
inline asm on purpose, so its completion waits are hand-placed and the compiler enforces nothing).  hipcc cross-compiles without a GPU."""
import shutil
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="needs hipcc")
def test_every_dma_tile_is_waited_for_once_and_only_at_its_publishing_barrier():
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "check_isa.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("ok  ") >= 4, r.stdout      # attn_dma_kernel x 2 element types, attn_split_kernel x 2 shapes


def test_checker_flags_a_missing_and_an_extra_wait():
    sys.path.insert(0, str(ROOT / "tools"))
    import check_isa
    good = ["global_load_lds_dwordx4 v[0:1], off", "v_mfma_f32_16x16x32_f16 v[0:3], v[4:7], v[8:11], v[0:3]", "s_waitcnt vmcnt(0)", "s_barrier", "s_endpgm"]
    assert check_isa.check_kernel("k", good) == []
    missing = [t for t in good if not t.startswith("s_waitcnt")]
    assert any("no vmcnt(0) wait" in e for e in check_isa.check_kernel("k", missing))
    extra = good[:1] + ["s_waitcnt vmcnt(1) lgkmcnt(7)"] + good[1:]
    assert any("extra vmcnt wait" in e for e in check_isa.check_kernel("k", extra))
    unreleased = ["global_load_lds_dwordx4 v[0:1], off", "s_endpgm"]
    assert any("before the kernel ends" in e for e in check_isa.check_kernel("k", unreleased))
