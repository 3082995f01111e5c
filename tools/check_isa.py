#!/usr/bin/env python3
"""Static check of the LDS-DMA wait discipline in the compiled attention and similarity kernels (`lds_dma16` hides `global_load_lds` from
hipcc's waitcnt pass, so every wait for a DMA'd tile is hand-placed and nothing enforced it).

Compiles lg_attention.hip and lg_sim.hip for gfx950 (device only, to assembly) and checks, for every kernel that issues `global_load_lds_*`:
  1. between a group of DMA instructions and the NEXT `s_barrier` in text order there is an `s_waitcnt` with vmcnt(0)  (a tile is never
     published to the other waves before this wave's pieces have landed);
  2. that wait is the ONLY s_waitcnt mentioning vmcnt in between  (a compiler-generated vmcnt wait right after the DMA issue would expose the
     DMA round trip: with an in-order counter, waiting for any older load also waits for the DMAs behind it — found in round 4 in the peeled
     first tile, where hipcc still waited for the Q fragment loads);
  3. from every loop header that is followed by a barrier, a vmcnt(0) wait precedes that barrier with no DMA in between (the back edge);
  4. the last DMA group of a kernel is followed by a vmcnt(0) wait before the kernel can end (no DMA may land in LDS that has been released).
Invariant the in-order argument rests on (documented next to lds_dma16 in lg_common.h): these kernels issue no VMEM STORE between a DMA and
its wait, and all their waits for DMAs are vmcnt(0).

usage: check_isa.py [--keep out.s]      exit code 0 = all kernels pass"""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
FLAGS = "--offload-arch=gfx950 --cuda-device-only -S -O3 -std=c++17 -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form -fno-slp-vectorize".split()


def compile_asm(src: Path, out: Path):
    subprocess.run(["hipcc", *FLAGS, str(src), "-o", str(out)], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=src.parent)


def kernels(asm: str):
    """yield (name, [instruction lines]) per function"""
    cur, body = None, []
    for line in asm.splitlines():
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur, body = m.group(1), []
            continue
        if cur is None:
            continue
        if line.startswith(".Lfunc_end"):
            yield cur, body
            cur = None
            continue
        t = line.strip()
        if t and not t.startswith(";") and not t.startswith(".") or re.match(r"^\.LBB\w+:", t):
            body.append(t)


def check_kernel(name, body):
    errs = []
    is_dma = lambda t: t.startswith("global_load_lds")
    is_bar = lambda t: t.startswith("s_barrier")
    is_vm = lambda t: t.startswith("s_waitcnt") and "vmcnt" in t
    is_vm0 = lambda t: is_vm(t) and "vmcnt(0)" in t
    n = len(body)
    i = 0
    while i < n:
        if not is_dma(body[i]):
            i += 1
            continue
        j = i
        while j + 1 < n and (is_dma(body[j + 1]) or not (is_bar(body[j + 1]) or is_vm(body[j + 1]) or body[j + 1].startswith("s_endpgm"))):
            j += 1                                    # extend over the group and whatever follows up to the first wait / barrier / end
        # collect the vm waits up to the next barrier (or the end of the kernel)
        k, waits, hit_bar = i + 1, [], False
        while k < n:
            if is_bar(body[k]):
                hit_bar = True
                break
            if is_vm(body[k]):
                waits.append((k, body[k]))
            k += 1
        if hit_bar:
            if not waits or not is_vm0(waits[-1][1]):
                errs.append(f"DMA at +{i}: no vmcnt(0) wait before the next s_barrier (+{k})")
            # an earlier wait is harmless when no matrix instruction sits between it and the publishing wait (e.g. hipcc's own vmcnt(0) for the operand
            # fragments right in front of the loop header, followed by the hand-placed one: nothing was held up that the publishing wait would not hold up anyway)
            extra = [w for w in waits[:-1] if any(t.startswith("v_mfma") for t in body[w[0]:waits[-1][0]])]
            if extra:
                errs.append(f"DMA at +{i}: {len(extra)} extra vmcnt wait(s) before the publishing wait: {extra[0][1]!r} at +{extra[0][0]} (exposes the DMA round trip)")
        else:
            if not any(is_vm0(w[1]) for w in waits):
                errs.append(f"last DMA group at +{i}: no vmcnt(0) wait before the kernel ends")
        # skip past this DMA group
        while i < n and (is_dma(body[i]) or not (is_vm(body[i]) or is_bar(body[i]))):
            i += 1
    # loop headers: the first barrier after a header must be preceded by a vmcnt(0) wait, no DMA in between
    for i, t in enumerate(body):
        if re.match(r"^\.LBB\w+:.*Loop Header", t):
            if not any(is_dma(x) for x in body[i + 1:]):
                continue                               # no DMA is issued from here on: nothing can be published through this loop's back edge
            seen_wait = False
            for k in range(i + 1, n):
                if is_dma(body[k]):
                    break                              # the loop issues DMAs before any barrier: nothing published from the back edge here
                if is_vm0(body[k]):
                    seen_wait = True
                if is_bar(body[k]):
                    if not seen_wait:
                        errs.append(f"loop header at +{i}: s_barrier at +{k} without a vmcnt(0) wait after the header")
                    break
    return errs


def main():
    keep = sys.argv[sys.argv.index("--keep") + 1] if "--keep" in sys.argv else None
    asm = ""
    with tempfile.TemporaryDirectory() as td:
        for i, src in enumerate(("lg_attention.hip", "lg_sim.hip")):      # every source whose kernels issue LDS-DMAs through lds_dma16 / lds_dma16_s
            out = Path(keep + (f".{i}" if i else "")) if keep else Path(td) / f"k{i}.s"
            compile_asm(ROOT / "lightglue_amd" / "csrc" / src, out)
            asm += out.read_text() + "\n"
    bad = 0
    checked = 0
    for name, body in kernels(asm):
        if not any(t.startswith("global_load_lds") for t in body):
            continue
        checked += 1
        errs = check_kernel(name, body)
        print(f"{'FAIL' if errs else 'ok  '} {name}: {sum(t.startswith('global_load_lds') for t in body)} DMA instructions, {sum(t.startswith('s_barrier') for t in body)} barriers")
        for e in errs:
            print("     ", e)
        bad += bool(errs)
    if checked == 0:
        print("no kernel with global_load_lds found: the check did not run")
        return 2
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
