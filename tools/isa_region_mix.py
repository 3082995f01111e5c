#!/usr/bin/env python3
"""Static instruction mix of one kernel of a .s file, per region between the kernel's stamps (s_memtime) / barriers.

    tools/isa_region_mix.py /tmp/isa/tail.s _ZN2lg11tail_kernelILi4ELi2EDF16_Li4ELb1EEEvNS_8TailArgsE [--barriers]

Static counts: a loop body is counted once (labels and backward branches are printed so that trip counts can be applied by hand).
Classes: mfma, valu (packed / transcendental / other), salu, lds, vmem, nop (s_nop N counts N + 1 issue cycles), wait."""
import re, sys
from collections import Counter

def classify(op):
    if op.startswith("v_mfma"): return "mfma"
    if op in ("s_nop",): return "nop"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    if op.startswith("v_pk_"): return "valu_pk"
    if re.match(r"v_(exp|log|rcp|rsq|sqrt|sin|cos)_", op): return "valu_trans"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_"): return "salu"
    return "other"

def main():
    path, kern = sys.argv[1], sys.argv[2]
    cut_barriers = "--barriers" in sys.argv
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith(kern + ":"))
    regions, cur, name = [], Counter(), "entry"
    for i in range(start + 1, len(lines)):
        l = lines[i].strip()
        if not l or l.startswith((";", ".", "//")):
            if re.match(r"\.LBB\d+_\d+:", l): cur["label"] += 1
            continue
        op = l.split()[0]
        if op == "s_endpgm":
            break
        if op == "s_memtime" or (cut_barriers and op == "s_barrier"):
            regions.append((name, cur)); cur = Counter(); name = f"{op}@{i - start}"
        c = classify(op)
        cur[c] += (int(l.split()[1]) + 1) if c == "nop" else 1
        if op.startswith("s_cbranch") or op == "s_branch": cur["branch"] += 1
    regions.append((name, cur))
    keys = ["mfma", "valu", "valu_pk", "valu_trans", "salu", "lds", "vmem", "nop", "wait", "branch", "label"]
    print(f"{'region':>18} " + " ".join(f"{k:>10}" for k in keys))
    tot = Counter()
    for n, c in regions:
        print(f"{n:>18} " + " ".join(f"{c[k]:>10}" for k in keys)); tot.update(c)
    print(f"{'total (static)':>18} " + " ".join(f"{tot[k]:>10}" for k in keys))

if __name__ == "__main__":
    main()
