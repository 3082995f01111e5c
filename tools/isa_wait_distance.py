#!/usr/bin/env python3
"""How far ahead of its first use was every global load issued?  Walks the gfx950 assembly of one kernel in text order, models the in-order
vmcnt counter, and prints for every `s_waitcnt vmcnt(N)` the distance (in MFMA instructions of this wave, and in all instructions) between the
wait and the issue of the YOUNGEST memory operation the wait forces to complete.  A small distance = the wave stalls for (most of) a memory
round trip there: an un-pipelined prefetch, usually hipcc sinking a load next to its use (or hoisting a consumer up to its load).
usage: isa_wait_distance.py file.s <kernel-name-prefix> [max_mfma_distance_to_print=8]"""
import re
import sys


def body_of(path, prefix):
    out, f = [], False
    for l in open(path).read().splitlines():
        if l.startswith(prefix):
            f = True
            continue
        if f and l.startswith(".Lfunc_end"):
            break
        t = l.strip()
        if f and t and not t.startswith(";"):
            out.append(t)
    return out


def main():
    body = body_of(sys.argv[1], sys.argv[2])
    lim = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    q = []          # outstanding VMEM ops: (index, mfma count at issue, text)
    mf = 0
    last_bar = 0
    for i, t in enumerate(body):
        if t.startswith("v_mfma"):
            mf += 1
        elif t.startswith("global_load") or t.startswith("global_store") or t.startswith("buffer_"):
            q.append((i, mf, t.split()[0]))
        elif t.startswith("s_barrier"):
            last_bar = i
        elif re.match(r"^\.LBB", t):
            # a join: the compiler's own model is conservative here, ours just keeps going in text order
            pass
        elif t.startswith("s_waitcnt") and "vmcnt" in t:
            n = int(re.search(r"vmcnt\((\d+)\)", t).group(1))
            if len(q) > n:
                must = q[: len(q) - n]
                y = must[-1]
                d_m, d_i = mf - y[1], i - y[0]
                if d_m <= lim:
                    nxt = next((b for b in body[i + 1:i + 4] if not b.startswith("s_")), "")
                    print(f"+{i:5d} {t:32s} waits for {y[2]} issued {d_i:4d} instr / {d_m:3d} MFMAs earlier (+{y[0]}); {len(must)} op(s) retire; since barrier {i - last_bar}; next: {nxt[:60]}")
                q = q[len(q) - n:]
    print(f"{len(body)} instructions, {mf} MFMAs")
    if "--lds" in sys.argv:   # the same for LDS reads (lgkmcnt; scalar loads share the counter and return out of order: prologue lines are approximate)
        q, mf = [], 0
        for i, t in enumerate(body):
            if t.startswith("v_mfma"):
                mf += 1
            elif t.startswith("ds_") or t.startswith("s_load") or t.startswith("s_buffer_load"):
                q.append((i, mf, t.split()[0]))
            elif t.startswith("s_waitcnt") and "lgkmcnt" in t:
                n = int(re.search(r"lgkmcnt\((\d+)\)", t).group(1))
                if len(q) > n:
                    y = q[len(q) - n - 1]
                    if mf - y[1] <= lim and y[2].startswith("ds_read"):
                        print(f"+{i:5d} {t:32s} waits for {y[2]} issued {i - y[0]:4d} instr / {mf - y[1]:3d} MFMAs earlier")
                    q = q[len(q) - n:]


if __name__ == "__main__":
    main()
