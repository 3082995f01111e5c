#!/usr/bin/env python3
"""Occupancy timeline of the fused tail kernel: every workgroup's life span on the 100 MHz wall clock and the CU it ran on
(profiling tap of the product library: engine option tail_timing = 1)."""
import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import gpu_util
from lightglue_amd import synthetic as synth
sd = synth.make_state_dict(0, recipe="A")
model = gpu_util.make_model(sd, "f16x3", depth_confidence=-1, width_confidence=-1)
data = gpu_util.to_torch(synth.make_batch(1, 32, 1024, 1024))
for kv in sys.argv[1:]:
    k, v = kv.split("="); model.set_option(k, int(v))
for _ in range(3): model(data)
model.set_option("tail_timing", 1); model(data); torch.cuda.synchronize()
d = model.debug_read("TAILDBG", np.int64).reshape(-1, 8, 8)
d = d[d[:, 0, 0] != 0]
t0 = (d[:, :, 6] & ((1 << 44) - 1)).min(1); raw = d[:, :, 7]; t1 = (raw & ((1 << 44) - 1)).max(1)   # bits 0..43: wall clock, 44..47: XCC_ID, 48..55: HW_ID bits 8..15
cu = (raw[:, 0] >> 44) & 0xFFF
cyc = (d[:, :, 5] - d[:, :, 0]).max(1)
base = t0.min(); life = (t1 - t0) / 100.0
assert 0 < t1.max() - base < 10_000_000, ("implausible span", int(t1.max() - base))   # (an unmasked id field once made this 1e14 ticks)
print("workgroups", len(d), " kernel span %.1f us" % ((t1.max() - base) / 100.0), " (the LAST tail launch of the forward: no fused projection)")
print("workgroup life us: median %.1f p10 %.1f p90 %.1f max %.1f; stamps 0->5 median %.0f cycles" % (np.median(life), np.percentile(life, 10), np.percentile(life, 90), life.max(), np.median(cyc)))
print("shader clock while a tail workgroup lives (stamp cycles / wall life; the stamps span slightly less than the life): median %.0f MHz" % np.median(cyc / life))
ts = np.linspace(0, t1.max() - base, 60)
print("live workgroups at 60 equally spaced times:", [int((((t0 - base) <= t) & ((t1 - base) > t)).sum()) for t in ts])
u = np.unique(cu); gaps = []; per = []
for c in u:
    sel = np.where(cu == c)[0]; o = sel[np.argsort(t0[sel])]; per.append(len(o))
    gaps += list((t0[o][1:] - t1[o][:-1]) / 100.0)
gaps = np.array(gaps)
print("distinct CUs", len(u), " workgroups per CU min/max", min(per), max(per))
if len(d) % 256 == 0:   # does workgroup id run on CU slot id % 256?  (no: only its XCD, id % 8, is fixed — which is what the tail's xcd_remap tile order relies on)
    ids = np.arange(len(d))
    same = sum(1 for c in u if np.unique(ids[cu == c] % 256).size == 1)
    print("CUs whose workgroups all have the same id %% 256: %d of %d" % (same, len(u)))
    xcc = (raw[:, 0] >> 44) & 0xF
    print("workgroups whose XCC_ID == id %% 8: %d of %d" % (int((xcc == ids % 8).sum()), len(d)))
print("gap between consecutive workgroups on a CU, us: median %.2f p10 %.2f p90 %.2f  (negative = overlap)" % (np.median(gaps), np.percentile(gaps, 10), np.percentile(gaps, 90)))
print("first start spread us: %.1f; last end - median end of final round %.1f" % (np.percentile((t0 - base) / 100.0, 24), (t1.max() - np.median(np.sort(t1)[-256:])) / 100.0))
