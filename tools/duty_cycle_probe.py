#!/usr/bin/env python3
"""Does the power limit set the two big kernels' launch time?    The same cfg #2 forward, the same kernels, but with an idle gap between forwards: the
board's average power falls with the duty cycle, and if the launch times are bound by the power limit they must fall with it (the clock the chip grants rises); if
they were bound by what the waves do cycle by cycle they would not move.  Per gap: HIP-event time of the fused tail and of the attention launches, board power and
sclk sampled over the whole loop (hwmon), duty cycle = GPU-busy time / wall time.

Result (profiles/r06t2_duty_cycle.log, r06t3_duty_cycle_b1.log): inconclusive for the power question, useful for operations — with gaps the launches get SLOWER (cfg #2: tail 232 -> 272 us
at 23 % duty and 468 W; B = 1, N = 2 048: attention 36.3 -> 39.9 us with sclk still at 2.39 GHz and 311 W), from the first millisecond of idling on.  The shader clock is not what drops; some other
power state of the chip (fabric / memory side) does and is paid back over the next forward.  A tight benchmark loop is therefore the chip's best case: a service with sporadic requests sees
+5 ... +15 % per forward unless the performance level is pinned (an operations setting, not this library's).

usage: duty_cycle_probe.py [--gaps 0,4,8,16,32] [--steps 60] [--pairs 32] [--kpts 1024]"""
import argparse
import glob
import sys
import threading
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import gpu_util
from lightglue_amd import synthetic as synth


def hwmon():
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    buf = C.create_string_buffer(64)
    if hip.hipDeviceGetPCIBusId(buf, 64, 0) != 0:
        return None
    hw = glob.glob(f"/sys/bus/pci/devices/{buf.value.decode().lower()}/hwmon/hwmon*")
    return hw[0] if hw else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gaps", default="0,4,8,16,32")
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--pairs", type=int, default=32)
    ap.add_argument("--kpts", type=int, default=1024)
    a = ap.parse_args()
    B, n = a.pairs, a.kpts
    model = gpu_util.make_model(synth.make_state_dict(0, recipe="A"), "f16x3", depth_confidence=-1, width_confidence=-1)
    data = gpu_util.to_torch(synth.make_batch(1, B, n, n))
    model.reserve(B, n, n, "cuda")
    for _ in range(10):
        model(data)
    hw = hwmon()
    rd = (lambda name: int(open(f"{hw}/{name}").read())) if hw else None
    for gap_ms in [float(x) for x in a.gaps.split(",")]:
        samples, stop = [], threading.Event()

        def sampler():
            while not stop.is_set():
                if rd:
                    samples.append((rd("power1_input") / 1e6, rd("freq1_input") / 1e6))
                time.sleep(0.02)
        for _ in range(20):            # settle at this duty cycle
            model(data); time.sleep(gap_ms / 1e3)
        th = threading.Thread(target=sampler, daemon=True); th.start()
        model.profile(True, "cuda")
        busy, t0 = 0.0, time.perf_counter()
        for _ in range(a.steps):
            t1 = time.perf_counter(); model(data); torch.cuda.synchronize(); busy += time.perf_counter() - t1
            time.sleep(gap_ms / 1e3)
        wall = time.perf_counter() - t0
        prof = model.profile_read("cuda"); model.profile(False, "cuda")
        stop.set(); th.join()
        per = {k: 1e3 * v[0] / v[1] for k, v in prof.items() if v[1] > 0}       # us per launch site
        W = np.median([s[0] for s in samples]) if samples else float("nan")
        mhz = np.median([s[1] for s in samples]) if samples else float("nan")
        print(f"gap {gap_ms:5.1f} ms: duty {100 * busy / wall:5.1f} %  forward {1e3 * busy / a.steps:6.3f} ms  tail {per.get('fused_tail', 0):6.1f} us  attention {0.5 * (per.get('attn_self', 0) + per.get('attn_cross', 0)):6.1f} us  "
              f"board {W:6.0f} W  sclk {mhz:5.0f} MHz (medians of {len(samples)} samples)", flush=True)


if __name__ == "__main__":
    main()
