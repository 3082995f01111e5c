#!/usr/bin/env python3
"""The timed CPU leg of bench.py (`cpu_baseline`), as its OWN PROCESS.  TEST / BASELINE INFRASTRUCTURE: like everything under oracle/ it is executed only by
bench.py's cpu_baseline leg (and tests); the product never imports it.

Why a process of its own: the figure wobbled 2x between rounds (6.43 / 3.21 / 5.65 pairs/s on one code path).  Inside bench.py the
CPU leg shared its process with the HIP runtime's helper threads, probed three thread counts (each probe spawning intra-op workers that kept the affinity mask they were
born with), and `os.sched_setaffinity(0, ...)` moved only the calling thread.  Here the affinity mask is set BEFORE torch is imported, so every thread of the process
inherits it; the thread count is FIXED; the cores are whole L3 domains (CCDs) of ONE socket read from /sys/devices/system/cpu/*/{topology,cache}, one hardware thread per
physical core, never the domain that holds CPU 0 (where the kernel parks its housekeeping); and the same k pairs are timed FIVE times — `value` is the median, the spread
is reported.

What is timed: the oracle's torch-kernel backend (`oracle/lightglue_oracle.py`, backend="torch": the restatement of the reference's CPU fp32 path on the ATen CPU kernels
the reference itself computes with, pinned against the reference's fixtures by tests/test_oracle_golden.py) — `kind: "port"`.  The reference is a PYTHON module: it can
be imported in the build container only and may not travel to the GPU box in any form, so `kind: "reference"` exists only where /root/reference is mounted (then the
unmodified module is timed beside the port, same cores, same method: /root/reference/benchmark.py:18-43 = warm-up runs, then timed repetitions of forward).

usage: python oracle/cpu_leg.py --n 1024 --m 1024 --out /tmp/x.npz [--threads 16] [--budget 20] [--dim 256] [--recipe A] [--wseed 0] [--conf '{"depth_confidence": -1, ...}']
prints ONE JSON line; the per-pair outputs of the first round go to --out (matches0/1, matching_scores0/1 per pair) for the parent's parity check."""
import argparse
import glob
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
REFERENCE_FILE = Path("/root/reference/lightglue/lightglue.py")


def _read(path, default=None):
    try:
        return open(path).read().strip()
    except OSError:
        return default


def _cpulist(text):
    out = []
    for part in (text or "").split(","):
        part = part.strip()
        if not part:
            continue
        a, _, b = part.partition("-")
        out += list(range(int(a), int(b or a) + 1))
    return out


def choose_cores(want: int):
    """`want` physical cores as whole L3 domains of one socket, one hardware thread each, avoiding CPU 0's domain when there is a choice.
    Returns (logical CPU ids, description)."""
    allowed = sorted(os.sched_getaffinity(0))
    phys = {}      # (package, core_id) -> lowest allowed logical cpu of that core
    l3_of = {}
    for cpu in allowed:
        base = f"/sys/devices/system/cpu/cpu{cpu}"
        pkg, core = _read(f"{base}/topology/physical_package_id", "0"), _read(f"{base}/topology/core_id", str(cpu))
        phys.setdefault((pkg, core), cpu)
        l3 = None
        for idx in glob.glob(f"{base}/cache/index*"):
            if _read(f"{idx}/level") == "3":
                l3 = _read(f"{idx}/shared_cpu_list")
        l3_of[cpu] = (pkg, l3 or f"pkg{pkg}")
    cores = sorted(phys.values())
    domains = {}
    for c in cores:
        domains.setdefault(l3_of[c], []).append(c)
    zero_dom = l3_of.get(0)
    by_pkg = {}
    for key, members in domains.items():
        by_pkg.setdefault(key[0], []).append((key, members))
    best = None
    for pkg, doms in sorted(by_pkg.items()):
        doms.sort(key=lambda kv: (kv[0] == zero_dom, min(kv[1])))       # CPU 0's domain last
        picked, used = [], []
        for key, members in doms:
            if len(picked) >= want:
                break
            picked += members; used.append(key[1])
        picked = picked[:want]
        if best is None or len(picked) > len(best[0]) or (len(picked) == len(best[0]) and 0 not in picked and 0 in best[0]):
            best = (picked, pkg, used)
    picked, pkg, used = best
    if 0 in picked and len(cores) > len(picked):      # small boxes: still keep CPU 0 out if another core is free
        spare = [c for c in cores if c not in picked]
        picked[picked.index(0)] = spare[0]
    return sorted(picked), f"{len(picked)} physical cores of socket {pkg}, L3 domains {used} (one hardware thread per core; {len(cores)} physical cores / {len(allowed)} logical CPUs allowed)"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, required=True); ap.add_argument("--m", type=int, required=True)
    ap.add_argument("--dim", type=int, default=256); ap.add_argument("--recipe", default="A"); ap.add_argument("--wseed", type=int, default=0)
    ap.add_argument("--conf", default='{"depth_confidence": -1, "width_confidence": -1}')
    ap.add_argument("--threads", type=int, default=16); ap.add_argument("--budget", type=float, default=20.0); ap.add_argument("--max-pairs", type=int, default=32)
    ap.add_argument("--rounds", type=int, default=5); ap.add_argument("--out", default=None); ap.add_argument("--no-reference", action="store_true")
    ap.add_argument("--checkpoint", default=None, help="a state dict file (module-tree names, torch.save) instead of the seeded recipe weights (bench.py --checkpoint)")
    args = ap.parse_args()

    cores, where = choose_cores(args.threads)
    os.sched_setaffinity(0, cores)                    # BEFORE torch is imported: every thread this process ever creates inherits the mask
    threads = len(cores)
    os.environ["OMP_NUM_THREADS"] = str(threads); os.environ["MKL_NUM_THREADS"] = str(threads)
    sys.path.insert(0, str(ROOT))
    import numpy as np
    import torch
    from lightglue_amd import synthetic
    from oracle import lightglue_oracle as O
    torch.set_num_threads(threads); torch.set_grad_enabled(False)
    try:
        torch.set_num_interop_threads(1)
    except RuntimeError:
        pass

    conf_kw = json.loads(args.conf)
    conf = O.make_conf(**conf_kw)
    kw = dict(synthetic.RECIPE_D_DATA) if args.recipe == "D" else {}
    sd = synthetic.make_state_dict(args.wseed, recipe=args.recipe, input_dim=args.dim)
    if args.checkpoint:
        sd = {k: v.float().numpy() for k, v in torch.load(args.checkpoint, map_location="cpu").items() if torch.is_tensor(v)}
    n, m = args.n, args.m
    fwd = lambda d: O.forward(sd, conf, d, backend="torch")

    def pinned_threads():
        ok = 0; tot = 0
        for t in os.listdir("/proc/self/task"):
            tot += 1
            try:
                ok += set(os.sched_getaffinity(int(t))) <= set(cores)
            except OSError:
                pass
        return ok, tot

    probe = synthetic.make_batch(999, 1, n, m, args.dim, **kw)
    fwd(probe)                                         # warm-up: thread pool, allocator, first-touch
    t_pair = float("inf")
    for _ in range(2):                                 # (the faster of two: one probe caught in a burst of foreign load would shrink the sample)
        t0 = time.perf_counter(); fwd(probe); t_pair = min(t_pair, time.perf_counter() - t0)
    k = int(max(1, min(args.max_pairs, args.budget / args.rounds / max(t_pair, 1e-3))))
    batches = [synthetic.make_batch(1 + i, 1, n, m, args.dim, **kw) for i in range(k)]       # == pairs 0..k-1 of rank 0's GPU batch
    rates, first, per_fwd = [], None, []
    for rnd in range(args.rounds):
        out, ts = [], []
        for d in batches:                              # every forward timed on its own: the host is shared, and a burst of foreign load must not decide a round
            t0 = time.perf_counter(); out.append(fwd(d)); ts.append(time.perf_counter() - t0)
        rates.append(k / sum(ts)); per_fwd.append(ts)
        if first is None:
            first = out
    # value = 1 / median forward time over all rounds x pairs (the reference benchmark's statistic is a mean over repetitions of ONE pair, benchmark.py:18-43; the
    # median is the robust form of it); the raw per-round rates and both spreads are reported beside it
    round_medians = [float(np.median(ts)) for ts in per_fwd]
    t_med = float(np.median(np.concatenate(per_fwd)))
    value = 1.0 / t_med
    res = {"value": value, "unit": "image-pairs/s", "cores": threads, "kind": "port",
           "sample": f"{k} pair(s) N={n} M={m} of the benchmark's own batch, each forward timed on its own, {args.rounds} rounds (value = 1 / median forward time over {k * args.rounds} forwards; "
                     f"1 warm-up pair), 9 layers, fp32 port of the reference CPU path (oracle/ on torch's CPU kernels) in its own process, {threads} threads on {where}",
           # per round, the SAME statistic as `value` (1 / median forward time of the round) and its spread; the mean-based rates (pairs / summed time) are kept beside
           # them: the host is shared with the pool's other tenants, and a burst of foreign load lengthens a few forwards of a round (p90 below) without moving its median
           "rounds_pairs_per_s": [round(1.0 / t, 3) for t in round_medians], "round_spread": round((max(round_medians) - min(round_medians)) / t_med, 4),
           "round_median_forward_ms": [round(t * 1e3, 2) for t in round_medians],
           "rounds_mean_pairs_per_s": [round(r, 3) for r in rates], "round_mean_spread": round((max(rates) - min(rates)) / float(np.median(rates)), 4),
           "forward_ms_p10_p50_p90": [round(float(np.percentile(np.concatenate(per_fwd), q)) * 1e3, 2) for q in (10, 50, 90)], "pairs_per_round": k,
           "threads_pinned": "%d of %d threads of the process inside the chosen cores" % pinned_threads(), "cpus": cores}
    # one thread (same pinning): the scalar-port figure
    torch.set_num_threads(1)
    if t_pair * threads < 30.0:
        fwd(batches[0]); t0 = time.perf_counter(); fwd(batches[0]); res["one_thread_pairs_per_s"] = round(1.0 / (time.perf_counter() - t0), 3)
    # cfg #1 (N = M = 512, B = 1, fp32, 256-d, pruning off) at 1 and `threads` threads, median of 5 after a warm-up
    if args.dim == 256:
        conf1 = O.make_conf(depth_confidence=-1, width_confidence=-1)
        sd1 = sd if args.recipe in ("A", "D") else synthetic.make_state_dict(0, recipe="A")
        d512 = synthetic.make_batch(1, 1, 512, 512)
        cfg1 = {}
        for th in (1, threads):
            torch.set_num_threads(th)
            O.forward(sd1, conf1, d512, backend="torch")
            ts = []
            for _ in range(5):
                t0 = time.perf_counter(); O.forward(sd1, conf1, d512, backend="torch"); ts.append(time.perf_counter() - t0)
            cfg1[f"{th} thread(s)"] = {"pairs_per_s": round(1.0 / float(np.median(ts)), 3), "ms_median": round(float(np.median(ts)) * 1e3, 2), "ms_min_max": [round(min(ts) * 1e3, 2), round(max(ts) * 1e3, 2)]}
        res["cfg1_n512_b1"] = cfg1
    torch.set_num_threads(threads)
    nonadaptive = conf_kw.get("depth_confidence", 1) <= 0 and conf_kw.get("width_confidence", 1) <= 0
    if REFERENCE_FILE.exists() and not args.no_reference and args.dim == 256 and n == m and nonadaptive:
        # build container only: the UNMODIFIED reference beside the port — same process, same cores, benchmark.py:18-43's method
        try:
            import importlib.util
            spec = importlib.util.spec_from_file_location("lg_ref", str(REFERENCE_FILE))
            lg = importlib.util.module_from_spec(spec); spec.loader.exec_module(lg)
            model = lg.LightGlue(features=None, depth_confidence=-1, width_confidence=-1).eval()
            model.load_state_dict({kk: torch.from_numpy(v) for kk, v in sd.items()}, strict=False)
            ref = {}
            for nn in (512, n):
                td = {a: {b: torch.from_numpy(v) for b, v in d.items()} for a, d in synthetic.make_batch(1, 1, nn, nn).items()}
                for th in (1, threads):
                    torch.set_num_threads(th)
                    model(td); model(td)
                    ts = []
                    for _ in range(5):
                        t0 = time.perf_counter(); model(td); ts.append(time.perf_counter() - t0)
                    ref[f"N={nn} {th} thread(s)"] = round(1.0 / float(np.median(ts)), 3)
            res["reference_pairs_per_s"] = ref
            res["port_pairs_per_s"] = value
            res["kind"] = "reference"
            res["value"] = ref[f"N={n} {threads} thread(s)"]
            res["sample"] = (f"UNMODIFIED reference (lightglue.py loaded standalone), CPU fp32, B=1, N=M={n}, 2 warm-up + 5 timed forwards (median), {threads} threads on {where}; "
                             f"the port's own timing is in port_pairs_per_s")
        except Exception as exc:  # pragma: no cover
            res["reference_error"] = repr(exc)[:200]
    if args.out:
        np.savez(args.out, **{f"{key}_{i}": np.asarray(o[key]) for i, o in enumerate(first) for key in ("matches0", "matches1", "matching_scores0", "matching_scores1")}, pairs=k)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
