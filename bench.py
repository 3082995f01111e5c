#!/usr/bin/env python3
"""Throughput benchmark of the MI355X-native LightGlue matcher forward path.

Metric (BASELINE.json): image-pairs/s at N=M=1024 SuperPoint-dim descriptors, 9 layers, pruning
and early-stop OFF, batch 32 pairs per GPU, inputs resident in HBM, synthetic data / seeded weights.
One "step" = one LightGlue.forward over one batch of 32 pairs (the whole reference forward: layers,
log-assignment, match filtering and the ragged match lists).

    python bench.py [--gpus N --steps K --warmup W] [--precision f16x3|bf16|fp16|fp32] [--attention fp16]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...        (no launcher: bench.py spawns the same N ranks itself on a free loopback port)

N > 1: one process per GPU, every rank matches its own 32 pairs (weak scaling, no data-path
collective) and the match indices are all-gathered over RCCL each step (lightglue_amd.parallel).

Prints ONE JSON line on rank 0 with the driver's contract plus
  roofline     — dominant kernel class, measured with HIP events on the launch stream over the timed
                 region (engine-side events, include/lightglue_amd.h lg_engine_profile_*)
  cpu_baseline — the port of the reference's CPU path (oracle/, on torch's CPU kernels) timed on this host's
                 cores on a bounded sample of the same workload (N=1, rank 0 only)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

from lightglue_amd import LightGlue, PairShardedMatcher  # noqa: E402
from lightglue_amd import synthetic  # noqa: E402

N_KPTS = 1024
PAIRS_PER_GPU = 32
D = 256
L = 9

# BASELINE.json `configs` (index = the --config number; 1 is the reference's own CPU case, timed in cpu_baseline.cfg1_*).  Per GPU and step:
# `pairs` image pairs of n x m keypoints with dim-d descriptors; adaptive = the reference's defaults depth_confidence 0.95 / width_confidence 0.99
# (recipe C weights: pairs stop at mixed depths and prune), else both off.  The default (2) is the configuration the metric is quoted on.
CONFIGS = {
    2: dict(label="BASELINE configs[1]: SuperPoint 256-d, N=M=1024, 9 layers, pruning OFF, batch=32 per GPU", pairs=PAIRS_PER_GPU, n=N_KPTS, m=N_KPTS, dim=256, recipe="A", wseed=0, adaptive=False),
    3: dict(label="BASELINE configs[2]: SuperPoint 256-d, N=M=2048, depth_confidence=0.95 width_confidence=0.99 (adaptive pruning ON), batch=16 per GPU", pairs=16, n=2048, m=2048, dim=256, recipe="C", wseed=0, adaptive=True),
    4: dict(label="BASELINE configs[3]: DISK 128-d, N=M=4096, 9 layers, pruning OFF, batch=32 per GPU (256 pairs pair-sharded over 8 GPUs, RCCL gather)", pairs=32, n=4096, m=4096, dim=128, recipe="A", wseed=0, adaptive=False),
    5: dict(label="BASELINE configs[4]: ALIKED 128-d, asymmetric N=2048 M=512, pruning ON, batch=64 per GPU (ragged / compaction stress)", pairs=64, n=2048, m=512, dim=128, recipe="C", wseed=2, adaptive=True),
}

# ---- algorithmic FLOPs (2 per MAC), SURVEY.md §8d.  Per launch of each kernel class over `pairs` pairs.
def flops_per_launch(pairs: int, n: int, m: int):
    pts = pairs * (n + m)
    return {
        "gemm_qkv_self": 2.0 * pts * 256 * 768,
        "gemm_qkv_cross": 2.0 * pts * 256 * 512,
        "gemm_out_proj": 2.0 * pts * 256 * 256,
        "gemm_ffn0": 2.0 * pts * 512 * 512,
        "gemm_ffn3_resid": 2.0 * pts * 512 * 256,
        "gemm_final_proj": 2.0 * pts * 256 * 256,
        # fused block tail: the reference's out_proj + ffn.0 + ffn.3 (the kernel folds out_proj into ffn.0 and
        # issues fewer MFMA FLOPs than this algorithmic count)
        "fused_tail": 2.0 * pts * (256 * 256 + 512 * 512 + 512 * 256),
        "attn_self": pairs * 4.0 * 256 * (n * n + m * m),          # QK^T + PV per image
        "attn_cross": pairs * 8.0 * 256 * n * m,                   # two directions, S computed twice
        "sim": pairs * 2.0 * 256 * n * m,
    }


# HBM bytes per launch measured with rocprofv3 PMC passes of THIS command (separate FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE x 2 per
# the gfx950 correction in MI355X_MICROARCH.md), written by tools/pmc_traffic.py to profiles/pmc_traffic.json together with a digest
# of the kernel sources they were measured on.  bench.py reports them only while that digest matches the tree (a changed kernel ->
# "traffic": null with the reason), so the constant cannot go stale silently.
PMC_TRAFFIC_FILE = ROOT / "profiles" / "pmc_traffic.json"


def kernel_source_digest() -> str:
    import hashlib
    h = hashlib.sha256()
    for f in sorted((ROOT / "lightglue_amd" / "csrc").glob("lg_*")):
        h.update(f.name.encode()); h.update(f.read_bytes())
    return h.hexdigest()[:16]


def pmc_traffic(key: str):
    """(bytes per launch or None, source note) for `key` = "<precision>/<attention>/B<pairs>/N<kpts>/<kernel class>"."""
    try:
        rec = json.loads(PMC_TRAFFIC_FILE.read_text())
    except (OSError, ValueError):
        return None, "profiles/pmc_traffic.json missing: run tools/pmc_traffic.py on the GPU box"
    if rec.get("kernel_source_digest") != kernel_source_digest():
        return None, f"stale: profiles/pmc_traffic.json was measured on kernel sources {rec.get('kernel_source_digest')}, the tree is {kernel_source_digest()} (re-run tools/pmc_traffic.py)"
    v = rec.get("bytes_per_launch", {}).get(key)
    return v, rec.get("source", "profiles/pmc_traffic.json") if v is not None else f"no entry {key} in profiles/pmc_traffic.json"


def hbm_bytes_assign(pairs: int, n: int, m: int) -> float:
    """two fp32 read sweeps of the similarity matrix (row/column LSE, then score/argmax)."""
    return pairs * 8.0 * n * m


def flops_per_pair(n: int, m: int) -> float:
    """80.53 GFLOP at n = m = 1024 (shared-S accounting for cross attention)."""
    per_pt_layer = 1_310_720 + 1_179_648
    return L * (per_pt_layer * (n + m) + 4 * D * (n * n + m * m) + 6 * D * n * m) + 2 * D * D * (n + m) + 2 * D * n * m


# time ratio port / real reference, measured side by side in the build container (profiles/r03_cpu_reference.md,
# tools/cpu_reference_table.py: 8 vCPU Intel Xeon @ 2.10 GHz, torch 2.10 CPU fp32, B = 1, pruning off; /root/reference loaded
# standalone as tools/make_golden.py does; re-measured by oracle/cpu_leg.py whenever /root/reference is mounted).  The timed port is the
# oracle's restatement running on torch's CPU kernels (oracle backend="torch": ATen linear / matmul / softmax / layer_norm / gelu and the
# fused fp32 SDPA the reference itself calls), so it runs within 1.0-1.35x of the reference on the same cores.
# `cpu_baseline.reference_estimate_pairs_per_s` = value x this ratio.
PORT_OVER_REFERENCE_TIME = {"1 thread": {"N=512": 1.00, "N=1024": 1.02}, "8 threads": {"N=512": 1.10, "N=1024": 1.34}}
CPU_LEG_THREADS = 16      # fixed: two whole L3 domains (CCDs) of one socket on the pool's EPYC 9575F hosts


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _port_over_reference(n, threads):
    """Measured time ratio port / reference for the nearest measured (N, threads) cell of profiles/r03_cpu_reference.md."""
    row = PORT_OVER_REFERENCE_TIME["1 thread" if threads <= 2 else "8 threads"]
    return row["N=512" if n <= 768 else "N=1024"]


def cpu_baseline(sd, n, m, gpu_out=None, budget_s=20.0, max_pairs=32, recipe="A", conf_kw=None, dim=256, wseed=0, checkpoint=None):
    """CPU leg (rank 0, N = 1 only; ~30 s in total) — run by oracle/cpu_leg.py IN ITS OWN PROCESS: affinity set before torch is imported (whole L3 domains of one
    socket, one hardware thread per physical core, not CPU 0's domain), a FIXED thread count, the same k pairs timed five times (median + spread), cfg #1 (N = 512,
    B = 1) at 1 and N threads beside it.  The port of the reference's CPU fp32 path (oracle/, torch-kernel backend) is timed on pairs of the SAME seeded batch the
    GPU matched (pair seeds 1, 2, ...), so the same calls also CHECK the GPU result (returned as the second value).  `kind` is "port" on the GPU box: the reference
    is a Python module, which may not travel there; where /root/reference is mounted (build container) the unmodified module is timed beside the port."""
    import subprocess
    import tempfile
    conf = conf_kw if conf_kw is not None else dict(depth_confidence=-1, width_confidence=-1)
    with tempfile.TemporaryDirectory() as tmp:
        out_npz = os.path.join(tmp, "cpu_leg.npz")
        cmd = [sys.executable, str(ROOT / "oracle" / "cpu_leg.py"), "--n", str(n), "--m", str(m), "--dim", str(dim), "--recipe", recipe, "--wseed", str(wseed),
               "--conf", json.dumps(conf), "--threads", str(CPU_LEG_THREADS), "--budget", str(budget_s), "--max-pairs", str(max_pairs), "--out", out_npz]
        if checkpoint:
            cmd += ["--checkpoint", str(checkpoint)]
        env = {k: v for k, v in os.environ.items() if k not in ("OMP_NUM_THREADS", "MKL_NUM_THREADS")}
        # glibc malloc: keep the forward's large temporaries (16 MB attention matrices, ...) in the heap instead of mmap / munmap per tensor — page-fault churn was
        # the main source of the round-to-round wobble, and costs the CPU side ~25 % (build container: 3.21 -> 4.08 pairs/s, round spread 13 % -> 3.5 %)
        env.update(MALLOC_TRIM_THRESHOLD_="4294967296", MALLOC_MMAP_THRESHOLD_="4294967296", MALLOC_TOP_PAD_="268435456")
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
        if p.returncode != 0:
            return {"error": p.stderr[-500:], "kind": "port"}, None
        res = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
        z = np.load(out_npz)
        k = int(z["pairs"])
        refs = [{key: z[f"{key}_{i}"] for key in ("matches0", "matches1", "matching_scores0", "matching_scores1")} for i in range(k)]
    res.update({"cpu_model": _cpu_model(), "logical_cores": os.cpu_count(), "allocator": "glibc malloc with MALLOC_MMAP_THRESHOLD_ / MALLOC_TRIM_THRESHOLD_ = 4 GiB (no mmap / munmap per tensor)",
                "port_over_reference_time_ratio": PORT_OVER_REFERENCE_TIME,
                "port_over_reference_source": "profiles/r03_cpu_reference.md (tools/cpu_reference_table.py: the unmodified reference and this port timed side by side in the build container); "
                                              "oracle/cpu_leg.py re-measures it wherever /root/reference is mounted (reference_pairs_per_s / port_pairs_per_s)"})
    if res.get("kind") == "port":
        # what the REAL reference would do on these cores, by the measured ratio (the GPU box has no /root/reference: a Python reference may not travel)
        res["reference_estimate_pairs_per_s"] = round(res["value"] * _port_over_reference(n, res["cores"]), 3)
        res["reference_estimate_note"] = (f"ratio measured at 1 / 8 threads on an 8-vCPU Xeon (build container), applied to {res['cores']} thread(s) of {_cpu_model()}: "
                                          "an extrapolation, not a measurement")
    parity = None
    if gpu_out is not None:
        parity = parity_block(gpu_out, refs, n, m, source=f"oracle (port of the reference CPU path on torch CPU kernels, fp32) on pairs 0..{k - 1} of the timed batch")
    return res, parity


def parity_block(gpu_out, refs, n, m, source, tol=1e-3, filter_threshold=0.1):
    """Index / score parity of the timed GPU batch against per-pair reference results (dicts with matches0/1, matching_scores0/1
    for B = 1).  A flip counts as explained only at a filter-threshold tie (score within tol of the threshold, one side -1)."""
    m0 = gpu_out["matches0"].cpu().numpy(); m1 = gpu_out["matches1"].cpu().numpy()
    s0 = gpu_out["matching_scores0"].cpu().numpy(); s1 = gpu_out["matching_scores1"].cpu().numpy()
    mism = unexplained = 0
    maxd = 0.0
    for b, r in enumerate(refs):
        for gm, gs, rm, rs in ((m0[b], s0[b], np.asarray(r["matches0"]).reshape(-1), np.asarray(r["matching_scores0"]).reshape(-1)),
                               (m1[b], s1[b], np.asarray(r["matches1"]).reshape(-1), np.asarray(r["matching_scores1"]).reshape(-1))):
            flip = gm != rm
            mism += int(flip.sum())
            thr_tie = ((gm == -1) | (rm == -1)) & ((np.abs(rs - filter_threshold) <= tol) | (np.abs(gs - filter_threshold) <= tol))
            unexplained += int((flip & ~thr_tie).sum())
            same = ~flip
            if same.any():
                maxd = max(maxd, float(np.abs(gs[same] - rs[same]).max()))
    return {"pairs": len(refs), "keypoints_compared": int(len(refs) * (n + m)), "index_mismatches": mism, "unexplained": unexplained,
            "max_dscore": maxd, "score_tolerance": tol, "source": source}


def batch_kwargs(recipe):
    """make_batch keywords of a recipe: "D" = trained-model statistics (descriptor norms in [0.5, 3], noisier copies)."""
    return dict(synthetic.RECIPE_D_DATA) if recipe == "D" else {}


def golden_parity(gpu_out, n, B, recipe="A", config=2, m=None):
    """The first pairs of the workload ARE a fixture produced by the real reference (tools/make_golden.py, weights seed 0, pair
    seeds 1, 2, ...): recipe A -> nonadaptive_1024_b4 (4 pairs), recipe D -> trained_stats_1024_b8 (8 pairs).  Only cfg #2's semantics (256-d,
    non-adaptive, weight seed 0, N = M = 1024, recipe A or D) ARE that fixture: any other --config / --kpts / --recipe returns None."""
    name, pairs = ("trained_stats_1024_b8", 8) if recipe == "D" else ("nonadaptive_1024_b4", 4)
    path = ROOT / "tests" / "golden" / f"{name}.npz"
    cfg = CONFIGS[config]
    if config != 2 or cfg["dim"] != 256 or cfg["adaptive"] or cfg["wseed"] != 0 or recipe not in ("A", "D") or n != 1024 or (m is not None and m != 1024) or B < pairs or not path.exists():
        return None
    z = np.load(path, allow_pickle=False)
    refs = [{k: z[k][b] for k in ("matches0", "matches1", "matching_scores0", "matching_scores1")} for b in range(pairs)]
    return parity_block(gpu_out, refs, n, n, source=f"tests/golden/{name}.npz (real reference, CPU fp32) = pairs 0..{pairs - 1} of the timed batch")


def bind_rank_to_gpu_numa(device_index: int) -> str:
    """Multi-GPU runs: keep a rank's host threads (Python, HIP runtime, RCCL proxy) on the CPUs of ITS GPU's NUMA node — on an 8-GPU node the launcher otherwise
    lets all ranks float over both sockets, and a rank whose launch thread sits on the far socket pays the cross-socket hop on every kernel launch and on the
    [3][B] host-block copy of every step.  The GPU's PCI address comes from the HIP runtime, its CPUs from sysfs (local_cpulist).  Never fatal: returns what it did."""
    try:
        import ctypes as C
        hip = C.CDLL("libamdhip64.so")
        buf = C.create_string_buffer(64)
        if hip.hipDeviceGetPCIBusId(buf, 64, int(device_index)) != 0:
            return "unchanged (hipDeviceGetPCIBusId failed)"
        bdf = buf.value.decode().lower()
        text = open(f"/sys/bus/pci/devices/{bdf}/local_cpulist").read().strip()
        cpus = []
        for part in text.split(","):
            if part:
                a, _, b = part.partition("-")
                cpus += list(range(int(a), int(b or a) + 1))
        cpus = sorted(set(cpus) & set(os.sched_getaffinity(0)))
        if not cpus:
            return f"unchanged (GPU {bdf}: empty local_cpulist)"
        os.sched_setaffinity(0, cpus)
        node = open(f"/sys/bus/pci/devices/{bdf}/numa_node").read().strip()
        return f"GPU {bdf} -> NUMA node {node}, {len(cpus)} CPUs ({text})"
    except Exception as exc:   # no sysfs entry, no permission, ...: run unpinned
        return f"unchanged ({type(exc).__name__}: {exc})"[:200]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--precision", default="f16x3", choices=["f16x3", "bf16", "fp16", "fp32"])
    ap.add_argument("--attention", default=None, choices=["fp16"], help="with --precision f16x3: the single-plane f16 attention (fast opt-in, outside the bar for sharp attention)")
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS), help="BASELINE.json config number (1-based as in DESIGN.md): 2 = the headline (default), 3 = N=M=2048 adaptive, "
                    "4 = DISK 128-d N=M=4096 (the pair-sharded 8-GPU config: use with --gpus N), 5 = ALIKED 128-d 2048x512 adaptive")
    ap.add_argument("--pairs", type=int, default=None, help="pairs per GPU per step (default: the config's)")
    ap.add_argument("--kpts", type=int, default=None, help="keypoints per image, N = M (default: the config's)")
    ap.add_argument("--recipe", default=None, choices=["A", "C", "D"], help="seeded weights / inputs (default: the config's): A (SURVEY 8c, the headline), C (adaptive: mixed stop depths) or D (trained-model "
                    "statistics: attention logit spread 25, LayerNorm gains in [0.5, 4], residual rms ~27, descriptor norms in [0.5, 3])")
    ap.add_argument("--checkpoint", default=None, help="weights from a state-dict file (module-tree names, e.g. tools/train_synthetic_checkpoint.py's) instead of the seeded recipe; the oracle leg loads the same file; "
                    "a labelled extra run, never the default line")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather-probe", action="store_true", help="one GPU: skip the K extra steps that measure what the (world-of-one) side-stream result gather costs a step")
    ap.add_argument("--no-pipeline", action="store_true", help="one GPU: synchronous forward per step (no deferred output assembly)")
    ap.add_argument("--inflight", type=int, default=None, help="one GPU: whole batches in flight (lightglue_amd.InflightMatcher: one engine + one HIP stream per lane, step i on lane i %% F).  Default: 1 at the "
                    "fixed-depth configs (2, 4: every launch fills the chip, a second lane measures -1 %% / +-0), 2 at the adaptive configs (3, 5), whose late layers cover a fraction of the chip once most pairs "
                    "have stopped (+3 ... +11 %%, profiles/r06t_ab_inflight.log, r06u_*)")
    ap.add_argument("--no-calibration", action="store_true", help="skip the 25 ms dense-MFMA spin that measures what the box sustains (profiling runs)")
    ap.add_argument("--no-fuse-next", action="store_true", help="run the q/k/v projections as their own kernels instead of inside the previous block's tail kernel")
    ap.add_argument("--unfused", action="store_true", help="use the per-op kernels instead of the fused block tail")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N` (no launcher): spawn the N ranks ourselves — one process per GPU through
        # torch.distributed.run on a free loopback port — and hand its exit code on.  Only rank 0 writes to stdout (the ONE JSON
        # line); the launcher's own chatter goes to stderr.  Under torchrun (WORLD_SIZE set) this branch is never taken.
        import socket
        import subprocess
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", "8")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
        sys.exit(subprocess.run(cmd, env=env).returncode)
    # The contract is ONE JSON line on stdout.  Native libraries write there too (RCCL prints its version banner on init, gloo its
    # "connected to N peer ranks" lines), so file descriptor 1 is pointed at stderr for the whole run and rank 0 writes the line to a
    # private duplicate of the original stdout.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # debugging aid: LG_BENCH_BACKEND=gloo LG_BENCH_ONE_GPU=1 runs several ranks on cuda:0 (flow test without a node)
    backend = os.environ.get("LG_BENCH_BACKEND", "nccl")
    if os.environ.get("LG_BENCH_ONE_GPU") == "1":
        local_rank = 0
    # bound BEFORE the process group exists, so that RCCL's proxy threads are born inside the mask as well
    rank_affinity = bind_rank_to_gpu_numa(local_rank) if world > 1 else "one GPU: not bound"
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run)"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    cfg = CONFIGS[args.config]
    n, m = (args.kpts, args.kpts) if args.kpts else (cfg["n"], cfg["m"])
    B = args.pairs or cfg["pairs"]
    dim, adaptive = cfg["dim"], cfg["adaptive"]
    args.recipe = args.recipe or cfg["recipe"]
    conf_kw = ({} if adaptive else dict(depth_confidence=-1, width_confidence=-1))
    if dim != 256:
        conf_kw["input_dim"] = dim
    sd = synthetic.make_state_dict(cfg["wseed"], recipe=args.recipe, input_dim=dim)
    if args.checkpoint:
        sd = {k: v.float().numpy() for k, v in torch.load(args.checkpoint, map_location="cpu").items() if torch.is_tensor(v)}
    model = LightGlue(features=None, precision=args.precision, attention_precision=args.attention, **conf_kw).eval()
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    if os.environ.get("LG_BENCH_ABLATION") == "1":   # timing ablations (variant libraries that compute garbage on purpose): no range guard, no parity block
        model.check_finite = False
    data_np = synthetic.make_batch(1 + rank * B, B, n, m, dim, **batch_kwargs(args.recipe))
    if os.environ.get("LG_BENCH_PERIODIC"):   # timing ablations only: every 64-key tile of an image holds the SAME keypoints, so a variant that races on K / V tiles still computes valid values
        per = int(os.environ["LG_BENCH_PERIODIC"])
        for img in ("image0", "image1"):
            for key in ("keypoints", "descriptors"):
                a = data_np[img][key]
                data_np[img][key] = np.ascontiguousarray(np.tile(a[:, :per], (1, a.shape[1] // per) + (1,) * (a.ndim - 2)))
    data = {k: {kk: torch.from_numpy(vv).to(dev) for kk, vv in v.items()} for k, v in data_np.items()}
    model.reserve(B, n, m, dev)
    for kv in os.environ.get("LG_BENCH_OPTS", "").split():   # A/B of engine options in one box, e.g. LG_BENCH_OPTS="tail_row_tiles=2 attn_rows=64"
        key, val = kv.split("=")
        model.set_option(key, int(val), dev)
    if args.no_fuse_next:
        model.set_option("fused_next", 0, dev)
    if args.unfused:
        model.set_option("fused_tail", 0, dev)
    sharded = PairShardedMatcher(model) if world > 1 else None
    if args.inflight is None:
        args.inflight = 2 if (adaptive and world == 1 and not args.no_pipeline) else 1
    lanes = None
    if args.inflight > 1 and world == 1 and not args.no_pipeline:
        from lightglue_amd import InflightMatcher
        lanes = InflightMatcher(model, args.inflight, dev)
        lanes.reserve(B, n, m)
        lane_pending = [None] * args.inflight
        lane_next = [0]

    pending = [None]

    def step():
        if lanes is not None:
            # F whole batches in flight: step i runs on lane i % F (its own engine and stream); its full output dict is built when the lane is used next
            k = lane_next[0]; lane_next[0] = (k + 1) % args.inflight
            prev, lane_pending[k] = lane_pending[k], lanes.submit(data)
            return prev.result() if prev is not None else None
        if sharded is not None:
            # the result gather of this step stays in flight on the side stream while the next step's forward runs;
            # every gather is waited for (and unpacked) one step later, the last one before the closing barrier
            prev, pending[0] = pending[0], sharded.issue_local(data, B * world)
            return prev.wait() if prev is not None else None
        if args.no_pipeline:
            return model(data)
        # one GPU: the forward's only host synchronisation (the ragged match lists need their sizes) is taken one step later
        # (LightGlue.forward_deferred), so Python assembles the outputs of step i while step i + 1 runs; every step's full
        # output dict is built inside the timed region, the last one before the closing barrier
        prev, pending[0] = pending[0], model.forward_deferred(data)
        return prev.result() if prev is not None else None

    def drain():
        if lanes is not None:
            last = None
            for i in range(args.inflight):
                k = (lane_next[0] + i) % args.inflight      # oldest first
                if lane_pending[k] is not None:
                    last, lane_pending[k] = lane_pending[k].result(), None
            return last
        if pending[0] is not None:
            last, pending[0] = pending[0], None
            return last.wait() if sharded is not None else last.result()
        return None

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize(dev)

    # Per-class kernel times (HIP events around every launch) are taken over the WARM-UP steps; in the timed region only
    # the dominant class keeps its events (the roofline leg needs exactly that kernel's launch duration, live), so the
    # headline is not taxed by ~90 event records per step.  With --warmup < 2 everything is timed in the timed region.
    warm_prof = {}
    dom_class = None
    if args.warmup >= 2:
        out = step()
        model.profile(True, dev)
        for _ in range(args.warmup - 1):
            out = step()
        torch.cuda.synchronize(dev)
        warm_prof = {k: (v[0] / (args.warmup - 1), v[1] / (args.warmup - 1)) for k, v in model.profile_read(dev).items() if v[1] > 0}
        model.profile(False, dev)
        fl0 = flops_per_launch(B, n, m)
        dom_class = max((k for k in warm_prof if k in fl0), key=lambda k: warm_prof[k][0])
    else:
        for _ in range(args.warmup):
            out = step()
    drain()

    def mfma_sustained():
        # what the matrix pipe of THIS box sustains (dense bf16 MFMA spin, ~25 ms): TFLOP/s and the shader clock it ran at
        import ctypes as C
        from lightglue_amd import _cabi
        tf, mhz = C.c_double(0.0), C.c_double(0.0)
        with torch.cuda.device(dev):
            _cabi.check(_cabi.load().lg_debug_mfma_sustained(C.byref(tf), C.byref(mhz), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        return round(tf.value, 1), round(mhz.value, 1)

    sustained_tflops, sustained_mhz = (None, None) if args.no_calibration else mfma_sustained()
    model.profile(True, dev, only=dom_class)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    last = drain()
    out = last if last is not None else out
    barrier()
    dt = time.perf_counter() - t0
    prof = model.profile_read(dev)
    model.profile(False, dev)

    def tail_clock_mhz():
        # the shader clock the chip gives the fused tail INSIDE a forward (power management): shader cycles between a workgroup's first and last stamp /
        # its life on the 100 MHz wall clock, median over the workgroups of the last tail launch (profiling tap of the product library, tools/tail_wall.py)
        try:
            model.set_option("tail_timing", 1, dev); model(data); torch.cuda.synchronize(dev)
            d = model.debug_read("TAILDBG", np.int64, dev).reshape(-1, 8, 8)
            d = d[d[:, 0, 0] != 0]
            t0 = (d[:, :, 6] & ((1 << 44) - 1)).min(1); t1 = (d[:, :, 7] & ((1 << 44) - 1)).max(1)
            cyc = (d[:, :, 5] - d[:, :, 0]).max(1)
            return round(float(np.median(cyc / ((t1 - t0) / 100.0))), 1) if len(d) else None
        except Exception:
            return None
        finally:
            model.set_option("tail_timing", 0, dev)

    def power_probe(seconds=1.6, settle=0.6):
        # the board's power sensor (hwmon of this GPU's PCI device) sampled while the SAME step loop runs: what the chip draws under this workload, against its
        # cap — the quantity that bounds the two big kernels — and the pairs per joule that follow.  Outside the timed region; never fatal.
        try:
            import ctypes as C
            import glob
            import threading
            hip = C.CDLL("libamdhip64.so")
            buf = C.create_string_buffer(64)
            if hip.hipDeviceGetPCIBusId(buf, 64, int(dev.index or 0)) != 0:
                return None
            hw = glob.glob(f"/sys/bus/pci/devices/{buf.value.decode().lower()}/hwmon/hwmon*")
            if not hw:
                return None
            rd = lambda name: int(open(f"{hw[0]}/{name}").read())
            cap = rd("power1_cap") / 1e6
            samples, stop = [], threading.Event()
            def sampler():
                t0 = time.perf_counter()
                while not stop.is_set():
                    samples.append((time.perf_counter() - t0, rd("power1_input") / 1e6, rd("freq1_input") / 1e6))
                    time.sleep(0.05)
            th = threading.Thread(target=sampler, daemon=True); th.start()
            t0 = time.perf_counter(); steps_done = 0; t_settled = None; steps_settled = 0
            while time.perf_counter() - t0 < seconds:
                step(); steps_done += 1
                if t_settled is None and time.perf_counter() - t0 >= settle:
                    torch.cuda.synchronize(dev); t_settled = time.perf_counter(); steps_settled = steps_done
            drain(); torch.cuda.synchronize(dev)
            t1 = time.perf_counter(); stop.set(); th.join(timeout=1.0)
            late = [(w, f) for (t, w, f) in samples if t >= settle]
            if not late or t_settled is None or steps_done == steps_settled:
                return None
            watts = float(np.median([w for w, _ in late])); mhz = float(np.median([f for _, f in late]))
            rate = B * (steps_done - steps_settled) / (t1 - t_settled)
            return {"board_power_w_median": round(watts, 1), "board_power_cap_w": cap, "sclk_mhz_median": round(mhz, 1), "pairs_per_s_during_probe": round(rate, 1),
                    "pairs_per_joule": round(rate / watts, 3), "samples": len(late),
                    "what": f"hwmon power1_input / freq1_input of this GPU sampled every 50 ms while the benchmark's own step loop ran for {seconds} s (first {settle} s discarded), after the timed region"}
        except Exception as exc:
            return {"error": repr(exc)[:200]}

    power = power_probe() if (rank == 0 and world == 1 and not args.no_calibration) else None
    kernel_clock = tail_clock_mhz() if (rank == 0 and not args.no_calibration and args.precision == "f16x3" and not adaptive) else None   # (every tile live: no stale stamps)
    rccl = None
    sync_value = None
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # what the collective looked like from inside the job (SCALE_r*.json then shows that RCCL saw N ranks): the result gather of
        # one step timed ALONE with events on this rank's stream (in the timed loop it runs on a side stream under the next forward)
        width = 3 * n + 3 * m + 2   # LG_WIRE_WIDTH: matches0 | scores0 | matches1 | scores1 | stop | status | prune0 | prune1
        send = torch.zeros((B, width), dtype=torch.int32, device=dev)
        recv = torch.empty((B * world, width), dtype=torch.int32, device=dev)
        via_host = dist.get_backend() == "gloo"
        def one_gather():
            if via_host:
                r = torch.empty((B * world, width), dtype=torch.int32); dist.all_gather_into_tensor(r, send.cpu())
            else:
                dist.all_gather_into_tensor(recv, send)
        for _ in range(3):
            one_gather()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(dev); e0.record()
        for _ in range(20):
            one_gather()
        e1.record(); torch.cuda.synchronize(dev)
        seen = [None] * world
        dist.all_gather_object(seen, {"rank": rank, "device": torch.cuda.current_device(), "name": torch.cuda.get_device_name(dev)})
        rccl = {"world": dist.get_world_size(), "backend": dist.get_backend() + (" (RCCL over xGMI)" if dist.get_backend() == "nccl" else " (debug backend, host copies)"),
                "ranks_seen": seen, "gather": "all_gather_into_tensor of one packed int32 buffer [pairs, 3n + 3m + 2] per step (matches0 | scores0 bits | matches1 | scores1 bits | stop | status | prune0 | prune1): every rank rebuilds the full output dict of forward() from it",
                "gather_bytes_sent_per_rank": int(send.numel() * 4), "gather_bytes_received_per_rank": int(recv.numel() * 4),
                "gather_ms_alone": e0.elapsed_time(e1) / 20, "gather_overlap": "gather, unpack kernel and the copy of the [3][B] host block run on a side stream behind an event, under step i+1's forward",
                "rank0_cpu_affinity": rank_affinity,
                "step_output_note": "the N > 1 step returns the same dict as the N = 1 step (see step_output): every rank rebuilds it for the WHOLE batch from the gathered rows"}
    elif not args.no_pipeline:
        # the reference's forward() is synchronous; the headline loop defers each step's host sync by one step.  The same K
        # steps once more with a synchronous forward per step, reported beside it
        torch.cuda.synchronize(dev)
        ts = time.perf_counter()
        for _ in range(args.steps):
            out_sync = model(data)
        torch.cuda.synchronize(dev)
        sync_value = B * args.steps / (time.perf_counter() - ts)
        del out_sync
    gather_probe = None
    if world == 1 and not args.no_gather_probe:
        # what the pair-sharded path's result gather costs a step when it runs beside the next forward: the same K steps through
        # PairShardedMatcher in a world of ONE over RCCL — engine-packed wire rows, all_gather_into_tensor on the side stream behind an event, unpack
        # kernel — against the plain pipelined loop above.  At N > 1 the collective is the same call with N - 1 more peers on the xGMI ring.
        try:
            import socket
            import torch.distributed as dist
            with socket.socket() as sock:
                sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port)); os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
            one = PairShardedMatcher(model, always_gather=True)
            pend = None
            for _ in range(3):
                prev, pend = pend, one.issue_local(data, B)
                if prev is not None: prev.wait()
            pend.wait(); torch.cuda.synchronize(dev)
            tg = time.perf_counter(); pend = None
            for _ in range(args.steps):
                prev, pend = pend, one.issue_local(data, B)
                if prev is not None: prev.wait()
            g_out = pend.wait(); torch.cuda.synchronize(dev)
            g_ms = (time.perf_counter() - tg) / args.steps * 1e3
            same = bool(torch.equal(g_out["matches0"], out["matches0"])) if "matches0" in out else None
            gather_probe = {"ms_per_step_with_world1_gather": g_ms, "ms_per_step_plain": dt / args.steps * 1e3, "delta_ms": g_ms - dt / args.steps * 1e3,
                            "gather_bytes": int(B * (3 * n + 3 * m + 2) * 4), "matches_equal_plain_loop": same,
                            "what": "K steps through PairShardedMatcher(always_gather) in a world of one over RCCL: engine-packed wire rows, all_gather_into_tensor on a side stream under the next forward, lg_unpack_wire"}
            dist.destroy_process_group()
        except Exception as exc:   # the probe must never cost the bench line
            gather_probe = {"error": repr(exc)[:300]}

    pcie_probe = None
    if world == 1 and not args.no_gather_probe and lanes is None and not args.no_pipeline:
        # PCIe-inclusive rate (never `value`: the reference's contract, and this boundary's, is tensors already in HBM): the same K pipelined steps with every batch
        # starting in PINNED HOST memory, (a) copied on the compute stream in front of its forward (what `batch_to_device` + forward does, ref utils.py:63-69),
        # (b) through lightglue_amd.prefetch_to_device: two batches ahead on a copy stream, i.e. under the previous forward
        try:
            from lightglue_amd import prefetch_to_device
            host = {k: {kk: vv.cpu().pin_memory() for kk, vv in v.items()} for k, v in data.items()}
            nbytes = sum(vv.numel() * vv.element_size() for v in host.values() for vv in v.values())

            def run(feed):
                pend = None
                for batch in feed:
                    prev, pend = pend, model.forward_deferred(batch)
                    if prev is not None: prev.result()
                res_ = pend.result(); torch.cuda.synchronize(dev)
                return res_
            inline = lambda k: ({kk: {k2: v2.to(dev, non_blocking=True) for k2, v2 in vv.items()} for kk, vv in host.items()} for _ in range(k))
            run(inline(3)); run(prefetch_to_device((host for _ in range(3)), dev, 2))
            t_i = time.perf_counter(); run(inline(args.steps)); t_i = (time.perf_counter() - t_i) / args.steps * 1e3
            t_p = time.perf_counter(); p_out = run(prefetch_to_device((host for _ in range(args.steps)), dev, 2)); t_p = (time.perf_counter() - t_p) / args.steps * 1e3
            pcie_probe = {"host_bytes_per_step": int(nbytes), "ms_per_step_resident": dt / args.steps * 1e3, "ms_per_step_copy_on_compute_stream": t_i, "ms_per_step_prefetched_on_copy_stream": t_p,
                          "pairs_per_s_copy_on_compute_stream": B / t_i * 1e3, "pairs_per_s_prefetched": B / t_p * 1e3,
                          "matches_equal_resident_loop": bool(torch.equal(p_out["matches0"], out["matches0"])) if "matches0" in out else None,
                          "what": "K pipelined steps with the inputs starting in pinned host memory: copied on the compute stream in front of each forward vs lightglue_amd.prefetch_to_device (copy stream, two batches ahead)"}
        except Exception as exc:   # never costs the bench line
            pcie_probe = {"error": repr(exc)[:300]}

    if rank == 0:
        total_pairs = B * world * args.steps
        value = total_pairs / dt
        fl = flops_per_launch(B, n, m)
        timed = {k: v for k, v in prof.items() if v[1] > 0}
        classes_seen = set(timed) | set(warm_prof)
        fused_next = "fused_tail" in classes_seen and "gemm_qkv_cross" not in classes_seen
        if fused_next:
            # the tail kernel also runs the NEXT block's q/k/v projection (L cross + L-1 self projections over 2L tail
            # launches per forward): charge their algorithmic FLOPs to the launches that execute them
            fl["fused_tail"] += (L * fl["gemm_qkv_cross"] + (L - 1) * fl["gemm_qkv_self"]) / (2 * L)
            if "gemm_final_proj" not in classes_seen:   # fixed depth: the LAST tail launch also runs the final projection of the log assignment
                fl["fused_tail"] += fl["gemm_final_proj"] / (2 * L)
        dom = max((k for k in timed if k in fl), key=lambda k: timed[k][0])
        dom_ms = timed[dom][0] / timed[dom][1]
        achieved = fl[dom] / (dom_ms * 1e-3) / 1e12
        peak = 157.3 if args.precision == "fp32" else 2500.0  # dense MFMA peak, MI355X_MICROARCH.md
        tkey = f"{args.precision}/{args.attention or args.precision}/B{B}/N{n}/"
        traffic_dom = pmc_traffic(tkey + dom + ("+next" if fused_next and dom == "fused_tail" else ""))
        traffic_assign = pmc_traffic(tkey + "assign_sweeps")
        if warm_prof:   # per-class table from the warm-up steps; the dominant class from the timed region
            kernel_ms = {k: round(v[0], 4) for k, v in warm_prof.items()}
            kernel_ms[dom] = round(timed[dom][0] / args.steps, 4)
            timed = {**{k: (v[0] * args.steps, v[1] * args.steps) for k, v in warm_prof.items()}, dom: timed[dom]}
        else:
            kernel_ms = {k: round(v[0] / args.steps, 4) for k, v in timed.items()}
        # MFMAs issued per algorithmic MAC: linear layers 3 in f16x3 (the projection's x operand is one plane in the fast opt-in: its
        # share of the tail's MACs runs 2), attention 3 with split operands, 1 otherwise
        mfma_per_mac_attn = 3 if (args.precision == "f16x3" and not args.attention) else 1
        mfma_per_mac_linear = (3 if not args.attention else round((3 * 393216 + 2 * 163840) / 557056, 3)) if args.precision == "f16x3" else 1
        res = {
            "metric": ("image-pairs/s at N=M=1024, 9 layers; match-index parity vs ref" if args.config == 2 else
                       f"image-pairs/s at N={n} M={m}, 9 layers ({'adaptive depth/width' if adaptive else 'pruning off'}, {dim}-d descriptors); match-index parity vs ref"),
            "value": value,
            "unit": "image-pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            # the arithmetic type of the matrix-core operands: split f16 (hi + lo planes, 22 bits, three v_mfma_f32_16x16x32_f16 per
            # product) is WIDER than the bf16 BASELINE cfg #2 names, at the same storage width and MFMA rate per instruction
            "dtype": {"f16x3": "f16", "bf16": "bf16", "fp16": "f16", "fp32": "f32"}[args.precision],
            "data": "synthetic",
            "config": {"baseline_config": cfg["label"] if (B, n, m) == (cfg["pairs"], cfg["n"], cfg["m"]) else f"{cfg['label']} — with --pairs / --kpts overrides: batch={B}, N={n}, M={m}",
                       "workload": (f"SuperPoint-dim 256-d descriptors, N=M={n}, 9 layers, pruning/early-stop OFF, batch={B} pairs per GPU, " if args.config == 2 else
                                    f"{dim}-d descriptors, N={n} M={m}, 9 layers, {'depth_confidence=0.95 width_confidence=0.99 (early stop + point pruning ON, every pair on its own)' if adaptive else 'pruning/early-stop OFF'}, batch={B} pairs per GPU, ")
                                   + (f"weights from {Path(args.checkpoint).name}" if args.checkpoint else f"seeded random weights (recipe {args.recipe}{', trained-model statistics' if args.recipe == 'D' else ''})") + f", precision={args.precision}"
                                   + (" (split-f16 operands, 3 MFMAs per product, for every contraction incl. q k^T and P V; fp32 accumulate / residual / softmax)" if args.precision == "f16x3" and not args.attention else "")
                                   + (", attention_precision=fp16 (single-plane f16 attention: the fast opt-in, outside the 1e-3 bar for sharp attention)" if args.attention else ""),
                       "pairs_per_gpu": B, "keypoints": n, "keypoints1": m, "descriptor_dim": dim, "parallelism": f"pair-sharded dp{world}", "batches_in_flight": (args.inflight if lanes is not None else 1),
                       "host_pipelining": ("result gather of step i overlaps the forward of step i+1" if world > 1 else
                                           "synchronous forward per step" if args.no_pipeline else
                                           f"{args.inflight} whole batches in flight (lightglue_amd.InflightMatcher: step i on lane i % {args.inflight}, one engine + one HIP stream per lane); a step's full output dict is built when its lane is used next, all K inside the timed region; "
                                           "event-timed launches (roofline legs) overlap the other lanes' kernels and read longer than alone" if lanes is not None else
                                           "output assembly (the forward's one host sync) of step i overlaps the forward of step i+1; all K outputs are built inside the timed region")},
            "roofline": {"bound": "mfma", "kernel": dom + (" (+ next block's q/k/v projection)" if fused_next and dom == "fused_tail" else ""), "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                         "traffic": traffic_dom[0], "traffic_source": traffic_dom[1],
                         "avg_launch_ms": dom_ms, "algorithmic_flops_per_launch": fl[dom],
                         "sustained_peak": sustained_tflops, "frac_of_sustained": (achieved / sustained_tflops if sustained_tflops else None),
                         # matrix-core work actually ISSUED per second: the split-f16 products cost 3 MFMAs per algorithmic MAC
                         "mfma_per_mac": mfma_per_mac_linear, "issued_tflops": achieved * mfma_per_mac_linear,
                         "note": ("ADAPTIVE config: pairs stop early and rows are pruned during the forward, so the FLOPs of the full N x M shape charged to every launch are an UPPER bound (achieved / frac overstate).  " if adaptive else "")
                                 + "algorithmic FLOPs (2/MAC) of one launch / HIP-event launch duration; split f16 issues 3 MFMAs per algorithmic MAC.  "
                                 "peak = nominal dense bf16 / f16 (2.4 GHz); sustained_peak = what a dense bf16 MFMA spin "
                                 "reaches on THIS box right before the timed region (power-managed clock, see effective_mfma_clock_mhz)"},
            # the HBM-bound stage of the path: dual log-softmax + argmax sweeps over the similarity matrix
            "roofline_hbm": ({"bound": "hbm", "kernel": "assign (lse_sweep + argmax_sweep + merges + finalize)",
                              "achieved": hbm_bytes_assign(B, n, m) / (timed["assign"][0] / timed["assign"][1] * 1e-3) / 1e9,
                              "peak": 8000.0, "unit": "GB/s",
                              "frac": hbm_bytes_assign(B, n, m) / (timed["assign"][0] / timed["assign"][1] * 1e-3) / 1e9 / 8000.0,
                              "algorithmic_bytes_per_launch": hbm_bytes_assign(B, n, m),
                              "traffic": traffic_assign[0], "traffic_source": traffic_assign[1]} if "assign" in timed else None),
            # the attention GEMMs against the same peak (north_star asks for it): algorithmic FLOPs of one self-attention launch (QK^T + PV
            # over both images) / its average launch time from the per-class events; the split attention issues 3 MFMAs per MAC
            "roofline_attention": ({"bound": "mfma", "kernel": "attn_self", "achieved": fl["attn_self"] / (kernel_ms["attn_self"] / L * 1e-3) / 1e12, "peak": peak, "unit": "TFLOP/s",
                                    "frac": fl["attn_self"] / (kernel_ms["attn_self"] / L * 1e-3) / 1e12 / peak, "avg_launch_ms": kernel_ms["attn_self"] / L,
                                    "algorithmic_flops_per_launch": fl["attn_self"], "mfma_per_mac": mfma_per_mac_attn, "issued_tflops": fl["attn_self"] / (kernel_ms["attn_self"] / L * 1e-3) / 1e12 * mfma_per_mac_attn,
                                    "traffic": pmc_traffic(tkey + "attention")[0]} if "attn_self" in kernel_ms else None),
            "kernel_ms_per_step": kernel_ms,
            "kernel_ms_per_step_source": ("HIP events around every launch during the warm-up steps; the roofline kernel's entry and "
                                          "avg_launch_ms come from events inside the timed region") if warm_prof else "HIP events inside the timed region",
            #  class attribution under fusion: what each event class brackets in the product configuration
            "kernel_ms_class_notes": {"prep": "input_dim 256 (fused_prep): only the bounding-box / state kernels — normalisation, rotary rows, index set and the descriptor copy run inside the FIRST gemm_qkv_self launch (proj_first_kernel), whose time and bytes (1 KB read + 1.3 KB write per keypoint) are in gemm_qkv_self",
                                      "fused_tail": "tail + the next block's q/k/v projection (+ the final projection in the last launch at fixed depth)",
                                      "gemm_qkv_self": "the first projection only (later ones run inside fused_tail) unless pruning forces standalone projections"},
            "gpu_ms_per_step_sum": round(sum(kernel_ms.values()), 3),
            "gpu_ms_per_step_sum_note": "sum of the per-class event times of the WARM-UP steps (each launch bracketed by its own event pair, ~90 pairs per step): an upper bound that is not additive with ms_per_step, which is the un-instrumented timed region",
            "algorithmic_tflops": value * flops_per_pair(n, m) / 1e12,
            # shader clock this box sustains under a matrix-core-dense load, measured right before the timed region: the pool's
            # boxes differ by up to ~20 % for one binary, and most of it is this clock
            "effective_mfma_clock_mhz": sustained_mhz, "sustained_dense_bf16_tflops": sustained_tflops,
            # the board's power limit is what bounds the two big kernels: the clock the fused tail actually ran at inside a forward, of 2 400 MHz nominal
            "shader_clock_mhz_inside_tail_kernel": kernel_clock,
            "power": power,
            "matches_per_pair": float((out["matches0"].cpu().numpy() > -1).sum(axis=1).mean()) if "matches0" in out else None,   # (one copy: no framework kernels behind the timed region either)
            "value_synchronous_forward": sync_value,   # pairs/s with model(data) per step (host sync inside every forward), N = 1 only
            "rccl": rccl,
            "gather_probe_one_gpu": gather_probe,
            "pcie_inclusive": pcie_probe,
            # what one step hands back: the same dict at every N — LightGlue.forward's keys / dtypes (ref :619-629), built inside the timed
            # region with ONE deferred host synchronisation per step (N = 1: forward_deferred; N > 1: PairShardedMatcher.issue_local -> Pending.wait)
            "step_output": {k: (str(v.dtype).replace("torch.", "") if torch.is_tensor(v) else f"list[{len(v)}] of {str(v[0].dtype).replace('torch.', '')}" if isinstance(v, list) and v else type(v).__name__)
                            for k, v in sorted((out or {}).items())},
        }
        # match-index parity of the batch that was just timed (rank 0's pairs): against the reference's own fixture for the
        # first 4 pairs and against the oracle on the pairs the CPU leg runs anyway
        default_weights = args.precision in ("f16x3", "fp32") and not args.checkpoint
        res["parity"] = golden_parity(out, n, B, args.recipe, args.config, m) if (default_weights and os.environ.get("LG_BENCH_ABLATION") != "1") else None
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"], res["parity_oracle"] = cpu_baseline(sd, n, m, gpu_out=out, recipe=args.recipe, conf_kw={k: v for k, v in conf_kw.items() if k != "input_dim"}, dim=dim, wseed=cfg["wseed"], checkpoint=args.checkpoint)
        json_out.write(json.dumps(res) + "\n")
        json_out.flush()
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
