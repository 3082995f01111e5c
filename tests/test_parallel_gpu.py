"""PairShardedMatcher around the HIP LightGlue on ONE GPU (SURVEY.md §8e without an 8-GPU node): the RCCL path with a
world of 1, and two ranks sharing cuda:0 over gloo — results must equal the plain model() bitwise, including a ragged batch
dealt by the work-balanced (non-contiguous) assignment."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import gpu_util
from conftest import require_gpu
from lightglue_amd import PairShardedMatcher
from lightglue_amd import synthetic as synth

pytestmark = pytest.mark.gpu

KEYS = ("matches0", "matches1", "matching_scores0", "matching_scores1")


def _model():
    return gpu_util.make_model(synth.make_state_dict(0, recipe="A"), "f16x3", depth_confidence=-1, width_confidence=-1)


def _batch(ragged):
    t = gpu_util.to_torch(synth.make_batch(77, 5, 320, 288))
    if ragged:
        t["image0"]["num_keypoints"] = torch.tensor([320, 40, 300, 64, 129], dtype=torch.int32, device="cuda")
        t["image1"]["num_keypoints"] = torch.tensor([288, 35, 200, 70, 288], dtype=torch.int32, device="cuda")
    return t


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _check(res, plain):
    """the sharded step returns the SAME product as LightGlue.forward: key set, dtypes, values, list semantics"""
    assert set(res) == set(plain), set(res) ^ set(plain)
    for k in KEYS + ("prune0", "prune1"):
        assert res[k].dtype == plain[k].dtype and torch.equal(res[k], plain[k]), k
    if torch.is_tensor(plain["stop"]):
        assert res["stop"].dtype == plain["stop"].dtype and torch.equal(res["stop"], plain["stop"])
    else:
        assert isinstance(res["stop"], int) and res["stop"] == plain["stop"]
    assert len(res["matches"]) == len(plain["matches"])
    for a, b, c, d in zip(res["matches"], plain["matches"], res["scores"], plain["scores"]):
        assert a.dtype == b.dtype and torch.equal(a, b) and c.dtype == d.dtype and torch.equal(c, d)


@pytest.mark.parametrize("ragged", [False, True])
def test_world_of_one_over_rccl_equals_plain_forward(ragged):
    require_gpu()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        model = _model()
        t = _batch(ragged)
        plain = model(t)
        sharded = PairShardedMatcher(model)
        _check(sharded(t), plain)
        pend = sharded.issue_local(t, 5)            # asynchronous form: gather in flight while the next forward runs
        again = model(t)
        _check(pend.wait(), plain)
        assert torch.equal(again["matches0"], plain["matches0"])
        matches, scores = PairShardedMatcher.ragged(sharded(t))
        for b in range(5):
            assert torch.equal(matches[b], plain["matches"][b]) and torch.equal(scores[b], plain["scores"][b])
        # adaptive depth / width: the prune counters and per-pair stop layers travel on the wire as well
        amodel = gpu_util.make_model(synth.make_state_dict(0, recipe="C"), "f16x3", pruning_min_kpts=64)
        aplain = amodel(t)
        assert aplain["prune0"].dtype == torch.int64
        _check(PairShardedMatcher(amodel)(t), aplain)
        # B = 1: `stop` is a Python int, as in forward (ref :604)
        one = {k: {kk: vv[:1] for kk, vv in v.items()} for k, v in t.items()}
        _check(sharded(one), model(one))
    finally:
        dist.destroy_process_group()


def test_status_of_a_poisoned_pair_raises_through_the_sharded_path():
    """the range guard's per-pair status rides on the wire row; the sharded step must raise like forward does."""
    require_gpu()
    from lightglue_amd import _cabi
    model = _model()
    model.check_finite = True
    t = _batch(False)
    t["image0"]["descriptors"][3] *= 1e6
    with pytest.raises(_cabi.LightGlueAmdError, match="pair 3"):
        PairShardedMatcher(model)(t)


def _worker(rank, world, port, q, ragged, poison=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model = _model()
        sharded = PairShardedMatcher(model)
        t = _batch(ragged)
        if poison:
            model.check_finite = True
            t["image0"]["descriptors"][4] *= 1e6            # the last pair: the LAST rank's shard
        if ragged and world == 2:
            assert sharded.assignment(t) != [[0, 1, 2], [3, 4]], "the balanced assignment must be non-contiguous for this test"
        try:
            res = sharded(t)
        except Exception as ex:
            q.put((rank, {"error": f"{type(ex).__name__}: {ex}"}))
            return
        conv = lambda v: [x.cpu().numpy() for x in v] if isinstance(v, list) else (v.cpu().numpy() if torch.is_tensor(v) else v)
        q.put((rank, {k: conv(v) for k, v in res.items()}))
    finally:
        dist.destroy_process_group()


def _run(world, ragged, poison=False):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, ragged, poison)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=900) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return got


def _check_np(got, plain):
    assert set(got) == set(plain), set(got) ^ set(plain)
    for k in KEYS + ("prune0", "prune1", "stop"):
        np.testing.assert_array_equal(got[k], plain[k].cpu().numpy(), err_msg=k)
        assert got[k].dtype == plain[k].cpu().numpy().dtype, k
    for k in ("matches", "scores"):
        assert len(got[k]) == len(plain[k])
        for a, b in zip(got[k], plain[k]):
            np.testing.assert_array_equal(a, b.cpu().numpy(), err_msg=k)


@pytest.mark.parametrize("ragged", [False, True])
def test_two_ranks_sharing_one_gpu_equal_plain_forward(ragged):
    require_gpu()
    plain = _model()(_batch(ragged))
    got = _run(2, ragged)
    for rank in (0, 1):
        _check_np(got[rank], plain)


def test_eight_ranks_sharing_one_gpu_equal_plain_forward():
    """Rehearsal of the 8-rank job on one GPU over gloo (5 pairs on 8 ranks: three ranks hold an EMPTY shard and still take part in the gather)."""
    require_gpu()
    plain = _model()(_batch(False))
    got = _run(8, False)
    for rank in range(8):
        _check_np(got[rank], plain)


def test_a_poisoned_pair_on_the_last_rank_raises_on_rank_0():
    require_gpu()
    got = _run(2, False, poison=True)
    for rank in (0, 1):
        assert "error" in got[rank] and "pair 4" in got[rank]["error"] and "LG_ERR_RANGE" in got[rank]["error"], got[rank]


def test_plain_bench_command_launches_its_own_ranks():
    """`python bench.py --gpus 2 ...` WITHOUT a launcher must spawn its two ranks itself (the driver's scaling run may
    use the plain form).  Rehearsal on one GPU: both ranks share cuda:0 and talk over gloo; the JSON line must be the only stdout line."""
    require_gpu()
    import json
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(LG_BENCH_ONE_GPU="1", LG_BENCH_BACKEND="gloo")
    p = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl"]["world"] == 2 and len(d["rccl"]["ranks_seen"]) == 2
    assert d["parity"]["index_mismatches"] == 0 and d["parity"]["unexplained"] == 0
    # the N > 1 step hands back forward()'s own dict for the WHOLE batch, and every rank is bound to its GPU's NUMA node (or says why not)
    assert d["step_output"] == {"matches": "list[64] of int64", "matches0": "int64", "matches1": "int64", "matching_scores0": "float32", "matching_scores1": "float32",
                                "prune0": "float32", "prune1": "float32", "scores": "list[64] of float32", "stop": "int64"}, d["step_output"]
    assert isinstance(d["rccl"]["rank0_cpu_affinity"], str) and d["rccl"]["rank0_cpu_affinity"]


def test_bench_config_4_two_ranks_on_one_gpu():
    """an N-GPU run must be able to measure BASELINE cfg #4 (DISK 128-d, N = M = 4096, pair-sharded, RCCL gather).  Rehearsal of
    the same entry point on one GPU at reduced N: two ranks share cuda:0 over gloo; the line names the config, carries the `rccl` block and a
    roofline for the shape's dominant kernel."""
    require_gpu()
    import json
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(LG_BENCH_ONE_GPU="1", LG_BENCH_BACKEND="gloo")
    p = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--config", "4", "--kpts", "512", "--pairs", "4", "--steps", "2", "--warmup", "2",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl"]["world"] == 2 and d["config"]["descriptor_dim"] == 128 and "DISK 128-d" in d["config"]["baseline_config"]
    assert d["roofline"]["kernel"] and d["roofline"]["frac"] > 0 and d["value"] > 0


def test_bench_default_is_config_2_and_reports_the_gather_probe():
    """The default workload stays BASELINE cfg #2 byte for byte (same seeds: the first four pairs ARE the reference fixture), and one GPU reports
    what the world-of-one result gather costs a step."""
    require_gpu()
    import json
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, str(root / "bench.py"), "--steps", "3", "--warmup", "2", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.strip()][-1])
    assert d["metric"] == "image-pairs/s at N=M=1024, 9 layers; match-index parity vs ref" and d["config"]["pairs_per_gpu"] == 32 and d["config"]["keypoints"] == 1024
    assert d["parity"]["pairs"] == 4 and d["parity"]["index_mismatches"] == 0
    assert d["step_output"] == {"matches": "list[32] of int64", "matches0": "int64", "matches1": "int64", "matching_scores0": "float32", "matching_scores1": "float32",
                                "prune0": "float32", "prune1": "float32", "scores": "list[32] of float32", "stop": "int64"}, d["step_output"]   # same keys / dtypes as at N > 1
    pw = d["power"]    # the board's power sensor under the same loop (None where the box exposes no hwmon): the cap is what bounds the big kernels
    assert pw is None or ("error" not in pw and 200 < pw["board_power_w_median"] <= pw["board_power_cap_w"] * 1.02 and pw["pairs_per_joule"] > 0), pw
    g = d["gather_probe_one_gpu"]
    assert "error" not in g, g
    assert g["matches_equal_plain_loop"] is True and g["ms_per_step_with_world1_gather"] > 0
