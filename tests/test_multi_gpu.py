"""Multi-GPU readiness (north star: "batches shard embarrassingly across the 8 GPUs of one node with RCCL only
to reduce the final throughput counter").  Two kinds of test:
  * on ONE GPU: the partition of a 65 536-graph BASELINE config 5 batch into the eight node-balanced shards an 8-GPU run
    would use, each shard run in turn through the HIP path and checked against the oracle -- what every rank of the first
    8-GPU run will see, minus the second device;
  * on >= 2 GPUs (skipped otherwise -- they run by themselves the moment two devices are visible): one process per GPU over
    RCCL, (a) the parity check per shard with the reductions bench.py uses, (b) `bench.py --gpus 2 --shard one-batch` on
    the HIP path itself."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

from gnnbuilder_amd import runtime, synthetic
from gnnbuilder_amd.batching import pack_graphs, shard_bounds
from helpers import canon, make_model, to_dev
from oracle import oracle as O

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    runtime.load_library(require_gpu=True)
    return torch.device("cuda:0")


def test_config5_batch_cut_into_its_eight_shards(dev):
    """BASELINE config 5: ONE batch of 65 536 molhiv-shaped graphs (1.67 M nodes), cut with shard_bounds(., 8) as
    `bench.py --gpus 8 --shard one-batch` cuts it: the shards tile the batch, every shard is within one graph of N / 8
    nodes, and each shard -- run in turn on this GPU with the config 5 model (2-layer GraphSAGE d = 256) -- matches the
    oracle on its first, last and 30 sampled graphs; the same graphs give the same rows whichever shard they sit in."""
    world = 8
    glob = synthetic.make_batch("molhiv", 65536, seed=2024)
    bounds = shard_bounds(glob.node_ptr, world)
    assert bounds[0][0] == 0 and bounds[-1][1] == glob.num_graphs and all(bounds[r][1] == bounds[r + 1][0] for r in range(world - 1))
    sizes = np.diff(glob.node_ptr)
    nodes = [int(glob.node_ptr[g1] - glob.node_ptr[g0]) for g0, g1 in bounds]
    assert sum(nodes) == glob.num_nodes
    assert max(abs(n - glob.num_nodes / world) for n in nodes) <= sizes.max()  # node-balanced up to one graph
    model = make_model("sage", in_dim=9, hidden=256, layers=2, task_out=1)
    spec, params = model.spec(), canon(model)
    cap = (max(g1 - g0 for g0, g1 in bounds), max(nodes), max(int(glob.edge_ptr[g1] - glob.edge_ptr[g0]) for g0, g1 in bounds))
    cm = runtime.CompiledModel.from_model(model, *cap)  # one workspace sized for the largest shard, as a rank allocates it
    rng = np.random.default_rng(8)
    for r, (g0, g1) in enumerate(bounds):
        shard = glob.slice(g0, g1)
        out = cm.forward(*to_dev(shard, dev)).cpu().numpy()
        cm.check()
        idx = np.unique(np.concatenate([[0, shard.num_graphs - 1, int(np.argmax(np.diff(shard.node_ptr)))],
                                        rng.choice(shard.num_graphs, 30, replace=False)]))
        sub = pack_graphs([shard.graph(int(g)) for g in idx])
        ref = O.forward_batched(spec, params, sub.x, sub.coo, sub.node_ptr, sub.edge_ptr)
        scale = max(1.0, float(np.abs(ref).max()))
        assert np.abs(out[idx] - ref).max() < TOL * scale, f"shard {r}"


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


needs_two = pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (runs by itself where they are visible)")


@needs_two
@pytest.mark.parametrize("conv,shape,graphs", [("gcn", "qm9", 2048), ("sage", "molhiv", 1024), ("gin", "molhiv", 1024), ("pna", "qm9", 512)])
def test_two_ranks_over_rccl_match_the_oracle(conv, shape, graphs):
    """One process per GPU (children of this test: the parent makes no GPU call of its own in them), each on its own device
    with its node-balanced shard of one global batch; every graph of every shard against the oracle, counters over RCCL."""
    world, port = 2, _free_port()
    procs = []
    for rank in range(world):
        env = _clean_env()
        env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(ROOT / "tests" / "multi_gpu_rank.py"), conv, shape, str(graphs)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-2000:] for o in outs]
    r = json.loads([l for l in outs[0][0].splitlines() if l.strip().startswith("{")][-1])
    assert r["rccl_ranks"] == 2 and r["graphs"] == r["global_graphs"] == graphs and r["nodes"] == r["global_nodes"]
    assert r["max_err"] < TOL
    assert r["max_nodes_per_rank"] <= r["global_nodes"] / 2 + r["largest_graph"]  # node-balanced up to one graph (the batch's largest)
    if conv in ("gcn", "gin"):
        assert r["path"].startswith("stack")


@needs_two
def test_bench_two_gpus_one_batch_shard_on_the_hip_path():
    """`python bench.py --gpus 2 --shard one-batch`: the command the driver's scaling run issues (its own ranks, nccl =
    RCCL, the HIP path -- no --dry-launch), on the plumbing-sized workload; one JSON line, both ranks counted."""
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--workload", "tiny", "--shard", "one-batch",
                        "--steps", "5", "--warmup", "2", "--repeats", "3", "--batches", "2", "--no-cpu-baseline"],
                       capture_output=True, text=True, env=_clean_env(), timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, p.stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["config"]["rccl_ranks"] == 2 and r["config"]["shard"] == "one-batch"
    assert "dry_launch" not in r and r["config"]["path"] in ("stack_zf", "stack", "layerwise")
    assert abs(r["value"] * r["ms_per_step"] * 1e-3 * r["steps"] - 5 * 128) < 1e-6 * 5 * 128
    assert "roofline" in r  # rank 0 finishes the single-GPU legs alone
