"""The N>1 path on CPU: two gloo ranks shard one batch by node count, each processes only its
graphs, and the throughput counters are reduced exactly as bench.py does over RCCL
(all_reduce SUM of graphs done, MAX of elapsed).  The per-graph compute stands in for the GPU
path with the PyTorch model forward -- what is under test is the sharding and the reductions."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gnnbuilder_amd import synthetic
from gnnbuilder_amd.batching import shard_batch, shard_bounds
from helpers import batch_vector, make_model


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, tmpdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    batch = synthetic.make_batch("qm9", 101, seed=4)           # every rank builds the same global batch
    mine = shard_batch(batch, world, rank)
    model = make_model("gcn", hidden=16, layers=2, task_out=3)  # same seed => same replicated weights
    with torch.no_grad():
        out = model(torch.from_numpy(mine.x), torch.from_numpy(mine.coo.T.astype(np.int64)),
                    torch.from_numpy(batch_vector(mine)))
    graphs = torch.tensor([float(mine.num_graphs)], dtype=torch.float64)
    elapsed = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(graphs, op=dist.ReduceOp.SUM)
    dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    gathered = [None] * world
    dist.all_gather_object(gathered, out.numpy())
    if rank == 0:
        np.save(os.path.join(tmpdir, "out.npy"), np.concatenate(gathered))
        np.save(os.path.join(tmpdir, "counters.npy"), np.array([graphs.item(), elapsed.item()]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_counter_reduction(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    batch = synthetic.make_batch("qm9", 101, seed=4)
    model = make_model("gcn", hidden=16, layers=2, task_out=3)
    with torch.no_grad():
        full = model(torch.from_numpy(batch.x), torch.from_numpy(batch.coo.T.astype(np.int64)),
                     torch.from_numpy(batch_vector(batch))).numpy()
    out = np.load(tmp_path / "out.npy")
    graphs, elapsed = np.load(tmp_path / "counters.npy")
    assert graphs == 101 and elapsed == 2.0          # SUM over ranks, MAX over ranks
    assert out.shape == full.shape and np.abs(out - full).max() < 1e-6  # shards are independent
    b = shard_bounds(batch.node_ptr, world)
    assert abs((batch.node_ptr[b[0][1]] - 0) - batch.num_nodes / 2) <= 29
