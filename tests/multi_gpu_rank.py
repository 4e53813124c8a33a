#!/usr/bin/env python3
"""One rank of the multi-GPU parity check (test infrastructure: uses the oracle; started by tests/test_multi_gpu.py, one
process per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment as bench.py's ranks get them).
Every rank draws the SAME global batch, keeps its node-balanced contiguous range (batching.shard_bounds -- the partition
of BASELINE config 5's 65 536 graphs over 8 GPUs), runs the HIP path on its own GPU and compares every graph of its shard
with the oracle; RCCL carries what bench.py sends over it: {graphs done, ranks} summed, the worst error maxed."""
import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import numpy as np
import torch
import torch.distributed as dist

from gnnbuilder_amd import runtime, synthetic
from gnnbuilder_amd.batching import shard_bounds
from helpers import canon, make_model, to_dev
from oracle import oracle as O


def main():
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    conv, shape, graphs = sys.argv[1], sys.argv[2], int(sys.argv[3])
    runtime.load_library(require_gpu=True)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    glob = synthetic.make_batch(shape, graphs, seed=99)
    g0, g1 = shard_bounds(glob.node_ptr, world)[rank]
    mine = glob.slice(g0, g1)
    model = make_model(conv, in_dim=synthetic.SHAPES[shape]["f_in"], hidden=64, layers=2, task_out=synthetic.SHAPES[shape]["out"])
    promise = int(np.diff(mine.node_ptr).max()) if conv in ("gcn", "gin") else 0
    cm = runtime.CompiledModel.from_model(model, mine.num_graphs, mine.num_nodes, max(mine.num_edges, 1), max_graph_nodes=promise)
    out = cm.forward(*to_dev(mine, dev)).cpu().numpy()
    cm.check()
    ref = O.forward_batched(model.spec(), canon(model), mine.x, mine.coo, mine.node_ptr, mine.edge_ptr)
    err = float(np.abs(out - ref).max()) if mine.num_graphs else 0.0
    c = torch.tensor([float(mine.num_graphs), 1.0, float(mine.num_nodes)], device=dev, dtype=torch.float64)
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    e = torch.tensor([err, float(mine.num_nodes)], device=dev, dtype=torch.float64)
    dist.all_reduce(e, op=dist.ReduceOp.MAX)
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"graphs": c[0].item(), "rccl_ranks": int(round(c[1].item())), "nodes": c[2].item(),
                          "max_err": e[0].item(), "max_nodes_per_rank": e[1].item(), "path": cm.last_path(),
                          "global_graphs": glob.num_graphs, "global_nodes": glob.num_nodes,
                          "largest_graph": int(np.diff(glob.node_ptr).max())}))


if __name__ == "__main__":
    main()
