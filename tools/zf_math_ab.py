#!/usr/bin/env python3
"""k_gcn2_zf alone (prepared topology, launches back to back on one stream, HIP events): math 0 (fp32 MFMA) vs math 2
(bf16x3 M1).  usage: zf_math_ab.py [workload]"""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
from gnnbuilder_amd import runtime, synthetic  # noqa: E402

w = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c2"]
model = bench.build_model(w)
batch = synthetic.make_batch(w["shape"], w["batch"], seed=3)
dev = torch.device("cuda:0")
cm = runtime.CompiledModel.from_model(model, batch.num_graphs, batch.num_nodes, batch.num_edges,
                                      max_graph_nodes=int(np.diff(batch.node_ptr).max()))
bd = tuple(torch.from_numpy(a).to(dev) for a in (batch.x, batch.coo, batch.node_ptr, batch.edge_ptr))
out = torch.empty((batch.num_graphs, model.spec().out_dim if hasattr(model.spec(), "out_dim") else 19), device=dev)
ref = None
for math in (0, 2, 3, 0, 2, 3):
    runtime.set_option("math", math)
    o = cm.forward(*bd)
    torch.cuda.synchronize()
    cm.graph_prep(bd[1], bd[2], bd[3], int(bd[0].shape[0]))
    for _ in range(20):
        cm.forward_prepared(bd[0], out=o)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        cm.forward_prepared(bd[0], out=o)
    e1.record()
    torch.cuda.synchronize()
    if ref is None:
        ref = o.clone()
    print("math %d: %.2f us per prepared forward (stack + head), path %s, max |diff to math 0| %.2e" % (
        math, e0.elapsed_time(e1) * 1e3 / 200, cm.last_path(), float((o - ref).abs().max())))
runtime.set_option("math", 0)
