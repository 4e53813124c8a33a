#!/usr/bin/env python3
"""Development only (library built with -DGNNB_ZF_ABLATE): launch-loop time of k_gcn2_zf with phases switched off
(results are WRONG by construction; what is measured is what each phase costs the kernel)."""
import os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np, torch
import bench
from gnnbuilder_amd import runtime, synthetic
w = bench.WORKLOADS["c2"]; dev = torch.device("cuda:0")
model = bench.build_model(w)
b = synthetic.make_batch(w["shape"], w["batch"], seed=0)
cm = runtime.CompiledModel.from_model(model, b.num_graphs, b.num_nodes, b.num_edges, max_graph_nodes=int(np.diff(b.node_ptr).max()))
bd = tuple(torch.from_numpy(a).to(dev) for a in (b.x, b.coo, b.node_ptr, b.edge_ptr))
for shape in (1, 0):
    runtime.set_option("zf_shape", shape)
    cm.graph_prep(bd[1], bd[2], bd[3], int(bd[0].shape[0]))
    for dbg, what in ((0, "everything"), (1, "no P1"), (3, "no P1, no P0'"), (4, "no M1"), (8, "no M0"), (16, "no Z write"),
                      (4 + 8, "no MFMA at all"), (1 + 2 + 16, "MFMA phases only"), (31, "skeleton: DMA, plan, barriers"), (32, "return at entry"), (64 + 128, "... after the first DMA, no W1 load"), (128, "everything but the W1 load"), (64, "return after the first DMA landed")):
        os.environ["GNNB_ZF_DBG"] = str(dbg)
        t = min(cm.gcn_stack_timed(bd[0], 100) for _ in range(3))
        print(f"shape {shape} dbg {dbg:2d} {what:32s} {t:6.2f} us")
os.environ["GNNB_ZF_DBG"] = "0"
