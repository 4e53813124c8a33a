#!/usr/bin/env python3
"""Development only (library built with -DGNNB_ZF_ABLATE): k_gcn2_zf with half of the workgroups started late, to see whether
the two workgroups of a CU run their MFMA phases against each other's narrow phases when they are out of phase."""
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
runtime.set_option("zf_shape", 0)
cm.graph_prep(bd[1], bd[2], bd[3], int(bd[0].shape[0]))
for sel in (0, 1, 2, 3):
    for d in (0, 8, 16, 24, 32, 40):
        os.environ["GNNB_ZF_DBG"] = str((d << 8) | (sel << 16))
        t = min(cm.gcn_stack_timed(bd[0], 100) for _ in range(3))
        print(f"half {sel} delay {d / 4:5.2f} us: {t:6.2f} us", flush=True)
os.environ["GNNB_ZF_DBG"] = "0"
