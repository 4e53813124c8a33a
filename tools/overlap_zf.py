#!/usr/bin/env python3
"""Development only (-DGNNB_ZF_ABLATE): does k_gcn2_zf's row walk (P1, VALU + LDS) overlap with an MFMA stream issued by the
other waves of the same SIMDs?  Waves 8 .. 15 run a whole M1's worth of MFMAs inside the P1 interval (results dropped)."""
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
cm.graph_prep(bd[1], bd[2], bd[3], int(bd[0].shape[0]))
D = 1 << 21
for dbg, what in ((0, "everything"), (2, "no P0'"), (2 | D, "no P0', + MFMA stream beside P1"), (2 | 1 | D, "no P0', no P1, MFMA stream alone"),
                  (2 | 1, "no P0', no P1"), (2 | 4, "no P0', no M1"), (2 | 4 | D, "no P0', no M1, + MFMA stream beside P1"), (0, "everything")):
    os.environ["GNNB_ZF_DBG"] = str(dbg)
    t = min(cm.gcn_stack_timed(bd[0], 100) for _ in range(3))
    print(f"{what:44s} {t:6.2f} us", flush=True)
os.environ["GNNB_ZF_DBG"] = "0"
