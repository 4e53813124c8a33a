#!/usr/bin/env python3
"""Development only (-DGNNB_ZF_ABLATE): the MFMA phases of k_gcn2_zf with ONE 8-wave workgroup per CU (GNNB_ZF_ONE=1) against
two -- does the matrix pipe saturate with two waves per SIMD?"""
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
for one in (0, 1):
    os.environ["GNNB_ZF_ONE"] = str(one)
    for dbg, what in ((0, "everything"), (19, "MFMA phases only"), (31, "skeleton"), (12, "no MFMA at all"), (4, "no M1")):
        os.environ["GNNB_ZF_DBG"] = str(dbg)
        t = min(cm.gcn_stack_timed(bd[0], 100) for _ in range(3))
        print(f"workgroups per CU {2 - one}: dbg {dbg:2d} {what:20s} {t:6.2f} us", flush=True)
