#!/usr/bin/env python3
"""Development: what does the graph prep cost ONE stream of forwards, by host?  Two workspaces, BASELINE config 2: a loop of
forward_prepared (stack + head: no prep) against a loop of forward_prepared_prep_next (the same + the other workspace's prep as
extra workgroups of the readout kernel) and a loop of forward (prep launch + stack + head).  us per call, median of five loops
(49.8 / 52.0 / 55.7 with one graph per prep wave, 58.9 with groups of four)."""
import os, sys, time, json
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench
from gnnbuilder_amd import runtime, synthetic
w = bench.WORKLOADS["c2"]; dev = torch.device("cuda:0")
model = bench.build_model(w)
bs = [synthetic.make_batch(w["shape"], w["batch"], seed=i) for i in range(2)]
mg = max(int(np.diff(b.node_ptr).max()) for b in bs)
cap = [max(getattr(b, a) for b in bs) for a in ("num_graphs", "num_nodes", "num_edges")]
cms = [runtime.CompiledModel.from_model(model, *cap, max_graph_nodes=mg) for _ in range(2)]
bd = [tuple(torch.from_numpy(a).to(dev) for a in (b.x, b.coo, b.node_ptr, b.edge_ptr)) for b in bs]
outs = [torch.empty(b.num_graphs, cms[0].out_dim, device=dev) for b in bs]
for i in range(2):
    cms[i].graph_prep(bd[i][1], bd[i][2], bd[i][3], bs[i].num_nodes)
def loop(kind, n=200):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        k = i & 1
        if kind == "prepared":
            cms[k].forward_prepared(bd[k][0], out=outs[k])
        elif kind == "prep_next":
            cms[k].forward_prepared_prep_next(bd[k][0], cms[k ^ 1], bd[k ^ 1][1], bd[k ^ 1][2], bd[k ^ 1][3], bs[k ^ 1].num_nodes, out=outs[k])
        else:
            cms[k].forward(*bd[k], out=outs[k])
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
res = {}
for kind in ("prepared", "prep_next", "batched", "prepared", "prep_next", "batched"):
    loop(kind, 50)
    res.setdefault(kind, []).append(float(np.median([loop(kind) for _ in range(5)])))
print(json.dumps({k: [round(x, 2) for x in v] for k, v in res.items()}), os.environ.get("GNNB_HIP_LIB", "shipped"))
