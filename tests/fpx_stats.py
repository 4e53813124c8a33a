#!/usr/bin/env python3
"""Development script (uses the oracle): how far the HIP fixed-point emulation lands from the oracle's, per conv / format."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, torch
from gnnbuilder_amd import runtime, synthetic
from oracle import oracle as O
from test_hip_parity import make_model, canon, to_dev
dev = torch.device("cuda:0")
for conv in ("gcn", "gin", "sage", "pna"):
    for W, I in ((32, 12), (16, 8), (12, 6)):
        worst, exact = 0.0, 1.0
        for seed in range(6):
            model = make_model(conv, in_dim=9, hidden=32, layers=3, task_out=3, seed=seed)
            batch = synthetic.make_batch("molhiv", 24, seed=W + seed)
            spec = dict(model.spec(), fpx=(W, I))
            ref = O.forward_batched(spec, canon(model), batch.x, batch.coo, batch.node_ptr, batch.edge_ptr)
            cm = runtime.CompiledModel.from_model(model, batch.num_graphs, batch.num_nodes, batch.num_edges, fpx=(W, I))
            out = cm.forward(*to_dev(batch, dev)).cpu().numpy()
            step = 2.0 ** -(W - I)
            worst = max(worst, float(np.abs(out - ref).max() / step))
            exact = min(exact, float(np.mean(out == ref)))
        print(f"{conv:5s} FPX({W},{I}): worst {worst:8.1f} steps apart, exact fraction >= {exact:.3f}")
