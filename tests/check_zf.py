#!/usr/bin/env python3
"""Development check (uses the oracle: lives under tests/) of the transform-first GCN stack kernel (k_gcn2_zf) on the BASELINE config 2 batch: parity
against the C oracle on sampled graphs and against k_gcn2_fused, launch-loop times of both (HIP events)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np, torch
import bench
from gnnbuilder_amd import runtime, synthetic
from oracle import oracle as O

wname = sys.argv[1] if len(sys.argv) > 1 else "c2"
w = bench.WORKLOADS[wname]; dev = torch.device("cuda:0")
model = bench.build_model(w)
for seed in (0, 5):
    b = synthetic.make_batch(w["shape"], w["batch"], seed=seed)
    mg = int(np.diff(b.node_ptr).max())
    cm = runtime.CompiledModel.from_model(model, b.num_graphs, b.num_nodes, b.num_edges, max_graph_nodes=mg)
    bd = tuple(torch.from_numpy(a).to(dev) for a in (b.x, b.coo, b.node_ptr, b.edge_ptr))
    outs = {}
    keep = []
    for zf in (1, 0, 2):
        runtime.set_option("fuse_zf", 1 if zf else 0)
        runtime.set_option("zf_shape", 0 if zf == 2 else 1)
        # (a workspace of its own per variant: the pooled matrix of another variant's run must not show through)
        cmv = runtime.CompiledModel.from_model(model, b.num_graphs, b.num_nodes, b.num_edges, max_graph_nodes=mg)
        outs[zf] = cmv.forward(*bd).cpu().numpy()
        cmv.check()
        print(f"seed {seed} zf={zf}: path {cmv.last_path()}, finite {np.isfinite(outs[zf]).all()}")
        keep.append(cmv)
    runtime.set_option("fuse_zf", 1)
    runtime.set_option("zf_shape", 2)
    print(f"  max |zf - fused| = {np.abs(outs[1] - outs[0]).max():.3e}, |zf shape 0 - fused| = {np.abs(outs[2] - outs[0]).max():.3e} (scale {np.abs(outs[0]).max():.3f})")
    idx = np.unique(np.concatenate([np.arange(0, 64), np.arange(b.num_graphs - 64, b.num_graphs),
                                    np.random.default_rng(1).integers(0, b.num_graphs, 128), [int(np.diff(b.node_ptr).argmax())]]))
    spec, params = model.spec(), [p.numpy() for p in model.canonical_params()]
    worst = 0.0
    for g in idx:
        sub = b.slice(int(g), int(g) + 1)
        ref = O.forward_batched(spec, params, sub.x, sub.coo, sub.node_ptr, sub.edge_ptr)
        worst = max(worst, float(np.abs(outs[1][g] - ref[0]).max()))
    print(f"  max |zf - oracle| over {len(idx)} graphs = {worst:.3e}")
    cm.graph_prep(bd[1], bd[2], bd[3], int(bd[0].shape[0]))
    for zf, shape in ((1, 1), (1, 0), (0, 0), (1, 1), (1, 0), (0, 0)):
        runtime.set_option("fuse_zf", zf)
        runtime.set_option("zf_shape", shape)
        cm.graph_prep(bd[1], bd[2], bd[3], int(bd[0].shape[0]))
        t = [cm.gcn_stack_timed(bd[0], 200) for _ in range(3)]
        print(f"  zf={zf} shape={shape}: " + " ".join(f"{v:.2f}" for v in t) + " us per launch")
    runtime.set_option("fuse_zf", 1)
    runtime.set_option("zf_shape", 2)
