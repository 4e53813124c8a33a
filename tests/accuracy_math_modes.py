#!/usr/bin/env python3
"""(Kept under tests/: it calls the CPU oracle, which only test infrastructure may do.)
Accuracy of the two math modes of the fused GCN stack against a float64 evaluation of the same model
(BASELINE config 2 shape): fp32 C oracle, HIP fp32-MFMA path (math 0), HIP bf16x6 path (math 1: the 2-layer stack keeps its
fp32 kernel), HIP bf16x3 path (math 2: the stack kernel's wide update on hi + mid bf16 pieces, reduced precision)."""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))  # helpers
import bench  # noqa: E402
from gnnbuilder_amd import runtime, synthetic  # noqa: E402
from helpers import canon  # noqa: E402
from oracle import oracle as O  # noqa: E402


def forward_f64(model, batch, x):
    """2-layer GCN + pooling + MLP head in float64 (PyG semantics: dinv = (1 + in_degree)^-1/2)."""
    sd = {k: v.detach().double().numpy() for k, v in model.state_dict().items()}
    N = batch.num_nodes
    src, dst = batch.coo[:, 0], batch.coo[:, 1]
    deg = np.bincount(dst, minlength=N).astype(np.float64) + 1.0
    dinv = deg ** -0.5

    def agg(h):
        out = h * (dinv * dinv)[:, None]
        np.add.at(out, dst, h[src] * (dinv[src] * dinv[dst])[:, None])
        return out

    h = x.astype(np.float64)
    for l in range(2):
        w, b = sd[f"gnn_convs.{l}.conv.lin.weight"], sd[f"gnn_convs.{l}.conv.bias"]
        h = np.maximum(agg(h) @ w.T + b, 0.0)
    pooled = []
    for g in range(batch.num_graphs):
        r = h[batch.node_ptr[g]:batch.node_ptr[g + 1]]
        pooled.append(np.concatenate([r.sum(0), r.mean(0), r.max(0)]) if len(r) else np.zeros(3 * h.shape[1]))
    z = np.stack(pooled)
    nl = sum(1 for k in sd if k.startswith("mlp_head.linear_layers.") and k.endswith(".weight"))
    for i in range(nl):
        z = z @ sd[f"mlp_head.linear_layers.{i}.weight"].T + sd[f"mlp_head.linear_layers.{i}.bias"]
        if i < nl - 1:
            z = np.maximum(z, 0.0)
    return z


def main():
    w = bench.WORKLOADS["c2"]
    model = bench.build_model(w)
    batch = synthetic.make_batch(w["shape"], w["batch"], seed=3)
    x = batch.x
    ref = forward_f64(model, batch, x)
    dev = torch.device("cuda:0")
    cm = runtime.CompiledModel.from_model(model, batch.num_graphs, batch.num_nodes, batch.num_edges,
                                          max_graph_nodes=int(np.diff(batch.node_ptr).max()))
    bd = tuple(torch.from_numpy(a).to(dev) for a in (x, batch.coo, batch.node_ptr, batch.edge_ptr))
    res = {"oracle fp32 (C, scalar order)": O.forward_batched(model.spec(), canon(model), x, batch.coo, batch.node_ptr, batch.edge_ptr)}
    for math, name in ((0, "HIP math 0: fp32 MFMA"), (1, "HIP math 1: bf16x6 MFMA"), (2, "HIP math 2: bf16x3 MFMA (reduced)"), (3, "HIP math 3: f16x3 MFMA (reduced)")):
        runtime.set_option("math", math)
        res[name] = cm.forward(*bd).cpu().numpy()
        print(f"  ({name}: path {cm.last_path()})")
    runtime.set_option("math", 0)
    scale = np.abs(ref).max()
    print(f"{batch.num_graphs} graphs, outputs |max| = {scale:.3f}; error against the float64 evaluation:")
    for name, out in res.items():
        e = np.abs(out - ref)
        print(f"  {name:36s} max abs {e.max():.3e}  mean abs {e.mean():.3e}  max rel-to-scale {e.max() / scale:.3e}")


if __name__ == "__main__":
    main()
