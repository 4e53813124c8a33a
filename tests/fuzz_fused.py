#!/usr/bin/env python3
"""Randomised parity sweep of the fused conv-stack kernel (GCN / GIN, any depth) against the oracle on one GPU.
    python tests/fuzz_fused.py [cases] [seed] [zf]  (lives under tests/: it uses the oracle, which is test infrastructure)
With a third argument "zf": only 2-layer GCN models (k_gcn2_zf), promise 4..169, the stage shape (zf_shape 0 / 1 / 2) drawn per
case, and the path the workspace reports is asserted.
Random model shapes (depth 2..6, width 32 / 64 / 128, F_in 1..32, activation, skip, pool order), random multigraph
batches (1..300 graphs of 0..promise nodes, promise 4..61: empty graphs, isolated nodes, self loops, duplicate edges, hubs)
with the promise exactly met by at least one graph.  Prints the worst error; exits non-zero on a mismatch or when the
fused path did not take a model it should have taken."""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from helpers import canon, make_model, to_dev  # noqa: E402
from gnnbuilder_amd import runtime  # noqa: E402
from gnnbuilder_amd.batching import pack_graphs  # noqa: E402
from oracle import oracle as O  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ZF = len(sys.argv) > 3 and sys.argv[3] == "zf"
dev = torch.device("cuda:0")
worst = worst_reduced = 0.0
for it in range(cases):
    conv = "gcn" if ZF else rng.choice(["gcn", "gin"])
    L = 2 if ZF else int(rng.integers(2, 7))
    h = int(rng.choice([32, 64, 128]))
    out = h if conv == "gin" else int(rng.choice([h, 4 * int(rng.integers(1, 33))]))
    fin = int(rng.integers(1, 33))
    act = str(rng.choice(["relu", "gelu", "sigmoid", "tanh"]))
    skip = bool(rng.integers(0, 2))
    pools = tuple(rng.permutation(["add", "mean", "max"])[: int(rng.integers(1, 4))])
    model = make_model(conv, in_dim=fin, hidden=h, layers=L, out_dim=out, act=act, skip=skip, pools=pools, task_out=int(rng.integers(1, 5)),
                       seed=it)
    if conv == "gin":
        eps = float(rng.uniform(-0.5, 0.5))
        for c in model.gnn_convs:
            c.eps = eps
            c.conv.eps.fill_(eps)
    promise = int(rng.integers(4, 170 if ZF else 62))
    shape = int(rng.integers(0, 3))
    # (round 5) the opt-in forms drawn per case: the stage-cut planner of k_gcn2_fused, the MLP head inside k_gcn2_zf
    runtime.set_option("stage_cut", int(rng.integers(0, 2)))
    runtime.set_option("zf_head", int(rng.integers(0, 2)))
    math = 0
    if not ZF and (conv == "gin" or L > 2):
        math = 3 * int(rng.integers(0, 2))  # the opt-in f16x3 form of the GIN / deep-GCN stack kernel in half of those cases
        runtime.set_option("math", math)
    elif not ZF:
        runtime.set_option("math", 0)
    if ZF:
        math = int(rng.choice([0, 0, 2, 3]))  # the opt-in reduced-precision forms of k_gcn2_zf's wide update (bf16x3, f16x3) in half of the cases
        runtime.set_option("math", math)
        runtime.set_option("zf_shape", shape)
        if shape == 0 or fin > 16:
            promise = min(promise, 89)  # the 96-row shape (and every input wider than 16) holds graphs of up to 89 nodes
    B = int(rng.integers(1, 301))
    graphs = []
    for g in range(B):
        n = promise if g == B // 2 else int(rng.integers(0, promise + 1))
        e = int(rng.integers(0, 4 * n + 1)) if n else 0
        coo = np.stack([rng.integers(0, max(n, 1), e), rng.integers(0, max(n, 1), e)], 1).astype(np.int32) if e else np.zeros((0, 2), np.int32)
        if n and rng.integers(0, 4) == 0:            # a hub: many edges into one node
            hub = int(rng.integers(0, n))
            extra = np.stack([rng.integers(0, n, 12), np.full(12, hub)], 1).astype(np.int32)
            coo = np.concatenate([coo, extra])
        graphs.append((rng.uniform(-1, 1, (n, fin)).astype(np.float32), coo))
    batch = pack_graphs(graphs)
    if batch.num_nodes == 0:
        continue
    ref = O.forward_batched(model.spec(), canon(model), batch.x, batch.coo, batch.node_ptr, batch.edge_ptr)
    cm = runtime.CompiledModel.from_model(model, batch.num_graphs, batch.num_nodes, max(batch.num_edges, 1), max_graph_nodes=promise)
    args = to_dev(batch, dev)
    got = cm.forward(*args).cpu().numpy()
    cm.check()
    # (round 6) every other case once more through the software-pipelined entry: this forward + the SAME batch's prep on a second
    # workspace in one call (the prep as a guest of the readout kernel where eligible, a launch of its own elsewhere -- e.g. with
    # zf_head, promise > 64), then the forward on that workspace: the same bits
    if it % 2 == 0:
        cm2 = runtime.CompiledModel.from_model(model, batch.num_graphs, batch.num_nodes, max(batch.num_edges, 1), max_graph_nodes=promise)
        again = cm.forward_prepared_prep_next(args[0], cm2, args[1], args[2], args[3], batch.num_nodes).cpu().numpy()
        piped = cm2.forward_prepared(args[0]).cpu().numpy()
        cm.check()
        cm2.check()
        if not (np.array_equal(again, got) and np.array_equal(piped, got)):
            print(f"FAIL case {it}: gnnb_forward_prepared_prep_next differs from gnnb_forward_batched (promise {promise}, B={B})")
            sys.exit(1)
        cm2.close()
    took = True
    try:
        cm.gcn_stack_timed(args[0], 1)
    except runtime.GnnbError:
        took = False
    err = float(np.abs(got - ref).max()) / max(1.0, float(np.abs(ref).max()))
    worst = max(worst, err) if math == 0 else worst
    worst_reduced = max(worst_reduced, err) if math else worst_reduced
    if ZF and cm.last_path() != "stack_zf":
        print(f"FAIL case {it}: path {cm.last_path()} (shape {shape}, promise {promise}, F={fin})")
        sys.exit(1)
    tag = f"{'shape ' + str(shape) + ' ' if ZF else ''}math {math} {conv} L={L} h={h} out={out} F={fin} {act} skip={int(skip)} pools={'/'.join(pools)} promise={promise} B={B} N={batch.num_nodes}"
    if not took or not err < 1e-4:
        print(f"FAIL case {it}: {tag}: fused={took} err={err:.3e}")
        sys.exit(1)
    if it % 10 == 0:
        print(f"case {it}: {tag}: err {err:.2e}", flush=True)
    cm.close()
runtime.set_option("stage_cut", 0)
runtime.set_option("zf_head", 0)
runtime.set_option("math", 0)
print(f"{cases} cases, worst relative error {worst:.3e}" + f"; math 2 / 3 (bf16x3 / f16x3) cases: {worst_reduced:.3e}")
