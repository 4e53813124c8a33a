#!/usr/bin/env python3
"""Randomised parity sweep of the layer-by-layer paths -- GraphSAGE and PNA whole models (ring-form narrow first layer, large-K
segmented GEMM incl. its stream-K tail and its N <= 64 form, pooling in the last GEMM's epilogue, PNA's folded `lin`) and
GCN / GIN without a promise -- against the oracle on one GPU.
    python tests/fuzz_layerwise.py [cases] [seed]      (lives under tests/: it uses the oracle, which is test infrastructure)
Random model shapes (depth 1..4, hidden 16 / 32 / 64 / 128 / 256, out any multiple of 4 up to hidden, F_in 1..32, activation,
skip, pool order), random batches (molecule-like graphs + empty graphs, isolated nodes, self loops, duplicate edges, hubs,
a few graphs of 100-400 nodes), PNA with and without a max_degree promise, every option combination of fuse_pool / pna_fold_lin / first_ring / gemm_tail_split
/ pna_pagg / pna_first / sage_first_mean drawn per case; half of the GraphSAGE / PNA cases keep every graph within 20 / 49 / 57 nodes and
set the max_graph_nodes promise (round 5: the LDS-staged kernels k_sage_first_mean, k_pna_pagg, k_pna_first need it).  Prints the worst error; exits non-zero on a mismatch."""
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

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda:0")
worst = worst_reduced = 0.0
MATH_DRAWN = len(sys.argv) > 3 and sys.argv[3] == "math"
OPTS = {"fuse_pool": 1, "pna_fold_lin": 1, "first_ring": 1, "gemm_tail_split": 2, "pna_pagg": 1, "pna_first": 1, "sage_first_mean": 1}
try:
    for it in range(cases):
        conv = str(rng.choice(["sage", "pna", "sage", "pna", "gcn", "gin"]))
        L = int(rng.integers(1, 5))
        h = int(rng.choice([16, 32, 64, 128, 256] if conv != "pna" else [16, 32, 64, 128]))
        out = h if (conv == "gin" or rng.integers(0, 2)) else 4 * int(rng.integers(1, h // 4 + 1))
        fin = int(rng.integers(1, 33))
        if conv == "pna" and rng.integers(0, 2):
            fin = int(rng.integers(8, 13))  # (the widths k_pna_first takes)
        small = conv in ("sage", "pna") and bool(rng.integers(0, 2))  # every graph within a stage: the promise can be made
        nmax = int(rng.choice([20, 49, 57])) if small else 60
        act = str(rng.choice(["relu", "gelu", "sigmoid", "tanh"]))
        skip = bool(rng.integers(0, 2))
        pools = tuple(rng.permutation(["add", "mean", "max"])[: int(rng.integers(1, 4))])
        model = make_model(conv, in_dim=fin, hidden=h, layers=L, out_dim=out, act=act, skip=skip, pools=pools,
                           task_out=int(rng.integers(1, 5)), seed=it)
        B = int(rng.integers(1, 2500 if h <= 64 else 900))
        mean_n = float(rng.choice([3.0, 12.0, 25.0]))
        graphs = []
        for g in range(B):
            r = rng.integers(0, 40)
            n = 0 if r == 0 else (int(rng.integers(100, 400)) if r == 1 and g % 7 == 0 and not small else int(np.clip(rng.normal(mean_n, mean_n / 3), 1, nmax)))
            e = int(rng.integers(0, 3 * n + 1)) if n else 0
            coo = np.stack([rng.integers(0, max(n, 1), e), rng.integers(0, max(n, 1), e)], 1).astype(np.int32) if e else np.zeros((0, 2), np.int32)
            if n and rng.integers(0, 30) == 0 and it % 3 == 0:   # a hub: many edges into one node (a third of the cases)
                hub = int(rng.integers(0, n))
                coo = np.concatenate([coo, np.stack([rng.integers(0, n, 20), np.full(20, hub)], 1).astype(np.int32)])
            graphs.append((rng.uniform(-1, 1, (n, fin)).astype(np.float32), coo))
        batch = pack_graphs(graphs)
        if batch.num_nodes == 0:
            continue
        opts = {k: int(rng.integers(0, v + 1)) if rng.integers(0, 3) == 0 else v for k, v in OPTS.items()}
        for k, v in opts.items():
            runtime.set_option(k, v)
        math = int(rng.choice([0, 0, 0, 1, 3])) if MATH_DRAWN else 0  # (argv[3] = "math": the opt-in modes in 2 of 5 cases)
        runtime.set_option("math", math)
        opts["math"] = math
        ref = O.forward_batched(model.spec(), canon(model), batch.x, batch.coo, batch.node_ptr, batch.edge_ptr)
        promise_n = int(np.diff(batch.node_ptr).max()) if small else 0
        cm = runtime.CompiledModel.from_model(model, batch.num_graphs, batch.num_nodes, max(batch.num_edges, 1), max_graph_nodes=promise_n)
        maxdeg = int(np.bincount(batch.coo[:, 1]).max()) if batch.num_edges else 0
        promise = 0
        if conv == "pna" and 0 < maxdeg <= 15 and rng.integers(0, 4) != 0:
            promise = maxdeg if rng.integers(0, 2) else 15   # (the degree-class form of the post-NN product)
            cm.set_max_degree(promise)
        args = to_dev(batch, dev)
        got = cm.forward(*args).cpu().numpy()
        cm.check()
        again = cm.forward(*args).cpu().numpy()
        if it % 2 == 0:
            # (round 6) the software-pipelined entry: this forward + the SAME batch's prep on a second workspace in one call (a guest
            # of the readout kernel where the workspace is eligible, a launch of its own elsewhere), then the forward there: the same bits
            cm2 = runtime.CompiledModel.from_model(model, batch.num_graphs, batch.num_nodes, max(batch.num_edges, 1), max_graph_nodes=promise_n)
            if promise:
                cm2.set_max_degree(promise)
            first = cm.forward_prepared_prep_next(args[0], cm2, args[1], args[2], args[3], batch.num_nodes).cpu().numpy()
            piped = cm2.forward_prepared(args[0]).cpu().numpy()
            cm2.check()
            if not (np.array_equal(first, got) and np.array_equal(piped, got)):
                print(f"FAIL case {it}: gnnb_forward_prepared_prep_next differs from gnnb_forward_batched ({conv}, max_graph_nodes={promise_n}, maxdeg promise {promise})")
                sys.exit(1)
            cm2.close()
        err = float(np.abs(got - ref).max()) / max(1.0, float(np.abs(ref).max()))
        worst = max(worst, err) if math != 3 else worst
        worst_reduced = max(worst_reduced, err) if math == 3 else worst_reduced
        tag = (f"{conv} L={L} h={h} out={out} F={fin} {act} skip={int(skip)} pools={'/'.join(pools)} B={B} N={batch.num_nodes} "
               f"opts={opts} maxdeg={maxdeg} promise={promise} max_graph_nodes={promise_n} path={cm.last_path()}")
        if not err < 1e-4 or not np.array_equal(got, again):
            print(f"FAIL case {it}: {tag}: err={err:.3e} repeatable={np.array_equal(got, again)}")
            sys.exit(1)
        if it % 10 == 0:
            print(f"case {it}: {tag}: err {err:.2e}", flush=True)
        cm.close()
finally:
    for k, v in OPTS.items():
        runtime.set_option(k, v)
    runtime.set_option("math", 0)
print(f"{cases} cases, worst relative error {worst:.3e}" + (f"; math 3 (f16x3) cases: {worst_reduced:.3e}" if MATH_DRAWN else ""))
