#!/usr/bin/env python3
"""Register-gather aggregate (k_aggregate_rg, `agg_form` 1) against the LDS ring (k_aggregate_ring): same results on
every kind / width it takes, then microseconds per launch in the HBM regime for a set of launch options.

    python tools/rg_vs_ring.py [--workload c2] [--sets '[{"agg_rg_r":2}, ...]']
"""
import argparse
import json
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from gnnbuilder_amd import runtime, synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c2")
    ap.add_argument("--kind", default="")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--sets", default="")
    ap.add_argument("--skip-check", action="store_true")
    args = ap.parse_args()
    w = bench.WORKLOADS[args.workload]
    dev = torch.device("cuda:0")
    model = bench.build_model(w)
    batch = synthetic.make_batch(w["shape"], w["batch"], seed=0)
    cm = runtime.CompiledModel.from_model(model, batch.num_graphs, batch.num_nodes, batch.num_edges)
    bd = tuple(torch.from_numpy(a).to(dev) for a in (batch.x, batch.coo, batch.node_ptr, batch.edge_ptr))
    N = batch.num_nodes
    cm.graph_prep(bd[1], bd[2], bd[3], N)
    if not args.skip_check:
        for width in (64, 128, 256):
            x = torch.rand(N, width, device=dev) * 2 - 1
            q = torch.rand(N, width, device=dev) * 2 - 1
            for kind in ("gcn", "sum", "mean", "simple", "pna_q"):
                k, st = ("pna", q) if kind == "pna_q" else (kind, None)
                runtime.set_option("agg_form", 0)
                ref = cm.aggregate(k, x, self_term=st, eps=0.25).clone()
                for r in (1, 2, 3, 4):
                    for flags in (0, 1):
                        runtime.set_option("agg_form", 1)
                        runtime.set_option("agg_rg_r", r)
                        runtime.set_option("agg_rg_flags", flags)
                        got = cm.aggregate(k, x, self_term=st, eps=0.25)
                        torch.cuda.synchronize()
                        err = (got - ref).abs().max().item()
                        same = torch.equal(got, ref)
                        if not err < 1e-6:
                            print(f"MISMATCH {kind} w={width} R={r} flags={flags}: {err:.3e}")
                        elif r == 2 and flags == 0:
                            print(f"ok {kind} w={width}: max |rg - ring| = {err:.1e} bit-identical={same}")
        runtime.set_option("agg_rg_r", 0)
        runtime.set_option("agg_rg_flags", 0)
    width = args.width or w["hidden"]
    kind = args.kind or bench.WORKLOAD_AGG[w["conv"]][0]
    sets = json.loads(args.sets) if args.sets else (
        [{"agg_form": 0}] + [{"agg_form": 1, "agg_rg_r": r, "agg_rg_wgs": g, "agg_rg_flags": f}
                             for r, gs in ((1, (8, 16, 32)), (2, (4, 5, 10, 20)), (3, (3, 6, 12)), (4, (2, 4, 8))) for g in gs for f in (0, 1)])
    base = dict(agg_form=0, agg_rg_r=0, agg_rg_wgs=0, agg_rg_flags=1, agg_nt_store=1)
    for opts in sets:
        for k, v in {**base, **opts}.items():
            runtime.set_option(k, v)
        alg, res = bench.measure_aggregate_roofline(cm, bd, width, dev, iters=200, kind=kind)
        print(json.dumps({"kind": kind, "width": width, "opts": opts, "alg_MB": round(alg / 1e6, 1), "hbm_us": round(res["hbm"]["us"], 2),
                          "hbm_frac": round(res["hbm"]["gbps"] / 8000, 3), "l3_us": round(res["l3_resident"]["us"], 2)}), flush=True)
    for k, v in base.items():
        runtime.set_option(k, v)


if __name__ == "__main__":
    main()
