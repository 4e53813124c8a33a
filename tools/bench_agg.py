#!/usr/bin/env python3
"""Sweep the gather-aggregate kernel's launch options on one GPU (tuning aid).

    python tools/bench_agg.py [--workload c2] [--width 128]
Prints microseconds per launch and algorithmic GB/s in the HBM regime (rotating buffers) and
with buffers resident in the Infinity Cache, for each option set.
"""
import argparse
import itertools
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from gnnbuilder_amd import runtime, synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c2")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--kind", default="gcn")
    ap.add_argument("--sets", default="", help='JSON list of option dicts, e.g. [{"agg_ring_waves":8}]')
    args = ap.parse_args()
    w = bench.WORKLOADS[args.workload]
    width = args.width or w["hidden"]
    dev = torch.device("cuda:0")
    model = bench.build_model(w)
    batch = synthetic.make_batch(w["shape"], w["batch"], seed=0)
    cm = runtime.CompiledModel.from_model(model, batch.num_graphs, batch.num_nodes, batch.num_edges)
    bd = tuple(torch.from_numpy(a).to(dev) for a in (batch.x, batch.coo, batch.node_ptr, batch.edge_ptr))
    defaults = dict(tile_rows=8, agg_lds_kb=0, agg_ring_waves=0, agg_ring_slots=2, agg_ring_wg_per_cu=1, agg_nt_store=1)
    sets = json.loads(args.sets) if args.sets else (
        [{}, {"agg_nt_store": 0}, {"agg_ring_waves": 8}, {"agg_ring_waves": 4}, {"agg_ring_slots": 3}, {"agg_ring_slots": 4},
         {"agg_ring_wg_per_cu": 2}, {"agg_ring_wg_per_cu": 2, "agg_ring_waves": 8}, {"tile_rows": 4}, {"tile_rows": 16}])
    for opts in sets:
        for k, v in {**defaults, **opts}.items():
            runtime.set_option(k, v)
        alg, res = bench.measure_aggregate_roofline(cm, bd, width, dev, iters=200, kind=args.kind)   # (graph prep inside: tile_rows applies)
        print(json.dumps({"kind": args.kind, "opts": opts, "width": width, "alg_MB": alg / 1e6,
                          "hbm_us": round(res["hbm"]["us"], 2), "hbm_GBps": round(res["hbm"]["gbps"]),
                          "l3_us": round(res["l3_resident"]["us"], 2), "l3_GBps": round(res["l3_resident"]["gbps"]),
                          "copy_us": round(res.get("copy_same_launch_shape", {}).get("us", 0), 2)}), flush=True)


if __name__ == "__main__":
    main()
