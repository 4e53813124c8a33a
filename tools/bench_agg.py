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
    ap.add_argument("--sets", default="", help='JSON list of option dicts, e.g. [{"agg_variant":0,"agg_rows_per_wg":32}]')
    args = ap.parse_args()
    w = bench.WORKLOADS[args.workload]
    width = args.width or w["hidden"]
    dev = torch.device("cuda:0")
    model = bench.build_model(w)
    batch = synthetic.make_batch(w["shape"], w["batch"], seed=0)
    cm = runtime.CompiledModel.from_model(model, batch.num_graphs, batch.num_nodes, batch.num_edges)
    bd = tuple(torch.from_numpy(a).to(dev) for a in (batch.x, batch.coo, batch.node_ptr, batch.edge_ptr))
    sets = json.loads(args.sets) if args.sets else (
        [{"agg_variant": 1}] + [{"agg_variant": 0, "agg_rows_per_wg": r} for r in (16, 32, 64, 128, 256)])
    for opts in sets:
        for k, v in opts.items():
            runtime.set_option(k, v)
        alg, res = bench.measure_aggregate_roofline(cm, bd, width, dev, iters=200)
        print(json.dumps({"opts": opts, "width": width, "alg_MB": alg / 1e6,
                          "hbm_us": round(res["hbm"]["us"], 2), "hbm_GBps": round(res["hbm"]["gbps"]),
                          "l3_us": round(res["l3_resident"]["us"], 2), "l3_GBps": round(res["l3_resident"]["gbps"])}))


if __name__ == "__main__":
    main()
