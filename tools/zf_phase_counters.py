#!/usr/bin/env python3
"""Dynamic per-phase instruction counts of k_gcn2_zf at BASELINE config 2, from hardware counters.

The development build of the kernel (-DGNNB_ZF_ABLATE, tools/build_variant.sh ablate k_stack_zf "-DGNNB_ZF_ABLATE") can skip
single phases (GNNB_ZF_DBG bits: 0 P1, 1 P0', 2 M1, 3 M0, 4 the Z write; 32 = return at once; 64 = return behind the prologue).
Run under the profiler, ONE counter group per pass, the program directly behind `--`:

    GNNB_HIP_LIB=$PWD/gnn-builder_amd/libgnnb_v_ablate.so rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS \
        --output-format csv -d gpurun_out/zfph/inst -o inst -- python3 tools/zf_phase_counters.py run
    python3 tools/zf_phase_counters.py summarize gpurun_out/zfph  profiles/r06_c2_gcn2_phase_counters.json

`run` launches the kernel REPS times per variant in a fixed order; `summarize` reads the per-dispatch counter rows back in
that order and forms  phase = everything - (everything without the phase).  Skipping a phase gives WRONG results; nothing here
is a measurement of the shipped library's speed.
"""
from __future__ import annotations

import csv
import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

REPS = 3
VARIANTS = [(0, "everything"), (1, "without P1 (aggregate of Z + pooling)"), (2, "without P0' (next stage's narrow aggregate + records)"),
            (4, "without M1"), (8, "without M0"), (16, "without the Z write"), (1 | 2 | 4 | 8 | 16, "without every phase (DMA, barriers, plan, prologue)"),
            (64, "launch + tables + weights + first DMA landed"), (32, "launch only")]


def run() -> None:
    import numpy as np
    import torch

    import bench
    from gnnbuilder_amd import runtime, synthetic

    w = bench.WORKLOADS["c2"]
    dev = torch.device("cuda:0")
    model = bench.build_model(w)
    b = synthetic.make_batch(w["shape"], w["batch"], seed=0)
    cm = runtime.CompiledModel.from_model(model, b.num_graphs, b.num_nodes, b.num_edges, max_graph_nodes=int(np.diff(b.node_ptr).max()))
    bd = tuple(torch.from_numpy(a).to(dev) for a in (b.x, b.coo, b.node_ptr, b.edge_ptr))
    cm.graph_prep(bd[1], bd[2], bd[3], int(bd[0].shape[0]))
    for dbg, what in VARIANTS:
        os.environ["GNNB_ZF_DBG"] = str(dbg)
        t = cm.gcn_stack_timed(bd[0], REPS)
        print(f"dbg {dbg:3d} {what:60s} {t:8.2f} us per launch (under the profiler when profiled)", flush=True)
    os.environ["GNNB_ZF_DBG"] = "0"


def summarize(src: str, out: str) -> None:
    per_counter: dict[str, list[float]] = {}
    for path in sorted(Path(src).rglob("*counter_collection.csv")):
        rows = list(csv.DictReader(path.open()))
        rows = [r for r in rows if "k_gcn2_zf" in r.get("Kernel_Name", "")]
        by_counter: dict[str, dict[int, float]] = {}
        for r in rows:
            by_counter.setdefault(r["Counter_Name"], {})
            d = int(r["Dispatch_Id"])
            by_counter[r["Counter_Name"]][d] = by_counter[r["Counter_Name"]].get(d, 0.0) + float(r["Counter_Value"])
        for name, vals in by_counter.items():
            per_counter[name] = [vals[k] for k in sorted(vals)]
    table = {}
    for name, seq in per_counter.items():
        per = len(seq) // len(VARIANTS) # (gnnb_gcn_stack_timed launches REPS warm-up launches in front of the REPS it times)
        if per * len(VARIANTS) != len(seq) or per < REPS:
            print(f"{name}: {len(seq)} dispatches, expected a multiple of {len(VARIANTS)}", file=sys.stderr)
            continue
        table[name] = {what: sum(seq[i * per:(i + 1) * per]) / per for i, (_, what) in enumerate(VARIANTS)}
    phases = {}
    for name, t in table.items():
        full = t["everything"]
        phases[name] = {"everything": full,
                        "P1": full - t[VARIANTS[1][1]], "P0'": full - t[VARIANTS[2][1]], "M1": full - t[VARIANTS[3][1]],
                        "M0": full - t[VARIANTS[4][1]], "Z write": full - t[VARIANTS[5][1]],
                        "skeleton (DMA issue, waits, barriers, plan, prologue)": t[VARIANTS[6][1]],
                        "prologue alone": t[VARIANTS[7][1]], "launch alone": t[VARIANTS[8][1]]}
    if "SQ_INSTS_VALU" in phases and "SQ_INSTS_MFMA" in phases:
        phases["VALU_without_MFMA"] = {k: phases["SQ_INSTS_VALU"][k] - phases["SQ_INSTS_MFMA"][k] for k in phases["SQ_INSTS_VALU"]}
    res = {"what": "k_gcn2_zf<RELU, 1, 8, 16 waves, 11 units, fp32> at BASELINE config 2, wave-instructions per launch, per phase = "
                   "everything - the launch without that phase (-DGNNB_ZF_ABLATE development build, rocprofv3 --pmc, one group per pass)",
           "reps_per_variant": REPS, "variants": table, "phases": phases}
    Path(out).write_text(json.dumps(res, indent=1))
    for name, p in phases.items():
        print(name, {k: round(v) for k, v in p.items()})


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "summarize":
        summarize(sys.argv[2], sys.argv[3])
    else:
        run()
