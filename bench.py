#!/usr/bin/env python3
"""bench.py -- the hot path on N MI355X GPUs of one node: graphs/sec + gather-aggregate HBM GB/s.

A "step" is one batched forward (graph prep + conv stack + pooling + MLP head) over one batch of
synthetic molecule-shaped graphs that is already resident in HBM.  Default workload = BASELINE.json
configs[1]: 2-layer GCN d=128, QM9-shaped graphs, batch 4096 per GPU.  One process per GPU
(torch.distributed / RCCL); graphs are independent, so ranks shard batches with no data-path
collective and only the throughput counters are reduced (weak scaling).

Prints ONE JSON line on rank 0 (contract in the task prompt) with these extra objects:
  roofline      -- the kernel that dominates the timed step.  Workload c2 (2-layer GCN with a
                   max_graph_nodes promise) runs the fused stack kernel k_gcn2_fused (both conv layers +
                   pooling, graphs staged in LDS, no HBM round trips): bound = fp32 MFMA, algorithmic
                   flops 2 N (F0 h0 + h0 h1) / launch duration from HIP events on the launch stream.
                   The other workloads run layer by layer and are dominated by the gather-aggregate
                   kernel: bound = HBM (see next).
  roofline_gather_aggregate -- the GCN gather-aggregate kernel at the full feature width (the
                   north-star kernel; every layer-by-layer model runs it): algorithmic bytes (SURVEY.md
                   8d) / measured launch duration (HIP events on the launch stream, rotating through
                   distinct buffers > 256 MiB so the Infinity Cache cannot serve the reads) against 8 TB/s.
  cpu_baseline  -- the reference's own C++ kernel library (oracle/_ref, compiled in place from
                   /root/reference; falls back to the C oracle port when it is absent) running the
                   same model on a bounded sample of the same graphs on ONE host core, rank 0 only.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

HBM_PEAK_GBPS = 8000.0       # MI355X_MICROARCH.md chip table (spec); 6290 measured float4 copy
FP32_MFMA_PEAK_TFLOPS = 157.3

WORKLOADS = {
    # BASELINE.json configs[1..4]
    "c2": dict(desc="2-layer GCN d=128, QM9-shaped synthetic graphs, batch=4096 per GPU", conv="gcn", shape="qm9",
               hidden=128, layers=2, pools=("add", "mean", "max"), batch=4096),
    "c3": dict(desc="3-layer GIN d=128 + sum-pool, ogbg-molhiv-shaped graphs, batch=4096 per GPU", conv="gin",
               shape="molhiv", hidden=128, layers=3, pools=("add",), batch=4096),
    "c4": dict(desc="3-layer PNA d=128, QM9-shaped graphs, batch=8192 per GPU", conv="pna", shape="qm9",
               hidden=128, layers=3, pools=("add", "mean", "max"), batch=8192),
    "c5": dict(desc="2-layer GraphSAGE d=256, ogbg-molhiv-shaped graphs, batch=8192 per GPU (65536 over 8)",
               conv="sage", shape="molhiv", hidden=256, layers=2, pools=("add", "mean", "max"), batch=8192),
}


def build_model(w, seed=0):
    import torch
    import gnnbuilder_amd as gnnb
    from gnnbuilder_amd import synthetic

    torch.manual_seed(seed)
    convs = {"gcn": gnnb.GCNConv_GNNB, "gin": gnnb.GINConv_GNNB, "sage": gnnb.SAGEConv_GNNB, "pna": gnnb.PNAConv_GNNB}
    shp = synthetic.SHAPES[w["shape"]]
    return gnnb.GNNModel(shp["f_in"], None, w["hidden"], w["layers"], w["hidden"], convs[w["conv"]], torch.nn.ReLU,
                         True, gnnb.GlobalPooling(list(w["pools"])),
                         gnnb.MLP(len(w["pools"]) * w["hidden"], shp["out"], 64, 2), None).eval()


def measure_aggregate_roofline(cm, batch_dev, width, dev, iters=200, regimes=("hbm", "l3_resident")):
    """GCN gather-aggregate at `width`, timed with HIP events on the launch stream.  Returns both
    the HBM regime (inputs/outputs rotate over > 256 MiB of distinct buffers) and the regime the
    kernel sees inside the pipeline (same buffers every launch: Infinity-Cache resident)."""
    import torch
    from gnnbuilder_amd import runtime

    x, coo, nptr, eptr = batch_dev
    N, E, B = int(x.shape[0]), int(coo.shape[0]), int(nptr.numel()) - 1
    cm.graph_prep(coo, nptr, eptr, N)
    # read every input row once + write every output row once + CSR + graph ptr
    alg_bytes = 4 * width * N + 4 * width * N + 4 * (N + 1) + 4 * E + 4 * (B + 1)
    per_pair = 2 * 4 * width * N
    nbuf = max(2, int(np.ceil(320 * 2**20 / per_pair)) + 1)
    ins = [torch.rand(N, width, device=dev) * 2 - 1 for _ in range(nbuf)]
    outs = [torch.empty(N, width, device=dev) for _ in range(nbuf)]
    res = {}
    for regime, n in (("hbm", nbuf), ("l3_resident", 1)):
        if regime not in regimes:
            continue
        us = cm.aggregate_timed("gcn", ins[:n], outs[:n], iters)  # launches issued from C
        res[regime] = dict(us=us, gbps=alg_bytes / (us * 1e-6) / 1e9)
    del ins, outs
    return alg_bytes, res


def measure_update_mfma(w, N, dev, iters=100):
    """The dense update of the full-width layer (X[N,d] . W[d,d]^T + b, ReLU) on fp32 MFMA."""
    import torch
    from gnnbuilder_amd import runtime

    d = w["hidden"]
    a = torch.rand(N, d, device=dev) - 0.5
    wt = (torch.rand(d, d, device=dev) - 0.5) / d ** 0.5
    b = torch.rand(d, device=dev)
    y = torch.empty(N, d, device=dev)
    us = runtime.linear_timed(a, wt, b, y, "relu", iters)
    flops = 2.0 * N * d * d
    return dict(us=us, tflops=flops / (us * 1e-6) / 1e12, frac=flops / (us * 1e-6) / 1e12 / FP32_MFMA_PEAK_TFLOPS)


def cpu_baseline(model, batches, budget_s=12.0):
    """Reference CPU path on ONE core over a bounded sample (~budget_s of CPU work) of the same
    workload: the per-graph loop of the reference testbench (model_tb.cpp.jinja:189-205)."""
    from oracle import oracle as O

    try:
        os.sched_setaffinity(0, {sorted(os.sched_getaffinity(0))[0]})  # as build_base_benchmarks.py:188-189
    except Exception:
        pass
    spec, params = model.spec(), [p.numpy() for p in model.canonical_params()]
    kind = "reference" if O.have_ref() else "port"
    chunk, done, t_total = 256, 0, 0.0
    bi, g0 = 0, 0
    while t_total < budget_s:
        batch = batches[bi]
        g1 = min(g0 + chunk, batch.num_graphs)
        sub = batch.slice(g0, g1)
        t0 = time.perf_counter()
        if kind == "reference":
            try:
                O.ref_forward_batched(spec, params, sub.x, sub.coo, sub.node_ptr, sub.edge_ptr)
            except ValueError:  # a graph beyond the reference build's MAX_NODES / an uninstantiated size
                kind = "port"
                continue
        else:
            O.forward_batched(spec, params, sub.x, sub.coo, sub.node_ptr, sub.edge_ptr, std="pyg")
        t_total += time.perf_counter() - t0
        done += g1 - g0
        g0 = g1
        if g0 >= batch.num_graphs:
            bi, g0 = (bi + 1) % len(batches), 0  # cycle: the sample is bounded by CPU time, not by graphs
    what = ("the reference's own C++ kernel library (gnn_builder_lib.h, float mode, g++ -O2) compiled in place "
            "as oracle/_ref" if kind == "reference" else "C oracle port (oracle/gnnb_oracle.c, gcc -O2)")
    try:
        cpu_model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        cpu_model = "unknown"
    return {"value": done / t_total, "unit": "graphs/s", "cores": 1, "kind": kind,
            "sample": f"{done} graphs of the timed workload (its batches in order, cycled), one graph per call, {t_total:.1f} s of CPU time; {what}",
            "host_cpu": cpu_model, "host_cores_available": os.cpu_count()}


def pmc_traffic(kind="aggregate"):
    """HBM bytes per launch of a kernel from the latest committed rocprofv3 PMC passes (FETCH_SIZE x2 on
    gfx950 + WRITE_SIZE, tools/profile_r.sh -> profiles/*_<kind>_pmc.json).  Counters cannot be read from
    inside this process; None when no profile is committed."""
    try:
        files = sorted((ROOT / "profiles").glob(f"*_{kind}_pmc.json"))
        d = json.loads(files[-1].read_text())
        t = d["hbm_traffic_bytes_per_launch"]
        return {"bytes_per_launch": t["total"], "read_bytes_fetch_size_x2": t["read_corrected_x2"],
                "write_bytes": t["write"], "over_algorithmic": t["over_algorithmic"],
                "source": f"profiles/{files[-1].name} (rocprofv3 --pmc, separate passes)"}
    except Exception:
        return None


def measure_fused_stack(cm, batch_dev, model_dims, iters=200):
    """The fused 2-layer GCN stack + pooling kernel on one prepared batch: launches issued back to back
    from C, HIP events on the launch stream.  Returns None when the path is not eligible."""
    from gnnbuilder_amd import runtime

    x, coo, nptr, eptr = batch_dev
    N, E, B = int(x.shape[0]), int(coo.shape[0]), int(nptr.numel()) - 1
    f0, h0, h1, npool = model_dims
    cm.graph_prep(coo, nptr, eptr, N)
    try:
        us = cm.gcn_stack_timed(x, iters)
    except RuntimeError:
        return None
    flops = 2.0 * N * (f0 * h0 + h0 * h1)          # the two dense updates (MFMA); aggregation flops not counted
    # HBM bytes the kernel has to move: x + node records + dinv + tile/graph tables in, pooled out
    alg_bytes = 4 * N * f0 + 32 * N + 4 * N + 4 * (B + 1) + 4 * B * npool * h1
    return dict(us=us, tflops=flops / (us * 1e-6) / 1e12, flops=flops, alg_bytes=alg_bytes)


def copy_ceiling(N, width, dev, iters=200):
    """Calibration beside the roofline: a plain streaming copy (torch's vectorised kernel) of the
    same [N, width] fp32 matrix = the same read + write bytes with no gather at all."""
    import torch

    nbuf = max(2, int(np.ceil(320 * 2**20 / (2 * 4 * width * N))) + 1)
    a = [torch.rand(N, width, device=dev) for _ in range(nbuf)]
    b = [torch.empty(N, width, device=dev) for _ in range(nbuf)]
    out = {}
    for regime, n in (("hbm", nbuf), ("l3_resident", 1)):
        for i in range(10):
            b[i % n].copy_(a[i % n])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(iters):
            b[i % n].copy_(a[i % n])
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / iters
        out[regime] = dict(us=us, gbps=2 * 4 * width * N / (us * 1e-6) / 1e9)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--batches", type=int, default=8, help="distinct synthetic batches rotated per rank")
    ap.add_argument("--streams", type=int, default=3,
                    help="batches in flight per GPU: each on its own HIP stream and workspace, so the MFMA-bound "
                         "update of one batch overlaps the HBM-bound gather / readout of the next")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--roofline-only", action="store_true",
                    help="only the HBM-regime gather-aggregate loop (for a rocprofv3 run whose kernel average is that loop)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from gnnbuilder_amd import runtime, synthetic

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            print(f"bench.py: --gpus {args.gpus} needs `python -m torch.distributed.run --nproc-per-node {args.gpus}`",
                  file=sys.stderr)
            sys.exit(2)
    runtime.load_library(require_gpu=True)  # no fallback: fail loudly without the HIP path
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    w = WORKLOADS[args.workload]
    model = build_model(w)
    if args.roofline_only:
        batch = synthetic.make_batch(w["shape"], w["batch"], seed=0)
        mg = int(np.diff(batch.node_ptr).max())
        cm = runtime.CompiledModel.from_model(model, batch.num_graphs, batch.num_nodes, batch.num_edges, max_graph_nodes=mg)
        bd = tuple(torch.from_numpy(a).to(dev) for a in (batch.x, batch.coo, batch.node_ptr, batch.edge_ptr))
        alg_bytes, agg = measure_aggregate_roofline(cm, bd, w["hidden"], dev, regimes=("hbm",))
        fused = measure_fused_stack(cm, bd, (int(batch.x.shape[1]), w["hidden"], w["hidden"], len(w["pools"]))) \
            if w["conv"] == "gcn" and w["layers"] == 2 else None
        print(json.dumps({"roofline_only": True, "algorithmic_bytes_per_launch": alg_bytes, **agg["hbm"],
                          "fused_stack": fused}))
        return
    # rank-distinct synthetic batches (weak scaling: every GPU gets its own `batch` graphs per step)
    batches = [synthetic.make_batch(w["shape"], w["batch"], seed=1000 * rank + i) for i in range(args.batches)]
    maxn = max(b.num_nodes for b in batches)
    maxe = max(b.num_edges for b in batches)
    nstreams = max(1, args.streams)
    # promise on the largest graph (validated on the device by every graph prep): lets molecule-sized
    # graphs be staged whole in LDS (fused conv stack)
    max_graph = int(max(np.diff(b.node_ptr).max() for b in batches))
    cms = [runtime.CompiledModel.from_model(model, w["batch"], maxn, maxe, max_graph_nodes=max_graph)
           for _ in range(nstreams)]
    cm = cms[0]
    streams = [torch.cuda.Stream(device=dev) for _ in range(nstreams)] if nstreams > 1 else [torch.cuda.current_stream()]
    dev_batches = [tuple(torch.from_numpy(a).to(dev) for a in (b.x, b.coo, b.node_ptr, b.edge_ptr)) for b in batches]
    outs = [torch.empty(w["batch"], cm.out_dim, device=dev) for _ in batches]
    torch.cuda.synchronize()

    def step(i):
        # step i = one batched forward of batch i (mod the rotation) on stream i mod nstreams
        k = i % len(dev_batches)
        cms[i % nstreams].forward(*dev_batches[k], out=outs[k], stream=streams[i % nstreams])

    for i in range(args.warmup):
        step(i)
    for c, st in zip(cms, streams):
        c.check(stream=st)  # device-side batch validation (synchronises)

    def barrier():
        if world > 1:
            dist.barrier()

    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    graphs_done = float(args.steps * w["batch"])
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)  # RCCL over xGMI: 8 bytes, latency only
        elapsed = float(t.item())
        c = torch.tensor([graphs_done], device=dev, dtype=torch.float64)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        graphs_done = float(c.item())

    # opt-in math mode (NOT the headline): the wide update of the fused GCN stack as six bf16 MFMA products of
    # an exact 3-way split of both fp32 operands, fp32 accumulate; same steps, same batches
    split_rate = None
    if w["conv"] == "gcn" and w["layers"] == 2 and not args.no_roofline:
        torch.cuda.synchronize()
        runtime.set_option("math", 1)
        for i in range(args.warmup):
            step(i)
        torch.cuda.synchronize()
        barrier()
        ts = time.perf_counter()
        for i in range(args.steps):
            step(i)
        torch.cuda.synchronize()
        el2 = time.perf_counter() - ts
        runtime.set_option("math", 0)
        if world > 1:
            t2 = torch.tensor([el2], device=dev, dtype=torch.float64)
            dist.all_reduce(t2, op=dist.ReduceOp.MAX)
            el2 = float(t2.item())
        split_rate = graphs_done / el2
        split_ms = el2 / args.steps * 1e3

    # the prep-excluded rate (topology tables re-used; only features change)
    x0, coo0, np0, ep0 = dev_batches[0]
    cm.graph_prep(coo0, np0, ep0, int(x0.shape[0]))
    for _ in range(5):
        cm.forward_prepared(x0, out=outs[0])
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    nprep = max(args.steps // 2, 10)
    for _ in range(nprep):
        cm.forward_prepared(x0, out=outs[0])
    torch.cuda.synchronize()
    ms_noprep = (time.perf_counter() - t1) / nprep * 1e3

    result = {
        "metric": "graphs/sec whole-node (batched QM9, GCN d=128)" if args.workload == "c2"
        else f"graphs/sec whole-node ({args.workload})",
        "value": graphs_done / elapsed,
        "unit": "graphs/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": w["desc"], "graphs_per_step_per_gpu": w["batch"],
                   "nodes_per_batch": int(np.mean([b.num_nodes for b in batches])),
                   "edges_per_batch": int(np.mean([b.num_edges for b in batches])),
                   "parallelism": f"graph-sharded x{world}, no data-path collective",
                   "batches_in_flight_per_gpu": nstreams, "max_graph_nodes_promise": max_graph,
                   "csr_build_in_timed_region": True},
        "ms_per_step_prepared_topology": ms_noprep,
    }
    if split_rate is not None:
        result["opt_in_math_bf16x6"] = {
            "value": split_rate, "unit": "graphs/s", "ms_per_step": split_ms,
            "how": "GNNB_MATH=1 / gnnb_set_option(\"math\", 1): A1.W1^T of the fused stack as 6 v_mfma_f32_16x16x32_bf16 per 32-wide "
                   "k block on an exact hi/mid/lo bf16 split of both operands, fp32 accumulate; NOT used for `value`",
            "accuracy": "max |out - float64 evaluation| on this workload: 9.4e-8 (fp32-MFMA path 6.3e-8, scalar fp32 "
                        "reference 1.4e-7; tests/accuracy_math_modes.py)",
        }

    if rank == 0 and not args.no_roofline:
        alg_bytes, agg = measure_aggregate_roofline(cm, dev_batches[0], w["hidden"], dev)
        gather = {
            "kernel": "k_aggregate_shot<GCN> (gather-aggregate, width %d)" % w["hidden"],
            "bound": "hbm", "achieved": agg["hbm"]["gbps"], "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": agg["hbm"]["gbps"] / HBM_PEAK_GBPS, "traffic": pmc_traffic("aggregate"),
            "algorithmic_bytes_per_launch": alg_bytes, "us_per_launch": agg["hbm"]["us"],
            "regime": "inputs/outputs rotate over >256 MiB of distinct buffers (HBM-served); launches issued "
                      "back to back from C, HIP events on the launch stream",
            "in_pipeline_l3_resident": agg["l3_resident"],
            "copy_ceiling_same_bytes": copy_ceiling(batches[0].num_nodes, w["hidden"], dev),
        }
        fused = measure_fused_stack(cm, dev_batches[0], (int(batches[0].x.shape[1]), w["hidden"], w["hidden"],
                                                        len(w["pools"]))) if w["conv"] == "gcn" and w["layers"] == 2 else None
        if fused is not None:
            # the step runs the fused stack: that kernel dominates it and is bound by the fp32 matrix rate
            result["roofline"] = {
                "kernel": "k_gcn2_fused (2 GCN layers + pooling in one persistent kernel, graphs staged in LDS)",
                "bound": "mfma", "achieved": fused["tflops"], "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": fused["tflops"] / FP32_MFMA_PEAK_TFLOPS, "traffic": pmc_traffic("gcn2"),
                "algorithmic_flops_per_launch": fused["flops"], "algorithmic_hbm_bytes_per_launch": fused["alg_bytes"],
                "us_per_launch": fused["us"], "share_of_step": fused["us"] / (ms_noprep * 1e3),
                "note": "flops = 2 N (F0 h0 + h0 h1), the two dense updates on v_mfma_f32_16x16x4_f32; HIP events on "
                        "the launch stream, launches issued back to back from C on one prepared batch",
            }
            result["roofline_gather_aggregate"] = gather
        else:
            result["roofline"] = gather
        result["roofline_update"] = dict(kernel="k_linear_reg (fp32 MFMA 16x16x4), full-width layer update",
                                         bound="mfma", peak=FP32_MFMA_PEAK_TFLOPS, unit="TFLOP/s",
                                         **measure_update_mfma(w, batches[0].num_nodes, dev))
    if world > 1:
        dist.barrier()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(model, batches)

    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
