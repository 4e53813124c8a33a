#!/usr/bin/env python3
"""bench.py -- the hot path on N MI355X GPUs of one node: graphs/sec + gather-aggregate HBM GB/s.

A "step" is one batched forward (graph prep + conv stack + pooling + MLP head) over one batch of
synthetic molecule-shaped graphs that is already resident in HBM.  Default workload = BASELINE.json
configs[1]: 2-layer GCN d=128, QM9-shaped graphs, batch 4096 per GPU.  One process per GPU
(torch.distributed / RCCL); graphs are independent, so ranks shard batches with no data-path
collective and only the throughput counters are reduced (weak scaling).

Launching: `python bench.py --gpus N` starts its own N rank processes (the parent never touches the
GPU: it only spawns children with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, relays rank 0's JSON
line and exits non-zero if any rank fails); under `python -m torch.distributed.run --nproc-per-node N`
the ranks are the launcher's.  WORLD_SIZE != --gpus is an error.

The timed region (K steps between barrier + synchronize on both sides, MAX over ranks) is repeated
`--repeats` times; `value` is computed from the MEDIAN repeat, min / max are reported beside it.

Prints ONE JSON line on rank 0 (contract in the task prompt) with these extra objects:
  roofline      -- the kernel that dominates the timed step.  c2 and c3 (GCN / GIN stacks, graphs within the
                   max_graph_nodes promise) run the fused stack kernel k_gcn2_fused: bound = fp32 MFMA (c3 without
                   the promise: the K=N=128 update GEMM, k_linear_wlds); c4 / c5: the large-K segmented GEMM (k_linear_dma);
                   all bound = fp32 MFMA, flops / launch duration from HIP events on the launch stream.
  roofline_gather_aggregate -- the GCN gather-aggregate kernel at the full feature width (the
                   north-star kernel; every layer-by-layer model runs it): algorithmic bytes (SURVEY.md
                   8d) / measured launch duration (HIP events on the launch stream, rotating through
                   distinct buffers > 256 MiB so the Infinity Cache cannot serve the reads) against 8 TB/s.
  cpu_baseline  -- the reference's own C++ kernel library (oracle/_ref, compiled in place from
                   /root/reference; the C oracle port when it is absent) running the same model on a
                   bounded sample of the same graphs on ONE host core, rank 0 only, at every N; beside it
                   (SURVEY 8d protocol, reference experiments/build_base_benchmarks.py:158-239) the
                   package's own PyTorch forward per graph on one pinned core and batched on all cores.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
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
    # config 3 on batches with the heavy tail of the real ogbg-molhiv (sizes up to 222 nodes; SURVEY's recipe stops at ~46):
    # graphs beyond the fused stack's 57-node stage limit are ordered last and run layer by layer (large segment)
    "c3t": dict(desc="3-layer GIN d=128 + sum-pool, ogbg-molhiv-shaped graphs WITH the data set's heavy tail (log-normal sizes, "
                     "max 222 nodes), batch=4096 per GPU", conv="gin", shape="molhiv_tail", hidden=128, layers=3, pools=("add",),
                batch=4096, large_limit=57),
    "c4": dict(desc="3-layer PNA d=128, QM9-shaped graphs, batch=8192 per GPU", conv="pna", shape="qm9",
               hidden=128, layers=3, pools=("add", "mean", "max"), batch=8192),
    "c5": dict(desc="2-layer GraphSAGE d=256, ogbg-molhiv-shaped graphs, batch=8192 per GPU (65536 over 8)",
               conv="sage", shape="molhiv", hidden=256, layers=2, pools=("add", "mean", "max"), batch=8192),
    # the reference's ONE published benchmark model (experiments/build_base_benchmarks.py:61-81: 6 conv layers, hidden 128, out 64,
    # skip connections, add|mean|max pooling, MLP head of 4 hidden layers of 64; its figures/runtime_results_pivot.csv rows are
    # this model on QM9 and friends) for the four convs, on QM9-shaped batches of 4096
    "ref6_gcn": dict(desc="reference benchmark model: 6-layer GCN 128/64 + MLP 4x64, QM9-shaped graphs, batch=4096 per GPU", conv="gcn",
                     shape="qm9", hidden=128, out_dim=64, layers=6, pools=("add", "mean", "max"), mlp_layers=4, batch=4096),
    "ref6_gin": dict(desc="reference benchmark model: 6-layer GIN 128/64 + MLP 4x64, QM9-shaped graphs, batch=4096 per GPU", conv="gin",
                     shape="qm9", hidden=128, out_dim=64, layers=6, pools=("add", "mean", "max"), mlp_layers=4, batch=4096),
    "ref6_sage": dict(desc="reference benchmark model: 6-layer GraphSAGE 128/64 + MLP 4x64, QM9-shaped graphs, batch=4096 per GPU",
                      conv="sage", shape="qm9", hidden=128, out_dim=64, layers=6, pools=("add", "mean", "max"), mlp_layers=4, batch=4096),
    "ref6_pna": dict(desc="reference benchmark model: 6-layer PNA 128/64 + MLP 4x64, QM9-shaped graphs, batch=4096 per GPU", conv="pna",
                     shape="qm9", hidden=128, out_dim=64, layers=6, pools=("add", "mean", "max"), mlp_layers=4, batch=4096),
    # plumbing-sized workload of the launcher test (--dry-launch); never a bench line
    "tiny": dict(desc="launcher test: 2-layer GCN d=16, 64 QM9-shaped graphs", conv="gcn", shape="qm9",
                 hidden=16, layers=2, pools=("add", "mean", "max"), batch=64),
}


def build_model(w, seed=0):
    import torch
    import gnnbuilder_amd as gnnb
    from gnnbuilder_amd import synthetic

    torch.manual_seed(seed)
    convs = {"gcn": gnnb.GCNConv_GNNB, "gin": gnnb.GINConv_GNNB, "sage": gnnb.SAGEConv_GNNB, "pna": gnnb.PNAConv_GNNB}
    shp = synthetic.SHAPES[w["shape"]]
    out_dim = w.get("out_dim", w["hidden"])
    return gnnb.GNNModel(shp["f_in"], None, w["hidden"], w["layers"], out_dim, convs[w["conv"]], torch.nn.ReLU,
                         True, gnnb.GlobalPooling(list(w["pools"])),
                         gnnb.MLP(len(w["pools"]) * out_dim, shp["out"], 64, w.get("mlp_layers", 2)), None).eval()


# --------------------------------------------------------------------------------------- roofline legs
# aggregate kind each workload's layer-by-layer forward runs at its full width (GNNB_AGG_*), and how many [N, w] matrices it writes
WORKLOAD_AGG = {"gcn": ("gcn", 1), "gin": ("sum", 1), "sage": ("mean", 1), "pna": ("pna", 4)}


def measure_aggregate_roofline(cm, batch_dev, width, dev, iters=200, regimes=("hbm", "l3_resident"), kind="gcn", pna_self_term=True):
    """Gather-aggregate of `kind` at `width`, timed with HIP events on the launch stream.  Returns both
    the HBM regime (inputs/outputs rotate over > 256 MiB of distinct buffers) and the regime the
    kernel sees inside the pipeline (same buffers every launch: Infinity-Cache resident).
    Algorithmic bytes as SURVEY.md 8(d): every input row read once + every output row written once (PNA: four output
    matrices -- max, min, mean, std) + CSR + graph ptr; PNA's per-destination term q [N, w] (the x_i half of the pre-NN,
    which the kernel reads beside the gathered p rows) is reported separately as `extra_read_bytes` -- and only when the
    timed forward reads it: under a max_degree promise the degree-class form folds q into x's class weights and the
    aggregate runs WITHOUT a destination term (`pna_self_term` False)."""
    import torch

    x, coo, nptr, eptr = batch_dev
    N, E, B = int(x.shape[0]), int(coo.shape[0]), int(nptr.numel()) - 1
    cm.graph_prep(coo, nptr, eptr, N)
    k_out = 4 if kind == "pna" else 1
    alg_bytes = 4 * width * N + 4 * width * N * k_out + 4 * (N + 1) + 4 * E + 4 * (B + 1)
    per_pair = 4 * width * N * (1 + k_out)
    nbuf = max(2, int(np.ceil(320 * 2**20 / per_pair)) + 1)
    ins = [torch.rand(N, width, device=dev) * 2 - 1 for _ in range(nbuf)]
    outs = [torch.empty(N, width * k_out, device=dev) for _ in range(nbuf)]
    selfq = torch.rand(N, width, device=dev) * 2 - 1 if kind == "pna" and pna_self_term else None
    res = {}
    for regime, n in (("hbm", nbuf), ("l3_resident", 1)):
        if regime not in regimes:
            continue
        us = cm.aggregate_timed(kind, ins[:n], outs[:n], iters, self_term=selfq)  # launches issued from C
        res[regime] = dict(us=us, gbps=alg_bytes / (us * 1e-6) / 1e9)
    if selfq is not None:
        res["extra_read_bytes"] = 4 * width * N
    # calibration with the SAME launch shape and bytes: the library's own float4 row copy (no gather, no CSR)
    if kind != "pna":
        try:
            us = cm.aggregate_timed("copy", ins, outs, iters)
            res["copy_same_launch_shape"] = dict(us=us, gbps=per_pair / (us * 1e-6) / 1e9)
        except Exception:
            pass
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
    return dict(us=us, us_per_launch=us, tflops=flops / (us * 1e-6) / 1e12, achieved=flops / (us * 1e-6) / 1e12,
                algorithmic_flops_per_launch=flops,
                frac=flops / (us * 1e-6) / 1e12 / FP32_MFMA_PEAK_TFLOPS, shape=f"M={N} K={d} N={d}")


def measure_segmented_gemm(w, N, dev, iters=50, pna_classes=False):
    """The large-K update of the workload's full-width layer as the forward runs it: SAGE [mean | x].[Wl|Wr]^T
    (2 segments, K = 2d) or PNA [x | A | amp.A | att.A].Wpost^T (4 segments, K = 13d, per-row scalers) -- under a
    max_degree promise PNA's degree-class form [x | A].W_class^T (2 segments, K = 5d: the same kernel, the rows taken
    through the class permutation in the forward, in place here: the row-class mode is reachable through a model only).
    Launches come from Python here (the kernel is > 300 us, the launch cost is ~10 us and overlaps);
    HIP events on the launch stream."""
    import torch
    from gnnbuilder_amd import runtime

    d = w["hidden"]
    if w["conv"] == "sage":
        segs = [(torch.rand(N, d, device=dev) - 0.5, None), (torch.rand(N, d, device=dev) - 0.5, None)]
        K, what = 2 * d, "SAGE [mean|x].[Wl|Wr]^T, 2 segments"
    elif pna_classes:
        segs = [(torch.rand(N, d, device=dev) - 0.5, None), (torch.rand(N, 4 * d, device=dev) - 0.5, None)]
        K, what = 5 * d, ("PNA degree-class form [x|A].W_class^T, 2 segments (max_degree promise: the 13d-wide product with the "
                         "degree scalers folded into one weight matrix per in-degree; flops counted at K = 5d)")
    else:
        agg = torch.rand(N, 4 * d, device=dev) - 0.5
        amp, att = torch.rand(N, device=dev) + 0.5, torch.rand(N, device=dev) + 0.5
        segs = [(torch.rand(N, d, device=dev) - 0.5, None), (agg, None), (agg, amp), (agg, att)]
        K, what = 13 * d, "PNA [x|A|amp.A|att.A].Wpost^T, 4 segments with row scalers"
    wt = (torch.rand(d, K, device=dev) - 0.5) / K ** 0.5
    b = torch.rand(d, device=dev)
    y = torch.empty(N, d, device=dev)
    act = "relu" if w["conv"] == "sage" else "none"
    for _ in range(3):
        runtime.linear(segs, wt, b, None, act, out=y)
    tm = runtime.HipTimer()
    torch.cuda.synchronize()
    tm.start()
    for _ in range(iters):
        runtime.linear(segs, wt, b, None, act, out=y)
    tm.stop()
    us = tm.elapsed_ms() * 1e3 / iters
    flops = 2.0 * N * K * d
    return dict(us_per_launch=us, achieved=flops / (us * 1e-6) / 1e12, algorithmic_flops_per_launch=flops,
                frac=flops / (us * 1e-6) / 1e12 / FP32_MFMA_PEAK_TFLOPS, shape=f"M={N} K={K} N={d}", what=what)


def measure_pna_product_aggregate(cm, batch_dev, width, dev, iters=100):
    """k_pna_pagg (PNA under both promises: the source-half product p = x . Wb^T and its max | min | mean | std aggregate in one
    kernel, p on chip) on the prepared batch: HIP events on the launch stream, buffers rotating over > 256 MiB.  Algorithmic HBM
    bytes: x read once + the four output matrices written once + node records + CSR + tile tables; algorithmic flops 2 N w^2."""
    import torch
    from gnnbuilder_amd import runtime

    x, coo, nptr, eptr = batch_dev
    N, E, B = int(x.shape[0]), int(coo.shape[0]), int(nptr.numel()) - 1
    cm.graph_prep(coo, nptr, eptr, N)
    per_pair = 4 * width * N * 5
    nbuf = max(2, int(np.ceil(320 * 2**20 / per_pair)) + 1)
    ins = [torch.rand(N, width, device=dev) * 2 - 1 for _ in range(nbuf)]
    outs = [torch.empty(N, 4 * width, device=dev) for _ in range(nbuf)]
    wpre = (torch.rand(width, 2 * width, device=dev) - 0.5) / width ** 0.5
    wb = wpre[:, width:]
    try:
        for i in range(nbuf):
            cm.pna_product_aggregate(ins[i], wb, out=outs[i])
    except runtime.GnnbError:
        return None
    tm = runtime.HipTimer()
    torch.cuda.synchronize()
    tm.start()
    for i in range(iters):
        cm.pna_product_aggregate(ins[i % nbuf], wb, out=outs[i % nbuf])
    tm.stop()
    us = tm.elapsed_ms() * 1e3 / iters
    alg_bytes = 4 * width * N * 5 + 32 * N + 4 * E + 8 * (N // 8 + 1) + 4 * (B + 1)
    flops = 2.0 * N * width * width
    return {"kernel": "k_pna_pagg<%d> (PNA pre-NN source half + max|min|mean|std aggregate in one kernel, p on chip; width %d)" % (width // 16, width),
            "bound": "hbm", "achieved": alg_bytes / (us * 1e-6) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": alg_bytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS, "traffic": None, "algorithmic_bytes_per_launch": alg_bytes,
            "algorithmic_flops_per_launch": flops, "us_per_launch": us,
            "note": "what the timed step runs instead of a p GEMM + k_aggregate_ring<PNA> on its full-width layers (max_degree + "
                    "max_graph_nodes promises); bound by its 4w-wide output stores; launches from Python (the kernel is ~90 us, the launch "
                    "cost overlaps); HIP events on the launch stream"}


def pmc_traffic(kind, workload, alg_bytes):
    """HBM bytes per launch of kernel `kind` AT THIS WORKLOAD from the latest committed rocprofv3 PMC passes
    (FETCH_SIZE x2 on gfx950 + WRITE_SIZE; tools/profile_r.sh -> profiles/<round>_<workload>_<kind>_pmc.json).
    Counters cannot be read from inside this process, so this is a replay of a committed profile
    (`measured_in_this_run`: false), and only of one taken on the same launch: the profile's algorithmic
    byte count must agree with this run's within 2 %.  None when there is no such profile."""
    try:
        files = sorted((ROOT / "profiles").glob(f"*_{workload}_{kind}_pmc.json"))
        if not files and workload == "c2":  # rounds 1-2 named the config-2 profiles without the workload
            files = sorted(f for f in (ROOT / "profiles").glob(f"r0[12]*_{kind}_pmc.json"))
        d = json.loads(files[-1].read_text())
        ref = d.get("algorithmic_bytes_per_launch", d.get("algorithmic_hbm_bytes_per_launch"))
        if not ref or abs(ref - alg_bytes) > 0.02 * alg_bytes:
            return None
        t = d["hbm_traffic_bytes_per_launch"]
        return {"bytes_per_launch": t["total"], "read_bytes_fetch_size_x2": t["read_corrected_x2"],
                "write_bytes": t["write"], "over_algorithmic": t["total"] / alg_bytes, "measured_in_this_run": False,
                "source": f"committed profile profiles/{files[-1].name} (rocprofv3 --pmc, separate passes, same workload)"}
    except Exception:
        return None


def measure_fused_stack(cm, batch_dev, model_dims, iters=200, conv="gcn", layers=2, seg=None):
    """The fused conv stack + pooling kernel (GCN / GIN, two or more layers, graphs within the promise) on one prepared
    batch: launches issued back to back from C, HIP events on the launch stream.  None when the path is not eligible."""
    x, coo, nptr, eptr = batch_dev
    N, E, B = int(x.shape[0]), int(coo.shape[0]), int(nptr.numel()) - 1
    f0, h0, h1, npool = model_dims
    cm.graph_prep(coo, nptr, eptr, N)
    try:
        # five loops of `iters` launches, the MEDIAN loop reported (and all five listed): the first loop after a pause runs
        # while the clocks ramp (round 6: 40.0 us for the first loop against 37.2 for the next six on the same box)
        runs = [cm.gcn_stack_timed(x, iters) for _ in range(5)]
    except RuntimeError:
        return None
    us = float(np.median(runs))
    if seg is not None:  # the stack kernel runs on the graphs in front of the large segment
        B, N, E = seg
    # the dense updates (MFMA); aggregation flops not counted.  GCN: one linear per layer; GIN: two (hidden = out)
    if conv == "gin":
        flops = 2.0 * N * (f0 * h0 + h0 * h0 + (layers - 2) * 2 * h0 * h0 + h0 * h1 + h1 * h1)  # (= f0 h0 + (2L - 1) h0^2 when out = hidden)
    else:
        flops = 2.0 * N * (f0 * h0 + (layers - 2) * h0 * h0 + h0 * h1)
    # HBM bytes the kernel has to move: x + node records + dinv + tile/graph tables in, pooled out
    alg_bytes = 4 * N * f0 + 32 * N + 4 * N + 4 * (B + 1) + 4 * B * npool * h1
    return dict(us=us, tflops=flops / (us * 1e-6) / 1e12, flops=flops, alg_bytes=alg_bytes, loops_us=[round(float(r), 2) for r in runs])


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


# --------------------------------------------------------------------------------------- CPU baseline legs
def _cpu_model_name():
    try:
        return [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        return "unknown"


def cpu_baseline(model, batches, budget_s=12.0):
    """CPU legs beside the GPU figure, rank 0, bounded samples of the same workload.
    `value`: the reference CPU path on ONE core -- the per-graph loop of the reference testbench
    (model_tb.cpp.jinja:189-205) over the reference's own C++ kernel library (oracle/_ref).
    Beside it the package's own PyTorch forward timed as the reference times PyG-CPU
    (experiments/build_base_benchmarks.py:158-239): per graph on one pinned core, and batched on all cores.
    The all-core leg runs FIRST: threads created after the process is pinned inherit the one-core mask."""
    from oracle import oracle as O

    all_cpus = sorted(os.sched_getaffinity(0))
    extra = {}
    try:
        extra.update(torch_allcores_leg(model, batches[0], len(all_cpus)))
    except Exception as e:  # never lose the bench line to a baseline leg
        extra["torch_allcores_batched"] = {"error": repr(e)}
    try:
        os.sched_setaffinity(0, {all_cpus[0]})  # as build_base_benchmarks.py:188-189
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
    res = {"value": done / t_total, "unit": "graphs/s", "cores": 1, "kind": kind,
           "sample": f"{done} graphs of the timed workload (its batches in order, cycled), one graph per call, {t_total:.1f} s of CPU time; {what}",
           "host_cpu": _cpu_model_name(), "host_cores_available": len(all_cpus)}
    try:
        extra.update(torch_1core_leg(model, batches[0]))
    except Exception as e:
        extra["torch_1core_per_graph"] = {"error": repr(e)}
    try:
        os.sched_setaffinity(0, set(all_cpus))
    except Exception:
        pass
    res.update(extra)
    return res


def torch_allcores_leg(model, batch, ncpus, budget_s=6.0):
    """GNNModel.forward (the model definition; the PyG-equivalent op sequence without PyG) on one whole batch on the host's
    cores.  torch's default pool (one thread per logical CPU) oversubscribes this op mix (index_add / scatter_reduce over
    ~10^5 short rows are serial or memory-bound; only F.linear scales), so the thread count is SWEPT and the best is
    reported with its count, every point beside it."""
    import torch

    nthreads0 = torch.get_num_threads()
    x = torch.from_numpy(batch.x)
    ei = torch.from_numpy(np.ascontiguousarray(batch.coo.T).astype(np.int64))
    bv = torch.from_numpy(np.repeat(np.arange(batch.num_graphs), np.diff(batch.node_ptr)).astype(np.int64))
    counts = sorted({c for c in (1, 4, 8, 16, 32, 64, nthreads0) if 1 <= c <= max(ncpus, 1)})
    sweep = {}
    try:
        with torch.no_grad():
            for c in counts:
                torch.set_num_threads(c)
                model(x, ei, bv)
                trials, t_spent = [], 0.0
                while len(trials) < 3 or (t_spent < budget_s / len(counts) and len(trials) < 30):
                    t0 = time.perf_counter()
                    model(x, ei, bv)
                    dt = time.perf_counter() - t0
                    trials.append(dt)
                    t_spent += dt
                sweep[c] = batch.num_graphs / float(np.median(trials))
    finally:
        torch.set_num_threads(nthreads0)
    best = max(sweep, key=sweep.get)
    return {"torch_allcores_batched": {
        "value": sweep[best], "unit": "graphs/s", "cores": ncpus, "torch_threads": best,
        "thread_sweep_graphs_per_s": {str(k): v for k, v in sweep.items()}, "graphs_per_call": batch.num_graphs,
        "protocol": "GNNModel.forward on one whole batch (index_add / scatter_reduce / F.linear), torch.set_num_threads swept, "
                    "median of the trials per count, best count reported",
        "note": "the batched PyTorch op mix does not scale with threads: the scatter / index_add passes over ~10^5 short rows "
                "are single-threaded or memory-bound on the host and only the dense F.linear parallelises; the 1-core "
                "reference-library figure stays `cpu_baseline.value`"}}


def torch_1core_leg(model, batch, budget_s=6.0):
    """One graph per call on ONE pinned core: torch.utils.benchmark.Timer(...).timeit(5) per graph, mean over the
    graphs (reference experiments/build_base_benchmarks.py:188-208, 221)."""
    import torch
    from torch.utils import benchmark

    nthreads0 = torch.get_num_threads()
    torch.set_num_threads(1)
    times, t_spent, g = [], 0.0, 0
    try:
        with torch.no_grad():
            while t_spent < budget_s and g < batch.num_graphs:
                xg, cg = batch.graph(g)
                x = torch.from_numpy(np.ascontiguousarray(xg))
                ei = torch.from_numpy(np.ascontiguousarray(cg.T).astype(np.int64))
                t0 = time.perf_counter()
                m = benchmark.Timer(stmt="model(x, ei)", globals={"model": model, "x": x, "ei": ei}, num_threads=1).timeit(5)
                t_spent += time.perf_counter() - t0
                times.append(m.mean)
                g += 1
    finally:
        torch.set_num_threads(nthreads0)
    return {"torch_1core_per_graph": {
        "value": 1.0 / float(np.mean(times)), "unit": "graphs/s", "cores": 1, "graphs_sampled": len(times),
        "protocol": "GNNModel.forward per graph (bs=1), torch.set_num_threads(1) + sched_setaffinity to one core, "
                    "torch.utils.benchmark.Timer.timeit(5).mean per graph, mean over graphs "
                    "(reference experiments/build_base_benchmarks.py:188-208)"}}


# --------------------------------------------------------------------------------------- the timed pipeline
def workload_promises(w, batches, segs):
    """The promises the timed forward is set up with (validated on the device by every graph prep): the largest graph of
    the batches in front of their large segments, and -- PNA -- the largest in-degree when the degree classes cover it."""
    max_graph = int(max(np.diff(b.node_ptr)[:(sg[0] if sg else b.num_graphs)].max() for b, sg in zip(batches, segs)))
    max_degree = 0
    if w["conv"] == "pna" and not os.environ.get("GNNB_BENCH_NO_DEGREE_PROMISE"):
        max_degree = int(max((np.bincount(b.coo[:, 1]).max() if b.num_edges else 0) for b in batches))
        if max_degree > 15:
            max_degree = 0
    return max_graph, max_degree


class Pipeline:
    """`nstreams` batches in flight on one GPU: one workspace + one HIP stream each; step i = one batched forward (graph
    prep included) of batch i (mod the rotation) on stream i mod nstreams."""

    def __init__(self, model, batches, segs, nstreams, dev, max_graph, max_degree):
        import torch
        from gnnbuilder_amd import runtime

        self.batches, self.segs, self.nstreams = batches, segs, nstreams
        maxn, maxe, maxb = (max(getattr(b, a) for b in batches) for a in ("num_nodes", "num_edges", "num_graphs"))
        self.cms = [runtime.CompiledModel.from_model(model, maxb, maxn, maxe, max_graph_nodes=max_graph) for _ in range(nstreams)]
        if max_degree:
            for c in self.cms:
                c.set_max_degree(max_degree)
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(nstreams)] if nstreams > 1 else [torch.cuda.current_stream()]
        self.dev_batches = [tuple(torch.from_numpy(a).to(dev) for a in (b.x, b.coo, b.node_ptr, b.edge_ptr)) for b in batches]
        self.outs = [torch.empty(b.num_graphs, self.cms[0].out_dim, device=dev) for b in batches]
        torch.cuda.synchronize()

    def enable_prep_next(self, model, max_graph, max_degree):
        """A second workspace per stream: step_prep_next alternates between the two (gnnb_forward_prepared_prep_next)."""
        from gnnbuilder_amd import runtime

        maxn, maxe, maxb = (max(getattr(b, a) for b in self.batches) for a in ("num_nodes", "num_edges", "num_graphs"))
        self.alt = [runtime.CompiledModel.from_model(model, maxb, maxn, maxe, max_graph_nodes=max_graph) for _ in range(self.nstreams)]
        if max_degree:
            for c in self.alt:
                c.set_max_degree(max_degree)

    def step_prep_next(self, i, steps):
        """Step i of a region of `steps` steps, software-pipelined per stream: the forward of batch i and the graph prep of the
        stream's NEXT batch (i + nstreams) in one call -- that prep runs as extra workgroups of the forward's readout kernel.  Every
        batch of the region is prepared exactly once and forwarded exactly once INSIDE the region: the first batch of each
        stream by a graph-prep launch of its own, the last one's call preps nothing."""
        S, nb = self.nstreams, len(self.dev_batches)
        self.alt_used = True
        i %= steps  # (a loop longer than the region starts the next region)
        j, k = i % S, i % nb
        pair = (self.cms[j], self.alt[j])
        cur, nxt = pair[(i // S) & 1], pair[((i // S) + 1) & 1]
        x, coo, nptr, eptr = self.dev_batches[k]
        if i < S:
            cur.graph_prep(coo, nptr, eptr, int(x.shape[0]), stream=self.streams[j])
        if i + S < steps:
            _, coo2, nptr2, eptr2 = self.dev_batches[(i + S) % nb]
            n2 = int(self.dev_batches[(i + S) % nb][0].shape[0])
            cur.forward_prepared_prep_next(x, nxt, coo2, nptr2, eptr2, n2, out=self.outs[k], stream=self.streams[j])
        else:
            cur.forward_prepared(x, out=self.outs[k], stream=self.streams[j])

    def step(self, i):
        k, j = i % len(self.dev_batches), i % self.nstreams
        if self.segs[k] is not None:
            self.cms[j].set_large_segment(*self.segs[k])
        self.cms[j].forward(*self.dev_batches[k], out=self.outs[k], stream=self.streams[j])

    def check(self):
        for c, st in zip(self.cms, self.streams):
            c.check(stream=st)  # device-side batch validation (synchronises)
        if getattr(self, "alt_used", False):
            for c, st in zip(self.alt, self.streams):
                c.check(stream=st)

    def prepare_topology(self):
        """workspace j keeps batch j prepared (tables + classes): the prep-EXCLUDED rate runs on the same streams"""
        for j in range(self.nstreams):
            k = j % len(self.dev_batches)
            x, coo, nptr, eptr = self.dev_batches[k]
            if self.segs[k] is not None:
                self.cms[j].set_large_segment(*self.segs[k])
            self.cms[j].graph_prep(coo, nptr, eptr, int(x.shape[0]), stream=self.streams[j])

    def step_prepared(self, i):
        j = i % self.nstreams
        k = j % len(self.dev_batches)
        self.cms[j].forward_prepared(self.dev_batches[k][0], out=self.outs[k], stream=self.streams[j])


def other_config_leg(name, dev, nstreams, budget_s=1.0):
    """One of BASELINE's other single-GPU configs, timed the way `value` is (same pipeline, CSR build included, median of
    the repeats) but briefly: three batches, no CPU leg.  Runs AFTER `value` is computed: the C2 timed region is untouched."""
    import torch
    from gnnbuilder_amd import synthetic

    w = WORKLOADS[name]
    model = build_model(w)
    batches = [synthetic.make_batch(w["shape"], w["batch"], seed=5000 + i) for i in range(max(3, nstreams))]
    segs = [None] * len(batches)
    max_graph, max_degree = workload_promises(w, batches, segs)
    pipe = Pipeline(model, batches, segs, nstreams, dev, max_graph, max_degree)
    for i in range(2 * len(batches)):
        pipe.step(i)
    pipe.check()
    # size the region from one probe region, then five repeats of it within the budget
    def region(k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(k):
            pipe.step(i)
        torch.cuda.synchronize()
        return time.perf_counter() - t0
    t_probe = region(6) / 6
    k = int(min(200, max(6, budget_s / 5 / max(t_probe, 1e-6))))
    times = [region(k) for _ in range(5)]
    el = float(np.median(times))
    graphs = float(sum(batches[i % len(batches)].num_graphs for i in range(k)))
    cm = pipe.cms[0]
    path = cm.last_path()
    res = {"workload": w["desc"], "name": name, "value": graphs / el, "unit": "graphs/s", "ms_per_step": el / k * 1e3, "steps": k, "repeats": 5,
           "value_min": graphs / max(times), "value_max": graphs / min(times),
           "path": path, "max_graph_nodes_promise": max_graph, "max_degree_promise": max_degree or None,
           "batches_in_flight": nstreams, "csr_build_in_timed_region": True}
    if w["conv"] in ("sage", "pna", "gin"):
        # the opt-in math modes on the GEMMs / wide products that dominate these (never `value`: see opt_in_math_* above)
        from gnnbuilder_amd import runtime
        notes = {1: ("opt_in_math_bf16x6", "gnnb_set_option(\"math\", 1): the layer-by-layer GEMMs as 6 bf16 MFMA products on an exact hi/mid/lo split of "
                                            "both operands, fp32 accumulate (fp32-equivalent: per GEMM no worse than 2x the fp32-MFMA kernel's error + "
                                            "1e-7 against a float64 product); NOT `value`"),
                 3: ("opt_in_math_f16x3_reduced_precision", "gnnb_set_option(\"math\", 3): the same GEMMs as 3 fp16 MFMA products on round-to-nearest hi + mid "
                                                             "fp16 pieces of both operands, fp32 accumulate: ~22 significant bits per product, fp16's RANGE "
                                                             "(values < 65504, pieces below 6e-8 lost); REDUCED precision, NOT `value`")}
        if w["conv"] == "gin":
            notes = {3: (notes[3][0], notes[3][1].replace("the same GEMMs", "the wide products of the GIN stack kernel"))}  # (math 1 leaves the GIN stack in fp32)
        for mode, (key, note) in notes.items():
            runtime.set_option("math", mode)
            try:
                for i in range(len(batches)):
                    pipe.step(i)
                pipe.check()
                t2 = [region(k) for _ in range(3)]
            finally:
                runtime.set_option("math", 0)
            e2 = float(np.median(t2))
            res[key] = {"value": graphs / e2, "unit": "graphs/s", "ms_per_step": e2 / k * 1e3, "repeats": 3, "note": note}
    if w["conv"] in ("gcn", "gin"):
        fused = measure_fused_stack(cm, pipe.dev_batches[0], (int(batches[0].x.shape[1]), w["hidden"], w.get("out_dim", w["hidden"]), len(w["pools"])),
                                    iters=50, conv=w["conv"], layers=w["layers"])
        if fused is not None:
            res["roofline"] = {"kernel": {"stack_zf": "k_gcn2_zf", "stack": "k_gcn2_fused<%s>" % w["conv"].upper()}.get(cm.last_path(), cm.last_path()),
                               "bound": "mfma", "us_per_launch": fused["us"], "achieved": fused["tflops"], "unit": "TFLOP/s",
                               "frac": fused["tflops"] / FP32_MFMA_PEAK_TFLOPS}
    else:
        g = measure_segmented_gemm(w, batches[0].num_nodes, dev, iters=10, pna_classes=bool(max_degree))
        res["roofline"] = {"kernel": "k_linear_dma (%s)" % g["what"], "bound": "mfma", "us_per_launch": g["us_per_launch"],
                           "achieved": g["achieved"], "unit": "TFLOP/s", "frac": g["frac"], "shape": g["shape"]}
    del pipe
    torch.cuda.synchronize()
    return res


# --------------------------------------------------------------------------------------- launcher
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(n, argv, poll_s=0.2, overall_timeout_s=None):
    """Parent of `python bench.py --gpus N` (no launcher around it): start N rank processes of this same
    script and relay rank 0's JSON line.  The parent makes NO GPU call (no torch.cuda, no HIP library
    load) and never exec()s.  It polls EVERY child: the first rank that exits non-zero (bad device, import
    error, a crash in a kernel) makes the parent terminate the others -- they would otherwise sit in
    init_process_group / barrier until the collective timeout -- and exit 1 within seconds.  An overall
    timeout (BENCH_LAUNCH_TIMEOUT_S, default 1800 s) bounds a hang of all ranks."""
    import tempfile
    import threading

    if overall_timeout_s is None:
        overall_timeout_s = float(os.environ.get("BENCH_LAUNCH_TIMEOUT_S", "1800"))
    # The rendezvous port: bind to port 0, read the number, close.  Another process could take it before rank 0
    # binds (a benign race: the ranks then fail fast with "address in use" and so does the parent).
    port = _free_port()
    procs = []
    out0 = tempfile.TemporaryFile(mode="w+")  # rank 0's stdout: a file, so that no pipe can fill up while we poll
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve())] + argv, env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL))

    def stop_all():
        for p in procs:  # the children we started, by handle: never by pattern
            if p.poll() is None:
                p.terminate()
        t_end = time.monotonic() + 5.0
        for p in procs:
            try:
                p.wait(timeout=max(0.0, t_end - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()

    t0 = time.monotonic()
    failed = None
    while True:
        rcs = [p.poll() for p in procs]
        bad = [(r, rc) for r, rc in enumerate(rcs) if rc not in (None, 0)]
        if bad:
            failed = f"ranks failed (rank, exit code): {bad}; the other ranks were terminated"
            break
        if all(rc == 0 for rc in rcs):
            break
        if time.monotonic() - t0 > overall_timeout_s:
            failed = f"no result after {overall_timeout_s:.0f} s: ranks terminated"
            break
        time.sleep(poll_s)
    if failed:
        stop_all()
    out0.seek(0)
    for l in out0.read().splitlines():
        # rank 0's JSON line goes to stdout; anything else a library printed there (gloo / RCCL banners) to stderr
        try:
            json.loads(l)
            if not failed:
                print(l)
        except ValueError:
            if l.strip():
                print(l, file=sys.stderr)
    sys.stdout.flush()
    if failed:
        print(f"bench.py: {failed}", file=sys.stderr)
        sys.exit(1)
    sys.exit(0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--repeats", type=int, default=11,
                    help="the timed region of K steps is run this many times; value = the median repeat (the first two or "
                         "three regions after start-up run ~5 %% slower -- clocks, caches -- and a 20-step region is only a "
                         "millisecond, so five repeats put the median ON that ramp; every repeat is in the line)")
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--batches", type=int, default=8, help="distinct synthetic batches rotated per rank")
    ap.add_argument("--streams", type=int, default=3,
                    help="batches in flight per GPU: each on its own HIP stream and workspace, so the MFMA-bound "
                         "update of one batch overlaps the HBM-bound gather / readout of the next")
    ap.add_argument("--shard", default="rank-batches", choices=("rank-batches", "one-batch"),
                    help="rank-batches: every rank draws its own batches of `batch` graphs; one-batch: every global "
                         "batch of batch x N graphs is cut into contiguous node-balanced ranges "
                         "(batching.shard_bounds), one per rank")
    ap.add_argument("--prep-next", default="0", choices=("0", "1"),
                    help="1: 2-layer GCN workloads on molecule-sized graphs run the software-pipelined step as `value` (gnnb_forward_prepared_prep_next: "
                         "forward of batch i + graph prep of the stream's next batch in one call, the prep as extra workgroups of the readout kernel); "
                         "0 (default): gnnb_forward_batched per step, the pipelined form reported beside it")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the brief c3 / c4 / c5 legs of the default (c2, one GPU) line")
    ap.add_argument("--roofline-only", action="store_true",
                    help="only the HBM-regime gather-aggregate loop (for a rocprofv3 run whose kernel average is that loop)")
    ap.add_argument("--dry-launch", action="store_true",
                    help="launcher / sharding / reduction plumbing on CPU (gloo): the step runs the PyTorch model "
                         "definition instead of the HIP path; the line is marked dry_launch and is not a measurement")
    args = ap.parse_args()

    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        self_launch(args.gpus, sys.argv[1:])  # does not return
    rank = int(os.environ.get("RANK", "0"))
    world = int(world_env or "1")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: WORLD_SIZE={world} but --gpus {args.gpus}: launch one rank per GPU "
              f"(`python bench.py --gpus N` starts its own ranks)", file=sys.stderr)
        sys.exit(2)

    import torch
    import torch.distributed as dist

    from gnnbuilder_amd import runtime, synthetic
    from gnnbuilder_amd.batching import shard_bounds

    dry = args.dry_launch
    if os.environ.get("BENCH_TEST_FAIL_RANK") == str(rank):  # launcher test hook: this rank dies at start-up
        print(f"bench.py: rank {rank} exiting 3 (BENCH_TEST_FAIL_RANK)", file=sys.stderr)
        sys.exit(3)
    if dry:
        dev = torch.device("cpu")
        torch.set_num_threads(1)
    else:
        # (device_count() does not initialise the GPU on this image: safe before anything else)
        ndev = torch.cuda.device_count()
        if ndev <= local_rank:
            print(f"bench.py: rank {rank}: LOCAL_RANK {local_rank} but only {ndev} GPU(s) visible", file=sys.stderr)
            sys.exit(4)
        runtime.load_library(require_gpu=True)  # no fallback: fail loudly without the HIP path
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if dry:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    def sync():
        if not dry:
            torch.cuda.synchronize()

    w = WORKLOADS[args.workload]
    model = build_model(w)
    if args.roofline_only:
        batch = synthetic.make_batch(w["shape"], w["batch"], seed=0)
        seg = None
        if w.get("large_limit"):
            from gnnbuilder_amd.batching import order_large_last
            batch, _, seg = order_large_last(batch, w["large_limit"])
        mg = int(np.diff(batch.node_ptr)[:(seg[0] if seg else batch.num_graphs)].max())
        cm = runtime.CompiledModel.from_model(model, batch.num_graphs, batch.num_nodes, batch.num_edges, max_graph_nodes=mg)
        if seg:
            cm.set_large_segment(*seg)
        _, md = workload_promises(w, [batch], [seg])
        if md:
            cm.set_max_degree(md)
        bd = tuple(torch.from_numpy(a).to(dev) for a in (batch.x, batch.coo, batch.node_ptr, batch.edge_ptr))
        # (round 6) the timed step's own kernels, each ALONE on the chip: whole forwards of one prepared batch back to back on
        # ONE stream -- under `rocprofv3 --kernel-trace --stats` every kernel of the step (the row-class / pooling GEMMs,
        # k_pna_pagg, k_pna_first, k_sage_first_mean, the readout) gets a per-launch average that is not stretched by the other
        # batches of the three-stream pipeline: what `other_configs[].roofline.frac` is recomputed from
        cm.graph_prep(bd[1], bd[2], bd[3], int(bd[0].shape[0]))
        outb = torch.empty(batch.num_graphs, cm.out_dim, device=dev)
        for _ in range(5):
            cm.forward_prepared(bd[0], out=outb)
        cm.check()
        tmf = runtime.HipTimer()
        torch.cuda.synchronize()
        tmf.start()
        nser = 40
        for _ in range(nser):
            cm.forward_prepared(bd[0], out=outb)
        tmf.stop()
        serial = {"us_per_forward_prepared_one_stream": tmf.elapsed_ms() * 1e3 / nser, "forwards": nser, "path": cm.last_path()}
        if w["conv"] in ("sage", "pna"):
            serial["dominant_gemm_stand_alone"] = measure_segmented_gemm(w, batch.num_nodes, dev, pna_classes=bool(md))
        if w["conv"] == "pna" and md:
            serial["pna_product_aggregate"] = measure_pna_product_aggregate(cm, bd, w["hidden"], dev)
        alg_bytes, agg = measure_aggregate_roofline(cm, bd, w["hidden"], dev, regimes=("hbm",))
        wkind = WORKLOAD_AGG[w["conv"]][0]
        own = None
        if wkind != "gcn":
            ab, ag = measure_aggregate_roofline(cm, bd, w["hidden"], dev, regimes=("hbm",), kind=wkind)
            own = {"kind": wkind, "algorithmic_bytes_per_launch": ab, **ag["hbm"]}
        fused = measure_fused_stack(cm, bd, (int(batch.x.shape[1]), w["hidden"], w.get("out_dim", w["hidden"]), len(w["pools"])),
                                    conv=w["conv"], layers=w["layers"], seg=seg) if w["conv"] in ("gcn", "gin") and w["layers"] >= 2 else None
        print(json.dumps({"roofline_only": True, "workload": args.workload, "algorithmic_bytes_per_launch": alg_bytes, **agg["hbm"],
                          "copy_same_launch_shape": agg.get("copy_same_launch_shape"), "workload_kind": own, "fused_stack": fused,
                          "stack_path": cm.last_path() if fused else None, "serial_forward": serial}))
        return

    if args.shard == "one-batch":
        # every rank draws the SAME global batches (batch x world graphs) and keeps its node-balanced range
        batches = []
        for i in range(args.batches):
            glob = synthetic.make_batch(w["shape"], w["batch"] * world, seed=77000 + i)
            g0, g1 = shard_bounds(glob.node_ptr, world)[rank]
            batches.append(glob.slice(g0, g1))
            del glob
    else:
        # rank-distinct synthetic batches (weak scaling: every GPU gets its own `batch` graphs per step)
        batches = [synthetic.make_batch(w["shape"], w["batch"], seed=1000 * rank + i) for i in range(args.batches)]
    # graphs beyond the stack kernels' stage limit go last in every batch and are named as its large segment
    segs = [None] * len(batches)
    if w.get("large_limit"):
        from gnnbuilder_amd.batching import order_large_last
        for i, b in enumerate(batches):
            batches[i], _, segs[i] = order_large_last(b, w["large_limit"])
    nstreams = max(1, args.streams)
    # promise on the largest graph (validated on the device by every graph prep): lets molecule-sized
    # graphs be staged whole in LDS (fused conv stack)
    # PNA: promise on the largest in-degree too (a bound where the reference's degree_guess is a hint; validated on the device): molecules stay far
    # below the 15 up to which the degree-class form of the post-NN product applies (gnnb_workspace_set_max_degree)
    max_graph, max_degree = workload_promises(w, batches, segs)
    prep_next = can_prep_next = False
    if dry:
        import torch.nn  # noqa: F401

        dev_batches = [(torch.from_numpy(b.x), torch.from_numpy(np.ascontiguousarray(b.coo.T).astype(np.int64)),
                        torch.from_numpy(np.repeat(np.arange(b.num_graphs), np.diff(b.node_ptr)).astype(np.int64)))
                       for b in batches]
        outs = [None] * len(batches)

        def step(i):
            k = i % len(dev_batches)
            with torch.no_grad():
                outs[k] = model(*dev_batches[k])
        pipe, cm = None, None
    else:
        pipe = Pipeline(model, batches, segs, nstreams, dev, max_graph, max_degree)
        cm, dev_batches, outs, step = pipe.cms[0], pipe.dev_batches, pipe.outs, pipe.step
        # software-pipelined graph prep (gnnb_forward_prepared_prep_next, ABI 104): where the next batch's prep can run inside the readout
        # kernel -- 2-layer GCN, molecule-sized graphs (promise <= 64 nodes), no large segment.  Same work per step (one graph prep +
        # one forward, every batch of the timed region prepared and forwarded inside it).  Measured: a gain on ONE stream
        # (52.0 vs 55.7 us per forward), none with three batches in flight (41.8 vs 41.2 us per step) -- so `value` stays on
        # gnnb_forward_batched and this form is reported beside it (--prep-next 1 swaps them)
        can_prep_next = (w["conv"] == "gcn" and w["layers"] == 2 and bool(max_graph) and max_graph <= 64 and
                         all(sg is None for sg in segs) and args.steps > nstreams)
        prep_next = can_prep_next and args.prep_next == "1"
        can_prep_next = can_prep_next and (prep_next or not args.no_roofline)  # (the other form is a side leg of the full line)
        if can_prep_next:
            pipe.enable_prep_next(model, max_graph, max_degree)

            def pipelined_step(i):
                pipe.step_prep_next(i, args.steps)
            plain_step = pipe.step
            if prep_next:
                step = pipelined_step

    for i in range(args.warmup):
        step(i)
    if pipe is not None:
        pipe.check()  # device-side batch validation (synchronises)

    def barrier():
        if world > 1:
            dist.barrier()

    def reduce_max(v):
        if world == 1:
            return v
        t = torch.tensor([v], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)  # RCCL over xGMI: 8 bytes, latency only
        return float(t.item())

    def timed_region(step_fn=None):
        step_fn = step_fn or step
        sync()
        barrier()
        sync()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step_fn(i)
        sync()
        barrier()
        sync()
        return reduce_max(time.perf_counter() - t0)

    repeats = max(1, args.repeats)
    times = [timed_region() for _ in range(repeats)]
    elapsed = float(np.median(times))

    graphs_done = float(sum(batches[i % len(batches)].num_graphs for i in range(args.steps)))
    # the same region in the other form of the step (see can_prep_next above)
    other_form = None
    if not dry and can_prep_next and not args.no_roofline:
        other = plain_step if prep_next else pipelined_step
        for i in range(args.warmup):
            other(i)
        other_form = float(np.median([timed_region(other) for _ in range(min(repeats, 5))]))
        for i in range(args.warmup):
            step(i)
    rccl_ranks = 1
    if world > 1:
        c = torch.tensor([graphs_done, 1.0], device=dev, dtype=torch.float64)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        graphs_done, rccl_ranks = float(c[0].item()), int(round(float(c[1].item())))

    # opt-in math mode (NOT the headline): the wide update of the fused GCN stack as six bf16 MFMA products of
    # an exact 3-way split of both fp32 operands, fp32 accumulate; same steps, same batches
    split_rate = None
    if not dry and not args.no_roofline:
        runtime.set_option("math", 1)
        for i in range(args.warmup):
            step(i)
        el2 = float(np.median([timed_region() for _ in range(repeats)]))  # (same statistic as `value`)
        runtime.set_option("math", 0)
        split_rate = graphs_done / el2
        split_ms = el2 / args.steps * 1e3
    # opt-in REDUCED-precision mode (never `value`): k_gcn2_zf's wide update on hi + mid bf16 pieces, three products ("bf16x3")
    reduced = None
    if not dry and not args.no_roofline and (w["conv"] in ("sage", "pna", "gin") or (w["conv"] == "gcn" and w["layers"] > 2)):
        reduced = {"how": "gnnb_set_option(\"math\", 3) / GNNB_MATH=3: the layer-by-layer LDS-DMA GEMMs / the wide products of the GIN and deep-GCN "
                          "stack kernel as 3 fp16 MFMA products (hi.hi + hi.mid + mid.hi) on "
                          "round-to-nearest hi + mid fp16 pieces of both operands, fp32 accumulate: ~22 significant bits per product, fp16's RANGE "
                          "(|values| < 65504, pieces below 6e-8 lost).  REDUCED precision, an accuracy-vs-throughput study mode (SURVEY 8 f-4); "
                          "NOT used for `value`",
                   "accuracy": "whole models against the oracle: within 2e-5 of the output scale (test_layer_by_layer_models_in_the_f16x3_math_mode, "
                               "test_gin_and_deep_gcn_stacks_in_the_f16x3_math_mode, test_full_size_configs_in_the_opt_in_math_modes); "
                               "per GEMM against a float64 product: < 4e-6 at K = 416 .. 832 (test_large_k_gemm_bf16x6_math_is_fp32_equivalent)"}
        runtime.set_option("math", 3)
        for i in range(args.warmup):
            step(i)
        el3 = float(np.median([timed_region() for _ in range(repeats)]))
        runtime.set_option("math", 0)
        reduced["f16x3"] = {"value": graphs_done / el3, "unit": "graphs/s", "ms_per_step": el3 / args.steps * 1e3}
    if not dry and not args.no_roofline and w["conv"] == "gcn" and w["layers"] == 2:
        reduced = {"how": "gnnb_set_option(\"math\", 2 | 3) / GNNB_MATH: H.W1^T of k_gcn2_zf as 3 MFMA products (hi.hi + hi.mid + mid.hi) on round-to-nearest "
                          "hi + mid 16-bit pieces of both operands, fp32 accumulate.  2 = bf16 pieces: ~18 significant bits per product, fp32's "
                          "range; 3 = fp16 pieces: ~22 bits, fp16's RANGE (|values| < 65504, pieces below 6e-8 lost).  REDUCED precision, "
                          "accuracy-vs-throughput study modes (SURVEY 8 f-4); NOT used for `value`",
                   "accuracy": "max |out - float64 evaluation| on BASELINE config 2 (outputs |max| 0.23): bf16x3 6.3e-7, f16x3 7.5e-8; fp32-MFMA path "
                               "7.1e-8, scalar fp32 reference 1.4e-7 (tests/accuracy_math_modes.py)"}
        for mode, key in ((2, "bf16x3"), (3, "f16x3")):
            runtime.set_option("math", mode)
            for i in range(args.warmup):
                step(i)
            el3 = float(np.median([timed_region() for _ in range(repeats)]))
            runtime.set_option("math", 0)
            reduced[key] = {"value": graphs_done / el3, "unit": "graphs/s", "ms_per_step": el3 / args.steps * 1e3}

    # the prep-EXCLUDED rate (both side by side): topology tables re-used, only features change.  Same pipeline as
    # `value` -- the same streams, one prepared batch per workspace, the same K-step region and statistic -- so the two are
    # comparable; the one-stream figure (a single workspace, forwards back to back) is kept beside it
    ms_noprep = ms_noprep_1s = ms_noprep_1s_lat = None
    if not dry:
        pipe.prepare_topology()
        for i in range(max(args.warmup, pipe.nstreams)):
            pipe.step_prepared(i)
        ms_noprep = float(np.median([timed_region(pipe.step_prepared) for _ in range(repeats)])) / args.steps * 1e3
        x0, coo0, np0, ep0 = dev_batches[0]
        for _ in range(5):
            cm.forward_prepared(x0, out=outs[0])
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        nprep = max(args.steps // 2, 10)
        for _ in range(nprep):
            cm.forward_prepared(x0, out=outs[0])
        torch.cuda.synchronize()
        ms_noprep_1s = (time.perf_counter() - t1) / nprep * 1e3
        # ... and with the library's settings for ONE batch in flight (head_pairs = 0: the readout's four-operand form; prep_group = 1 is
        # not involved here): the defaults trade ~2 us of this figure for the pipeline's step
        runtime.set_option("head_pairs", 0)
        try:
            for _ in range(5):
                cm.forward_prepared(x0, out=outs[0])
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(nprep):
                cm.forward_prepared(x0, out=outs[0])
            torch.cuda.synchronize()
            ms_noprep_1s_lat = (time.perf_counter() - t1) / nprep * 1e3
        finally:
            runtime.set_option("head_pairs", 1)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()  # every collective is done: rank 0 finishes the single-GPU legs alone
    if rank != 0:
        return

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
                   "shard": args.shard, "rccl_ranks": rccl_ranks,
                   "batches_in_flight_per_gpu": nstreams, "max_graph_nodes_promise": max_graph,
                   "csr_build_in_timed_region": True,
                   "graph_prep": ("software-pipelined per stream over two alternating workspaces (gnnb_forward_prepared_prep_next): step i = forward of "
                                  "batch i + CSR build of the stream's next batch, the build running as extra workgroups of the readout kernel (k_head_small, GUEST); every "
                                  "batch of the timed region is built once and forwarded once inside the region (the first per stream by a "
                                  "k_graph_prep launch)") if prep_next else "k_graph_prep launch in front of every forward (gnnb_forward_batched)",
                   "max_degree_promise": max_degree or None,
                   "path": None if dry else cm.last_path()},
        "repeats": {"n": repeats, "statistic": "median", "steps_per_repeat": args.steps,
                    "value_min": graphs_done / max(times), "value_max": graphs_done / min(times),
                    "ms_per_step_all": [t / args.steps * 1e3 for t in times]},
        "ms_per_step_prepared_topology": ms_noprep,
        "prepared_topology": None if ms_noprep is None else {
            "ms_per_step": ms_noprep, "value": graphs_done / (ms_noprep * 1e-3 * args.steps), "unit": "graphs/s",
            "batches_in_flight_per_gpu": nstreams, "ms_per_step_single_stream": ms_noprep_1s,
            "ms_per_step_single_stream_head_pairs_0": ms_noprep_1s_lat,
            "note": "CSR build EXCLUDED: every workspace keeps one batch's tables prepared, only the features are read anew; "
                    "same streams / region / statistic as `value` (which includes the CSR build); single_stream = one workspace, "
                    "forwards back to back on one stream (head_pairs_0: with the option set for one batch in flight, gnnb_hip.h)"},
    }
    if other_form is not None:
        result["prep_next_pipeline" if not prep_next else "forward_batched_per_step"] = {
            "value": graphs_done / other_form, "unit": "graphs/s", "ms_per_step": other_form / args.steps * 1e3,
            "note": ("the same region, software-pipelined per stream over two alternating workspaces (gnnb_forward_prepared_prep_next): step i = "
                     "forward of batch i + CSR build of the stream's next batch as extra workgroups of the readout kernel; every batch of the "
                     "region is built once and forwarded once inside it.  A gain with ONE batch in flight, none with three (DESIGN 3.6): not `value`")
            if not prep_next else "the same region with gnnb_forward_batched per step (k_graph_prep launch in front of every forward)"}
    if segs[0] is not None:
        result["config"]["large_segment"] = {
            "limit_nodes": w["large_limit"],
            "graphs_per_batch_mean": float(np.mean([b.num_graphs - sg[0] for b, sg in zip(batches, segs)])),
            "nodes_per_batch_mean": float(np.mean([b.num_nodes - sg[1] for b, sg in zip(batches, segs)])),
            "largest_graph": int(max(np.diff(b.node_ptr).max() for b in batches)),
            "how": "graphs beyond the stage limit are ordered last (batching.order_large_last) and run layer by layer; the rest "
                   "of the batch stays in the LDS-resident stack (gnnb_workspace_set_large_segment)"}
    if dry:
        result["dry_launch"] = True
        result["dtype"] = "f32 (CPU stand-in)"
        result["config"]["note"] = ("launcher / sharding / reduction plumbing on CPU over gloo; the step is the PyTorch "
                                    "model definition, NOT the HIP path: not a measurement")
        print(json.dumps(result))
        return
    if split_rate is not None:
        result["opt_in_math_bf16x6"] = {
            "value": split_rate, "unit": "graphs/s", "ms_per_step": split_ms,
            "how": "GNNB_MATH=1 / gnnb_set_option(\"math\", 1): the wide updates (A1.W1^T of the fused GCN stack, the K <= 128 "
                   "GEMMs, the large-K segmented GEMM) as 6 bf16 MFMA products per k block on an exact hi/mid/lo bf16 split of "
                   "both operands, fp32 accumulate; NOT used for `value`.  (The 2-layer GCN stack keeps its fp32 kernel k_gcn2_zf, "
                   "which is faster than the bf16x6 form of k_gcn2_fused; the fused GIN and deeper-than-two GCN stacks "
                   "are fp32-only and stay on their fp32 stack kernel: the mode only changes the layer-by-layer GEMMs)",
            "accuracy": ("max |out - float64 evaluation| on this workload: 9.4e-8 (fp32-MFMA path 6.3e-8, scalar fp32 "
                         "reference 1.4e-7; tests/accuracy_math_modes.py)") if args.workload == "c2" else
                        ("per GEMM against a float64 product: no worse than 2x the fp32-MFMA kernel's error + 1e-7 "
                         "(tests: *_bf16x6_math_is_fp32_equivalent)"),
        }

    if reduced is not None:
        result["opt_in_math_reduced_precision"] = reduced

    if not args.no_roofline:
        alg_bytes, agg = measure_aggregate_roofline(cm, dev_batches[0], w["hidden"], dev)
        ceiling = copy_ceiling(batches[0].num_nodes, w["hidden"], dev)
        if "copy_same_launch_shape" in agg:
            ceiling["own_float4_copy_same_launch_shape_hbm"] = agg["copy_same_launch_shape"]
        # what plain byte-moving kernels reach on this pool's boxes (tools/copy_forms.py, committed; not re-measured here): the
        # best form at the C5-sized and the C2-sized launch, and what the ring kernel's own ownership pattern costs
        cal = ROOT / "profiles" / "r06_copy_calibration.json"
        if cal.exists():
            c = json.loads(cal.read_text())
            ceiling["pool_copy_ceiling_tbps"] = {**c["pool_copy_ceiling_tbps"], "measured_in_this_run": False,
                                                 "file": "profiles/r06_copy_calibration.json", "findings": c["findings"][1:4]}
        gather = {
            "kernel": "k_aggregate_ring<GCN> (gather-aggregate, width %d)" % w["hidden"],
            "bound": "hbm", "achieved": agg["hbm"]["gbps"], "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": agg["hbm"]["gbps"] / HBM_PEAK_GBPS, "traffic": pmc_traffic("aggregate", args.workload, alg_bytes),
            "algorithmic_bytes_per_launch": alg_bytes, "us_per_launch": agg["hbm"]["us"],
            "regime": "inputs/outputs rotate over >256 MiB of distinct buffers (HBM-served); launches issued "
                      "back to back from C, HIP events on the launch stream",
            "in_pipeline_l3_resident": agg["l3_resident"],
            "copy_ceiling_same_bytes": ceiling,
        }
        wkind, k_out = WORKLOAD_AGG[w["conv"]]
        if wkind != "gcn":
            # the aggregate this workload's layer-by-layer forward actually runs at its full width (SUM / MEAN / PNA with four
            # output matrices), beside the GCN kind the north star names: same batch, same protocol
            ab, ag = measure_aggregate_roofline(cm, dev_batches[0], w["hidden"], dev, kind=wkind, pna_self_term=not max_degree)
            gather["workload_kind"] = {
                "kernel": "k_aggregate_ring<%s> (width %d, %d output matri%s)" % (wkind.upper(), w["hidden"], k_out, "x" if k_out == 1 else "ces"),
                "bound": "hbm", "achieved": ag["hbm"]["gbps"], "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": ag["hbm"]["gbps"] / HBM_PEAK_GBPS, "algorithmic_bytes_per_launch": ab, "us_per_launch": ag["hbm"]["us"],
                "traffic": pmc_traffic("aggregate_%s" % wkind, args.workload, ab),
                "in_pipeline_l3_resident": ag.get("l3_resident"), "extra_read_bytes": ag.get("extra_read_bytes", 0),
                "destination_term": bool(wkind == "pna" and not max_degree),
                "in_timed_step": not (wkind == "pna" and max_degree and cm.last_path() == "layerwise" and max_graph + 7 <= 64),
                "note": "algorithmic bytes per SURVEY 8(d): 4 w N (1 + k_out) + CSR + graph ptr; PNA WITHOUT a max_degree promise "
                        "reads its per-destination term q [N, w] on top (extra_read_bytes, not counted); with the promise -- "
                        "what the timed step runs -- the aggregate takes no destination term"}
        if w["conv"] == "pna" and max_degree:
            pa = measure_pna_product_aggregate(cm, dev_batches[0], w["hidden"], dev)
            if pa is not None:
                gather["pna_product_aggregate"] = pa
        fused = measure_fused_stack(cm, dev_batches[0], (int(batches[0].x.shape[1]), w["hidden"], w.get("out_dim", w["hidden"]), len(w["pools"])),
                                    conv=w["conv"], layers=w["layers"], seg=segs[0]) if w["conv"] in ("gcn", "gin") and w["layers"] >= 2 else None
        upd = dict(kernel="k_linear_wlds (fp32 MFMA, weights in LDS), full-width layer update", bound="mfma", peak=FP32_MFMA_PEAK_TFLOPS,
                   unit="TFLOP/s", traffic=None, **measure_update_mfma(w, batches[0].num_nodes, dev))
        if fused is not None:
            stack_path = cm.last_path()  # which stack kernel the timed launches ran
            # the step runs the fused stack: that kernel dominates it and is bound by the fp32 matrix rate
            step_us = elapsed / args.steps * 1e3 * 1e3 / 1.0
            result["roofline"] = {
                "kernel": "%s (%d %s layers + pooling in one persistent kernel, graphs staged in LDS)" % ({"stack_zf": "k_gcn2_zf", "stack": "k_gcn2_fused"}.get(stack_path, stack_path), w["layers"], w["conv"].upper()),
                "bound": "mfma", "achieved": fused["tflops"], "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": fused["tflops"] / FP32_MFMA_PEAK_TFLOPS,
                "traffic": pmc_traffic("gcn2", args.workload, fused["alg_bytes"]),
                "algorithmic_flops_per_launch": fused["flops"], "algorithmic_hbm_bytes_per_launch": fused["alg_bytes"],
                "us_per_launch": fused["us"], "loops_us": fused.get("loops_us"),  # (median of five 200-launch loops; all five listed)
                # the solo launch time against the one-stream prepared forward (NOT against the timed step: with
                # several batches in flight the stack kernels of consecutive batches overlap their edges)
                "share_of_single_stream_forward": fused["us"] / (ms_noprep_1s * 1e3),
                # the same flops against the timed step itself: what the kernel delivers inside the pipeline
                "in_pipeline": {"us_per_step": step_us, "achieved": fused["flops"] / (step_us * 1e-6) / 1e12,
                                "frac": fused["flops"] / (step_us * 1e-6) / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                                "note": "algorithmic flops of one launch / time per timed step (graph prep and readout of "
                                        "other batches run beside the kernel; a solo launch can be LONGER than a step "
                                        "when consecutive launches overlap on the chip)"},
                "note": "flops = the dense updates on v_mfma_f32_16x16x4_f32 (GCN: 2 N (F0 h0 + h0 h1); GIN: 2 N (F0 h + (2L - 1) h^2)); "
                        "HIP events on the launch stream, launches issued back to back from C on one prepared batch",
            }
        elif w["conv"] in ("sage", "pna"):
            # layer-wise workloads with a wide concatenated update: the large-K segmented GEMM dominates the step
            result["roofline"] = dict(kernel="k_linear_dma (fp32 MFMA 32x32x2, chunks global -> LDS by DMA)", bound="mfma",
                                      peak=FP32_MFMA_PEAK_TFLOPS, unit="TFLOP/s", traffic=None,
                                      **measure_segmented_gemm(w, batches[0].num_nodes, dev, pna_classes=bool(max_degree)))
        elif w["conv"] == "gin":
            result["roofline"] = upd
        else:
            result["roofline"] = gather
        result["roofline_gather_aggregate"] = gather
        result["roofline_update"] = upd
    if world == 1 and args.workload == "c2" and not args.no_other_configs:
        # BASELINE's other single-GPU configs, briefly (appended after `value` was computed: the C2 region is untouched)
        del pipe, cm
        torch.cuda.synchronize()
        oc = []
        for name in ("c3", "c4", "c5"):
            try:
                oc.append(other_config_leg(name, dev, nstreams))
            except Exception as e:  # never lose the bench line to a side leg
                oc.append({"name": name, "error": repr(e)})
        result["other_configs"] = oc
    if not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(model, batches)

    print(json.dumps(result))


if __name__ == "__main__":
    main()
