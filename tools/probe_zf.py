#!/usr/bin/env python3
"""Diagnostic: phase cycles of the transform-first 2-layer GCN stack kernel k_gcn2_zf (probe build, `make probe`)."""
import ctypes as C, os, sys, json
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("GNNB_HIP_LIB", str(ROOT / "gnn-builder_amd" / "libgnnb_hip_probe.so"))
import bench
from gnnbuilder_amd import runtime, synthetic
w = bench.WORKLOADS["c2"]; dev = torch.device("cuda:0")
model = bench.build_model(w)
b = synthetic.make_batch(w["shape"], w["batch"], seed=0)
cm = runtime.CompiledModel.from_model(model, b.num_graphs, b.num_nodes, b.num_edges, max_graph_nodes=int(np.diff(b.node_ptr).max()))
bd = tuple(torch.from_numpy(a).to(dev) for a in (b.x, b.coo, b.node_ptr, b.edge_ptr))
for _ in range(10):
    cm.forward(*bd)
torch.cuda.synchronize()
assert cm.last_path() == "stack_zf", cm.last_path()
lib = runtime.load_library(); n = 16 * 8192
buf = (C.c_ulonglong * n)(); lib.gnnb_probe_read(buf, n)
NW = 16 if int(os.environ.get("GNNB_ZF_SHAPE", "1")) else 8
pall = np.frombuffer(buf, dtype=np.uint64)[8 * 8192:8 * 8192 + 16 * 4096].reshape(4096 // NW, NW, 16).astype(np.float64)
p = pall[:, 0, :]
p = p[p[:, 14] > 0]
life = (p[:, 1] - p[:, 0]) / 100
out = {"workgroups": len(p), "stages_per_wg_mean": p[:, 14].mean(), "stages_per_wg_max": p[:, 14].max(),
       "span_us": (p[:, 1].max() - p[:, 0].min()) / 100, "lifetime_us_mean": life.mean(), "lifetime_us_max": life.max(),
       "last_start_us": (p[:, 0].max() - p[:, 0].min()) / 100, "clock_mhz": float(np.median(p[:, 13] / life))}
print(json.dumps(out))
names = ["prologue (tables, weights, first DMA, P0 of stage 0)", "issue next DMA", "M0", "barrier (H complete)", "M1 (Z in accumulators)",
         "DMA wait + barrier (H read)", "Z write + barrier", "P1 + pooling", "P0 of the next stage", "barrier (end of stage)", "(inside P1: the row loop alone)"]
for i, nm in enumerate(names):
    per = p[:, 2 + i] / (1 if i == 0 else p[:, 14])
    print(f"  {nm:52s} {100 * (p[:, 2 + i] / p[:, 13]).mean():5.1f}%  {per.mean():8.0f} cycles" + ("" if i == 0 else "/stage"))
rows = (p[:, 15].astype(np.uint64) & np.uint64(0xffffffff)).astype(np.float64)
graphs = ((p[:, 15].astype(np.uint64) >> np.uint64(32)) & np.uint64(0xffff)).astype(np.float64)
units = (p[:, 15].astype(np.uint64) >> np.uint64(48)).astype(np.float64)
end = (p[:, 1] - p[:, 0].min()) / 100
start = (p[:, 0] - p[:, 0].min()) / 100
print("lifetime us percentiles 5/25/50/75/95/100:", np.percentile(life, [5, 25, 50, 75, 95, 100]).round(2).tolist())
print("start us percentiles 50/95/100:", np.percentile(start, [50, 95, 100]).round(2).tolist(), " end us 50/95/100:", np.percentile(end, [50, 95, 100]).round(2).tolist())
print(f"rows per workgroup {rows.min():.0f}..{rows.max():.0f} (mean {rows.mean():.1f}); corr(lifetime, rows) {np.corrcoef(life, rows)[0, 1]:.2f}, corr(lifetime, graphs) {np.corrcoef(life, graphs)[0, 1]:.2f}, corr(lifetime, start) {np.corrcoef(life, start)[0, 1]:.2f}")
print(f"units per workgroup {units.min():.0f}..{units.max():.0f} (mean {units.mean():.2f}); corr(lifetime, units) {np.corrcoef(life, units)[0, 1]:.2f}")
for u in np.unique(units):
    m = units == u
    print(f"  units {u:.0f}: {m.sum()} workgroups, lifetime mean {life[m].mean():.2f} us, 95% {np.percentile(life[m], 95):.2f}, max {life[m].max():.2f}; rows {rows[m].mean():.0f}")
xcd = np.arange(len(life)) % 8
print("lifetime mean per XCD (block id mod 8):", [round(float(life[xcd == k].mean()), 2) for k in range(8)])
print("per-wave cycles/stage (mean over workgroups):")
pa = pall[pall[:, 0, 14] > 0]
for wv in range(NW):
    print(f"  wave {wv}: " + " ".join(f"{(pa[:, wv, 2 + i] / (1 if i == 0 else pa[:, wv, 14])).mean():7.0f}" for i in range(11)))
