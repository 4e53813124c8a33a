#!/usr/bin/env python3
"""Diagnostic: stage timeline of the ring-form gather-aggregate kernel from the -DGNNB_PROBE build
(make -C gnn-builder_amd/csrc probe): wave 0 of every workgroup logs wall-clock stamps when a stage is issued,
waited for, landed and reduced.  python tools/probe_ring.py ['{"agg_ring_waves":16}'] [kind] [workload]"""
import ctypes as C
import json
import os
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("GNNB_HIP_LIB", str(ROOT / "gnn-builder_amd" / "libgnnb_hip_probe.so"))
import bench  # noqa: E402
from gnnbuilder_amd import runtime, synthetic  # noqa: E402

opts = json.loads(sys.argv[1]) if len(sys.argv) > 1 else {}
kind = sys.argv[2] if len(sys.argv) > 2 else "gcn"
w = bench.WORKLOADS[sys.argv[3] if len(sys.argv) > 3 else "c2"]
dev = torch.device("cuda:0")
model = bench.build_model(w)
batch = synthetic.make_batch(w["shape"], w["batch"], seed=0)
cm = runtime.CompiledModel.from_model(model, batch.num_graphs, batch.num_nodes, batch.num_edges)
x, coo, nptr, eptr = (torch.from_numpy(a).to(dev) for a in (batch.x, batch.coo, batch.node_ptr, batch.edge_ptr))
for k, v in opts.items():
    runtime.set_option(k, v)
cm.graph_prep(coo, nptr, eptr, batch.num_nodes)
width = w["hidden"]
nbuf = 9
ins = [torch.rand(batch.num_nodes, width, device=dev) for _ in range(nbuf)]
outs = [torch.empty(batch.num_nodes, width, device=dev) for _ in range(nbuf)]
for i in range(30):
    cm.aggregate(kind, ins[i % nbuf], out=outs[i % nbuf])
torch.cuda.synchronize()
lib = runtime.load_library()
n = 2048 * 64
buf = (C.c_ulonglong * n)()
lib.gnnb_probe_read(buf, n)
a = np.frombuffer(buf, dtype=np.uint64).reshape(2048, 64).astype(np.int64)
a = a[a[:, 0] > 0]
t0 = a[:, 0].min()
print(f"opts={opts} kind={kind} workgroups={a.shape[0]}; starts spread over {(a[:, 0].max() - t0) / 100:.2f} us")
names = {1: "issued", 2: "wait", 3: "landed", 4: "reduced"}
nev = int(np.median(a[:, 1]))
print(f"events per workgroup: median {nev}, max {a[:, 1].max()}")
sel = a[a[:, 1] == nev]
for e in range(nev):
    t = (sel[:, 2 + 2 * e] - t0) / 100.0
    code = sel[:, 3 + 2 * e]
    print(f"  ev{e:2d} {names[int(np.median(code)) // 1000000]:8s} rows {np.median(code % 1000000):5.0f}   t = {np.median(t):6.2f} us (p10 {np.percentile(t, 10):6.2f}, p90 {np.percentile(t, 90):6.2f})")
end = np.array([r[2 + 2 * (int(r[1]) - 1)] for r in a]) - t0
print(f"last event: median {np.median(end) / 100:.2f} us, max {end.max() / 100:.2f} us")
ends = np.array([r[2 + 2 * (int(r[1]) - 1)] for r in a]) - t0
starts = a[:, 0] - t0
idx = np.arange(a.shape[0])
for xcd in range(8):
    m = idx % 8 == xcd
    print(f"  xcd {xcd}: start median {np.median(starts[m]) / 100:.2f}, end median {np.median(ends[m]) / 100:.2f} max {ends[m].max() / 100:.2f} us")
order = np.argsort(ends)[-12:]
print("  latest workgroups:", [(int(i), round(float(ends[i]) / 100, 2)) for i in order])
for i in order[-4:]:
    r = a[i]
    print(f"  wg {int(i)}: {int(r[1])} events:", [(names[int(r[3 + 2 * e]) // 1000000][:3], int(r[3 + 2 * e]) % 1000000, round(float(r[2 + 2 * e] - t0) / 100, 2)) for e in range(int(r[1]))])
# per-workgroup work (rows, edges, rows of degree > 4) against its end time
TR = 8
ntiles = (batch.num_nodes + TR - 1) // TR
nptr_h, eptr_h = batch.node_ptr.astype(np.int64), batch.edge_ptr.astype(np.int64)
tf = nptr_h[np.searchsorted(nptr_h, np.arange(ntiles + 1) * TR, side="left").clip(max=len(nptr_h) - 1)]
tf[-1] = batch.num_nodes
gidx = np.searchsorted(nptr_h, tf, side="left")
te = eptr_h[gidx.clip(max=len(eptr_h) - 1)]
deg = np.bincount(batch.coo[:, 1] if kind != "x" else batch.coo[:, 0], minlength=batch.num_nodes)
nwg = a.shape[0]
wt0 = (np.arange(nwg) * ntiles) // nwg
wt1 = ((np.arange(nwg) + 1) * ntiles) // nwg
rows = tf[wt1] - tf[wt0]
edges = te[wt1] - te[wt0]
cdeg = np.concatenate([[0], np.cumsum(deg > 4)])
big = cdeg[tf[wt1]] - cdeg[tf[wt0]]
print(f"  rows/wg {rows.min()}..{rows.max()}, edges/wg {edges.min()}..{edges.max()}, deg>4 rows/wg {big.min()}..{big.max()}")
for nm, v in (("rows", rows), ("edges", edges), ("deg>4", big), ("wg index", idx)):
    print(f"  corr(end, {nm}) = {np.corrcoef(ends, v)[0, 1]:.2f}")
