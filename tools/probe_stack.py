#!/usr/bin/env python3
"""Diagnostic: phase cycles of k_gcn2_fused (probe build); PROBE_WL = c2 | c3 | ref6_gcn | ref6_gin (GNNB_FUSE_ZF=0 for c2)."""
import ctypes as C, os, sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("GNNB_HIP_LIB", str(ROOT / "gnn-builder_amd" / "libgnnb_hip_probe.so"))
import bench
from gnnbuilder_amd import runtime, synthetic
w = bench.WORKLOADS[os.environ.get('PROBE_WL','c2')]; dev = torch.device("cuda:0")
model = bench.build_model(w)
b = synthetic.make_batch(w["shape"], w["batch"], seed=0)
cm = runtime.CompiledModel.from_model(model, b.num_graphs, b.num_nodes, b.num_edges, max_graph_nodes=int(np.diff(b.node_ptr).max()))
bd = tuple(torch.from_numpy(a).to(dev) for a in (b.x, b.coo, b.node_ptr, b.edge_ptr))
for _ in range(10):
    cm.forward(*bd)
torch.cuda.synchronize()
lib = runtime.load_library(); n = 16 * 8192
buf = (C.c_ulonglong * n)(); lib.gnnb_probe_read(buf, n)
pall = np.frombuffer(buf, dtype=np.uint64)[8 * 8192:8 * 8192 + 16 * 4096].reshape(512, 8, 16).astype(np.float64)
p = pall[:, 0, :]
p = p[p[:, 14] > 0]
life = (p[:, 1] - p[:, 0]) / 100
print(f"workgroups {len(p)}, stages/WG {p[:, 14].mean():.2f} (max {p[:, 14].max():.0f}); span {(p[:, 1].max() - p[:, 0].min()) / 100:.2f} us, lifetime mean {life.mean():.2f} max {life.max():.2f}, last start +{(p[:, 0].max() - p[:, 0].min()) / 100:.2f}")
names = ["wait DMA + barrier(1)", "issue next DMA", "P0 (agg F0)", "barrier(2)", "M0", "barrier(3)", "GIN: in-place product 0 | GCN: P1", "GIN: m_mid + barrier | GCN: barrier(4)", "GIN: P1 (all layers)", "GIN: barrier + in-place products", "M1 + pooling"]
for i, nm in enumerate(names):
    print(f"  {nm:24s} {100 * (p[:, 2 + i] / p[:, 13]).mean():5.1f}%  {(p[:, 2 + i] / p[:, 14]).mean():8.0f} cycles/stage")
print(f"  clock {np.median(p[:, 13] / life):.0f} MHz")

print("per-wave cycles/stage (mean over workgroups):")
pa = pall[pall[:, 0, 14] > 0]
for w in range(8):
    print(f"  wave {w}: " + " ".join(f"{(pa[:, w, 2 + i] / pa[:, w, 14]).mean():7.0f}" for i in range(11)))
