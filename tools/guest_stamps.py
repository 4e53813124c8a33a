#!/usr/bin/env python3
"""Development (variant library built with -DZF_GUEST_DEV=8): wall-clock stamps of the guest prep inside k_gcn2_zf."""
import ctypes as C, os, sys, json
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench
from gnnbuilder_amd import runtime, synthetic
w = bench.WORKLOADS["c2"]; dev = torch.device("cuda:0")
model = bench.build_model(w)
bs = [synthetic.make_batch(w["shape"], w["batch"], seed=i) for i in range(2)]
mg = max(int(np.diff(b.node_ptr).max()) for b in bs)
cap = [max(getattr(b, a) for b in bs) for a in ("num_graphs", "num_nodes", "num_edges")]
cms = [runtime.CompiledModel.from_model(model, *cap, max_graph_nodes=mg) for _ in range(2)]
bd = [tuple(torch.from_numpy(a).to(dev) for a in (b.x, b.coo, b.node_ptr, b.edge_ptr)) for b in bs]
for i in range(2):
    cms[i].graph_prep(bd[i][1], bd[i][2], bd[i][3], bs[i].num_nodes)
for i in range(20):
    k = i & 1
    cms[k].forward_prepared_prep_next(bd[k][0], cms[k ^ 1], bd[k ^ 1][1], bd[k ^ 1][2], bd[k ^ 1][3], bs[k ^ 1].num_nodes)
torch.cuda.synchronize()
lib = runtime.load_library()
buf = (C.c_ulonglong * (256 * 16 * 8))()
assert lib.gnnb_guest_dbg_read(buf) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(256, 16, 8).astype(np.float64) / 100.0  # us
t0 = t[:, :, 0].min()
def col(i, waves):
    v = t[:, waves, i]; v = v[v > 0] - t0; return v
print("kernel span: entry min 0, last stamp", round(float((t[:, :, 7].max() - t0)), 2), "us")
for name, waves in (("P1 waves 0-7", slice(0, 8)), ("prep waves 8-11", slice(8, 12)), ("idle waves 12-15", slice(12, 16))):
    print(name)
    for i, lab in ((0, "entry"), (1, "last stage: P1 done / tail start"), (2, "guest: start"), (3, "guest: params in"), (4, "guest: fetched inputs waited"),
                   (5, "guest: prep_one_graph returned"), (6, "guest: stores acked"), (7, "after the last barrier")):
        v = col(i, waves)
        if len(v):
            print(f"   {lab:36s} mean {v.mean():7.2f}  p50 {np.percentile(v, 50):7.2f}  p95 {np.percentile(v, 95):7.2f}  max {v.max():7.2f}")
# per-wave deltas on prep waves
pw = t[:, 8:12, :]
ok = pw[:, :, 5] > 0
for a, b, lab in ((1, 2, "break -> guest start"), (2, 3, "params"), (3, 4, "dma wait"), (4, 5, "prep_one_graph"), (5, 6, "store ack"), (6, 7, "barrier")):
    d = (pw[:, :, b] - pw[:, :, a])[ok]
    print(f"   delta {lab:24s} mean {d.mean():6.2f}  p95 {np.percentile(d, 95):6.2f}  max {d.max():6.2f} us")
