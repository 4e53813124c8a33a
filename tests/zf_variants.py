#!/usr/bin/env python3
"""Development check (uses the oracle: lives under tests/): parity (oracle, sampled graphs) + launch-loop time of k_gcn2_zf for ONE library build
(GNNB_HIP_LIB=<variant .so> python tests/zf_variants.py [tag]); tools/zf_variants.sh loops over the variant builds."""
import os, sys, json
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np, torch
import bench
from gnnbuilder_amd import runtime, synthetic
from oracle import oracle as O
tag = sys.argv[1] if len(sys.argv) > 1 else os.environ.get("GNNB_HIP_LIB", "default")
w = bench.WORKLOADS[os.environ.get("ZFV_WORKLOAD", "c2")]  # (c2: k_gcn2_zf; c3: the GIN stack kernel)
dev = torch.device("cuda:0")
model = bench.build_model(w)
spec, params = model.spec(), [p.numpy() for p in model.canonical_params()]
res = {"tag": tag}
for seed in (0,):
    b = synthetic.make_batch(w["shape"], w["batch"], seed=seed)
    mg = int(np.diff(b.node_ptr).max())
    bd = tuple(torch.from_numpy(a).to(dev) for a in (b.x, b.coo, b.node_ptr, b.edge_ptr))
    idx = np.unique(np.concatenate([np.arange(0, 48), np.arange(b.num_graphs - 48, b.num_graphs),
                                    np.random.default_rng(1).integers(0, b.num_graphs, 96), [int(np.diff(b.node_ptr).argmax())]]))
    refs = {int(g): O.forward_batched(spec, params, *(lambda s: (s.x, s.coo, s.node_ptr, s.edge_ptr))(b.slice(int(g), int(g) + 1)))[0] for g in idx}
    for shape in ((1, 0) if w["conv"] == "gcn" else (2,)):
        runtime.set_option("zf_shape", shape)
        cm = runtime.CompiledModel.from_model(model, b.num_graphs, b.num_nodes, b.num_edges, max_graph_nodes=mg)
        out = cm.forward(*bd).cpu().numpy()
        cm.check()
        assert cm.last_path().startswith("stack"), cm.last_path()
        worst = max(float(np.abs(out[g] - refs[g]).max()) for g in refs)
        cm.graph_prep(bd[1], bd[2], bd[3], int(bd[0].shape[0]))
        t = sorted(cm.gcn_stack_timed(bd[0], 200) for _ in range(7))
        res[f"shape{shape}"] = {"err": worst, "us_min": t[0], "us_med": t[3]}
print(json.dumps(res))
