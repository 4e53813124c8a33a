"""End to end through the reference's workflow on a real MI355X: Project -> generated shim +
testbench -> hipcc -> ./result -> MAE against the PyTorch golden (reference code_gen.py:339-395,
model_tb.cpp.jinja:157-265).  The reference only *reports* MAE; here it is asserted."""
import ctypes as C

import numpy as np
import pytest
import torch

import gnnbuilder_amd as gnnb
from gnnbuilder_amd import synthetic
from gnnbuilder_amd.data import ListDataset
from helpers import make_model

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("conv", ["gcn", "pna"])
def test_project_testbench_end_to_end(tmp_path, conv):
    model = make_model(conv, in_dim=11, hidden=32, layers=3, task_out=19)
    ds = ListDataset.from_batch(synthetic.make_batch("qm9", 24, seed=5), y_dim=19)
    proj = gnnb.Project(f"tb_{conv}", model, "regression", None, tmp_path, dataset=ds, max_nodes=40, max_edges=120)
    proj.gen_hw_model()
    proj.gen_testbench()
    proj.gen_makefile()
    res = proj.build_and_run_testbench()
    assert set(res) >= {"model_output_mae", "model_runtime"}
    assert res["model_output_mae"] < 1e-5 and res["model_output_mae_batched"] < 1e-5
    assert 0 < res["model_runtime_batched"] < res["model_runtime"]


@pytest.mark.parametrize("conv,mode,bound", [("gcn", "f16x3", 2e-5), ("sage", "bf16x6", 1e-5), ("gin", "f16x3", 2e-5)])
def test_project_math_modes_end_to_end(tmp_path, conv, mode, bound):
    """Project(math=...) (MI355X only: the throughput side of the reference's float_or_fixed switch, code_gen.py:39-52) through
    the generated shim and testbench: the mode is set by the generated code itself (a separate process), MAE against the
    PyTorch golden inside the mode's bound."""
    model = make_model(conv, in_dim=11, hidden=128, layers=2, task_out=19)
    ds = ListDataset.from_batch(synthetic.make_batch("qm9", 24, seed=6), y_dim=19)
    proj = gnnb.Project(f"tbm_{conv}", model, "regression", None, tmp_path, dataset=ds, max_nodes=40, max_edges=120, math=mode)
    proj.gen_hw_model()
    proj.gen_testbench()
    proj.gen_makefile()
    res = proj.build_and_run_testbench()
    assert res["model_output_mae"] < bound and res["model_output_mae_batched"] < bound


def test_generated_top_symbol_via_ctypes(tmp_path):
    """Bind the generated `<name>_top` exactly as the reference's C testbench calls it."""
    import subprocess
    model = make_model("sage", in_dim=9, hidden=16, layers=2, task_out=1)
    ds = ListDataset.from_batch(synthetic.make_batch("esol", 3, seed=2), y_dim=1)
    proj = gnnb.Project("abi_sage", model, "regression", None, tmp_path, dataset=ds, max_nodes=64, max_edges=200)
    proj.gen_hw_model()
    proj.gen_makefile()
    subprocess.run(["make", "-f", "makefile_testbench", "libabi_sage.so"], cwd=proj.model_dir, check=True,
                   capture_output=True)
    lib = C.CDLL(str(proj.model_dir / "libabi_sage.so"))
    params = [np.ascontiguousarray(p.detach().numpy()) for _, p in model.layer_parameters_flat]
    g = ds[1]
    x = np.zeros((64, 9), np.float32)
    x[:g.num_nodes] = g.x.numpy()
    e = np.zeros((200, 2), np.int32)
    e[:g.num_edges] = g.edge_index.T.numpy()
    out = np.zeros(1, np.float32)
    args = [a.ctypes.data_as(C.c_void_p) for a in (x, e, out)] + [g.num_nodes, g.num_edges, 1] + \
           [p.ctypes.data_as(C.c_void_p) for p in params]
    lib.abi_sage_top(*args)
    assert lib.abi_sage_status() == 0
    with torch.no_grad():
        ref = model(g.x, g.edge_index).view(-1).numpy()
    assert np.abs(out - ref).max() < 1e-5
    lib.abi_sage_release()


@pytest.mark.gpu
def test_bench_line_keeps_the_contract(tmp_path):
    """bench.py prints ONE JSON line with the contract's keys plus the roofline / cpu_baseline objects
    (short run, bounded CPU sample)."""
    import json
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--batches", "4"],
                       capture_output=True, text=True, timeout=600, cwd=str(root))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["unit"] == "graphs/s"
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32"
    assert "workload" in d["config"] and "model" not in d["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in d["roofline"], k
    assert abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 1e-9
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in d["cpu_baseline"], k
    assert d["cpu_baseline"]["kind"] in ("reference", "port") and d["cpu_baseline"]["cores"] == 1
    assert d["value"] > 1e6 and 0.0 < d["roofline"]["frac"] < 1.0
    # BASELINE's other single-GPU configs ride along (brief legs behind the C2 timed region), each on the route the full-size
    # parity tests prove (tests/test_hip_parity.py::test_full_size_configs_3_4_5_on_the_routes_bench_times)
    oc = {o["name"]: o for o in d["other_configs"]}
    assert sorted(oc) == ["c3", "c4", "c5"]
    for name, o in oc.items():
        assert "error" not in o, o
        for k in ("workload", "value", "ms_per_step", "path", "max_degree_promise", "roofline"):
            assert k in o, (name, k)
        assert o["value"] > 1e6 and 0.0 < o["roofline"]["frac"] < 1.0 and "kernel" in o["roofline"]
    assert oc["c3"]["path"] == "stack" and oc["c4"]["max_degree_promise"] and oc["c4"]["path"] == "layerwise"
    # CSR-included and CSR-excluded rates on the SAME pipeline
    pt = d["prepared_topology"]
    assert pt["batches_in_flight_per_gpu"] == d["config"]["batches_in_flight_per_gpu"] and pt["ms_per_step"] <= d["ms_per_step"] * 1.1
