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
