"""``Project`` keeps the reference's surface (code_gen.py:62-489) and emits the MI355X host shim,
the testbench harness, the hipcc makefile and tb_data in the reference's on-disk format."""
import json
import subprocess

import numpy as np
import pytest
import torch

import gnnbuilder_amd as gnnb
from gnnbuilder_amd import runtime, synthetic
from gnnbuilder_amd.data import ListDataset
from helpers import make_model


@pytest.fixture(scope="module")
def project(tmp_path_factory):
    build = tmp_path_factory.mktemp("build")
    model = make_model("sage", in_dim=9, hidden=16, layers=2, task_out=1, pools=("add", "max"))
    ds = ListDataset.from_batch(synthetic.make_batch("esol", 5, seed=1), y_dim=1)
    proj = gnnb.Project("demo_sage", model, "regression", None, build, dataset=ds, max_nodes=64, max_edges=160,
                        num_nodes_guess=13, num_edges_guess=28, degree_guess=3, float_or_fixed="float",
                        fpx=gnnb.FPX(32, 16), fpga_part="xcu280-fsvh2892-2L-e", n_jobs=4)
    proj.gen_hw_model()
    proj.gen_testbench()
    proj.gen_makefile()
    return proj


def test_emitted_files_exist(project):
    for f in ("model.h", "model.cpp", "model_tb.cpp", "makefile_testbench", "model_desc.json"):
        assert (project.model_dir / f).exists()


def test_top_signature_matches_the_reference_abi(project):
    h = (project.model_dir / "model.h").read_text()
    assert 'void demo_sage_top(' in h
    assert "F_TYPE node_feature_table_input[64][9]" in h and "int edge_list_input[160][2]" in h
    assert "F_TYPE model_output[1]" in h and "int copy_parameters_flag" in h
    # one trailing pointer per parameter, mlp_head first then convs (reference models.py:615-624)
    names = project.model.layer_parameter_names_flat
    pos = [h.index(f"{n}_fixed_in") for n in names]
    assert pos == sorted(pos)
    assert names[0].startswith("mlp_head") and "W_TYPE gnn_convs_0_conv_lin_l_weight_fixed_in[16][9]" in h


def test_manifest_describes_the_model(project):
    m = json.loads((project.model_dir / "model_desc.json").read_text())
    assert m["spec"]["conv"] == "sage" and m["spec"]["pools"] == ["add", "max"]
    assert m["canonical_order"][0] == "gnn_convs_0_conv_lin_l_weight"
    assert {p["name"] for p in m["parameters"]} == set(m["canonical_order"])


def test_tb_data_uses_the_reference_layout(project):
    tb = project.model_dir / "tb_data"
    lines = (tb / "dataset_info.txt").read_text().split("\n")
    assert lines[0] == "num_graphs 5" and lines[1:6] == ["0", "1", "2", "3", "4"]
    w = np.fromfile(tb / "model_parameters" / "gnn_convs_0_conv_lin_l_weight.bin", dtype="<f4")
    assert np.array_equal(w.reshape(16, 9), project.model.gnn_convs[0].conv.lin_l.weight.detach().numpy())
    g = project.dataset[3]
    info = np.fromfile(tb / "graphs" / "graph_3_info.bin", dtype="<i4")
    assert info.tolist() == [g.num_nodes, g.num_edges]
    coo = np.fromfile(tb / "graphs" / "graph_3_coo.bin", dtype="<i4").reshape(-1, 2)
    assert np.array_equal(coo, g.edge_index.T.numpy())
    x = np.fromfile(tb / "graphs" / "graph_3_node_features.bin", dtype="<f4").reshape(-1, 9)
    assert np.array_equal(x, g.x.numpy())
    gold = np.fromfile(tb / "graphs" / "graph_3_model_golden_output.bin", dtype="<f4")
    with torch.no_grad():
        assert np.array_equal(gold, project.model(g.x, g.edge_index).view(-1).numpy())
    assert np.fromfile(tb / "graphs" / "graph_3_task_golden_output.bin", dtype="<f4").shape == (1,)


def test_generated_sources_compile_with_hipcc(project):
    if not runtime.LIB_PATH.exists():
        runtime.build_library()
    proc = subprocess.run(["make", "-f", "makefile_testbench", "result", "libdemo_sage.so"], cwd=project.model_dir,
                          capture_output=True, text=True)
    assert proc.returncode == 0, proc.stderr
    out = subprocess.run(["nm", "-D", "--defined-only", str(project.model_dir / "libdemo_sage.so")],
                         capture_output=True, text=True).stdout
    for sym in ("demo_sage_top", "demo_sage_batched", "demo_sage_status", "demo_sage_release"):
        assert f" T {sym}" in out


@pytest.mark.skipif(torch.cuda.is_available(), reason="needs a machine WITHOUT a GPU")
def test_testbench_fails_loudly_without_gpu(project):
    with pytest.raises(Exception, match="Testbench execution failed"):
        project.build_and_run_testbench()


def test_constructor_validation_matches_the_reference():
    model = make_model("gcn", hidden=8)
    with pytest.raises(ValueError):
        gnnb.Project("p", model, "bogus", None, "/tmp")
    with pytest.raises(ValueError):
        gnnb.Project("p", model, "regression", None, "/tmp", clock_speed=0)
    with pytest.raises(ValueError):
        gnnb.Project("p", model, "regression", None, "/tmp", fpga_part="nope")
    with pytest.raises(ValueError):
        gnnb.Project("p", model, "regression", None, "/tmp", n_jobs=0)
    # float_or_fixed = "fixed": the ap_fixed<W, I> emulation is rendered into the model description ...
    fx = gnnb.Project("pfx", model, "regression", None, "/tmp", float_or_fixed="fixed", fpx=gnnb.FPX(16, 10))
    assert fx.template_dict["desc"]["fpx_w"] == 16 and fx.template_dict["desc"]["fpx_i"] == 10
    assert gnnb.Project("pfl", model, "regression", None, "/tmp").template_dict["desc"]["fpx_w"] == 0
    # ... other rounding / overflow modes than the reference's defaults are not emulated
    with pytest.raises(NotImplementedError):
        gnnb.Project("p", model, "regression", None, "/tmp", float_or_fixed="fixed", fpx=gnnb.FPX(16, 10, Q="AP_RND"))
    with pytest.raises(Exception, match="I must be <= 33"):
        gnnb.FPX(64, 40)
    for w_, i_ in ((8, 9), (8, 0)):  # I > W, I < 1: refused by the Project, not later by gnnb_model_create
        with pytest.raises(ValueError):
            gnnb.Project("p", model, "regression", None, "/tmp", float_or_fixed="fixed", fpx=gnnb.FPX(w_, i_))
    # math (MI355X only): a known mode name, rendered as the runtime option in front of gnnb_model_create
    with pytest.raises(ValueError):
        gnnb.Project("p", model, "regression", None, "/tmp", math="fp8")
    assert gnnb.Project("p", model, "regression", None, "/tmp").template_dict["math_mode"] == 0
    assert gnnb.Project("p", model, "regression", None, "/tmp", math="f16x3").template_dict["math_mode"] == 3
    p = gnnb.Project("p", model, "regression", None, "/tmp")
    with pytest.raises(NotImplementedError):
        p.run_vitis_hls_synthesis()
    with pytest.raises(Exception, match="does not exist"):
        gnnb.Project("never_generated", model, "regression", None, "/tmp/gnnb_nowhere").build_and_run_testbench()


def test_math_mode_is_rendered_into_the_generated_shim(tmp_path):
    """Project(math=...) -> the `math` field of the generated design's gnnb_model_desc (its own arithmetic, captured by
    gnnb_model_create: no process-wide option is touched); the reduced modes' shims ask the workspace for GNNB_ERR_RANGE."""
    model = make_model("gcn", hidden=8)
    ds = ListDataset.from_batch(synthetic.make_batch("esol", 2, seed=1), y_dim=1)
    for mode, n in (("fp32", 0), ("bf16x6", 1), ("f16x3", 3)):
        proj = gnnb.Project(f"m_{mode}", model, "regression", None, tmp_path, dataset=ds, max_nodes=64, max_edges=200, math=mode)
        proj.gen_hw_model()
        src = (proj.model_dir / "model.cpp").read_text()
        assert "gnnb_set_option" not in src
        desc = src[src.index("k_desc = {"):src.index("};", src.index("k_desc = {"))]
        assert f"/* math           */ {n}," in desc
        assert ("gnnb_workspace_check(g_ws, NULL)" in src) == (n >= 2)


def test_tb_data_reader_round_trips_what_the_writer_wrote(project):
    """``load_tb_data`` reads the reference's on-disk layout back into the batched form
    (graphs in dataset_info.txt order) together with the golden outputs."""
    from gnnbuilder_amd.data import load_tb_data

    batch, golden, indices = load_tb_data(project.model_dir / "tb_data", num_features=9, out_dim=1)
    assert indices == [0, 1, 2, 3, 4] and batch.num_graphs == 5
    batch.validate()
    for k, idx in enumerate(indices):
        g = project.dataset[idx]
        x, coo = batch.graph(k)
        assert np.array_equal(x, g.x.numpy()) and np.array_equal(coo, g.edge_index.T.numpy())
        with torch.no_grad():
            assert np.array_equal(golden[k], project.model(g.x, g.edge_index).view(-1).numpy())
