"""Whole-model parity against golden vectors produced by the REFERENCE'S OWN generated C++.

tests/golden/ref_models/*.npz were written by tools/gen_ref_model_fixtures.py in the build container: the
reference's templates/model.{h,cpp}.jinja rendered in place with a duck-typed model object, compiled with
g++ against the reference's gnn_builder_lib.h, `<name>_top` called per graph through ctypes.  They pin what
the reference's tb_data fixtures cannot: layer dimensions, skip placement (model.cpp.jinja:264-311),
activation after every conv (:313-322), pooling concat order (:440-448), the MLP head (:454-530) and the
parameter naming.  Only numbers are stored (inputs, weights by reference name, per-graph outputs).

CPU:  the C oracle must reproduce them (PNA with the library's std flavour, std="hls": SURVEY finding 5).
GPU:  the HIP path through the C ABI must reproduce the non-PNA ones within the north-star tolerance 1e-4
      (its PNA follows PyG's std; HIP PNA is checked against the oracle's std="pyg" in test_hip_parity.py and
      against the reference's PyG golden tb_pna_output.bin).
"""
import json
from pathlib import Path

import numpy as np
import pytest

from oracle import oracle as O

GOLDEN = Path(__file__).resolve().parent / "golden" / "ref_models"
CASES = sorted(p.stem for p in GOLDEN.glob("*.npz"))

# reference parameter suffixes -> canonical slot order of the C ABI / oracle (include/gnnb_hip.h)
CONV_SLOTS = {
    "gcn": ["conv_lin_weight", "conv_bias"],
    "gin": ["mlp_linear_0_weight", "mlp_linear_0_bias", "mlp_linear_1_weight", "mlp_linear_1_bias"],
    "sage": ["conv_lin_l_weight", "conv_lin_l_bias", "conv_lin_r_weight"],
    "pna": ["conv_pre_nns_0_0_weight", "conv_pre_nns_0_0_bias", "conv_post_nns_0_0_weight", "conv_post_nns_0_0_bias",
            "conv_lin_weight", "conv_lin_bias"],
}


def load_case(name):
    z = np.load(GOLDEN / f"{name}.npz")
    spec = json.loads(str(z["spec"]))
    params = [z[f"w__gnn_convs_{l}_{s}"] for l in range(spec["num_layers"]) for s in CONV_SLOTS[spec["conv"]]]
    for i in range(spec["mlp_hidden_layers"] + 1):
        params += [z[f"w__mlp_head_linear_layers_{i}_weight"], z[f"w__mlp_head_linear_layers_{i}_bias"]]
    # every stored weight is consumed exactly once
    assert len(params) == len([k for k in z.files if k.startswith("w__")])
    return spec, params, z["x"], z["coo"], z["node_ptr"], z["edge_ptr"], z["out"]


def test_fixture_set_covers_the_grid():
    specs = [load_case(c)[0] for c in CASES]
    assert len(CASES) >= 16
    assert {s["conv"] for s in specs} == {"gcn", "gin", "sage", "pna"}
    for conv in ("gcn", "gin", "sage", "pna"):
        depths = {s["num_layers"] for s in specs if s["conv"] == conv}
        assert {1, 4} <= depths and (2 in depths or 3 in depths), (conv, depths)
    assert {s["activation"] for s in specs} == {"relu", "sigmoid", "tanh"}
    assert any(s["skip"] and s["num_layers"] >= 3 for s in specs) and any(not s["skip"] for s in specs)
    assert len({tuple(s["pools"]) for s in specs}) >= 6
    # the reference's published benchmark model (experiments/build_base_benchmarks.py:61-81): six conv layers with skip,
    # out != hidden, a head of four hidden layers -- for every conv
    ref6 = [s for s in specs if s["num_layers"] == 6]
    assert {s["conv"] for s in ref6} == {"gcn", "gin", "sage", "pna"}
    assert all(s["skip"] and s["out_dim"] != s["hidden_dim"] and s["mlp_hidden_layers"] == 4 and s["pools"] == ["add", "mean", "max"] for s in ref6)


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_the_reference_generated_model(name):
    spec, params, x, coo, node_ptr, edge_ptr, want = load_case(name)
    got = O.forward_batched(spec, params, x, coo, node_ptr, edge_ptr, std="hls")
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= 1e-6, (name, np.abs(got - want).max())
    # per graph through the single-graph entry as well (the reference's call pattern)
    for g in (0, want.shape[0] - 1):
        n0, n1, e0, e1 = node_ptr[g], node_ptr[g + 1], edge_ptr[g], edge_ptr[g + 1]
        one = O.forward(spec, params, x[n0:n1], coo[e0:e1] - n0, std="hls")
        assert np.abs(one - want[g]).max() <= 1e-6


@pytest.mark.parametrize("name", [c for c in CASES if "pna" not in c])
def test_torch_model_definition_reproduces_the_reference_generated_model(name):
    """The package's GNNModel.forward (what Project.gen_testbench_data records as the golden) against the
    same vectors: the Python model definition and the reference's generated C++ agree."""
    import torch

    import gnnbuilder_amd as gnnb
    from helpers import ACTS, CONVS

    spec, params, x, coo, node_ptr, edge_ptr, want = load_case(name)
    gw = spec["out_dim"] if spec["num_layers"] else spec["in_dim"]
    model = gnnb.GNNModel(spec["in_dim"], None, spec["hidden_dim"], spec["num_layers"], gw, CONVS[spec["conv"]],
                          ACTS[spec["activation"]], spec["skip"], gnnb.GlobalPooling(spec["pools"]),
                          gnnb.MLP(len(spec["pools"]) * gw, spec["mlp_out"], spec["mlp_hidden"], spec["mlp_hidden_layers"],
                                   activation=ACTS[spec["mlp_activation"]]), None).eval()
    if spec["conv"] == "gin":
        for c in model.gnn_convs:      # GNNModel never passes eps (reference models.py:546-548): set it on the layers
            c.eps = spec["gin_eps"]
            c.conv.eps.fill_(spec["gin_eps"])
    with torch.no_grad():
        for p, v in zip(model.canonical_params(), params):
            p.copy_(torch.from_numpy(np.asarray(v)))
        bv = torch.from_numpy(np.repeat(np.arange(len(node_ptr) - 1), np.diff(node_ptr)).astype(np.int64))
        got = model(torch.from_numpy(x), torch.from_numpy(coo.T.astype(np.int64)), bv).numpy()
    assert np.abs(got - want).max() <= 2e-6, (name, np.abs(got - want).max())


@pytest.mark.gpu
@pytest.mark.parametrize("name", [c for c in CASES if "pna" not in c])
def test_hip_reproduces_the_reference_generated_model(name):
    import torch

    from gnnbuilder_amd import runtime

    spec, params, x, coo, node_ptr, edge_ptr, want = load_case(name)
    dev = torch.device("cuda:0")
    B, N, E = len(node_ptr) - 1, x.shape[0], coo.shape[0]
    args = tuple(torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (x, coo, node_ptr, edge_ptr))
    for promise in (0, int(np.diff(node_ptr).max())):     # layer-wise route, and the LDS-resident routes
        cm = runtime.CompiledModel(spec, params, B, N, max(E, 1), max_graph_nodes=promise)
        got = cm.forward(*args)
        cm.check()
        err = np.abs(got.cpu().numpy() - want).max()
        assert err <= 1e-4, (name, promise, err)
        cm.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", [c for c in CASES if "pna" in c])
def test_hip_pna_against_reference_generated_model_outside_the_std_block(name):
    """PNA from the reference's generated C++ uses the library's std (sqrt(var + 1e-5)); the HIP path follows
    PyG's (clamp + mask).  The two differ by up to ~5e-4 per std value (SURVEY finding 5), so this is a
    plumbing check of concat order / scalers / dims at a looser bound; the tight checks are HIP == oracle
    std="pyg" (test_hip_parity.py) and oracle std="hls" == these fixtures (above)."""
    import torch

    from gnnbuilder_amd import runtime

    spec, params, x, coo, node_ptr, edge_ptr, want = load_case(name)
    dev = torch.device("cuda:0")
    B, N, E = len(node_ptr) - 1, x.shape[0], coo.shape[0]
    args = tuple(torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (x, coo, node_ptr, edge_ptr))
    cm = runtime.CompiledModel(spec, params, B, N, max(E, 1))
    got = cm.forward(*args).cpu().numpy()
    cm.check()
    pyg = O.forward_batched(spec, params, x, coo, node_ptr, edge_ptr, std="pyg")
    assert np.abs(got - pyg).max() <= 1e-4
    assert np.abs(got - want).max() <= 5e-3
