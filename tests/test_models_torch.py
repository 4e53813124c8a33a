"""The PyTorch model API (gnnbuilder_amd/models.py) against the reference's PyG golden vectors
and against the CPU oracle -- three independent statements of the same semantics."""
import numpy as np
import pytest
import torch

import gnnbuilder_amd as gnnb
import golden_util as G
from gnnbuilder_amd import synthetic
from helpers import batch_vector, canon, make_model
from oracle import oracle as O

STATE_NAMES = {
    "gcn": (gnnb.GCNConv_GNNB, ["conv.lin.weight", "conv.bias"], {}),
    "gin": (gnnb.GINConv_GNNB, ["mlp.linear_0.weight", "mlp.linear_0.bias", "mlp.linear_1.weight", "mlp.linear_1.bias"],
            {"eps": G.conv_kwargs("gin")["eps"]}),
    "sage": (gnnb.SAGEConv_GNNB, ["conv.lin_l.weight", "conv.lin_l.bias", "conv.lin_r.weight"], {}),
    "pna": (gnnb.PNAConv_GNNB, ["conv.pre_nns.0.0.weight", "conv.pre_nns.0.0.bias", "conv.post_nns.0.0.weight",
                                "conv.post_nns.0.0.bias", "conv.lin.weight", "conv.lin.bias"],
            {"delta": G.conv_kwargs("pna")["delta"]}),
}


@pytest.mark.parametrize("kind", ["gcn", "gin", "sage", "pna"])
def test_conv_modules_reproduce_pyg_golden(kind):
    cls, names, kw = STATE_NAMES[kind]
    conv = cls(G.F, G.F, **kw)
    sd = conv.state_dict()
    assert all(n in sd for n in names), "parameter names must match the reference's state_dict"
    for n, w in zip(names, G.conv_weights(kind)):
        sd[n] = torch.from_numpy(np.array(w))
    conv.load_state_dict(sd)
    x, coo = G.graph()
    y = conv(torch.from_numpy(x), torch.from_numpy(coo.T.astype(np.int64))).detach().numpy()
    assert np.abs(y - G.conv_golden(kind)).max() < 1e-6


def test_parameter_order_is_head_first_then_convs():
    # reference: mlp_head is assigned before gnn_convs (models.py:497,510), SURVEY 3.2
    m = make_model("sage", hidden=16, layers=2)
    names = m.layer_parameter_names_flat
    assert names[0] == "mlp_head_linear_layers_0_weight"
    assert names.index("gnn_convs_0_conv_lin_l_weight") > names.index("mlp_head_linear_layers_2_bias")
    assert len(names) == len(set(names)) == len(m.canonical_param_names())
    assert set(names) == set(m.canonical_param_names())


def test_layer_dims_follow_the_reference_rule():
    assert make_model("gcn", in_dim=11, hidden=32, layers=1, out_dim=8).gnn_layer_sizes == [(11, 8)]
    assert make_model("gcn", in_dim=11, hidden=32, layers=3, out_dim=8).gnn_layer_sizes == [(11, 32), (32, 32), (32, 8)]
    with pytest.raises(ValueError):
        make_model("gcn", in_dim=11, hidden=32, layers=0, out_dim=8)  # models.py:512-518


def test_unsupported_configurations_raise():
    with pytest.raises(ValueError):
        gnnb.MLP(4, 2, activation=torch.nn.ELU)
    with pytest.raises(NotImplementedError):
        gnnb.MLP(4, 2, norm_layer=torch.nn.LayerNorm)
    with pytest.raises(ValueError):
        gnnb.GlobalPooling([])
    with pytest.raises(NotImplementedError):
        gnnb.GlobalPooling(["median"])
    with pytest.raises(ValueError):
        gnnb.GNNModel(4, None, 8, 1, 8, torch.nn.Linear, torch.nn.ReLU, False, gnnb.GlobalPooling(["add"]),
                      gnnb.MLP(8, 1), None)
    with pytest.raises(NotImplementedError):
        gnnb.GATConv_GNNB(4, 4)


@pytest.mark.parametrize("conv,act,skip,pools,layers", [
    ("gcn", "relu", True, ("add", "mean", "max"), 2), ("gin", "tanh", True, ("add",), 4),
    ("sage", "sigmoid", True, ("mean", "max"), 3), ("pna", "relu", False, ("max", "add", "mean"), 2),
    ("sage", "gelu", False, ("add",), 1), ("gcn", "relu", False, ("max",), 0)])
def test_torch_forward_matches_oracle(conv, act, skip, pools, layers):
    in_dim = 11
    m = make_model(conv, in_dim=in_dim, hidden=11 if layers == 0 else 24, layers=layers, act=act, skip=skip,
                   pools=pools, task_out=5)
    b = synthetic.make_batch("qm9", 12, seed=layers)
    ref = O.forward_batched(m.spec(), canon(m), b.x, b.coo, b.node_ptr, b.edge_ptr)
    with torch.no_grad():
        got = m(torch.from_numpy(b.x), torch.from_numpy(b.coo.T.astype(np.int64)),
                torch.from_numpy(batch_vector(b))).numpy()
        xg, cg = b.graph(5)
        one = m(torch.from_numpy(xg), torch.from_numpy(cg.T.astype(np.int64))).numpy()
    assert np.abs(got - ref).max() < 2e-5
    assert one.shape == (1, 5) and np.abs(one[0] - ref[5]).max() < 2e-5


def test_output_activation_softmax_oracle_equals_torch():
    """GNNModel(output_activation=nn.Softmax / nn.LogSoftmax) (reference models.py:500-502, 572-573): the oracle's
    output map equals the torch module applied to each graph's row."""
    import gnnbuilder_amd as gnnb
    from gnnbuilder_amd import synthetic
    from helpers import batch_vector, canon
    from oracle import oracle as O

    batch = synthetic.make_batch("qm9", 12, seed=3)
    for cls, name in ((torch.nn.Softmax, "softmax"), (torch.nn.LogSoftmax, "log_softmax")):
        torch.manual_seed(1)
        model = gnnb.GNNModel(11, None, 16, 2, 16, gnnb.GCNConv_GNNB, torch.nn.ReLU, True, gnnb.GlobalPooling(["add", "max"]),
                              gnnb.MLP(32, 5, 16, 1), cls).eval()
        assert model.spec()["output_activation"] == name
        with torch.no_grad():
            want = model(torch.from_numpy(batch.x), torch.from_numpy(batch.coo.T.astype(np.int64)),
                         torch.from_numpy(batch_vector(batch))).numpy()
        got = O.forward_batched(model.spec(), canon(model), batch.x, batch.coo, batch.node_ptr, batch.edge_ptr)
        assert np.abs(got - want).max() < 1e-6
        if name == "softmax":
            assert np.abs(got.sum(1) - 1).max() < 1e-6
    with pytest.raises(NotImplementedError):
        gnnb.GNNModel(11, None, 16, 1, 16, gnnb.GCNConv_GNNB, torch.nn.ReLU, True, gnnb.GlobalPooling(["add"]),
                      gnnb.MLP(16, 5, 16, 1), torch.nn.Softmin).spec()
