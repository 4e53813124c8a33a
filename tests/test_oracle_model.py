"""Whole-model oracle (oracle/gnnb_oracle.c) against the reference's own compiled kernels
(oracle/_ref), composed in the order its generated top uses -- pins the parts the reference's
committed fixtures do not reach: multi-layer stacks, skip, activations, pooling, MLP head."""
import numpy as np
import pytest

from gnnbuilder_amd import synthetic
from gnnbuilder_amd.batching import pack_graphs
from helpers import canon, make_model
from oracle import oracle as O

needs_ref = pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")

CASES = [
    ("gcn", 11, 128, 2, "relu", True, ("add", "mean", "max"), "qm9", 19),
    ("gin", 9, 128, 3, "relu", True, ("add",), "molhiv", 1),
    ("sage", 9, 256, 2, "relu", True, ("add", "mean", "max"), "molhiv", 1),
    ("pna", 11, 128, 3, "relu", True, ("add", "mean", "max"), "qm9", 19),
    ("gcn", 11, 16, 4, "tanh", True, ("max", "add"), "qm9", 5),
    ("sage", 9, 16, 3, "sigmoid", False, ("mean",), "esol", 1),
    ("gin", 11, 16, 2, "gelu", False, ("add", "max", "mean"), "qm9", 5),
]


@needs_ref
@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c[0]}-L{c[3]}-d{c[2]}-{c[4]}")
def test_oracle_bit_exact_with_reference_kernels(case):
    conv, fin, hid, layers, act, skip, pools, shape, out = case
    m = make_model(conv, in_dim=fin, hidden=hid, layers=layers, act=act, skip=skip, pools=pools, task_out=out,
                   mlp_hidden=16 if hid == 16 else 64)
    b = synthetic.make_batch(shape, 6, seed=hid)
    got = O.forward_batched(m.spec(), canon(m), b.x, b.coo, b.node_ptr, b.edge_ptr, std="hls")
    ref = O.ref_forward_batched(m.spec(), canon(m), b.x, b.coo, b.node_ptr, b.edge_ptr)
    assert np.array_equal(got, ref)
    if conv != "pna":  # the std flavour only exists in PNA
        assert np.array_equal(got, O.forward_batched(m.spec(), canon(m), b.x, b.coo, b.node_ptr, b.edge_ptr))


def test_batched_equals_per_graph():
    m = make_model("pna", hidden=16, layers=2, task_out=3)
    b = synthetic.make_batch("qm9", 9, seed=2)
    whole = O.forward_batched(m.spec(), canon(m), b.x, b.coo, b.node_ptr, b.edge_ptr)
    for g in range(b.num_graphs):
        x, coo = b.graph(g)
        assert np.array_equal(O.forward(m.spec(), canon(m), x, coo), whole[g])


def test_degenerate_inputs_are_finite():
    m = make_model("pna", in_dim=4, hidden=8, layers=2, task_out=2)
    rng = np.random.default_rng(0)
    graphs = [(rng.uniform(-1, 1, (1, 4)), np.zeros((0, 2))), (rng.uniform(-1, 1, (0, 4)), np.zeros((0, 2))),
              (rng.uniform(-1, 1, (4, 4)), np.array([[0, 0], [1, 0], [1, 0], [2, 3]]))]
    b = pack_graphs([(np.asarray(x, np.float32), np.asarray(c, np.int32)) for x, c in graphs])
    out = O.forward_batched(m.spec(), canon(m), b.x, b.coo, b.node_ptr, b.edge_ptr)
    assert np.isfinite(out).all()
    # PNA std of a degree-1 node is exactly 0 under the PyG semantics (SURVEY finding 5)
    with pytest.raises(RuntimeError):
        bad = b.coo.copy()
        bad[0, 0] = 3  # edge of graph 2 pointing at a node of... itself is fine; make it leave the graph
        bad[0, 0] = 99
        O.forward_batched(m.spec(), canon(m), b.x, bad, b.node_ptr, b.edge_ptr)


@pytest.mark.parametrize("conv", ["gcn", "gin", "sage", "pna"])
def test_fixed_point_emulation_in_the_oracle(conv):
    """FPX(W, I) emulation (reference code_gen.py:39-52, model.h.jinja:38-62): every output lies on the
    ap_fixed<W, I> grid, a fine grid reproduces the float result, a coarse one departs from it by a few steps per
    layer, and wrap-around (AP_WRAP) keeps values inside [-2^(I-1), 2^(I-1))."""
    from gnnbuilder_amd import synthetic
    from helpers import canon, make_model

    model = make_model(conv, in_dim=9, hidden=16, layers=2, task_out=3)
    batch = synthetic.make_batch("esol", 12, seed=1)
    args = (canon(model), batch.x, batch.coo, batch.node_ptr, batch.edge_ptr)
    ref = O.forward_batched(model.spec(), *args)
    for W, I, tol in ((32, 8, 1e-3), (16, 8, 0.5)):
        spec = dict(model.spec(), fpx=(W, I))
        out = O.forward_batched(spec, *args)
        step = 2.0 ** -(W - I)
        assert np.abs(out / step - np.round(out / step)).max() < 1e-3          # on the grid
        assert np.abs(out).max() < 2.0 ** (I - 1)
        assert np.abs(out - ref).max() < tol, (W, I, np.abs(out - ref).max())
    coarse = O.forward_batched(dict(model.spec(), fpx=(12, 6)), *args)
    assert np.abs(coarse - ref).max() > 1e-3                                   # the grid really bites
