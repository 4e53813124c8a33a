"""Parity tests proper: the HIP path (through the C ABI of libgnnb_hip.so) against the CPU
oracle and the reference's golden vectors.  Needs a real MI355X: run with ``-m gpu``.

Tolerance: 1e-4 absolute on O(1) outputs (BASELINE.json north_star: "outputs within 1e-4 of the
PyTorch reference"); integer tables bit-exact.
"""
import numpy as np
import pytest
import torch

import golden_util as G
from gnnbuilder_amd import runtime, synthetic
from gnnbuilder_amd.batching import GraphBatch, pack_graphs
from helpers import canon, make_model, to_dev
from oracle import oracle as O

pytestmark = pytest.mark.gpu

TOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    runtime.load_library(require_gpu=True)  # fails loudly: no fallback
    return torch.device("cuda:0")


def fixture_batch():
    x, coo = G.graph()
    return pack_graphs([(x, coo)])


def oracle_tables_batched(batch: GraphBatch):
    row_ptr, cols = [0], []
    for g in range(batch.num_graphs):
        xg, cg = batch.graph(g)
        in_deg, _, offsets, nbrs = O.tables(cg, xg.shape[0])
        n0 = int(batch.node_ptr[g])
        for d in in_deg:
            row_ptr.append(row_ptr[-1] + int(d))
        cols.append(nbrs + n0)
    return np.asarray(row_ptr, np.int32), (np.concatenate(cols) if cols else np.zeros(0, np.int32)).astype(np.int32)


def plain_model(conv, fin, fout):
    # one layer; identity-ish tail so stage tests can reuse CompiledModel's workspace
    return make_model(conv, in_dim=fin, hidden=fout, layers=1, out_dim=fout, task_out=3, mlp_layers=0)


# --------------------------------------------------------------------------- graph prep
def edge_case_batch():
    rng = np.random.default_rng(5)
    graphs = []
    graphs.append((rng.uniform(-1, 1, (1, 8)), np.zeros((0, 2), np.int32)))                    # single node, E=0
    graphs.append((rng.uniform(-1, 1, (5, 8)), np.array([[0, 1], [1, 0], [2, 2], [3, 1], [3, 1]])))  # self loop, dup edge, isolated node 4
    graphs.append((rng.uniform(-1, 1, (0, 8)), np.zeros((0, 2), np.int32)))                    # empty graph
    n = 150                                                                                     # > 64 nodes: several lane chunks
    e = np.stack([rng.integers(0, n, 700), rng.integers(0, n, 700)], 1)                         # > 512 edges: uncached path
    graphs.append((rng.uniform(-1, 1, (n, 8)), e))
    graphs.append((rng.uniform(-1, 1, (3, 8)), np.array([[0, 1], [1, 2], [2, 0]])))
    return pack_graphs([(np.asarray(x, np.float32), np.asarray(c, np.int32)) for x, c in graphs])


def molecule_path_batch():
    """The corners of graph prep's molecule path (<= 64 nodes and <= 64 edges, matched on the bits of the destination index):
    exactly 64 nodes / 64 edges, 63 / 65 (just outside: the general path), node counts either side of a power of two (the
    number of matched bits), every edge into ONE node, every edge the same edge, self loops only, isolated nodes."""
    rng = np.random.default_rng(11)
    graphs = []

    def g(n, coo):
        graphs.append((rng.uniform(-1, 1, (n, 8)).astype(np.float32), np.asarray(coo, np.int32).reshape(-1, 2)))

    for n, e in ((64, 64), (64, 63), (63, 64), (64, 65), (65, 64), (32, 64), (33, 64), (31, 60), (17, 40), (16, 40), (2, 64), (1, 7)):
        g(n, np.stack([rng.integers(0, n, e), rng.integers(0, n, e)], 1))
    g(40, np.stack([rng.integers(0, 40, 64), np.full(64, 39)], 1))      # a hub takes all 64 edges (rank up to 63)
    g(40, np.tile([[3, 5]], (64, 1)))                                     # 64 copies of one edge
    g(12, np.stack([np.arange(12), np.arange(12)], 1))                    # self loops only
    g(50, np.zeros((0, 2)))                                               # no edges
    g(9, [[0, 8], [8, 0], [8, 8], [1, 8]])
    return pack_graphs(graphs)


@pytest.mark.parametrize("which", ["fixture", "qm9", "molhiv", "edge", "molecule_path", "c2_full"])
def test_graph_prep_bit_exact(dev, which):
    batch = {"fixture": fixture_batch, "qm9": lambda: synthetic.make_batch("qm9", 300, 1),
             "molhiv": lambda: synthetic.make_batch("molhiv", 200, 2), "edge": edge_case_batch,
             "molecule_path": molecule_path_batch, "c2_full": lambda: synthetic.make_batch("qm9", 4096, 0)}[which]()
    # (a workspace bound to a non-GCN model keeps every edge = the reference's tables; GCN: next test)
    cm = runtime.CompiledModel.from_model(plain_model("gin", batch.x.shape[1], 8), batch.num_graphs + 1,
                                          batch.num_nodes + 1, batch.num_edges + 1)
    _, coo, nptr, eptr = to_dev(batch, dev)
    cm.graph_prep(coo, nptr, eptr, batch.num_nodes)
    cm.check()
    row_ptr, col, in_deg = cm.tables_to_host()
    rp_ref, col_ref = oracle_tables_batched(batch)
    assert np.array_equal(row_ptr, rp_ref)
    assert np.array_equal(col, col_ref)
    assert np.array_equal(in_deg, np.diff(rp_ref))
    if which == "fixture":  # the reference's own committed tables (test.cpp:884-1054)
        assert np.array_equal(in_deg, G.i32("tb_in_degree_table"))
        assert np.array_equal(row_ptr[:-1], G.i32("tb_neighbor_table_offsets"))
        assert np.array_equal(col, G.i32("tb_neighbor_table"))


def test_gcn_workspace_drops_explicit_self_loops(dev):
    """Tables of a GCN-bound workspace: an edge (v, v) is not entered (PyG add_remaining_self_loops); the degree
    is the node record's, row_ptr holds row starts inside the graph's own CSR segment."""
    batch = edge_case_batch()
    cm = runtime.CompiledModel.from_model(plain_model("gcn", 8, 8), batch.num_graphs, batch.num_nodes, batch.num_edges)
    _, coo, nptr, eptr = to_dev(batch, dev)
    cm.graph_prep(coo, nptr, eptr, batch.num_nodes)
    cm.check()
    row_ptr, col, in_deg = cm.tables_to_host()
    keep = batch.coo[:, 0] != batch.coo[:, 1]
    assert (~keep).sum() >= 1
    want_deg = np.bincount(batch.coo[keep, 1], minlength=batch.num_nodes)
    assert np.array_equal(in_deg, want_deg)
    for g in range(batch.num_graphs):
        e0, e1 = batch.edge_ptr[g], batch.edge_ptr[g + 1]
        for v in range(batch.node_ptr[g], batch.node_ptr[g + 1]):
            srcs = batch.coo[e0:e1][(batch.coo[e0:e1, 1] == v) & (batch.coo[e0:e1, 0] != v), 0]
            assert e0 <= row_ptr[v] and row_ptr[v] + in_deg[v] <= e1
            assert np.array_equal(col[row_ptr[v]:row_ptr[v] + in_deg[v]], srcs)   # stable COO order


def test_gcn_explicit_self_loops_follow_pyg(dev):
    """The chosen semantics, stated independently of the oracle: GCNConv = D^-1/2 (A' + I) D^-1/2 X W^T + b where
    A' is the input adjacency WITHOUT its self loops (PyG gcn_norm / add_remaining_self_loops) and D = deg(A') + 1.
    The reference C++ would add the loop's message on top of its self term (gnn_builder_lib.h:1266-1278): the oracle's
    self_loops="hls" flavour reproduces that and must differ here."""
    rng = np.random.default_rng(3)
    n, fin, fout = 9, 8, 8
    coo = np.array([[0, 1], [1, 0], [2, 2], [2, 3], [3, 2], [4, 4], [4, 4], [5, 6], [6, 5], [7, 5], [1, 2]], np.int32)
    x = rng.uniform(-1, 1, (n, fin)).astype(np.float32)
    model = plain_model("gcn", fin, fout)
    W, b = [q.numpy().astype(np.float64) for q in model.canonical_params()[:2]]
    keep = coo[:, 0] != coo[:, 1]
    A = np.zeros((n, n))
    np.add.at(A, (coo[keep, 1], coo[keep, 0]), 1.0)          # A[dst, src], duplicates counted
    dinv = 1.0 / np.sqrt(A.sum(1) + 1.0)
    want = (dinv[:, None] * (A + np.eye(n)) * dinv[None, :]) @ x.astype(np.float64) @ W.T + b
    batch = pack_graphs([(x, coo)])
    cm = runtime.CompiledModel.from_model(model, 1, n, coo.shape[0])
    xd, cood, nptr, eptr = to_dev(batch, dev)
    cm.graph_prep(cood, nptr, eptr, n)
    got = runtime.linear([(cm.aggregate("gcn", xd), None)], torch.from_numpy(W.astype(np.float32)).to(dev),
                         torch.from_numpy(b.astype(np.float32)).to(dev)).cpu().numpy()
    assert np.abs(got - want).max() < 2e-6
    pyg = O.forward(model.spec(), canon(model), x, coo)
    hls = O.forward(model.spec(), canon(model), x, coo, self_loops="hls")
    assert np.abs(pyg - hls).max() > 1e-3                      # the two semantics really differ on this input
    whole = cm.forward(xd, cood, nptr, eptr).cpu().numpy()[0]
    assert np.abs(whole - pyg).max() < TOL
    with torch.no_grad():                                      # and the torch model definition agrees
        t = model(torch.from_numpy(x), torch.from_numpy(coo.T.astype(np.int64))).numpy()[0]
    assert np.abs(t - pyg).max() < 1e-5
    # the three dropped self-loop edges leave three CSR slots unowned: they read -1 in the host copies, every other slot
    # holds a source of its row / the COO row it came from (gnnb_hip.h, gnnb_graph_prep)
    row_ptr, col, in_deg = cm.tables_to_host()
    eid = cm.edge_index_table_to_host()
    assert (col == -1).sum() == 3 and (eid == -1).sum() == 3 and np.array_equal(col == -1, eid == -1)
    assert in_deg.sum() == int(keep.sum())
    for v in range(n):
        slots = np.arange(row_ptr[v], row_ptr[v] + in_deg[v])
        assert np.array_equal(coo[eid[slots], 1], np.full(len(slots), v)) and np.array_equal(coo[eid[slots], 0], col[slots])


def test_stage_entry_points_refuse_mismatched_tensors(dev):
    """Raw pointers cross the C ABI (advisor finding): the GINE aggregate reads edge_term[edge * width] for every CSR slot
    and x[source * width] -- a short, narrower, int64, CPU or strided tensor, or an `out` of another shape, is refused by
    the binding instead of being read out of bounds.  The sizes of the prepared batch are known after forward() too."""
    model = plain_model("gin", 8, 8)
    x, coo = G.graph()
    batch = pack_graphs([(x, coo)])
    cm = runtime.CompiledModel.from_model(model, 1, G.N, G.E)
    xd, cood, nptr, eptr = to_dev(batch, dev)
    cm.forward(xd, cood, nptr, eptr)                           # (prepares the batch: no separate graph_prep call)
    assert cm.edge_index_table_to_host().shape[0] == G.E
    et = torch.rand(G.E, 8, device=dev)
    good = cm.aggregate_edges(xd, et)
    assert good.shape == xd.shape
    for bad_x, bad_et, bad_out in ((xd, et[:-1], None), (xd, et[:, :4].contiguous(), None), (xd, et.double(), None),
                                   (xd, et.cpu(), None), (xd, et.T.contiguous().T, None), (xd[:-1], et, None),
                                   (xd, et, torch.empty(G.N, 4, device=dev)), (xd, et, torch.empty(G.N - 1, 8, device=dev))):
        with pytest.raises(runtime.GnnbError):
            cm.aggregate_edges(bad_x, bad_et, out=bad_out)
    with pytest.raises(runtime.GnnbError):
        cm.aggregate("sum", xd, out=torch.empty(G.N, 4, device=dev))
    with pytest.raises(runtime.GnnbError):
        cm.forward_prepared(xd, out=torch.empty(3, cm.out_dim, device=dev))
    with pytest.raises(runtime.GnnbError):
        cm.forward(xd, cood, nptr, eptr, out=torch.empty(1, cm.out_dim + 1, device=dev))


@pytest.mark.parametrize("kind,golden", [("simple", "tb_simple_output"), ("lg", "tb_lgconv_output")])
def test_weight_free_convs_match_reference_golden(dev, kind, golden):
    """SimpleConv (plain neighbour sum) and LGConv (sum_j x_j / sqrt(d_i d_j), no self term) on the reference's
    fixture graph against its PyG-generated goldens (test.cpp:1728-1919 accepts 1e-3), all variants of the kernel."""
    x, coo = G.graph()
    batch = pack_graphs([(x, coo)])
    cm = runtime.CompiledModel.from_model(plain_model("gin", 8, 8), 1, G.N, G.E)
    xd, cood, nptr, eptr = to_dev(batch, dev)
    cm.graph_prep(cood, nptr, eptr, G.N)
    want = G.f32(golden).reshape(G.N, 8)
    try:
        for waves in (0, 2):
            runtime.set_option("agg_ring_waves", waves)
            got = cm.aggregate(kind, xd).cpu().numpy()
            assert np.abs(got - want).max() < 2e-6, (kind, waves)
            assert np.abs(got - O.conv(kind, x, coo, [])).max() < 2e-6
    finally:
        runtime.set_option("agg_ring_waves", 0)


def test_gine_conv_matches_reference_golden(dev):
    """GINEConv on the reference's fixture graph with its edge features and weights against its PyG golden
    (test.cpp:1287-1455 accepts 1e-3), and the edge-index table bit-exact against tb_edge_index_table.bin."""
    x, coo = G.graph()
    batch = pack_graphs([(x, coo)])
    cm = runtime.CompiledModel.from_model(plain_model("gin", 8, 8), 1, G.N, G.E)
    xd, cood, nptr, eptr = to_dev(batch, dev)
    cm.graph_prep(cood, nptr, eptr, G.N)
    assert np.array_equal(cm.edge_index_table_to_host(), G.i32("tb_edge_index_table"))
    w = [torch.from_numpy(np.array(t)).to(dev) for t in G.gine_weights()]
    ea = torch.from_numpy(G.edge_features()).to(dev)
    y = cm.gine_conv(xd, ea, *w, eps=G.conv_kwargs("gine")["eps"])
    torch.cuda.synchronize()
    assert np.abs(y.cpu().numpy() - G.f32("tb_gine_output", (G.N, G.F))).max() < 2e-6


def test_gine_conv_on_batches_matches_oracle(dev):
    """GINE over a batch with degenerate graphs (isolated nodes, E = 0, self loops, duplicate edges, a hub) and molhiv-
    sized ones, widths that are / are not multiples of 4, against the oracle graph by graph."""
    rng = np.random.default_rng(11)
    for width, edim in ((8, 16), (32, 5), (7, 3)):
        small = edge_case_batch()
        big = synthetic.make_batch("molhiv", 20, seed=width)
        graphs = [(rng.uniform(-1, 1, (small.graph(g)[0].shape[0], width)).astype(np.float32), small.graph(g)[1])
                  for g in range(small.num_graphs)]
        graphs += [(rng.uniform(-1, 1, (big.graph(g)[0].shape[0], width)).astype(np.float32), big.graph(g)[1])
                   for g in range(big.num_graphs)]
        batch = pack_graphs(graphs)
        ea = rng.uniform(-1, 1, (batch.num_edges, edim)).astype(np.float32)
        ws = [rng.uniform(-0.5, 0.5, s).astype(np.float32) for s in ((width, edim), (width,), (16, width), (16,), (12, 16), (12,))]
        cm = runtime.CompiledModel.from_model(plain_model("gin", width, 8), batch.num_graphs, batch.num_nodes, batch.num_edges)
        xd, cood, nptr, eptr = to_dev(batch, dev)
        cm.graph_prep(cood, nptr, eptr, batch.num_nodes)
        y = cm.gine_conv(xd, torch.from_numpy(ea).to(dev), *[torch.from_numpy(t).to(dev) for t in ws], eps=0.3).cpu().numpy()
        ref = []
        for g in range(batch.num_graphs):
            xg, cg = batch.graph(g)
            if xg.shape[0]:
                ref.append(O.gine_conv(xg, cg, ea[batch.edge_ptr[g]:batch.edge_ptr[g + 1]], ws, eps=0.3))
        assert np.abs(y - np.concatenate(ref)).max() < 1e-5, (width, edim)


def test_weight_free_convs_on_degenerate_and_large_graphs(dev):
    """LG / Simple / copy on isolated nodes, empty graphs, hubs and graphs beyond an LDS stage, vs the oracle."""
    batch = edge_case_batch()
    big = synthetic.make_batch("molhiv", 30, seed=5)
    both = pack_graphs([batch.graph(g) for g in range(batch.num_graphs)] +
                       [(big.graph(g)[0][:, :8], big.graph(g)[1]) for g in range(big.num_graphs)])
    cm = runtime.CompiledModel.from_model(plain_model("gin", 8, 8), both.num_graphs, both.num_nodes, both.num_edges)
    xd, cood, nptr, eptr = to_dev(both, dev)
    cm.graph_prep(cood, nptr, eptr, both.num_nodes)
    for kind in ("simple", "lg"):
        ref = np.concatenate([O.conv(kind, *both.graph(g), []) for g in range(both.num_graphs) if both.graph(g)[0].shape[0]])
        try:
            for opts in (dict(), dict(agg_lds_kb=8), dict(agg_ring_waves=2, agg_ring_slots=3)):
                for k, v in opts.items():
                    runtime.set_option(k, v)
                got = cm.aggregate(kind, xd).cpu().numpy()
                assert np.abs(got - ref).max() < 2e-6, (kind, opts)
        finally:
            runtime.set_option("agg_lds_kb", 0)
            runtime.set_option("agg_ring_waves", 0)
            runtime.set_option("agg_ring_slots", 2)
    assert torch.equal(cm.aggregate("copy", xd), xd)


@pytest.mark.parametrize("shape,graphs,width", [("molhiv_tail", 2500, 128), ("qm9", 4096, 128), ("molhiv", 1500, 20), ("qm9", 3000, 256)])
def test_row_balanced_aggregate_ranges_change_nothing(dev, shape, graphs, width):
    """Round 4 (opt-in, `agg_balance`): the ring kernel's workgroups take an exact 1 / grid share of the ROWS (graph prep's
    cut table; the boundary graph is staged by both neighbours, each reduces its own rows) instead of whole-graph runs.
    Every aggregate kind, on batches large enough to have a range per workgroup (incl. graphs larger than an LDS stage: the
    direct path is clipped too), float4 and scalar rows: identical to the whole-graph ranges -- bit for bit where a row's
    arithmetic does not depend on whether its graph was staged in LDS or read from L2 (every kind but the two with
    coefficients, GCN and LG: there the staged path multiplies by a precomputed coefficient) --, and SUM / MEAN equal to a
    plain scatter formulation in fp64 to fp32 rounding."""
    batch = synthetic.make_batch(shape, graphs, seed=13)
    rng = np.random.default_rng(width)
    x = rng.uniform(-1, 1, (batch.num_nodes, width)).astype(np.float32)
    q = rng.uniform(-1, 1, (batch.num_nodes, width)).astype(np.float32)
    outs = {}
    try:
        for bal in (1, 0):
            runtime.set_option("agg_balance", bal)
            cm = runtime.CompiledModel.from_model(plain_model("pna", 8, 8), batch.num_graphs, batch.num_nodes, batch.num_edges)
            _, cood, nptr, eptr = to_dev(batch, dev)
            cm.graph_prep(cood, nptr, eptr, batch.num_nodes)
            xd, qd = torch.from_numpy(x).to(dev), torch.from_numpy(q).to(dev)
            for kind in ("gcn", "sum", "mean", "pna", "lg", "simple", "copy"):
                outs[(kind, bal)] = cm.aggregate(kind, xd, self_term=qd if kind == "pna" else None, eps=0.25).cpu().numpy()
            cm.check()
    finally:
        runtime.set_option("agg_balance", 0)
    for kind in ("sum", "mean", "pna", "simple", "copy"):
        assert np.array_equal(outs[(kind, 1)], outs[(kind, 0)]), kind
    for kind in ("gcn", "lg"):
        assert np.abs(outs[(kind, 1)] - outs[(kind, 0)]).max() < 1e-6, kind
    assert np.array_equal(outs[("copy", 1)], x)
    # independent check of two kinds in float64 (scatter form)
    src, dst = batch.coo[:, 0], batch.coo[:, 1]
    deg = np.bincount(dst, minlength=batch.num_nodes).astype(np.float64)
    mean = np.zeros((batch.num_nodes, width))
    np.add.at(mean, dst, x[src].astype(np.float64))
    ssum = mean + 1.25 * x
    mean = np.where(deg[:, None] > 0, mean / np.maximum(deg, 1)[:, None], 0.0)
    assert np.abs(outs[("mean", 1)] - mean).max() < 1e-5 and np.abs(outs[("sum", 1)] - ssum).max() < 2e-5


@pytest.mark.parametrize("hidden,layers,pools,act", [(256, 2, ("add", "mean", "max"), "relu"), (128, 3, ("max", "add"), "tanh"),
                                                     (256, 1, ("mean",), "sigmoid"), (128, 2, ("add", "mean", "max"), "gelu")])
def test_pooling_in_the_last_gemm_epilogue(dev, hidden, layers, pools, act):
    """GraphSAGE: the last layer's large-K GEMM pools in its epilogue (32-row blocks, graphs crossing blocks combined in
    row order by k_pool_combine) and never writes its [N, d] output -- against the separate pooling pass (fuse_pool = 0)
    and the oracle, EVERY graph.  Batch: heavy-tailed sizes (graphs of 100+ rows span several blocks and tiles), runs of
    one- and two-node graphs (many graphs inside one block), graphs that start exactly on a block boundary, empty graphs at
    both ends and in the middle, a row count that is no multiple of the 128-row tile; and the same graphs in reverse order."""
    model = make_model("sage", in_dim=9, hidden=hidden, layers=layers, act=act, pools=pools, task_out=3, seed=hidden + layers)
    rng = np.random.default_rng(layers)
    base = synthetic.make_batch("molhiv_tail", 700, seed=17)
    empty = (np.zeros((0, 9), np.float32), np.zeros((0, 2), np.int32))

    def chain(k):  # a path graph of k nodes
        e = np.array([[i, i + 1] for i in range(k - 1)] + [[i + 1, i] for i in range(k - 1)], np.int32).reshape(-1, 2)
        return rng.uniform(-1, 1, (k, 9)).astype(np.float32), e

    tiny = [chain(k) for k in (1, 2, 1, 1, 3, 2, 1, 1, 1, 2) * 4]
    aligned = [chain(32), chain(64), chain(40), chain(24), chain(96), chain(33)]  # (the first starts on row 0: block-aligned heads)
    graphs = aligned + [empty, empty] + [base.graph(g) for g in range(350)] + tiny + [empty] + [base.graph(g) for g in range(350, 700)] + [empty]
    for order in (1, -1):
        batch = pack_graphs(graphs[::order])
        assert batch.num_nodes % 128 != 0 and int(np.diff(batch.node_ptr).max()) >= 96
        ref = O.forward_batched(model.spec(), canon(model), batch.x, batch.coo, batch.node_ptr, batch.edge_ptr)
        outs = {}
        try:
            for fuse in (1, 0):
                runtime.set_option("fuse_pool", fuse)
                cm = runtime.CompiledModel.from_model(model, batch.num_graphs, batch.num_nodes, batch.num_edges)
                outs[fuse] = cm.forward(*to_dev(batch, dev)).cpu().numpy()
                cm.check()
                outs[(fuse, 2)] = cm.forward(*to_dev(batch, dev)).cpu().numpy()  # (a second forward on the same workspace)
        finally:
            runtime.set_option("fuse_pool", 1)
        scale = max(1.0, float(np.abs(ref).max()))
        assert np.isfinite(outs[1]).all()
        assert np.abs(outs[1] - ref).max() < TOL * scale and np.abs(outs[0] - ref).max() < TOL * scale
        assert np.abs(outs[1] - outs[0]).max() < 2e-5 * scale
        assert np.array_equal(outs[1], outs[(1, 2)])  # deterministic: the same batch gives the same bits


@pytest.mark.parametrize("hidden,out,layers,pools,act,skip", [(128, 128, 3, ("add", "mean", "max"), "relu", True), (128, 64, 2, ("max", "add"), "tanh", False),
                                                              (32, 32, 3, ("mean",), "gelu", True), (128, 128, 1, ("add",), "sigmoid", False),
                                                              (64, 128, 4, ("add", "mean", "max"), "relu", True)])
def test_pna_lin_folded_into_the_post_nn(dev, hidden, out, layers, pools, act, skip):
    """PNA (round 4): `lin` folded into the post-NN at upload -- one 13F-wide GEMM per layer, skip + activation in its
    epilogue, the last layer of 128-wide models pooling there too -- against the reference's two products (pna_fold_lin = 0),
    against the separate pooling pass, and the oracle, every graph.  Heavy-tailed batch with empty graphs, graphs crossing the
    32-row pooling blocks, isolated nodes (degree 0: amplification 0, attenuation on the clamped degree)."""
    model = make_model("pna", in_dim=11, hidden=hidden, out_dim=out, layers=layers, act=act, pools=pools, task_out=2, skip=skip, seed=hidden + layers)
    base = synthetic.make_batch("molhiv_tail", 300, seed=5)
    empty = (np.zeros((0, 11), np.float32), np.zeros((0, 2), np.int32))
    rng = np.random.default_rng(layers)

    def regraph(g):  # the same topology with 11 features
        x, e = base.graph(g)
        return rng.uniform(-1, 1, (x.shape[0], 11)).astype(np.float32), e

    lone = (rng.uniform(-1, 1, (3, 11)).astype(np.float32), np.zeros((0, 2), np.int32))  # three isolated nodes
    graphs = [empty] + [regraph(g) for g in range(150)] + [lone, empty] + [regraph(g) for g in range(150, 300)] + [empty]
    batch = pack_graphs(graphs)
    ref = O.forward_batched(model.spec(), canon(model), batch.x, batch.coo, batch.node_ptr, batch.edge_ptr)
    outs = {}
    try:
        for fold, fuse in ((1, 1), (0, 1), (1, 0)):
            runtime.set_option("pna_fold_lin", fold)
            runtime.set_option("fuse_pool", fuse)
            cm = runtime.CompiledModel.from_model(model, batch.num_graphs, batch.num_nodes, batch.num_edges)
            outs[(fold, fuse)] = cm.forward(*to_dev(batch, dev)).cpu().numpy()
            cm.check()
    finally:
        runtime.set_option("pna_fold_lin", 1)
        runtime.set_option("fuse_pool", 1)
    scale = max(1.0, float(np.abs(ref).max()))
    for k, v in outs.items():
        assert np.isfinite(v).all() and np.abs(v - ref).max() < TOL * scale, k
    assert np.abs(outs[(1, 1)] - outs[(0, 1)]).max() < 3e-5 * scale and np.abs(outs[(1, 1)] - outs[(1, 0)]).max() < 3e-5 * scale


@pytest.mark.parametrize("hidden,out,layers,pools,act,skip,fin", [(128, 128, 3, ("add", "mean", "max"), "relu", True, 11), (128, 64, 2, ("max", "add"), "tanh", False, 32),
                                                                  (64, 128, 4, ("mean",), "gelu", True, 64), (128, 96, 1, ("add",), "sigmoid", False, 128),
                                                                  (32, 128, 3, ("add", "mean", "max"), "relu", True, 32),
                                                                  (128, 48, 3, ("add", "max"), "relu", True, 11), (64, 36, 2, ("mean",), "tanh", False, 20)])
def test_pna_degree_classes(dev, hidden, out, layers, pools, act, skip, fin):
    """PNA under a max_degree promise (round 4): the 13 F-wide post-NN product as the 5 F-wide [x | A] . W_class^T over rows
    sorted into degree classes (one pre-combined matrix per in-degree 1 .. 15) -- against the general form (no promise),
    against pna_classes = 0, and the oracle, every graph.  Batch: molecules, isolated nodes (degree 0 counts as 1), a node
    of degree exactly the promise, empty graphs, a row count that is no multiple of the tile; layers whose widths do not
    take the form (out <= 32) run the general one in the same model; 33 .. 64 output columns and input widths that are no whole
    32-wide chunks take it through the narrow / generic kernels.  A broken promise is flagged (32)."""
    model = make_model("pna", in_dim=fin, hidden=hidden, out_dim=out, layers=layers, act=act, pools=pools, task_out=2, skip=skip, seed=hidden + layers + fin)
    base = synthetic.make_batch("qm9", 600, seed=9)
    rng = np.random.default_rng(fin)
    empty = (np.zeros((0, fin), np.float32), np.zeros((0, 2), np.int32))

    def regraph(g):
        x, e = base.graph(g)
        return rng.uniform(-1, 1, (x.shape[0], fin)).astype(np.float32), e

    lone = (rng.uniform(-1, 1, (3, fin)).astype(np.float32), np.zeros((0, 2), np.int32))
    star = (rng.uniform(-1, 1, (14, fin)).astype(np.float32), np.array([[i, 0] for i in range(1, 14)] + [[0, i] for i in range(1, 14)], np.int32))
    graphs = [empty] + [regraph(g) for g in range(300)] + [lone, star, empty] + [regraph(g) for g in range(300, 600)] + [empty]
    batch = pack_graphs(graphs)
    maxdeg = int(np.bincount(batch.coo[:, 1]).max())
    assert 13 <= maxdeg <= 15 and batch.num_nodes % 128 != 0
    ref = O.forward_batched(model.spec(), canon(model), batch.x, batch.coo, batch.node_ptr, batch.edge_ptr)
    outs = {}
    try:
        for promise, classes in ((maxdeg, 1), (0, 1), (maxdeg, 0), (15, 1)):
            runtime.set_option("pna_classes", classes)
            cm = runtime.CompiledModel.from_model(model, batch.num_graphs, batch.num_nodes, batch.num_edges)
            cm.set_max_degree(promise)
            outs[(promise, classes)] = cm.forward(*to_dev(batch, dev)).cpu().numpy()
            cm.check()
            if (promise, classes) == (maxdeg, 1):
                again = cm.forward(*to_dev(batch, dev)).cpu().numpy()
                assert np.array_equal(outs[(promise, classes)], again)  # (the order inside a class does not reach the results)
    finally:
        runtime.set_option("pna_classes", 1)
    scale = max(1.0, float(np.abs(ref).max()))
    for k, v in outs.items():
        assert np.isfinite(v).all() and np.abs(v - ref).max() < TOL * scale, k
    assert np.abs(outs[(maxdeg, 1)] - outs[(0, 1)]).max() < 3e-5 * scale
    assert np.array_equal(outs[(0, 1)], outs[(maxdeg, 0)])
    # a promise the batch breaks: flagged by graph prep
    cm = runtime.CompiledModel.from_model(model, batch.num_graphs, batch.num_nodes, batch.num_edges)
    cm.set_max_degree(maxdeg - 1)
    cm.forward(*to_dev(batch, dev))
    with pytest.raises(runtime.GnnbError, match="0x20"):
        cm.check()


@pytest.mark.parametrize("conv,fin,hidden,act", [("sage", 9, 256, "relu"), ("sage", 16, 100, "tanh"), ("gcn", 11, 128, "relu"),
                                                 ("gcn", 20, 64, "gelu"), ("gin", 9, 128, "relu"), ("gin", 32, 256, "sigmoid"),
                                                 ("sage", 4, 16, "relu")])
def test_narrow_first_layer_in_ring_form(dev, conv, fin, hidden, act):
    """k_conv_first (round 4): the narrow first layer with whole graphs staged in LDS and all output columns from one stage,
    against the round-3 form (inside k_linear_reg's A stage, `first_ring` = 0), the unfused form and the oracle.  Batch:
    molecules, graphs of 130-300 nodes (beyond a stage: taken in pieces from L2) incl. a hub of degree 200, isolated nodes,
    one-node and empty graphs at both ends, duplicate edges and self loops."""
    model = make_model(conv, in_dim=fin, hidden=hidden, layers=2, act=act, pools=("add", "max"), task_out=3, seed=fin + hidden)
    rng = np.random.default_rng(fin)
    base = synthetic.make_batch("molhiv_tail", 260, seed=fin)
    empty = (np.zeros((0, fin), np.float32), np.zeros((0, 2), np.int32))

    def rnd(n, e, hub=False):
        coo = np.stack([rng.integers(0, n, e), rng.integers(0, n, e)], 1).astype(np.int32)
        if hub:
            coo = np.concatenate([coo, np.stack([np.arange(1, 201), np.zeros(200, np.int64)], 1).astype(np.int32)])
        return rng.uniform(-1, 1, (n, fin)).astype(np.float32), coo

    graphs = [empty, rnd(1, 0)] + [(rng.uniform(-1, 1, (base.graph(g)[0].shape[0], fin)).astype(np.float32), base.graph(g)[1]) for g in range(130)]
    graphs += [rnd(300, 900, hub=True), rnd(131, 260), empty, rnd(129, 10)]
    graphs += [(rng.uniform(-1, 1, (base.graph(g)[0].shape[0], fin)).astype(np.float32), base.graph(g)[1]) for g in range(130, 260)] + [rnd(2, 5), empty]
    batch = pack_graphs(graphs)
    ref = O.forward_batched(model.spec(), canon(model), batch.x, batch.coo, batch.node_ptr, batch.edge_ptr)
    outs = {}
    try:
        for ring, narrow in ((1, 1), (0, 1), (0, 0)):
            runtime.set_option("first_ring", ring)
            runtime.set_option("fuse_narrow", narrow)
            cm = runtime.CompiledModel.from_model(model, batch.num_graphs, batch.num_nodes, batch.num_edges)
            outs[(ring, narrow)] = cm.forward(*to_dev(batch, dev)).cpu().numpy()
            cm.check()
            assert cm.last_path() == "layerwise"
    finally:
        runtime.set_option("first_ring", 1)
        runtime.set_option("fuse_narrow", 1)
    scale = max(1.0, float(np.abs(ref).max()))
    for k, v in outs.items():
        assert np.isfinite(v).all() and np.abs(v - ref).max() < TOL * scale, k
    assert np.abs(outs[(1, 1)] - outs[(0, 0)]).max() < 2e-5 * scale


def test_malformed_batch_is_reported(dev):
    batch = synthetic.make_batch("qm9", 8, 0)
    bad = batch.coo.copy()
    bad[3, 0] = batch.num_nodes - 1  # edge of graph 0 pointing into the last graph
    cm = runtime.CompiledModel.from_model(plain_model("gcn", 11, 8), 8, batch.num_nodes, batch.num_edges)
    _, _, nptr, eptr = to_dev(batch, dev)
    cm.graph_prep(torch.from_numpy(bad).to(dev), nptr, eptr, batch.num_nodes)
    with pytest.raises(runtime.GnnbError):
        cm.check()


def test_flagged_batch_is_reported_lazily_without_a_check(dev):
    """A caller that never calls check(): the forward after a flagged batch has RUN raises (host-mapped flag read without
    synchronisation) and clears the flags; a flag is reported once, by whichever comes first -- that lazy report or check().  Covers a
    broken max_graph_nodes promise on the fused GCN stack, the case the advisor singled out."""
    model = make_model("gcn", in_dim=11, hidden=32, layers=2, task_out=1)
    good = synthetic.make_batch("qm9", 64, seed=0)
    cm = runtime.CompiledModel.from_model(model, good.num_graphs, good.num_nodes, good.num_edges,
                                          max_graph_nodes=int(np.diff(good.node_ptr).max()) - 3)   # promise too small
    args = to_dev(good, dev)
    cm.forward(*args)                    # runs (contained), flags the batch on the device
    torch.cuda.synchronize()
    with pytest.raises(runtime.GnnbError, match="earlier batch"):
        cm.forward(*args)                # (reported; the batch of THIS call was not enqueued)
    cm.check()                           # reported once: the lazy report cleared the device flag too (advisor, round 2)
    cm.forward(*args)                    # flagged again ...
    with pytest.raises(runtime.GnnbError, match="malformed batch"):
        cm.check()                       # ... and this time the check is what reports (and clears) it
    cm.set_max_graph_nodes(int(np.diff(good.node_ptr).max()))
    out = cm.forward(*args).cpu().numpy()
    cm.check()
    ref = O.forward_batched(model.spec(), canon(model), good.x, good.coo, good.node_ptr, good.edge_ptr)
    assert np.abs(out - ref).max() < TOL * max(1.0, float(np.abs(ref).max()))


def test_capacity_is_checked(dev):
    batch = synthetic.make_batch("qm9", 8, 0)
    cm = runtime.CompiledModel.from_model(plain_model("gcn", 11, 8), 4, batch.num_nodes, batch.num_edges)
    with pytest.raises(runtime.GnnbError, match="exceeds workspace"):
        cm.forward(*to_dev(batch, dev))


# --------------------------------------------------------------------------- single conv layers vs the PyG goldens
@pytest.mark.parametrize("kind", ["gcn", "gin", "sage", "pna"])
def test_conv_layer_matches_reference_golden(dev, kind):
    """One conv layer on the reference's 100-node fixture graph, weights from its tb_data,
    expected output = its PyG-generated golden (test.cpp:1056-1726 accepts 1e-3 / 1e-2)."""
    x, coo = G.graph()
    batch = pack_graphs([(x, coo)])
    w = [torch.from_numpy(np.array(t)).to(dev) for t in G.conv_weights(kind)]
    cm = runtime.CompiledModel.from_model(plain_model(kind, 8, 8), 1, G.N, G.E)
    xd, cood, nptr, eptr = to_dev(batch, dev)
    cm.desc.pna_delta = G.conv_kwargs("pna")["delta"] if kind == "pna" else 1.0
    cm.graph_prep(cood, nptr, eptr, G.N)
    if kind == "gcn":
        y = runtime.linear([(cm.aggregate("gcn", xd), None)], w[0], w[1])
    elif kind == "gin":
        z = cm.aggregate("sum", xd, eps=G.conv_kwargs("gin")["eps"])
        y = runtime.linear([(runtime.linear([(z, None)], w[0], w[1], act="relu"), None)], w[2], w[3])
    elif kind == "sage":
        m = cm.aggregate("mean", xd)
        y = runtime.linear([(m, None), (xd, None)], torch.cat([w[0], w[2]], 1).contiguous(), w[1])
    else:
        wpre = w[0]
        q = runtime.linear([(xd, None)], wpre[:, :8], w[1])
        p = runtime.linear([(xd, None)], wpre[:, 8:], None)
        a = cm.aggregate("pna", p, self_term=q)
        deg = torch.from_numpy(G.i32("tb_in_degree_table")).to(dev).clamp(min=1).float()
        delta = G.conv_kwargs("pna")["delta"]
        amp = (torch.log(deg + 1) / delta).contiguous()
        att = (delta / torch.log(deg + 1)).contiguous()
        hid = runtime.linear([(xd, None), (a, None), (a, amp), (a, att)], w[2], w[3])
        y = runtime.linear([(hid, None)], w[4], w[5])
    torch.cuda.synchronize()
    assert np.abs(y.cpu().numpy() - G.conv_golden(kind)).max() < 2e-6


# --------------------------------------------------------------------------- dense update vs a torch fp32 reference
@pytest.mark.parametrize("M,N,K", [(1, 1, 1), (37, 19, 11), (128, 64, 32), (300, 128, 128), (257, 130, 100),
                                   (1000, 256, 143), (4096, 64, 384)])
@pytest.mark.parametrize("act", ["none", "relu", "gelu", "sigmoid", "tanh"])
def test_linear_matches_torch(dev, M, N, K, act):
    g = torch.Generator().manual_seed(M * 131 + N * 7 + K)
    a = torch.rand(M, K, generator=g) * 2 - 1
    w = (torch.rand(N, K, generator=g) * 2 - 1) / max(K, 1) ** 0.5
    b = torch.rand(N, generator=g) * 2 - 1
    skip = torch.rand(M, N, generator=g) * 2 - 1
    ref = a.double() @ w.double().T + b.double() + skip.double()
    ref = {"none": lambda t: t, "relu": torch.relu, "gelu": lambda t: torch.nn.functional.gelu(t),
           "sigmoid": torch.sigmoid, "tanh": torch.tanh}[act](ref).float()
    y = runtime.linear([(a.to(dev), None)], w.to(dev), b.to(dev), skip=skip.to(dev), act=act)
    torch.cuda.synchronize()
    assert (y.cpu() - ref).abs().max().item() < 2e-5


@pytest.mark.parametrize("M", [1, 15, 16, 17, 1000, 4097, 73763])
@pytest.mark.parametrize("N,K", [(128, 128), (64, 128), (128, 64), (64, 64)])
def test_linear_weights_in_lds_kernel(dev, M, N, K):
    """k_linear_wlds (K, N in {64, 128}: weights in LDS, per-wave rings, swapped MFMA operands) vs float64, with and
    without the skip operand, every activation on the small shapes; and bit-identical to the register-resident
    kernel's result up to the fp32 summation order (both accumulate k in the same MFMA order: exact equality)."""
    g = torch.Generator().manual_seed(M * 7 + N + K)
    a = torch.rand(M, K, generator=g) * 2 - 1
    w = (torch.rand(N, K, generator=g) * 2 - 1) / K ** 0.5
    b = torch.rand(N, generator=g) * 2 - 1
    skip = torch.rand(M, N, generator=g) * 2 - 1
    ad, wd, bd, sd = a.to(dev), w.to(dev), b.to(dev), skip.to(dev)
    acts = ("relu", "none", "gelu", "sigmoid", "tanh") if M <= 1000 else ("relu",)
    fn = {"none": lambda t: t, "relu": torch.relu, "gelu": torch.nn.functional.gelu, "sigmoid": torch.sigmoid, "tanh": torch.tanh}
    for act in acts:
        for use_skip in (False, True):
            ref = a.double() @ w.double().T + b.double() + (skip.double() if use_skip else 0)
            ref = fn[act](ref).float()
            try:
                runtime.set_option("gemm_wlds", 1)
                y = runtime.linear([(ad, None)], wd, bd, skip=sd if use_skip else None, act=act)
                runtime.set_option("gemm_wlds", 0)
                y0 = runtime.linear([(ad, None)], wd, bd, skip=sd if use_skip else None, act=act)
            finally:
                runtime.set_option("gemm_wlds", 1)
            torch.cuda.synchronize()
            assert (y.cpu() - ref).abs().max().item() < 2e-5, (act, use_skip)
            assert (y - y0).abs().max().item() < 2e-6, (act, use_skip)
    # no bias
    y = runtime.linear([(ad, None)], wd, None, act="none")
    assert (y.cpu() - (a.double() @ w.double().T).float()).abs().max().item() < 2e-5


@pytest.mark.parametrize("M,N,K", [(37, 19, 11), (300, 128, 128), (129, 64, 64), (1000, 33, 16), (50, 200, 36)])
def test_linear_lds_tiled_kernel_on_small_k(dev, M, N, K):
    """K <= 128 normally takes the register-resident-weight kernel; force the LDS-tiled one too."""
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.rand(M, K, generator=g) - 0.5
    w = torch.rand(N, K, generator=g) - 0.5
    b = torch.rand(N, generator=g)
    ref = (a.double() @ w.double().T + b.double()).float()
    try:
        runtime.set_option("gemm_variant", 1)
        y1 = runtime.linear([(a.to(dev), None)], w.to(dev), b.to(dev)).cpu()
    finally:
        runtime.set_option("gemm_variant", 0)
    y0 = runtime.linear([(a.to(dev), None)], w.to(dev), b.to(dev)).cpu()
    assert (y1 - ref).abs().max().item() < 2e-5 and (y0 - ref).abs().max().item() < 2e-5


def test_linear_segments_and_rowscale(dev):
    g = torch.Generator().manual_seed(7)
    M, N = 333, 96
    ks = [11, 44, 44, 44]  # the PNA shape at F=11: [x | A | amp.A | att.A]
    x = torch.rand(M, 11, generator=g) - 0.5
    A = torch.rand(M, 44, generator=g) - 0.5
    amp, att = torch.rand(M, generator=g) + 0.5, torch.rand(M, generator=g) + 0.5
    w = (torch.rand(N, sum(ks), generator=g) - 0.5) / 6
    b = torch.rand(N, generator=g)
    cat = torch.cat([x, A, A * amp[:, None], A * att[:, None]], 1).double()
    ref = (cat @ w.double().T + b.double()).float()
    Ad = A.to(dev)
    y = runtime.linear([(x.to(dev), None), (Ad, None), (Ad, amp.to(dev)), (Ad, att.to(dev))], w.to(dev), b.to(dev))
    torch.cuda.synchronize()
    assert (y.cpu() - ref).abs().max().item() < 2e-5


# --------------------------------------------------------------------------- pooling
@pytest.mark.parametrize("d", [8, 64, 128, 100, 7])
def test_global_pool_matches_oracle(dev, d):
    batch = synthetic.make_batch("molhiv", 50, 4)
    rng = np.random.default_rng(d)
    h = rng.uniform(-1, 1, (batch.num_nodes, d)).astype(np.float32)
    cm = runtime.CompiledModel.from_model(plain_model("gcn", 9, 8), 50, batch.num_nodes, batch.num_edges)
    _, coo, nptr, eptr = to_dev(batch, dev)
    cm.graph_prep(coo, nptr, eptr, batch.num_nodes)
    pools = ["max", "add", "mean"]
    out = cm.global_pool(torch.from_numpy(h).to(dev), pools)
    torch.cuda.synchronize()
    ref = np.stack([np.concatenate([O.global_pool(h[batch.node_ptr[g]:batch.node_ptr[g + 1]], k) for k in pools])
                    for g in range(50)])
    assert np.abs(out.cpu().numpy() - ref).max() < 1e-5


# --------------------------------------------------------------------------- whole models vs the oracle
MODEL_CASES = [
    # conv, in, hidden, layers, act, skip, pools, shape, graphs
    ("gcn", 11, 128, 2, "relu", True, ("add", "mean", "max"), "qm9", 256),      # BASELINE config 2 (shape)
    ("gcn", 9, 64, 2, "relu", True, ("add", "mean", "max"), "esol", 40),        # BASELINE config 1 (shape)
    ("gin", 9, 128, 3, "relu", True, ("add",), "molhiv", 128),                  # config 3
    ("pna", 11, 128, 3, "relu", True, ("add", "mean", "max"), "qm9", 96),       # config 4
    ("sage", 9, 256, 2, "relu", True, ("add", "mean", "max"), "molhiv", 128),   # config 5
    ("gcn", 11, 32, 4, "tanh", True, ("max", "add"), "qm9", 64),
    ("gin", 11, 48, 4, "gelu", True, ("mean",), "qm9", 64),
    ("sage", 9, 40, 3, "sigmoid", False, ("mean", "max"), "molhiv", 64),
    ("pna", 9, 24, 2, "tanh", False, ("max",), "molhiv", 48),
    ("sage", 11, 16, 1, "relu", True, ("add",), "qm9", 32),
    ("gcn", 11, 11, 0, "relu", False, ("add", "max"), "qm9", 32),               # gnn_num_layers = 0
]


@pytest.mark.parametrize("case", MODEL_CASES, ids=lambda c: f"{c[0]}-L{c[3]}-d{c[2]}-{c[4]}")
def test_whole_model_matches_oracle(dev, case):
    conv, fin, hidden, layers, act, skip, pools, shape, ngraphs = case
    model = make_model(conv, in_dim=fin, hidden=hidden, layers=layers, act=act, skip=skip, pools=pools,
                       task_out=synthetic.SHAPES[shape]["out"], seed=layers * 17 + hidden)
    batch = synthetic.make_batch(shape, ngraphs, seed=hidden)
    cm = runtime.CompiledModel.from_model(model, batch.num_graphs, batch.num_nodes, batch.num_edges)
    out = cm.forward(*to_dev(batch, dev))
    cm.check()
    torch.cuda.synchronize()
    ref = O.forward_batched(model.spec(), canon(model), batch.x, batch.coo, batch.node_ptr, batch.edge_ptr)
    got = out.cpu().numpy()
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() < TOL, f"max err {np.abs(got - ref).max():.3e}, |ref| {np.abs(ref).max():.3e}"


@pytest.mark.parametrize("conv,promise", [("gcn", 0), ("gcn", 65), ("gin", 0), ("sage", 0), ("pna", 0)])
def test_molecule_path_corners_through_whole_models(dev, conv, promise):
    """Node records (first four sources, degrees), normalisers and PNA scalers written by graph prep's molecule path on
    its corner cases (hubs of degree 64, 64 copies of one edge, self loops only ...), checked through whole models --
    layer by layer and, with a promise, through the conv-stack kernel -- against the oracle."""
    batch = molecule_path_batch()
    model = make_model(conv, in_dim=8, hidden=32, layers=2, act="relu", skip=True, pools=("add", "mean", "max"), task_out=3, seed=7)
    cm = runtime.CompiledModel.from_model(model, batch.num_graphs, batch.num_nodes, batch.num_edges, max_graph_nodes=promise)
    got = cm.forward(*to_dev(batch, dev)).cpu().numpy()
    cm.check()
    ref = O.forward_batched(model.spec(), canon(model), batch.x, batch.coo, batch.node_ptr, batch.edge_ptr)
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() < TOL * max(1.0, float(np.abs(ref).max()))
    if promise:
        assert cm.last_path().startswith("stack")


@pytest.mark.parametrize("cls", [torch.nn.Softmax, torch.nn.LogSoftmax])
@pytest.mark.parametrize("conv,promise", [("gcn", 29), ("gcn", 0), ("sage", 0)])
def test_output_activation(dev, cls, conv, promise):
    """GNNModel.output_activation (softmax / log_softmax over each graph's output row) on every route of the forward."""
    import gnnbuilder_amd as gnnb

    torch.manual_seed(5)
    from helpers import CONVS
    model = gnnb.GNNModel(11, None, 64, 2, 64, CONVS[conv], torch.nn.ReLU, True, gnnb.GlobalPooling(["add", "mean", "max"]),
                          gnnb.MLP(192, 19, 64, 2), cls).eval()
    batch = synthetic.make_batch("qm9", 70, seed=2)
    cm = runtime.CompiledModel.from_model(model, batch.num_graphs, batch.num_nodes, batch.num_edges, max_graph_nodes=promise)
    out = cm.forward(*to_dev(batch, dev)).cpu().numpy()
    cm.check()
    ref = O.forward_batched(model.spec(), canon(model), batch.x, batch.coo, batch.node_ptr, batch.edge_ptr)
    assert np.abs(out - ref).max() < TOL


@pytest.mark.parametrize("conv", ["gcn", "gin", "sage", "pna"])
@pytest.mark.parametrize("W,I", [(32, 12), (16, 8), (12, 6)])
def test_fixed_point_emulation(dev, conv, W, I):
    """gnnb_model_desc.fpx_w / fpx_i (the reference's float_or_fixed = "fixed" with FPX(W, I)): the HIP forward against
    the oracle's emulation.  Both quantise the same tensors; their fp32 sums differ in order, so a value that lands
    within rounding of a grid line may fall one step apart (and a step apart at one layer moves later ones)."""
    model = make_model(conv, in_dim=9, hidden=32, layers=3, task_out=3)
    batch = synthetic.make_batch("molhiv", 24, seed=W)
    spec = dict(model.spec(), fpx=(W, I))
    ref = O.forward_batched(spec, canon(model), batch.x, batch.coo, batch.node_ptr, batch.edge_ptr)
    cm = runtime.CompiledModel.from_model(model, batch.num_graphs, batch.num_nodes, batch.num_edges, fpx=(W, I))
    out = cm.forward(*to_dev(batch, dev)).cpu().numpy()
    cm.check()
    step = 2.0 ** -(W - I)
    assert np.abs(out / step - np.round(out / step)).max() < 1e-3              # on the grid
    frac_exact = np.mean(out == ref)
    # measured over six seeds per case (tests/fpx_stats.py): never more than ONE grid step apart, >= 95 % of the outputs
    # identical for W <= 16; at FPX(32, 12) a step (2^-20) is the size of an fp32 rounding of the values themselves, so
    # the share of identical outputs is lower there (>= 76 % measured) while the distance stays one step
    assert np.abs(out - ref).max() <= 2 * step, (np.abs(out - ref).max() / step, frac_exact)
    assert frac_exact >= (0.95 if W <= 16 else 0.6), frac_exact
    # fine grid == the float model
    if W == 32:
        flt = O.forward_batched(model.spec(), canon(model), batch.x, batch.coo, batch.node_ptr, batch.edge_ptr)
        assert np.abs(out - flt).max() < 1e-3


def test_host_entry_and_single_graph(dev):
    """forward_batched_host (what <name>_top uses) == device entry; a graph alone == inside a batch."""
    model = make_model("gcn", hidden=64)
    batch = synthetic.make_batch("qm9", 33, seed=9)
    cm = runtime.CompiledModel.from_model(model, batch.num_graphs, batch.num_nodes, batch.num_edges)
    out_d = cm.forward(*to_dev(batch, dev)).cpu().numpy()
    out_h = cm.forward_host(batch.x, batch.coo, batch.node_ptr, batch.edge_ptr)
    assert np.array_equal(out_d, out_h)
    one = batch.slice(7, 8)
    out_1 = cm.forward_host(one.x, one.coo, one.node_ptr, one.edge_ptr)
    assert np.abs(out_1[0] - out_d[7]).max() < 1e-6


def test_degenerate_graphs(dev):
    """Isolated nodes, E=0 graphs, an empty graph, self loops, duplicate edges (SURVEY section 4 gaps)."""
    batch = edge_case_batch()
    for conv in ("gcn", "gin", "sage", "pna"):
        model = make_model(conv, in_dim=8, hidden=16, layers=2, task_out=3)
        cm = runtime.CompiledModel.from_model(model, batch.num_graphs, batch.num_nodes, batch.num_edges)
        out = cm.forward(*to_dev(batch, dev)).cpu().numpy()
        ref = O.forward_batched(model.spec(), canon(model), batch.x, batch.coo, batch.node_ptr, batch.edge_ptr)
        assert np.isfinite(out).all()
        assert np.abs(out - ref).max() < TOL, conv


# --------------------------------------------------------------------------- full BASELINE sizes: properties + sampled oracle
def test_full_size_config2_properties(dev):
    """BASELINE config 2 at full size (B=4096), layer by layer: (1) EVERY graph against the oracle (the C restatement does
    the 4096 graphs in well under a second), (2) batch-composition independence: reversing the order of the graphs
    permutes the outputs."""
    model = make_model("gcn", in_dim=11, hidden=128, layers=2)
    batch = synthetic.make_batch("qm9", 4096, seed=0)
    cm = runtime.CompiledModel.from_model(model, batch.num_graphs, batch.num_nodes, batch.num_edges)
    out = cm.forward(*to_dev(batch, dev)).cpu().numpy()
    cm.check()
    ref = O.forward_batched(model.spec(), canon(model), batch.x, batch.coo, batch.node_ptr, batch.edge_ptr)
    assert ref.shape == out.shape
    assert np.abs(out - ref).max() < TOL
    rev = pack_graphs([batch.graph(g) for g in range(4095, -1, -1)])
    out_rev = cm.forward(*to_dev(rev, dev)).cpu().numpy()
    assert np.abs(out_rev[::-1] - out).max() < 1e-5


def test_fused_stack_keeps_very_large_batches(dev):
    """32 768 QM9-shaped graphs (590 k rows): graph prep coarsens the node tiles so that every workgroup's run of the
    tile table still fits its LDS table and the batch STAYS on the fused stack (it used to fall back to the layer-wise
    path past 258 k rows); sampled graphs against the oracle, and the same graphs inside a small batch give the same rows."""
    model = make_model("gcn", in_dim=11, hidden=128, layers=2)
    batch = synthetic.make_batch("qm9", 32768, seed=3)
    cm = runtime.CompiledModel.from_model(model, batch.num_graphs, batch.num_nodes, batch.num_edges,
                                          max_graph_nodes=int(np.diff(batch.node_ptr).max()))
    args = to_dev(batch, dev)
    out = cm.forward(*args).cpu().numpy()
    cm.check()
    assert cm.gcn_stack_timed(args[0], 2) > 0.0                       # the fused stack took it
    idx = np.sort(np.random.default_rng(1).choice(batch.num_graphs, 160, replace=False))
    sub = pack_graphs([batch.graph(int(g)) for g in idx])
    ref = O.forward_batched(model.spec(), canon(model), sub.x, sub.coo, sub.node_ptr, sub.edge_ptr)
    assert np.abs(out[idx] - ref).max() < TOL
    small = runtime.CompiledModel.from_model(model, sub.num_graphs, sub.num_nodes, sub.num_edges, max_graph_nodes=29)
    assert np.abs(small.forward(*to_dev(sub, dev)).cpu().numpy() - out[idx]).max() < 1e-5


AGG_OPTION_SETS = [
    # launch geometries of the ring-form gather-aggregate kernel: waves per workgroup, stages in the ring, LDS budget
    # (small budgets push tiles to the direct-from-L2 path and make the ring cycle), tile granularity, workgroups per CU,
    # plain / non-temporal stores
    dict(), dict(agg_ring_waves=1), dict(agg_ring_waves=2), dict(agg_ring_waves=4), dict(agg_ring_waves=8),
    dict(agg_ring_slots=1), dict(agg_ring_slots=3), dict(agg_ring_slots=4, agg_ring_waves=2),
    dict(agg_lds_kb=8), dict(agg_lds_kb=24, agg_ring_waves=4), dict(agg_lds_kb=150, agg_ring_waves=1, agg_ring_slots=1),
    dict(tile_rows=4), dict(tile_rows=16, agg_ring_slots=3), dict(tile_rows=64), dict(agg_ring_wg_per_cu=2),
    dict(agg_ring_wg_per_cu=4, agg_ring_waves=4, agg_lds_kb=20), dict(agg_nt_store=0), dict(agg_nt_store=0, agg_ring_waves=2, tile_rows=4),
]


@pytest.mark.parametrize("opt", AGG_OPTION_SETS, ids=lambda o: "-".join(f"{k[4:] if k.startswith('agg_') else k}{v}" for k, v in o.items()))
def test_tiling_options_do_not_change_results(dev, opt):
    """Both forms of the gather-aggregate kernel (incl. the direct-from-L2 path and rings that cycle) agree
    with the oracle at every launch geometry, for a PNA model (4 output rows, q in the stage) and a GCN model."""
    for conv in ("pna", "gcn"):
        _tiling_case(dev, opt, conv)


def _tiling_case(dev, opt, conv):
    model = make_model(conv, in_dim=9, hidden=48 if conv == "gcn" else 32, layers=3, task_out=1)  # (GCN: wide enough for the aggregate kernel)
    batch = synthetic.make_batch("molhiv", 64, seed=11)
    ref = O.forward_batched(model.spec(), canon(model), batch.x, batch.coo, batch.node_ptr, batch.edge_ptr)
    defaults = dict(tile_rows=8, agg_lds_kb=0, agg_ring_waves=0, agg_ring_slots=2, agg_ring_wg_per_cu=1, agg_nt_store=1)
    try:
        for k, v in opt.items():
            runtime.set_option(k, v)
        cm = runtime.CompiledModel.from_model(model, batch.num_graphs, batch.num_nodes, batch.num_edges)
        out = cm.forward(*to_dev(batch, dev)).cpu().numpy()
    finally:
        for k, v in defaults.items():
            runtime.set_option(k, v)
    assert np.abs(out - ref).max() < TOL


@pytest.mark.parametrize("mlp_hidden,mlp_layers,task_out,pools,hidden", [
    (64, 2, 19, ("add", "mean", "max"), 128), (50, 3, 7, ("max", "add"), 32), (16, 0, 3, ("mean",), 64),
    (33, 1, 1, ("add",), 20), (64, 2, 19, ("add", "mean", "max"), 256)])
def test_fused_readout_equals_separate_kernels(dev, mlp_hidden, mlp_layers, task_out, pools, hidden):
    """Pooling + MLP head as one kernel vs k_global_pool + three GEMMs vs the oracle (odd widths too)."""
    model = make_model("gcn", in_dim=11, hidden=hidden, layers=2, pools=pools, mlp_hidden=mlp_hidden,
                       mlp_layers=mlp_layers, task_out=task_out, mlp_act="tanh")
    batch = synthetic.make_batch("qm9", 100, seed=hidden)
    ref = O.forward_batched(model.spec(), canon(model), batch.x, batch.coo, batch.node_ptr, batch.edge_ptr)
    cm = runtime.CompiledModel.from_model(model, batch.num_graphs, batch.num_nodes, batch.num_edges)
    outs = []
    try:
        for fuse in (1, 0):
            runtime.set_option("fuse_head", fuse)
            outs.append(cm.forward(*to_dev(batch, dev)).cpu().numpy())
    finally:
        runtime.set_option("fuse_head", 1)
    assert np.abs(outs[0] - ref).max() < TOL and np.abs(outs[1] - ref).max() < TOL
    assert np.abs(outs[0] - outs[1]).max() < 1e-5


@pytest.mark.parametrize("conv,fin", [("gcn", 11), ("gin", 9), ("gcn", 17), ("gin", 32), ("gcn", 4), ("sage", 9), ("sage", 16), ("sage", 3)])
def test_fused_narrow_layer_equals_separate_kernels(dev, conv, fin):
    """First layer with aggregate + update in one kernel vs the two-kernel path vs the oracle
    (incl. degree > 4 nodes and empty graphs)."""
    model = make_model(conv, in_dim=fin, hidden=64, layers=2, task_out=3)
    rng = np.random.default_rng(fin)
    star = np.array([[i, 0] for i in range(1, 9)] + [[0, i] for i in range(1, 9)])  # node 0 has in-degree 8
    graphs = [(rng.uniform(-1, 1, (9, fin)), star), (rng.uniform(-1, 1, (0, fin)), np.zeros((0, 2)))]
    b0 = synthetic.make_batch("qm9", 60, seed=fin)
    graphs += [(rng.uniform(-1, 1, (b0.graph(g)[0].shape[0], fin)), b0.graph(g)[1]) for g in range(60)]
    batch = pack_graphs([(np.asarray(x, np.float32), np.asarray(c, np.int32)) for x, c in graphs])
    ref = O.forward_batched(model.spec(), canon(model), batch.x, batch.coo, batch.node_ptr, batch.edge_ptr)
    cm = runtime.CompiledModel.from_model(model, batch.num_graphs, batch.num_nodes, batch.num_edges)
    outs = []
    try:
        for fuse in (1, 0):
            runtime.set_option("fuse_narrow", fuse)
            outs.append(cm.forward(*to_dev(batch, dev)).cpu().numpy())
    finally:
        runtime.set_option("fuse_narrow", 1)
    assert np.abs(outs[0] - ref).max() < TOL and np.abs(outs[1] - ref).max() < TOL
    assert np.abs(outs[0] - outs[1]).max() < 1e-5


@pytest.mark.parametrize("fin,h0,h1,act,pools", [(11, 128, 128, "relu", ("add", "mean", "max")), (9, 64, 64, "tanh", ("max", "add")),
                                                 (11, 32, 128, "gelu", ("mean",)), (20, 128, 64, "sigmoid", ("add", "mean", "max")),
                                                 (11, 64, 20, "relu", ("add",)),
                                                 # (either side of the three-k-step form of the narrow product: widths <= 12 / 13..16)
                                                 (12, 64, 128, "relu", ("add", "max")), (13, 128, 64, "relu", ("mean", "max")),
                                                 (16, 32, 32, "tanh", ("add",)), (4, 128, 128, "relu", ("max",))])
@pytest.mark.parametrize("pad_to_tile", [False, True])
@pytest.mark.parametrize("zf", [1, 2, 0])
def test_fused_gcn_stack_equals_layerwise_and_oracle(dev, fin, h0, h1, act, pools, pad_to_tile, zf):
    """The persistent 2-layer GCN kernels (graphs staged once in LDS: k_gcn2_zf, the default, which transforms the last
    layer before aggregating it, and k_gcn2_fused) vs the layer-by-layer path vs the oracle, on molecule batches plus
    degenerate graphs (empty, isolated node, in-degree 8, self loop)."""
    model = make_model("gcn", in_dim=fin, hidden=h0, layers=2, out_dim=h1, act=act, pools=pools, task_out=5)
    rng = np.random.default_rng(h0 + h1)
    star = np.array([[i, 0] for i in range(1, 9)] + [[0, i] for i in range(1, 9)] + [[3, 3]])
    graphs = [(rng.uniform(-1, 1, (9, fin)), star), (rng.uniform(-1, 1, (0, fin)), np.zeros((0, 2))),
              (rng.uniform(-1, 1, (1, fin)), np.zeros((0, 2))), (rng.uniform(-1, 1, (0, fin)), np.zeros((0, 2)))]
    b0 = synthetic.make_batch("qm9", 300, seed=fin + h0)
    graphs += [(rng.uniform(-1, 1, (b0.graph(g)[0].shape[0], fin)), b0.graph(g)[1]) for g in range(300)]
    if pad_to_tile:  # N a multiple of the node-tile size: the trailing empty graphs sit exactly on the last tile edge
        n_now = sum(g[0].shape[0] for g in graphs)
        graphs.append((rng.uniform(-1, 1, ((-n_now) % 16 or 16, fin)), np.zeros((0, 2))))
        graphs.append((rng.uniform(-1, 1, (0, fin)), np.zeros((0, 2))))
    graphs.append((rng.uniform(-1, 1, (0, fin)), np.zeros((0, 2))))  # batch ends with an empty graph
    batch = pack_graphs([(np.asarray(x, np.float32), np.asarray(c, np.int32)) for x, c in graphs])
    ref = O.forward_batched(model.spec(), canon(model), batch.x, batch.coo, batch.node_ptr, batch.edge_ptr)
    cm = runtime.CompiledModel.from_model(model, batch.num_graphs, batch.num_nodes, batch.num_edges, max_graph_nodes=29)
    try:
        runtime.set_option("fuse_zf", 1 if zf else 0)
        runtime.set_option("zf_shape", 0 if zf == 2 else 1)  # (1: one 16-wave workgroup per CU; 2 here: two 8-wave ones)
        fused = cm.forward(*to_dev(batch, dev)).cpu().numpy()
        assert cm.last_path() == ("stack_zf" if zf else "stack")
    finally:
        runtime.set_option("fuse_zf", 1)
        runtime.set_option("zf_shape", 2)
    cm.check()
    cm.set_max_graph_nodes(0)  # no promise -> layer-by-layer path
    layerwise = cm.forward(*to_dev(batch, dev)).cpu().numpy()
    assert cm.last_path() == "layerwise"
    assert np.isfinite(fused).all()
    assert np.abs(fused - ref).max() < TOL and np.abs(layerwise - ref).max() < TOL
    assert np.abs(fused - layerwise).max() < 2e-5


def test_wrong_tensor_layouts_are_refused(dev):
    """A raw pointer crosses the C ABI: another dtype / layout / device would be silently reinterpreted (advisor
    finding).  The binding refuses a PyG edge_index ([2, E] int64), CPU tensors, non-contiguous views, a feature width
    that is not the model's."""
    model = make_model("gcn", in_dim=11, hidden=16, layers=1)
    batch = synthetic.make_batch("qm9", 8, seed=0)
    cm = runtime.CompiledModel.from_model(model, batch.num_graphs, batch.num_nodes, batch.num_edges)
    x, coo, nptr, eptr = to_dev(batch, dev)
    cm.forward(x, coo, nptr, eptr)
    for bad in ((x, coo.T.contiguous().long(), nptr, eptr),          # PyG edge_index
                (x, coo.long(), nptr, eptr),                         # int64 rows
                (x.cpu(), coo, nptr, eptr),                          # host tensor
                (x[:, :9].contiguous(), coo, nptr, eptr),            # another feature width
                (x.double(), coo, nptr, eptr),
                (x, coo.T.contiguous().T, nptr, eptr),               # non-contiguous view
                (x, coo, nptr[:-1], eptr)):
        with pytest.raises(runtime.GnnbError):
            cm.forward(*bad)


def test_broken_max_graph_nodes_promise_is_detected(dev):
    model = make_model("gcn", in_dim=9, hidden=64, layers=2)  # (molhiv-shaped features)
    batch = synthetic.make_batch("molhiv", 40, seed=3)  # graphs of up to 222 nodes
    assert np.diff(batch.node_ptr).max() > 29
    cm = runtime.CompiledModel.from_model(model, batch.num_graphs, batch.num_nodes, batch.num_edges, max_graph_nodes=29)
    cm.forward(*to_dev(batch, dev))
    with pytest.raises(runtime.GnnbError, match="malformed batch"):
        cm.check()


def test_full_size_config2_fused_path(dev):
    """BASELINE config 2 at full size through the fused stack (k_gcn2_zf, both shapes, and k_gcn2_fused): EVERY one of the
    4096 graphs vs the oracle."""
    model = make_model("gcn", in_dim=11, hidden=128, layers=2)
    batch = synthetic.make_batch("qm9", 4096, seed=1)
    ref = O.forward_batched(model.spec(), canon(model), batch.x, batch.coo, batch.node_ptr, batch.edge_ptr)
    try:
        for fuse_zf, shape, path in ((1, 1, "stack_zf"), (1, 0, "stack_zf"), (0, 2, "stack")):
            runtime.set_option("fuse_zf", fuse_zf)
            runtime.set_option("zf_shape", shape)
            cm = runtime.CompiledModel.from_model(model, batch.num_graphs, batch.num_nodes, batch.num_edges, max_graph_nodes=29)
            out = cm.forward(*to_dev(batch, dev)).cpu().numpy()
            cm.check()
            assert cm.last_path() == path
            assert out.shape == ref.shape and np.abs(out - ref).max() < TOL, (fuse_zf, shape)
    finally:
        runtime.set_option("fuse_zf", 1)
        runtime.set_option("zf_shape", 2)


@pytest.mark.parametrize("fin,h0,h1,pools,mlp_layers,act,shape", [(11, 128, 128, ("add", "mean", "max"), 2, "relu", 2), (11, 128, 128, ("add", "mean", "max"), 2, "relu", 0),
                                                                  (9, 64, 32, ("max",), 4, "tanh", 2), (20, 128, 64, ("mean", "add"), 0, "gelu", 2),
                                                                  (16, 32, 128, ("add", "mean", "max"), 3, "sigmoid", 2)])
def test_mlp_head_inside_the_gcn_stack_kernel(dev, fin, h0, h1, pools, mlp_layers, act, shape):
    """`zf_head` 1 (round 5, opt-in): k_gcn2_zf runs the MLP head on the graphs each workgroup pooled -- conv stack + pooling +
    head in ONE launch, as the reference's top does (model.cpp.jinja:737-765) -- against the default (head as its own launch
    beside the next batch's stack kernel: measured faster) and the oracle.  Both kernel shapes; a batch whose
    workgroups own between 0 and ~40 graphs (runs of one-node and empty graphs), a large segment behind the stack (its
    graphs keep the separate readout), heads of one layer (not fused: same results)."""
    # (mlp_layers = hidden layers of the head: 0 is a single Linear, which the kernel leaves to the separate readout)
    model = make_model("gcn", in_dim=fin, hidden=h0, layers=2, out_dim=h1, act=act, pools=pools, task_out=5, seed=fin + h1,
                       mlp_layers=mlp_layers, mlp_act=act)
    rng = np.random.default_rng(fin)
    base = synthetic.make_batch("qm9", 700, seed=8)
    one = lambda: (rng.uniform(-1, 1, (1, fin)).astype(np.float32), np.zeros((0, 2), np.int32))
    empty = (np.zeros((0, fin), np.float32), np.zeros((0, 2), np.int32))

    def regraph(g):
        x, e = base.graph(g)
        return rng.uniform(-1, 1, (x.shape[0], fin)).astype(np.float32), e

    graphs = [empty] + [regraph(g) for g in range(300)] + [one() for _ in range(90)] + [empty] * 5 + [regraph(g) for g in range(300, 700)] + [empty]
    batch = pack_graphs(graphs)
    ref = O.forward_batched(model.spec(), canon(model), batch.x, batch.coo, batch.node_ptr, batch.edge_ptr)
    outs = {}
    try:
        runtime.set_option("zf_shape", shape)
        for head in (1, 0):
            runtime.set_option("zf_head", head)
            cm = runtime.CompiledModel.from_model(model, batch.num_graphs, batch.num_nodes, batch.num_edges, max_graph_nodes=29)
            outs[head] = cm.forward(*to_dev(batch, dev)).cpu().numpy()
            cm.check()
            assert cm.last_path() == "stack_zf"
            again = cm.forward(*to_dev(batch, dev)).cpu().numpy()
            assert np.array_equal(outs[head], again)
        # a large segment behind the stack: its graphs take the separate readout, the rest the in-kernel one
        runtime.set_option("zf_head", 1)
        big = [(rng.uniform(-1, 1, (120, fin)).astype(np.float32), np.stack([np.arange(119), np.arange(1, 120)], 1).astype(np.int32)) for _ in range(3)]
        from gnnbuilder_amd.batching import order_large_last
        b2, order, seg = order_large_last(pack_graphs(graphs[:200] + big + graphs[200:400]), 29)
        ref2 = O.forward_batched(model.spec(), canon(model), b2.x, b2.coo, b2.node_ptr, b2.edge_ptr)
        cm = runtime.CompiledModel.from_model(model, b2.num_graphs, b2.num_nodes, b2.num_edges, max_graph_nodes=29)
        cm.set_large_segment(*seg)
        got2 = cm.forward(*to_dev(b2, dev)).cpu().numpy()
        cm.check()
        assert np.abs(got2 - ref2).max() < TOL * max(1.0, float(np.abs(ref2).max()))
    finally:
        runtime.set_option("zf_head", 0)
        runtime.set_option("zf_shape", 2)
    scale = max(1.0, float(np.abs(ref).max()))
    assert np.abs(outs[1] - ref).max() < TOL * scale and np.abs(outs[0] - ref).max() < TOL * scale
    assert np.abs(outs[1] - outs[0]).max() < 2e-5 * scale


@pytest.mark.parametrize("conv,promise", [("gcn", 29), ("gcn", 0), ("sage", 0)])
def test_forward_is_hip_graph_capturable(dev, conv, promise):
    """gnnb_forward_batched does no allocation and no synchronisation, so a caller can capture it
    (graph prep included) into a hipGraph and replay it with new features in the same buffers
    (SURVEY 8e / DESIGN 3.6).  Both the fused GCN stack and the layer-by-layer path."""
    model = make_model(conv, in_dim=11, hidden=128, layers=2, out_dim=128, act="relu", pools=("add", "mean", "max"), task_out=19)
    batch = synthetic.make_batch("qm9", 256, seed=5)
    cm = runtime.CompiledModel.from_model(model, batch.num_graphs, batch.num_nodes, batch.num_edges, max_graph_nodes=promise)
    x, coo, nptr, eptr = to_dev(batch, dev)
    out = torch.empty(batch.num_graphs, 19, device=dev)
    side = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):
        cm.forward(x, coo, nptr, eptr, out=out, stream=side)  # warm-up: one-time function attributes, lazy module load
    side.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        cm.forward(x, coo, nptr, eptr, out=out, stream=torch.cuda.current_stream())
    rng = np.random.default_rng(0)
    for _ in range(2):  # replay on fresh features written into the captured input buffer
        xn = rng.uniform(-1, 1, batch.x.shape).astype(np.float32)
        x.copy_(torch.from_numpy(xn))
        out.zero_()
        g.replay()
        torch.cuda.synchronize()
        ref = O.forward_batched(model.spec(), canon(model), xn, batch.coo, batch.node_ptr, batch.edge_ptr)
        assert np.abs(out.cpu().numpy() - ref).max() < TOL
    cm.check()


@pytest.mark.parametrize("fin,h0,h1,act", [(11, 128, 128, "relu"), (9, 64, 64, "tanh"), (20, 128, 64, "gelu"), (11, 32, 128, "relu")])
def test_fused_gcn_stack_bf16x6_math_is_fp32_equivalent(dev, fin, h0, h1, act):
    """Opt-in math mode 1: the wide update of the fused stack as six bf16 MFMA products of an exact 3-way
    split of both operands.  Must agree with the oracle inside the north-star tolerance AND
    sit at fp32 rounding level next to the fp32-MFMA path (the dropped partial products are < 2^-24 of a product)."""
    model = make_model("gcn", in_dim=fin, hidden=h0, layers=2, out_dim=h1, act=act, pools=("add", "mean", "max"), task_out=7)
    batch = synthetic.make_batch("qm9", 500, seed=h0 + fin)
    rng = np.random.default_rng(1)
    x = rng.uniform(-1, 1, (batch.num_nodes, fin)).astype(np.float32)
    ref = O.forward_batched(model.spec(), canon(model), x, batch.coo, batch.node_ptr, batch.edge_ptr)
    cm = runtime.CompiledModel.from_model(model, batch.num_graphs, batch.num_nodes, batch.num_edges, max_graph_nodes=29)
    xd = torch.from_numpy(x).to(dev)
    _, coo, nptr, eptr = to_dev(batch, dev)
    try:
        runtime.set_option("fuse_zf", 0)  # (the bf16x6 form lives in k_gcn2_fused; with k_gcn2_zf on, math = 1 keeps that fp32 kernel)
        runtime.set_option("math", 1)
        split = cm.forward(xd, coo, nptr, eptr).cpu().numpy()
        cm.check()
        assert cm.last_path() == "stack"
        runtime.set_option("math", 0)
        exact = cm.forward(xd, coo, nptr, eptr).cpu().numpy()
        runtime.set_option("fuse_zf", 1)
        runtime.set_option("math", 1)
        never_slower = cm.forward(xd, coo, nptr, eptr).cpu().numpy()
        assert cm.last_path() == "stack_zf" and np.abs(never_slower - ref).max() < TOL
    finally:
        runtime.set_option("math", 0)
        runtime.set_option("fuse_zf", 1)
    scale = max(1.0, float(np.abs(ref).max()))
    assert np.abs(split - ref).max() < TOL and np.abs(exact - ref).max() < TOL
    assert np.abs(split - exact).max() < 4e-6 * scale, np.abs(split - exact).max()


@pytest.mark.parametrize("math,bound", [(2, 2e-5), (3, 2e-6)])
@pytest.mark.parametrize("fin,h0,h1,act,count", [(11, 128, 128, "relu", 500), (9, 64, 64, "tanh", 300), (16, 128, 20, "gelu", 257),
                                                 (11, 32, 128, "relu", 64), (5, 128, 64, "sigmoid", 1)])
def test_gcn_stack_bf16x3_math_is_the_reduced_precision_mode(dev, fin, h0, h1, act, count, math, bound):
    """Opt-in math modes 2 and 3 (the analogue of the reference's float_or_fixed switch, code_gen.py:39-52):
    k_gcn2_zf's wide update H.W1^T on the 16-bit matrix cores, both operands as hi + mid pieces (round to nearest), three
    products, fp32 accumulate -- bf16 pieces (2: ~18 significant bits per product, fp32's range) or fp16 pieces (3: ~22 bits,
    fp16's range).  REDUCED precision by design: it must (a) actually run (its output differs from the fp32 form), (b) stay
    well inside the north-star tolerance of 1e-4 against the oracle (bounds: 2e-5 / 2e-6 of the output scale), and (c) leave
    everything else of the kernel alone (empty graphs, ragged stages, pooling: the same batch)."""
    model = make_model("gcn", in_dim=fin, hidden=h0, layers=2, out_dim=h1, act=act, pools=("add", "mean", "max"), task_out=7)
    batch = synthetic.make_batch("qm9", count, seed=h0 + fin)
    rng = np.random.default_rng(2)
    x = rng.uniform(-1, 1, (batch.num_nodes, fin)).astype(np.float32)
    ref = O.forward_batched(model.spec(), canon(model), x, batch.coo, batch.node_ptr, batch.edge_ptr)
    cm = runtime.CompiledModel.from_model(model, batch.num_graphs, batch.num_nodes, batch.num_edges, max_graph_nodes=29)
    xd = torch.from_numpy(x).to(dev)
    _, coo, nptr, eptr = to_dev(batch, dev)
    try:
        runtime.set_option("math", math)
        reduced = cm.forward(xd, coo, nptr, eptr).cpu().numpy()
        cm.check()
        assert cm.last_path() == "stack_zf"
        again = cm.forward(xd, coo, nptr, eptr).cpu().numpy()
        runtime.set_option("math", 0)
        exact = cm.forward(xd, coo, nptr, eptr).cpu().numpy()
    finally:
        runtime.set_option("math", 0)
    scale = max(1.0, float(np.abs(ref).max()))
    assert np.array_equal(reduced, again)  # deterministic
    assert np.abs(exact - ref).max() < TOL * scale
    assert np.abs(reduced - ref).max() < bound * scale, np.abs(reduced - ref).max()
    if count > 1:
        assert np.abs(reduced - exact).max() > 0.0  # (the mode ran: the fp32 form is bit-stable, this one rounds differently)


@pytest.mark.parametrize("act,skip,hidden,layers,conv", [("relu", False, 128, 3, "gin"), ("tanh", True, 64, 4, "gin"), ("gelu", False, 32, 2, "gin"),
                                                         ("relu", True, 128, 3, "gcn"), ("sigmoid", False, 64, 5, "gcn")])
def test_gin_and_deep_gcn_stacks_in_the_f16x3_math_mode(dev, act, skip, hidden, layers, conv):
    """Opt-in math mode 3 in k_gcn2_fused's GIN and deep-GCN variants (round 5): every 128-wide product reads hi + mid fp16 pieces
    that its producer (the aggregate phase or the product before) wrote into the same padded rows; three fp16 MFMA products,
    fp32 accumulate.  REDUCED precision: 2e-5 of the output scale against the oracle; the mode must have run (its output
    differs from the fp32 kernel's) on the stack path; empty graphs, ragged stages and the skip term are the fp32 kernel's."""
    model = make_model(conv, in_dim=9, hidden=hidden, layers=layers, out_dim=hidden, act=act, skip=skip, pools=("add", "mean", "max"), task_out=3, seed=7)
    batch = synthetic.make_batch("molhiv", 400, seed=layers)
    ref = O.forward_batched(model.spec(), canon(model), batch.x, batch.coo, batch.node_ptr, batch.edge_ptr)
    cm = runtime.CompiledModel.from_model(model, batch.num_graphs, batch.num_nodes, batch.num_edges, max_graph_nodes=int(np.diff(batch.node_ptr).max()))
    args = to_dev(batch, dev)
    try:
        runtime.set_option("math", 3)
        got = cm.forward(*args).cpu().numpy()
        cm.check()
        assert cm.last_path() == "stack"
    finally:
        runtime.set_option("math", 0)
    exact = cm.forward(*args).cpu().numpy()
    scale = max(1.0, float(np.abs(ref).max()))
    assert np.abs(exact - ref).max() < TOL * scale
    assert np.abs(got - ref).max() < 2e-5 * scale, (np.abs(got - ref).max(), scale)
    assert np.abs(got - exact).max() > 0.0


@pytest.mark.parametrize("conv,shape,hidden,promise_degree", [("sage", "molhiv", 256, False), ("pna", "qm9", 128, True), ("pna", "molhiv", 128, False),
                                                         ("gin", "molhiv", 128, False)])
def test_layer_by_layer_models_in_the_f16x3_math_mode(dev, conv, shape, hidden, promise_degree):
    """Opt-in math mode 3 on whole layer-by-layer models (the BASELINE config 4 / 5 shapes, small batches): the LDS-DMA GEMMs
    multiply hi + mid fp16 pieces (three products, fp32 accumulate); aggregates, first-layer kernels and the readout stay fp32.
    REDUCED precision: the bound is 2e-5 of the output scale against the oracle (north star: 1e-4) -- PNA's sum / std
    aggregates times the degree scalers stay far below fp16's 65504 on molecule-shaped inputs."""
    batch = synthetic.make_batch(shape, 700, seed=5)
    model = make_model(conv, in_dim=int(batch.x.shape[1]), hidden=hidden, layers=3 if conv != "sage" else 2, out_dim=hidden, act="relu",
                       pools=("add",) if conv == "gin" else ("add", "mean", "max"), task_out=3, seed=11)
    ref = O.forward_batched(model.spec(), canon(model), batch.x, batch.coo, batch.node_ptr, batch.edge_ptr)
    deg = int(np.bincount(batch.coo[:, 1]).max())  # (global node ids)
    cm = runtime.CompiledModel.from_model(model, batch.num_graphs, batch.num_nodes, batch.num_edges)
    if promise_degree and deg <= 15:
        cm.set_max_degree(deg)  # (the degree-class form of PNA: k_linear_dma's row-class mode)
    args = to_dev(batch, dev)
    try:
        runtime.set_option("math", 3)
        got = cm.forward(*args).cpu().numpy()
        cm.check()
    finally:
        runtime.set_option("math", 0)
    exact = cm.forward(*args).cpu().numpy()
    scale = max(1.0, float(np.abs(ref).max()))
    assert np.isfinite(got).all()
    assert np.abs(exact - ref).max() < TOL * scale
    assert np.abs(got - ref).max() < 2e-5 * scale, (np.abs(got - ref).max(), scale)


def _random_graphs(rng, count, n_max, fin, dense):
    """Arbitrary directed multigraphs: empty graphs, isolated nodes, self loops, repeated edges, hubs."""
    graphs = []
    for _ in range(count):
        n = int(rng.integers(0, n_max + 1))
        if n == 0:
            graphs.append((np.zeros((0, fin), np.float32), np.zeros((0, 2), np.int32)))
            continue
        e = int(rng.integers(0, dense * n + 1))
        src = rng.integers(0, n, e)
        dst = rng.integers(0, n, e) if rng.random() < 0.7 else np.full(e, rng.integers(0, n))  # sometimes one hub takes it all
        graphs.append((rng.uniform(-1, 1, (n, fin)).astype(np.float32), np.stack([src, dst], 1).astype(np.int32)))
    return graphs


@pytest.mark.parametrize("conv,layers,n_max,promise", [("gcn", 2, 33, 33), ("gcn", 2, 33, 0), ("gcn", 3, 90, 0), ("gin", 2, 70, 0),
                                                       ("sage", 2, 70, 0), ("pna", 2, 40, 0)])
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_random_multigraphs_match_oracle(dev, conv, layers, n_max, promise, seed):
    """Randomised structure sweep (nothing molecule-like about it): in-degrees far above the four inline
    neighbour slots, self loops, duplicate edges, empty graphs anywhere in the batch -- every conv family,
    the fused GCN stack (promise = 33, its largest admissible graph) and the layer-by-layer path."""
    rng = np.random.default_rng(100 * seed + layers + n_max)
    fin = int(rng.integers(3, 20))
    model = make_model(conv, in_dim=fin, hidden=64, layers=layers, out_dim=64, act="relu", pools=("add", "mean", "max"), task_out=5)
    batch = pack_graphs(_random_graphs(rng, 120, n_max, fin, dense=4))
    ref = O.forward_batched(model.spec(), canon(model), batch.x, batch.coo, batch.node_ptr, batch.edge_ptr)
    cm = runtime.CompiledModel.from_model(model, batch.num_graphs, batch.num_nodes, max(batch.num_edges, 1), max_graph_nodes=promise)
    got = cm.forward(*to_dev(batch, dev)).cpu().numpy()
    cm.check()
    scale = max(1.0, float(np.abs(ref).max()))
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() < TOL * scale, (np.abs(got - ref).max(), scale)


@pytest.mark.parametrize("M,N,K", [(3000, 128, 128), (777, 64, 64), (1000, 32, 32), (513, 64, 128)])
def test_linear_bf16x6_math_is_fp32_equivalent(dev, M, N, K):
    """Opt-in math mode 1 in the register-resident-weight GEMM: six bf16 MFMA products of an exact 3-way
    split of both operands.  Against a float64 product it must be as good as the native fp32-MFMA kernel."""
    torch.manual_seed(M + N)
    a = torch.rand(M, K, device=dev) - 0.5
    w = (torch.rand(N, K, device=dev) - 0.5) / K ** 0.5
    b = torch.rand(N, device=dev)
    ref = torch.tanh(a.double() @ w.double().T + b.double())
    try:
        runtime.set_option("math", 1)
        split = runtime.linear([(a, None)], w, b, act="tanh")
    finally:
        runtime.set_option("math", 0)
    exact = runtime.linear([(a, None)], w, b, act="tanh")
    e_split = (split.double() - ref).abs().max().item()
    e_exact = (exact.double() - ref).abs().max().item()
    assert e_split < 2e-6 and e_split < 2.0 * e_exact + 1e-7, (e_split, e_exact)


@pytest.mark.parametrize("M,N,F", [(4000, 256, 32), (70000, 128, 32), (2000, 200, 64)])
def test_large_k_gemm_bf16x6_math_is_fp32_equivalent(dev, M, N, F):
    """Opt-in math mode 1 in the LDS-DMA GEMM (the SAGE / PNA shapes): four segments of width F .. 4F, two of them
    row-scaled (the PNA update), bias + skip + tanh; six v_mfma_f32_32x32x16_bf16 products of an exact 3-way split of both
    operands per 16-wide k block.  Against a float64 product it must be as good as the native fp32-MFMA kernel; tail
    slices (M = 70 000: 547 tiles) included."""
    g = torch.Generator().manual_seed(M + N)
    x = torch.rand(M, F, generator=g) - 0.5
    A = torch.rand(M, 4 * F, generator=g) - 0.5
    amp, att = torch.rand(M, generator=g) + 0.5, torch.rand(M, generator=g) + 0.5
    w = (torch.rand(N, 13 * F, generator=g) - 0.5) / (13 * F) ** 0.5
    b, skip = torch.rand(N, generator=g), torch.rand(M, N, generator=g) - 0.5
    rows = torch.cat([torch.arange(0, min(M, 3000)), torch.arange(max(M - 3000, 0), M)])
    cat = torch.cat([x, A, A * amp[:, None], A * att[:, None]], 1)[rows].double()
    ref = torch.tanh(cat @ w.double().T + b.double() + skip[rows].double())
    Ad = A.to(dev)
    segs = [(x.to(dev), None), (Ad, None), (Ad, amp.to(dev)), (Ad, att.to(dev))]
    wd, bd, sd = w.to(dev), b.to(dev), skip.to(dev)
    try:
        runtime.set_option("math", 1)
        split = runtime.linear(segs, wd, bd, skip=sd, act="tanh").cpu()
    finally:
        runtime.set_option("math", 0)
    exact = runtime.linear(segs, wd, bd, skip=sd, act="tanh").cpu()
    e_split = (split[rows].double() - ref).abs().max().item()
    e_exact = (exact[rows].double() - ref).abs().max().item()
    assert e_split < 2e-6 and e_split < 2.0 * e_exact + 1e-7, (e_split, e_exact)
    # math 3 (opt-in, REDUCED precision): fp16 hi + mid pieces, three products -- the same shapes, inside fp16's range here:
    # ~22 significant bits per product; the bound leaves room for the 2^-21 per-term error over K = 13 F terms
    try:
        runtime.set_option("math", 3)
        half3 = runtime.linear(segs, wd, bd, skip=sd, act="tanh").cpu()
    finally:
        runtime.set_option("math", 0)
    e_half3 = (half3[rows].double() - ref).abs().max().item()
    assert e_half3 < 4e-6, (e_half3, e_exact)
    assert not torch.equal(half3, exact)  # (the mode ran)


@pytest.mark.parametrize("mlp_hidden,mlp_layers,task_out,pools,h1", [
    (64, 2, 19, ("add", "mean", "max"), 128), (16, 1, 1, ("max",), 64), (128, 3, 33, ("add", "max"), 32),
    (50, 2, 7, ("mean",), 64), (64, 0, 5, ("add", "mean", "max"), 20), (100, 2, 3, ("add",), 128)])
def test_small_readout_kernel_equals_lds_kernel_and_oracle(dev, mlp_hidden, mlp_layers, task_out, pools, h1):
    """The readout behind the fused GCN stack: small-footprint kernel (operands straight from L2, co-resident
    with the next batch's conv kernel) vs the weights-in-LDS kernel vs the oracle, over head shapes incl. a
    single linear, widths that are not multiples of 16 and one the small form must decline (hidden 50)."""
    model = make_model("gcn", in_dim=11, hidden=64, layers=2, out_dim=h1, act="relu", pools=pools, mlp_hidden=mlp_hidden,
                       mlp_layers=mlp_layers, task_out=task_out, mlp_act="tanh")
    batch = synthetic.make_batch("qm9", 203, seed=mlp_hidden + task_out)  # 203: the last 16-graph tile is ragged
    ref = O.forward_batched(model.spec(), canon(model), batch.x, batch.coo, batch.node_ptr, batch.edge_ptr)
    cm = runtime.CompiledModel.from_model(model, batch.num_graphs, batch.num_nodes, batch.num_edges, max_graph_nodes=29)
    outs = []
    try:
        for small in (1, 0):
            runtime.set_option("head_small", small)
            outs.append(cm.forward(*to_dev(batch, dev)).cpu().numpy())
    finally:
        runtime.set_option("head_small", 1)
    cm.check()
    for o in outs:
        assert np.abs(o - ref).max() < TOL
    assert np.abs(outs[0] - outs[1]).max() < 1e-5


@pytest.mark.parametrize("M,N,F", [(1000, 128, 64), (333, 200, 32), (4097, 256, 128)])
def test_linear_dma_path_segments_rowscale_and_ragged_edges(dev, M, N, F):
    """The LDS-DMA form of the tiled GEMM (segment widths whole 32-wide chunks, N > 64): four segments with two
    row-scaled ones (the PNA shape), M and N that are not multiples of the tile, skip + activation -- against
    a float64 product, and bit-for-bit against the register-staged kernel (same MFMA order)."""
    g = torch.Generator().manual_seed(M + N)
    x = torch.rand(M, F, generator=g) - 0.5
    A = torch.rand(M, 4 * F, generator=g) - 0.5
    amp, att = torch.rand(M, generator=g) + 0.5, torch.rand(M, generator=g) + 0.5
    w = (torch.rand(N, 13 * F, generator=g) - 0.5) / (13 * F) ** 0.5
    b, skip = torch.rand(N, generator=g), torch.rand(M, N, generator=g) - 0.5
    cat = torch.cat([x, A, A * amp[:, None], A * att[:, None]], 1).double()
    ref = torch.tanh(cat @ w.double().T + b.double() + skip.double())
    Ad, segs = A.to(dev), None
    segs = [(x.to(dev), None), (Ad, None), (Ad, amp.to(dev)), (Ad, att.to(dev))]
    outs = []
    try:
        runtime.set_option("gemm_tail_split", 1)  # (row slices keep the summation order; the default stream-K tail has its own test)
        for dma in (1, 0):
            runtime.set_option("gemm_dma", dma)
            outs.append(runtime.linear(segs, w.to(dev), b.to(dev), skip=skip.to(dev), act="tanh").cpu())
    finally:
        runtime.set_option("gemm_dma", 1)
        runtime.set_option("gemm_tail_split", 2)
    assert (outs[0].double() - ref).abs().max().item() < 2e-5
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("M,N,K", [(256 * 300 - 5, 256, 64), (256 * 561 + 70, 128, 96), (256 * 265, 200, 32),
                                   (256 * 40 + 3, 128, 160), (256 * 100 - 130, 256, 64), (5000, 130, 64), (77, 255, 96)])
def test_linear_dma_tail_split_is_bit_identical(dev, M, N, K):
    """Tiles of the last, partial round go out as two 128-row or four 64-row slices (600 = 2 x 256 + 88 tiles -> halves,
    562 = 2 x 256 + 50 -> quarters, 530 = 2 x 256 + 18 -> quarters, 41 and 200 tiles < 256 CUs -> all tiles sliced):
    same MFMA order per output element, so bit-identical to the unsplit launch, and right against a float64
    product; ragged M (the last slices are partly or wholly past M), skip + activation."""
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.rand(M, K, generator=g) - 0.5
    w = (torch.rand(N, K, generator=g) - 0.5) / K ** 0.5
    b, skip = torch.rand(N, generator=g), torch.rand(M, N, generator=g) - 0.5
    ad, wd, bd, sd = a.to(dev), w.to(dev), b.to(dev), skip.to(dev)
    outs = []
    try:
        for split in (1, 0):
            runtime.set_option("gemm_tail_split", split)
            outs.append(runtime.linear([(ad, None)], wd, bd, skip=sd, act="relu").cpu())
    finally:
        runtime.set_option("gemm_tail_split", 2)
    rows = torch.cat([torch.arange(0, min(4096, M)), torch.arange(max(M - 70000, 0), M)])     # head + the whole tail round
    ref = torch.relu(a[rows].double() @ w.double().T + b.double() + skip[rows].double())
    assert (outs[0][rows].double() - ref).abs().max().item() < 2e-5
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("M,N,K,segs", [(147494, 128, 1664, 4), (128 * 1153, 128, 1664, 1), (128 * 300 - 5, 256, 1024, 2),
                                        (128 * 40 + 3, 128, 1280, 1), (5000, 130, 1600, 1), (77, 255, 1024, 2),
                                        (128 * 255, 128, 1152, 3), (128 * 257, 128, 1152, 1), (128 * 513 + 1, 192, 1056, 1),
                                        (128 * 300 - 5, 256, 512, 2)])
def test_linear_dma_stream_k_tail(dev, M, N, K, segs):
    """The default tail of k_linear_dma (gemm_tail_split 2): the last, partial round of tiles cut along K into equal runs over
    all resident workgroups, parts added up in run order by the last workgroup at each tile.  Another summation order than
    the unsplit launch -- compared with a float64 product at the same tolerance, and with the unsplit launch --, one order
    per shape: two launches give the same bits; counters are left cleared (a third launch is right again).  Shapes: C4's
    13F GEMM (1153 tiles, 4 segments with row scalers), fewer tiles than CUs, runs that span two tiles, ragged M and N; the
    last one (K = 512 < 1024) takes row slices under the same option."""
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.rand(M, K, generator=g) - 0.5
    w = (torch.rand(N, K, generator=g) - 0.5) / K ** 0.5
    b, skip = torch.rand(N, generator=g), torch.rand(M, N, generator=g) - 0.5
    sc = [None] + [torch.rand(M, generator=g) + 0.5 for _ in range(segs - 1)]
    ks = K // segs // 32 * 32
    cuts = [i * ks for i in range(segs)] + [K]
    ad, wd, bd, sd = a.to(dev), w.to(dev), b.to(dev), skip.to(dev)
    seg_list = [(ad[:, cuts[i]:cuts[i + 1]], sc[i].to(dev) if sc[i] is not None else None) for i in range(segs)]
    outs = []
    try:
        for split in (2, 2, 0, 2):
            runtime.set_option("gemm_tail_split", split)
            outs.append(runtime.linear(seg_list, wd, bd, skip=sd, act="relu").cpu())
    finally:
        runtime.set_option("gemm_tail_split", 2)
    rows = torch.cat([torch.arange(0, min(4096, M)), torch.arange(max(M - 40000, 0), M)])
    a64 = a[rows].double().clone()
    for i in range(1, segs):
        a64[:, cuts[i]:cuts[i + 1]] *= sc[i][rows].double()[:, None]
    ref = torch.relu(a64 @ w.double().T + b.double() + skip[rows].double())
    assert (outs[0][rows].double() - ref).abs().max().item() < 3e-5
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[3])
    assert (outs[0] - outs[2]).abs().max().item() < 2e-5
    # the arrival counters are indexed by the first workgroup of a shared tile (< 512), never by the tile (1153 here): every
    # counter back at zero and the guard region behind the 1024 counters untouched (round-4 advisor finding: tile 1025 wrote
    # past the allocation)
    runtime.stream_k_guard()


@pytest.mark.parametrize("promise,math,zf", [(34, 0, 0), (50, 0, 0), (55, 0, 0), (57, 0, 0), (58, 0, 0), (61, 0, 0), (62, 0, 0),
                                             (41, 1, 0), (45, 1, 0), (46, 1, 0), (46, 1, 1),
                                             (34, 0, 1), (62, 0, 1), (89, 0, 1), (120, 0, 1), (169, 0, 1), (172, 0, 1), (173, 0, 1), (174, 0, 1),
                                             (89, 0, 2), (93, 0, 2), (94, 0, 2),
                                             (29, 0, 3), (93, 0, 3), (94, 0, 3), (173, 0, 3), (174, 0, 3)])
def test_fused_gcn_stack_takes_graphs_up_to_61_nodes(dev, promise, math, zf):
    """k_gcn2_fused: graphs of up to 57 nodes fit one 64-row stage with the default 8-row node tiles, up to 61 with 4-row
    tiles (graph prep picks the tile size from the promise): ESOL-sized molecules (n_max 55) take the fused stack.  62 is
    past the limit: the layer-by-layer path answers, same numbers.  The opt-in bf16x6 mode keeps 48-row stages (limit 45).
    k_gcn2_zf (the default for two fp32 GCN layers) has 176-row stages (one 16-wave workgroup per CU): 169 nodes with 8-row
    tiles, 173 with 4-row tiles; zf = 2 selects its other shape (two 8-wave workgroups per CU, 96-row stages: 89 / 93); zf = 3
    the default: the 176-row shape wherever it exists (input widths up to 16)."""
    model = make_model("gcn", in_dim=9, hidden=128, layers=2, out_dim=128, act="relu", pools=("add", "mean", "max"), task_out=4)
    rng = np.random.default_rng(promise)
    graphs = []
    for _ in range(150):
        n = int(rng.integers(1, promise + 1))
        e = int(rng.integers(0, 3 * n))
        graphs.append((rng.uniform(-1, 1, (n, 9)).astype(np.float32),
                       np.stack([rng.integers(0, n, e), rng.integers(0, n, e)], 1).astype(np.int32)))
    graphs.append((rng.uniform(-1, 1, (promise, 9)).astype(np.float32), np.zeros((0, 2), np.int32)))  # one of the largest size
    batch = pack_graphs(graphs)
    ref = O.forward_batched(model.spec(), canon(model), batch.x, batch.coo, batch.node_ptr, batch.edge_ptr)
    try:
        runtime.set_option("math", math)
        runtime.set_option("fuse_zf", 1 if zf else 0)
        runtime.set_option("zf_shape", {2: 0, 3: 2}.get(zf, 1))  # (zf = 3: the default)
        cm = runtime.CompiledModel.from_model(model, batch.num_graphs, batch.num_nodes, max(batch.num_edges, 1), max_graph_nodes=promise)
        got = cm.forward(*to_dev(batch, dev)).cpu().numpy()
        cm.check()
        assert np.abs(got - ref).max() < TOL * max(1.0, float(np.abs(ref).max()))
        # which path ran: reported by the workspace; the stack's timed entry refuses when it is not eligible
        xd = torch.from_numpy(batch.x).to(dev)
        limit = 45 if (math and not zf) else {0: 61, 1: 173, 2: 93, 3: 173}[zf]  # (math = 1 leaves k_gcn2_zf on: fp32, faster)
        if promise <= limit:
            assert cm.last_path() == ("stack_zf" if zf else "stack")
            assert cm.gcn_stack_timed(xd, 2) > 0.0
        else:
            assert cm.last_path() == "layerwise"
            with pytest.raises(runtime.GnnbError):
                cm.gcn_stack_timed(xd, 2)
    finally:
        runtime.set_option("math", 0)
        runtime.set_option("fuse_zf", 1)
        runtime.set_option("zf_shape", 2)


@pytest.mark.parametrize("conv,shape,promise", [("gcn", "qm9", True), ("gin", "molhiv", False), ("pna", "qm9", False)])
def test_forward_is_graph_capturable(dev, conv, shape, promise):
    """The whole batched forward (graph prep included) is enqueued on the caller's stream without host synchronisation,
    allocation or host-side reads: it can be captured into a HIP graph and replayed.  The replay on NEW inputs written
    into the captured buffers gives the oracle's numbers (fused GCN stack and layer-by-layer paths)."""
    fin = synthetic.SHAPES[shape]["f_in"]
    model = make_model(conv, in_dim=fin, hidden=64, layers=2, out_dim=64, act="relu", pools=("add", "max"), task_out=3)
    b0, b1 = synthetic.make_batch(shape, 96, seed=1), synthetic.make_batch(shape, 96, seed=2)
    # the captured buffers are sized for the larger batch; graph boundaries / sizes of a replay must equal the capture's,
    # so the second batch reuses the first one's topology with new features
    x1 = np.random.default_rng(5).uniform(-1, 1, b0.x.shape).astype(np.float32)
    kw = dict(max_graph_nodes=int(np.diff(b0.node_ptr).max())) if promise else {}
    cm = runtime.CompiledModel.from_model(model, b0.num_graphs, b0.num_nodes, b0.num_edges, **kw)
    xd, coo, nptr, eptr = to_dev(b0, dev)
    out = torch.empty((b0.num_graphs, cm.out_dim), dtype=torch.float32, device=dev)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        cm.forward(xd, coo, nptr, eptr, out=out)          # warm-up outside the capture (kernel attributes, caches)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        cm.forward(xd, coo, nptr, eptr, out=out)
    for xs in (b0.x, x1):
        xd.copy_(torch.from_numpy(xs))
        out.zero_()
        g.replay()
        torch.cuda.synchronize()
        ref = O.forward_batched(model.spec(), canon(model), xs, b0.coo, b0.node_ptr, b0.edge_ptr)
        assert np.abs(out.cpu().numpy() - ref).max() < TOL * max(1.0, float(np.abs(ref).max()))
    cm.check()


@pytest.mark.parametrize("layers,hidden,out_dim,act,skip,shape", [
    (3, 128, 128, "relu", True, "qm9"), (3, 128, 128, "relu", False, "qm9"), (4, 64, 32, "tanh", True, "esol"),
    (5, 32, 64, "gelu", True, "qm9"), (6, 128, 20, "sigmoid", True, "esol"), (3, 64, 128, "relu", True, "qm9")])
def test_fused_gcn_stack_of_more_than_two_layers(dev, layers, hidden, out_dim, act, skip, shape):
    """GCN stacks of 3..6 layers in ONE kernel (reference compute_gnn_head, model.cpp.jinja:151-359, any depth): the
    middle layers repeat the aggregate / update pair inside the stage, skip connection on exactly these, their weight
    slices re-read per layer.  Against the oracle, against the layer-by-layer path, on molecule batches and on random
    multigraphs (hubs, duplicate edges, empty graphs); the timed entry proves the fused stack is what ran."""
    fin = synthetic.SHAPES[shape]["f_in"]
    model = make_model("gcn", in_dim=fin, hidden=hidden, layers=layers, out_dim=out_dim, act=act, skip=skip, task_out=3)
    rng = np.random.default_rng(layers * 100 + hidden)
    for batch in (synthetic.make_batch(shape, 400, seed=layers), pack_graphs(_random_graphs(rng, 120, 40, fin, dense=4))):
        promise = int(np.diff(batch.node_ptr).max())
        ref = O.forward_batched(model.spec(), canon(model), batch.x, batch.coo, batch.node_ptr, batch.edge_ptr)
        scale = max(1.0, float(np.abs(ref).max()))
        cm = runtime.CompiledModel.from_model(model, batch.num_graphs, batch.num_nodes, max(batch.num_edges, 1), max_graph_nodes=promise)
        args = to_dev(batch, dev)
        got = cm.forward(*args).cpu().numpy()
        cm.check()
        assert cm.gcn_stack_timed(args[0], 2) > 0.0
        assert np.abs(got - ref).max() < TOL * scale, (np.abs(got - ref).max(), scale)
        try:
            runtime.set_option("fuse_gcn2", 0)
            lw = runtime.CompiledModel.from_model(model, batch.num_graphs, batch.num_nodes, max(batch.num_edges, 1))
            assert np.abs(lw.forward(*args).cpu().numpy() - got).max() < 5e-5 * scale
        finally:
            runtime.set_option("fuse_gcn2", 1)


@pytest.mark.parametrize("layers,hidden,act,skip,shape,eps", [
    (2, 128, "relu", True, "qm9", 0.0), (3, 128, "relu", True, "qm9", 0.2), (3, 64, "tanh", False, "esol", 0.0),
    (4, 32, "gelu", True, "qm9", -0.3), (5, 128, "sigmoid", True, "esol", 0.1)])
def test_fused_gin_stack(dev, layers, hidden, act, skip, shape, eps):
    """GIN stacks (hidden = out, graphs within the promise) in ONE kernel: (1 + eps) x_i + sum_j x_j, two linears per
    layer with ReLU between them (reference gin_conv, gnn_builder_lib.h:1389-1544), skip connection and the model's
    activation behind the second.  Against the oracle and the layer-by-layer path; the timed entry proves which ran."""
    fin = synthetic.SHAPES[shape]["f_in"]
    model = make_model("gin", in_dim=fin, hidden=hidden, layers=layers, out_dim=hidden, act=act, skip=skip, task_out=3)
    for c in model.gnn_convs:
        c.eps = eps
        c.conv.eps.fill_(eps)
    rng = np.random.default_rng(layers * 10 + hidden)
    for batch in (synthetic.make_batch(shape, 300, seed=layers), pack_graphs(_random_graphs(rng, 100, 40, fin, dense=4))):
        promise = int(np.diff(batch.node_ptr).max())
        ref = O.forward_batched(model.spec(), canon(model), batch.x, batch.coo, batch.node_ptr, batch.edge_ptr)
        scale = max(1.0, float(np.abs(ref).max()))
        cm = runtime.CompiledModel.from_model(model, batch.num_graphs, batch.num_nodes, max(batch.num_edges, 1), max_graph_nodes=promise)
        args = to_dev(batch, dev)
        got = cm.forward(*args).cpu().numpy()
        cm.check()
        assert cm.gcn_stack_timed(args[0], 2) > 0.0
        assert np.abs(got - ref).max() < TOL * scale, (np.abs(got - ref).max(), scale)
        try:
            runtime.set_option("fuse_gcn2", 0)
            lw = runtime.CompiledModel.from_model(model, batch.num_graphs, batch.num_nodes, max(batch.num_edges, 1))
            assert np.abs(lw.forward(*args).cpu().numpy() - got).max() < 5e-5 * scale
        finally:
            runtime.set_option("fuse_gcn2", 1)


@pytest.mark.parametrize("layers,hidden,out_dim,act,skip,pools", [(6, 128, 64, "relu", True, ("add", "mean", "max")), (2, 64, 32, "tanh", True, ("add",)),
                                                                  (3, 128, 20, "sigmoid", False, ("max", "mean")), (4, 32, 4, "gelu", True, ("mean",)),
                                                                  (6, 128, 64, "relu", True, ("add", "mean", "max"))])
def test_fused_stacks_with_a_last_layer_narrower_than_hidden(dev, layers, hidden, out_dim, act, skip, pools):
    """The reference's one published benchmark model has out != hidden (6 layers, 128 / 64: experiments/
    build_base_benchmarks.py:61-81).  GIN stacks take it in the LDS-resident kernel through hidden x hidden zero-padded
    copies of the last layer's matrices (gnnb_model_create), GCN stacks through the kernel's own last-layer slice; the
    pooled rows are out_dim wide.  Sigmoid / tanh / GELU make the padded columns act(0) != 0 -- they must not leak.  Against
    the oracle and the layer-by-layer path."""
    for conv in ("gin", "gcn"):
        model = make_model(conv, in_dim=11, hidden=hidden, layers=layers, out_dim=out_dim, act=act, skip=skip, pools=pools,
                           mlp_hidden=64, mlp_layers=4, task_out=19, seed=layers + out_dim)
        batch = synthetic.make_batch("qm9", 400, seed=out_dim)
        promise = int(np.diff(batch.node_ptr).max())
        ref = O.forward_batched(model.spec(), canon(model), batch.x, batch.coo, batch.node_ptr, batch.edge_ptr)
        scale = max(1.0, float(np.abs(ref).max()))
        cm = runtime.CompiledModel.from_model(model, batch.num_graphs, batch.num_nodes, batch.num_edges, max_graph_nodes=promise)
        args = to_dev(batch, dev)
        got = cm.forward(*args).cpu().numpy()
        cm.check()
        assert cm.last_path().startswith("stack"), (conv, cm.last_path())
        assert np.abs(got - ref).max() < TOL * scale, (conv, np.abs(got - ref).max(), scale)
        lw = runtime.CompiledModel.from_model(model, batch.num_graphs, batch.num_nodes, batch.num_edges)
        out_lw = lw.forward(*args).cpu().numpy()
        assert lw.last_path() == "layerwise" and np.abs(out_lw - got).max() < 5e-5 * scale, conv


@pytest.mark.parametrize("conv,layers,hidden", [("gcn", 3, 64), ("gin", 3, 128), ("gin", 2, 32), ("gcn", 4, 32)])
def test_fused_stacks_with_wide_input_features(dev, conv, layers, hidden):
    """F_in = 20 (two 16-wide k blocks in the first layer's product) and 30 on the deep GCN / GIN fused stacks, random
    multigraphs with graphs of up to 57 nodes (the largest the default 8-row node tiles admit)."""
    for fin in (20, 30):
        model = make_model(conv, in_dim=fin, hidden=hidden, layers=layers, out_dim=hidden, act="relu", task_out=2)
        rng = np.random.default_rng(fin + layers)
        batch = pack_graphs(_random_graphs(rng, 150, 57, fin, dense=3))
        promise = int(np.diff(batch.node_ptr).max())
        ref = O.forward_batched(model.spec(), canon(model), batch.x, batch.coo, batch.node_ptr, batch.edge_ptr)
        cm = runtime.CompiledModel.from_model(model, batch.num_graphs, batch.num_nodes, max(batch.num_edges, 1), max_graph_nodes=promise)
        args = to_dev(batch, dev)
        got = cm.forward(*args).cpu().numpy()
        cm.check()
        assert cm.gcn_stack_timed(args[0], 2) > 0.0
        assert np.abs(got - ref).max() < TOL * max(1.0, float(np.abs(ref).max()))


def test_config1_esol_takes_the_fused_stack(dev):
    """BASELINE config 1 (2-layer GCN d=64, ESOL: graphs of up to 55 nodes) with the reference's MAX_NODES as the promise:
    the 64-row stages take it on the fused stack (one graph of exactly 55 nodes included), same numbers as the oracle and as
    the layer-by-layer path."""
    model = make_model("gcn", in_dim=9, hidden=64, layers=2, out_dim=64, act="relu", pools=("add", "mean", "max"), task_out=1)
    rng = np.random.default_rng(55)
    base = synthetic.make_batch("esol", 600, seed=1)
    graphs = [base.graph(g) for g in range(base.num_graphs)]
    ring = np.stack([np.arange(55), (np.arange(55) + 1) % 55], 1)
    graphs.insert(300, (rng.uniform(-1, 1, (55, 9)).astype(np.float32), np.concatenate([ring, ring[:, ::-1]]).astype(np.int32)))
    batch = pack_graphs(graphs)
    assert int(np.diff(batch.node_ptr).max()) == 55
    ref = O.forward_batched(model.spec(), canon(model), batch.x, batch.coo, batch.node_ptr, batch.edge_ptr)
    cm = runtime.CompiledModel.from_model(model, batch.num_graphs, batch.num_nodes, batch.num_edges, max_graph_nodes=55)
    got = cm.forward(*to_dev(batch, dev)).cpu().numpy()
    cm.check()
    assert cm.gcn_stack_timed(torch.from_numpy(batch.x).to(dev), 2) > 0.0          # the fused stack is what ran
    assert np.abs(got - ref).max() < TOL * max(1.0, float(np.abs(ref).max()))
    try:
        runtime.set_option("fuse_gcn2", 0)
        lw = runtime.CompiledModel.from_model(model, batch.num_graphs, batch.num_nodes, batch.num_edges)
        assert np.abs(lw.forward(*to_dev(batch, dev)).cpu().numpy() - got).max() < 2e-5 * max(1.0, float(np.abs(ref).max()))
    finally:
        runtime.set_option("fuse_gcn2", 1)


# --------------------------------------------------------------------------- BASELINE configs 3, 4, 5 at full size
FULL_SIZE = [
    # name, conv, shape, hidden, layers, pools, graphs per GPU
    ("c3", "gin", "molhiv", 128, 3, ("add",), 4096),
    ("c4", "pna", "qm9", 128, 3, ("add", "mean", "max"), 8192),
    ("c5", "sage", "molhiv", 256, 2, ("add", "mean", "max"), 8192),
]


def _full_size_check(dev, case, as_bench, tol=None):
    tol = TOL if tol is None else tol
    name, conv, shape, hidden, layers, pools, B = case
    fin, out = synthetic.SHAPES[shape]["f_in"], synthetic.SHAPES[shape]["out"]
    model = make_model(conv, in_dim=fin, hidden=hidden, layers=layers, pools=pools, task_out=out, seed=B + hidden)
    batch = synthetic.make_batch(shape, B, seed=31)
    indeg = np.bincount(batch.coo[:, 1], minlength=batch.num_nodes)
    promise = int(np.diff(batch.node_ptr).max()) if as_bench else 0
    cm = runtime.CompiledModel.from_model(model, batch.num_graphs, batch.num_nodes, batch.num_edges, max_graph_nodes=promise)
    if as_bench and conv == "pna":
        assert int(indeg.max()) <= 15
        cm.set_max_degree(int(indeg.max()))                      # bench.py:589-593: the batches' largest in-degree
    out_d = cm.forward(*to_dev(batch, dev)).cpu().numpy()
    cm.check()
    if conv == "pna":
        cm.stream_k_guard()                                      # (13F form: 1153 tiles through the stream-K space)
    assert np.isfinite(out_d).all()
    idx = np.sort(np.random.default_rng(B).choice(B, 256, replace=False))
    idx[0], idx[-1] = 0, B - 1                                   # the batch's two ends are always checked
    big = int(np.argmax(np.diff(batch.node_ptr)))                # its largest graph
    hub = int(np.searchsorted(batch.node_ptr, int(np.argmax(indeg)), side="right") - 1)  # the graph of its highest-degree node
    idx[1], idx[2] = big, hub
    idx = np.unique(idx)
    sub = pack_graphs([batch.graph(int(g)) for g in idx])
    ref = O.forward_batched(model.spec(), canon(model), sub.x, sub.coo, sub.node_ptr, sub.edge_ptr)
    err = np.abs(out_d[idx] - ref).max()
    assert err < tol * max(1.0, float(np.abs(ref).max())), f"{name}: max err {err:.3e} (|ref| max {np.abs(ref).max():.3e})"
    order = np.arange(B - 1, -1, -1)
    rev = pack_graphs([batch.graph(int(g)) for g in order])
    out_rev = cm.forward(*to_dev(rev, dev)).cpu().numpy()
    cm.check()
    assert np.abs(out_rev[::-1] - out_d).max() < 2e-5 * max(1.0, float(np.abs(out_d).max()))
    return cm, model, batch, out_d


@pytest.mark.parametrize("case", FULL_SIZE, ids=lambda c: c[0])
def test_full_size_configs_3_4_5(dev, case):
    """BASELINE configs 3 / 4 / 5 at the per-GPU batch the bench runs (tile tables, grids and the large-K GEMM at
    M = 104 k / 147 k / 209 k rows) WITHOUT any promise -- the general layer-by-layer forms (PNA: the 13F-wide product with
    its stream-K tail) --: 256 sampled graphs (both ends, the largest graph, the graph of the highest-degree node) against
    the oracle at the north-star tolerance, and batch-composition independence (the same graphs in reverse order give the
    same rows)."""
    _full_size_check(dev, case, as_bench=False)


@pytest.mark.parametrize("case", FULL_SIZE, ids=lambda c: c[0])
def test_full_size_configs_3_4_5_on_the_routes_bench_times(dev, case):
    """The same three configs set up EXACTLY as bench.py sets them up (bench.py:584-612): the largest graph of the batch as the
    max_graph_nodes promise and -- PNA -- the batch's largest in-degree as the max_degree promise.  So C4 runs the
    degree-class form at B = 8192 (rows counting-sorted by in-degree over 147 k rows, 5F-wide k_linear_dma in row-class mode,
    its partial last round of tiles), C3 the fused GIN stack, C5 the ring-form first layer + the pooling GEMM.  Same sample
    (both ends, largest graph, highest-degree node's graph) against the oracle, reversed-order independence, and the route is
    asserted, not assumed."""
    cm, model, batch, out_d = _full_size_check(dev, case, as_bench=True)
    assert cm.last_path() == {"c3": "stack", "c4": "layerwise", "c5": "layerwise"}[case[0]]
    if case[1] == "pna":
        # the degree-class form is what ran: switching it off changes the rounding (not the mathematics)
        try:
            runtime.set_option("pna_classes", 0)
            gen = runtime.CompiledModel.from_model(model, batch.num_graphs, batch.num_nodes, batch.num_edges)
            out_g = gen.forward(*to_dev(batch, dev)).cpu().numpy()
        finally:
            runtime.set_option("pna_classes", 1)
        diff = np.abs(out_g - out_d).max()
        assert 0.0 < diff < 5e-5 * max(1.0, float(np.abs(out_d).max()))


@pytest.mark.parametrize("case,math,bound", [
    (("c2", "gcn", "qm9", 128, 2, ("add", "mean", "max"), 4096), 2, 2e-5), (("c2", "gcn", "qm9", 128, 2, ("add", "mean", "max"), 4096), 3, 2e-6),
    (FULL_SIZE[0], 3, 2e-5), (FULL_SIZE[1], 1, None), (FULL_SIZE[1], 3, 2e-5), (FULL_SIZE[2], 1, None), (FULL_SIZE[2], 3, 2e-5)],
    ids=lambda v: v[0] if isinstance(v, tuple) else str(v))
def test_full_size_configs_in_the_opt_in_math_modes(dev, case, math, bound):
    """The opt-in math legs of the bench line at the sizes and on the routes the bench times them: BASELINE config 2 with the
    reduced-precision forms of k_gcn2_zf (math 2 / 3), config 3 with the f16x3 form of the GIN stack kernel, configs 4 and 5 with the bf16x6 (1: fp32-equivalent, the north-star
    tolerance) and f16x3 (3: reduced, 2e-5 of the output scale) GEMMs -- the same sample of 256 graphs against the oracle and
    the same reversed-order check as the fp32 tests."""
    try:
        runtime.set_option("math", math)
        cm, _, _, _ = _full_size_check(dev, case, as_bench=True, tol=bound)
        assert cm.last_path() == {"c2": "stack_zf", "c3": "stack"}.get(case[0], "layerwise")
    finally:
        runtime.set_option("math", 0)


@pytest.mark.parametrize("width", [64, 128, 256])
def test_register_gather_aggregate_matches_the_ring(dev, width):
    """k_aggregate_rg (round 5: no LDS, no barrier; a lane group per destination row, the gather straight into registers) against
    k_aggregate_ring on every kind and launch depth it takes: same sums in the same order (SUM / MEAN / SIMPLE / PNA bit for
    bit; GCN within one rounding -- its coefficient products contract differently), on a batch with isolated nodes, hubs of
    degree > 4 (the CSR tail), empty graphs and a row count that divides by nothing.  Whole models on that form against the
    oracle: `agg_form` 1 on a layer-wise GCN and on PNA with its destination term."""
    rng = np.random.default_rng(width)
    base = synthetic.make_batch("molhiv", 300, seed=4)
    graphs = [base.graph(g) for g in range(300)]
    star = (rng.uniform(-1, 1, (20, 9)).astype(np.float32), np.array([[i, 0] for i in range(1, 20)] + [[0, i] for i in range(1, 20)], np.int32))
    lone = (rng.uniform(-1, 1, (3, 9)).astype(np.float32), np.zeros((0, 2), np.int32))
    empty = (np.zeros((0, 9), np.float32), np.zeros((0, 2), np.int32))
    batch = pack_graphs([empty, star] + graphs[:150] + [lone, empty] + graphs[150:] + [star])
    model = make_model("sage", in_dim=9, hidden=32, layers=1, pools=("add",), task_out=1)
    cm = runtime.CompiledModel.from_model(model, batch.num_graphs, batch.num_nodes, batch.num_edges)
    _, coo, nptr, eptr = to_dev(batch, dev)
    cm.graph_prep(coo, nptr, eptr, batch.num_nodes)
    x = torch.rand(batch.num_nodes, width, device=dev) * 2 - 1
    q = torch.rand(batch.num_nodes, width, device=dev) * 2 - 1
    try:
        for kind, st in (("gcn", None), ("sum", None), ("mean", None), ("simple", None), ("pna", q)):
            runtime.set_option("agg_form", 0)
            ref = cm.aggregate(kind, x, self_term=st, eps=0.25).clone()
            for r, wgs, flags in ((1, 0, 1), (1, 3, 0), (2, 0, 1), (2, 5, 0), (3, 0, 1), (4, 2, 0), (1, 64, 1)):
                runtime.set_option("agg_form", 1)
                runtime.set_option("agg_rg_r", r)
                runtime.set_option("agg_rg_wgs", wgs)
                runtime.set_option("agg_rg_flags", flags)
                got = cm.aggregate(kind, x, self_term=st, eps=0.25)
                if kind == "gcn":
                    assert (got - ref).abs().max().item() < 5e-7, (kind, r, wgs, flags)
                else:
                    assert torch.equal(got, ref), (kind, r, wgs, flags)
    finally:
        for k, v in (("agg_form", 0), ("agg_rg_r", 0), ("agg_rg_wgs", 0), ("agg_rg_flags", 1)):
            runtime.set_option(k, v)
    if width != 128:
        return
    try:
        runtime.set_option("agg_form", 1)
        for conv in ("gcn", "pna", "sage", "gin"):
            m = make_model(conv, in_dim=9, hidden=128, layers=3, pools=("add", "mean", "max"), task_out=2, seed=11)
            ref = O.forward_batched(m.spec(), canon(m), batch.x, batch.coo, batch.node_ptr, batch.edge_ptr)
            c2 = runtime.CompiledModel.from_model(m, batch.num_graphs, batch.num_nodes, batch.num_edges)
            got = c2.forward(*to_dev(batch, dev)).cpu().numpy()
            c2.check()
            assert c2.last_path() == "layerwise"
            assert np.abs(got - ref).max() < TOL * max(1.0, float(np.abs(ref).max())), conv
    finally:
        runtime.set_option("agg_form", 0)


@pytest.mark.parametrize("hidden,out,layers,fin", [(128, 128, 3, 11), (64, 128, 3, 64), (32, 64, 2, 32), (128, 64, 4, 128)])
def test_pna_product_and_aggregate_in_one_kernel(dev, hidden, out, layers, fin):
    """k_pna_pagg (round 5): a full-width PNA layer's source-half product p = x . Wb^T and its max | min | mean | std aggregate
    in one kernel -- whole graphs staged in LDS, p never in HBM -- under the degree promise (no destination term) + the
    max_graph_nodes promise (a graph must fit a 64-row stage).  Against the two-kernel route (pna_pagg = 0: GEMM -> [N, F] ->
    k_aggregate_ring<PNA>), the general form and the oracle, every graph.  Batch: molecules, isolated nodes, a star of degree
    13 (the CSR tail), empty graphs, one-node graphs, a graph of exactly the promised size; widths 128 / 64 / 32; a promise
    beyond the stage (57 + 7 > 64) keeps the two-kernel route with the same numbers."""
    model = make_model("pna", in_dim=fin, hidden=hidden, out_dim=out, layers=layers, act="relu", pools=("add", "mean", "max"), task_out=2, seed=hidden + layers + fin)
    base = synthetic.make_batch("qm9", 500, seed=19)
    rng = np.random.default_rng(fin + hidden)
    empty = (np.zeros((0, fin), np.float32), np.zeros((0, 2), np.int32))

    def regraph(g):
        x, e = base.graph(g)
        return rng.uniform(-1, 1, (x.shape[0], fin)).astype(np.float32), e

    lone = (rng.uniform(-1, 1, (3, fin)).astype(np.float32), np.zeros((0, 2), np.int32))
    one = (rng.uniform(-1, 1, (1, fin)).astype(np.float32), np.zeros((0, 2), np.int32))
    star = (rng.uniform(-1, 1, (14, fin)).astype(np.float32), np.array([[i, 0] for i in range(1, 14)] + [[0, i] for i in range(1, 14)], np.int32))
    ring = np.stack([np.arange(40), (np.arange(40) + 1) % 40], 1)
    big = (rng.uniform(-1, 1, (40, fin)).astype(np.float32), np.concatenate([ring, ring[:, ::-1]]).astype(np.int32))
    graphs = [empty, star] + [regraph(g) for g in range(250)] + [lone, one, one, big, empty] + [regraph(g) for g in range(250, 500)] + [star, empty]
    batch = pack_graphs(graphs)
    maxdeg = int(np.bincount(batch.coo[:, 1]).max())
    maxn = int(np.diff(batch.node_ptr).max())
    assert maxdeg == 13 and maxn == 40
    ref = O.forward_batched(model.spec(), canon(model), batch.x, batch.coo, batch.node_ptr, batch.edge_ptr)
    outs = {}
    try:
        runtime.set_option("pna_first", 0)  # (the narrow first layer's own one-kernel form has its test below: only pna_pagg varies here)
        for name, pagg, promise_n, promise_d in (("one_kernel", 1, maxn, maxdeg), ("two_kernels", 0, maxn, maxdeg), ("beyond_stage", 1, 58, maxdeg), ("general", 1, maxn, 0)):
            runtime.set_option("pna_pagg", pagg)
            cm = runtime.CompiledModel.from_model(model, batch.num_graphs, batch.num_nodes, batch.num_edges, max_graph_nodes=promise_n)
            cm.set_max_degree(promise_d)
            outs[name] = cm.forward(*to_dev(batch, dev)).cpu().numpy()
            cm.check()
            assert cm.last_path() == "layerwise"
            if name == "one_kernel":
                assert np.array_equal(outs[name], cm.forward(*to_dev(batch, dev)).cpu().numpy())
                # the opt-in f16x3 mode (REDUCED precision): the kernel's product on fp16 pieces, the rows split in place once per
                # stage; the class GEMMs in their f16x3 form too -- 2e-5 of the output scale against the oracle
                runtime.set_option("math", 3)
                reduced = cm.forward(*to_dev(batch, dev)).cpu().numpy()
                cm.check()
                runtime.set_option("math", 0)
    finally:
        runtime.set_option("pna_pagg", 1)
        runtime.set_option("pna_first", 1)
        runtime.set_option("math", 0)
    scale = max(1.0, float(np.abs(ref).max()))
    assert np.isfinite(reduced).all() and np.abs(reduced - ref).max() < 2e-5 * scale, np.abs(reduced - ref).max()
    assert not np.array_equal(reduced, outs["one_kernel"])
    for k, v in outs.items():
        assert np.isfinite(v).all() and np.abs(v - ref).max() < TOL * scale, k
    assert np.abs(outs["one_kernel"] - outs["two_kernels"]).max() < 2e-5 * scale
    assert np.array_equal(outs["two_kernels"], outs["beyond_stage"])
    if hidden == 128 and fin != 128:
        assert not np.array_equal(outs["one_kernel"], outs["two_kernels"])  # (the new kernel is what ran: another summation order)


@pytest.mark.parametrize("fin,hidden,out,layers,act", [(9, 256, 256, 2, "relu"), (16, 128, 64, 3, "tanh"), (4, 64, 64, 2, "gelu"), (11, 256, 128, 4, "sigmoid")])
def test_sage_first_layer_forms_the_next_layers_mean(dev, fin, hidden, out, layers, act):
    """k_sage_first_mean (round 5): GraphSAGE's narrow first layer with the stage's output rows kept in LDS, from which the NEXT
    layer's mean aggregate is taken -- out and mean leave the chip once each, the aggregate kernel of layer 1 is not run.  Needs
    the max_graph_nodes promise (<= 49).  Against the two-kernel route (`sage_first_mean` 0: bit-identical -- the same sums in
    the same order) and the oracle, every graph; batch with isolated nodes, a hub of degree 30 (CSR tail), empty and one-node
    graphs, a 49-node ring; a promise beyond the stage keeps the old route."""
    model = make_model("sage", in_dim=fin, hidden=hidden, out_dim=out, layers=layers, act=act, pools=("add", "mean", "max"), task_out=3, seed=fin + hidden)
    rng = np.random.default_rng(fin)
    base = synthetic.make_batch("molhiv", 400, seed=13)
    empty = (np.zeros((0, fin), np.float32), np.zeros((0, 2), np.int32))

    def regraph(g):
        x, e = base.graph(g)
        return rng.uniform(-1, 1, (x.shape[0], fin)).astype(np.float32), e

    one = (rng.uniform(-1, 1, (1, fin)).astype(np.float32), np.zeros((0, 2), np.int32))
    lone = (rng.uniform(-1, 1, (4, fin)).astype(np.float32), np.array([[0, 1]], np.int32))
    star = (rng.uniform(-1, 1, (31, fin)).astype(np.float32), np.array([[i, 0] for i in range(1, 31)] + [[0, i] for i in range(1, 31)], np.int32))
    ring = np.stack([np.arange(49), (np.arange(49) + 1) % 49], 1)
    big = (rng.uniform(-1, 1, (49, fin)).astype(np.float32), np.concatenate([ring, ring[:, ::-1]]).astype(np.int32))
    graphs = [empty, star] + [regraph(g) for g in range(200)] + [one, lone, big, empty, empty] + [regraph(g) for g in range(200, 400)] + [one]
    batch = pack_graphs(graphs)
    maxn = int(np.diff(batch.node_ptr).max())
    assert maxn == 49
    ref = O.forward_batched(model.spec(), canon(model), batch.x, batch.coo, batch.node_ptr, batch.edge_ptr)
    outs = {}
    try:
        for name, on, promise in (("one_kernel", 1, maxn), ("two_kernels", 0, maxn), ("beyond_stage", 1, 50), ("no_promise", 1, 0)):
            runtime.set_option("sage_first_mean", on)
            cm = runtime.CompiledModel.from_model(model, batch.num_graphs, batch.num_nodes, batch.num_edges, max_graph_nodes=promise)
            outs[name] = cm.forward(*to_dev(batch, dev)).cpu().numpy()
            cm.check()
            assert cm.last_path() == "layerwise"
    finally:
        runtime.set_option("sage_first_mean", 1)
    scale = max(1.0, float(np.abs(ref).max()))
    for k, v in outs.items():
        assert np.isfinite(v).all() and np.abs(v - ref).max() < TOL * scale, k
    for k in ("two_kernels", "beyond_stage", "no_promise"):
        assert np.array_equal(outs["one_kernel"], outs[k]), k


@pytest.mark.parametrize("fin,hidden,out,layers,act,delta", [(11, 128, 128, 3, "relu", 1.0), (9, 128, 64, 2, "tanh", 2.5), (12, 64, 64, 2, "gelu", 1.0),
                                                             (10, 128, 128, 1, "sigmoid", 0.7), (11, 64, 128, 2, "relu", 1.0)])
def test_narrow_pna_layer_in_one_kernel(dev, fin, hidden, out, layers, act, delta):
    """k_pna_first (round 5): a PNA layer with a narrow input (the first) as ONE kernel -- pre-NN per node, the four statistics with
    the destination term, degree scalers, the 13F-wide lin-folded post-NN product on the matrix cores -- with whole graphs
    staged in LDS (max_graph_nodes promise).  Against the layer-by-layer kernels (`pna_first` 0), with and without the degree
    promise, and the oracle, every graph; isolated nodes (degree 0: the scalers of degree 1, all statistics 0), a hub of degree
    13 (the CSR tail), empty / one-node graphs, a 57-node ring (the largest graph a stage takes with 8-row tiles), delta != 1,
    hidden 64 (first layer 9 / 12 -> 64)."""
    model = make_model("pna", in_dim=fin, hidden=hidden, out_dim=out, layers=layers, act=act, pools=("add", "mean", "max"), task_out=2, seed=fin + hidden + layers)
    for conv in model.gnn_convs:  # (GNNModel never passes delta: reference models.py:546-548; set it on the layers)
        conv.delta_scaler = delta
        conv.conv.aggr_module.avg_deg_log = torch.Tensor([delta])
    rng = np.random.default_rng(fin * 7)
    base = synthetic.make_batch("qm9", 400, seed=23)
    empty = (np.zeros((0, fin), np.float32), np.zeros((0, 2), np.int32))

    def regraph(g):
        x, e = base.graph(g)
        return rng.uniform(-1, 1, (x.shape[0], fin)).astype(np.float32), e

    one = (rng.uniform(-1, 1, (1, fin)).astype(np.float32), np.zeros((0, 2), np.int32))
    lone = (rng.uniform(-1, 1, (3, fin)).astype(np.float32), np.zeros((0, 2), np.int32))
    star = (rng.uniform(-1, 1, (14, fin)).astype(np.float32), np.array([[i, 0] for i in range(1, 14)] + [[0, i] for i in range(1, 14)], np.int32))
    ring = np.stack([np.arange(57), (np.arange(57) + 1) % 57], 1)
    big = (rng.uniform(-1, 1, (57, fin)).astype(np.float32), np.concatenate([ring, ring[:, ::-1]]).astype(np.int32))
    graphs = [empty, star] + [regraph(g) for g in range(200)] + [one, lone, big, empty] + [regraph(g) for g in range(200, 400)] + [star, one]
    batch = pack_graphs(graphs)
    maxn, maxdeg = int(np.diff(batch.node_ptr).max()), int(np.bincount(batch.coo[:, 1]).max())
    assert maxn == 57 and maxdeg == 13
    assert abs(model.spec()["pna_delta"] - delta) < 1e-6
    ref = O.forward_batched(model.spec(), canon(model), batch.x, batch.coo, batch.node_ptr, batch.edge_ptr)
    outs = {}
    try:
        for name, on, promise_n, promise_d in (("one_kernel", 1, maxn, 0), ("layer_by_layer", 0, maxn, 0), ("one_kernel_classes", 1, maxn, maxdeg),
                                               ("layer_by_layer_classes", 0, maxn, maxdeg), ("beyond_stage", 1, 58, 0)):
            runtime.set_option("pna_first", on)
            cm = runtime.CompiledModel.from_model(model, batch.num_graphs, batch.num_nodes, batch.num_edges, max_graph_nodes=promise_n)
            cm.set_max_degree(promise_d)
            outs[name] = cm.forward(*to_dev(batch, dev)).cpu().numpy()
            cm.check()
            if name == "one_kernel":
                assert np.array_equal(outs[name], cm.forward(*to_dev(batch, dev)).cpu().numpy())
    finally:
        runtime.set_option("pna_first", 1)
    scale = max(1.0, float(np.abs(ref).max()))
    for k, v in outs.items():
        assert np.isfinite(v).all() and np.abs(v - ref).max() < TOL * scale, k
    assert np.abs(outs["one_kernel"] - outs["layer_by_layer"]).max() < 3e-5 * scale
    assert np.array_equal(outs["layer_by_layer"], outs["beyond_stage"])
    assert not np.array_equal(outs["one_kernel"], outs["layer_by_layer"])  # (the new kernel is what ran)


@pytest.mark.parametrize("conv,fin,hidden", [("pna", 11, 128), ("sage", 9, 256), ("pna", 10, 64)])
def test_lds_staged_layers_on_dense_graphs(dev, conv, fin, hidden):
    """The LDS-staged layer kernels of round 5 (k_pna_first, k_pna_pagg, k_sage_first_mean) stage a slice of the CSR `col` array
    for rows of degree > 4 -- 448 / 512 entries per stage.  Cliques of 12 nodes (132 directed edges, degree 11) put 600+ edges
    into a stage: the slice does not fit and those rows read `col` from global memory instead.  Every graph against the oracle
    and against the layer-by-layer kernels."""
    model = make_model(conv, in_dim=fin, hidden=hidden, layers=3, act="relu", pools=("add", "mean", "max"), task_out=2, seed=fin)
    rng = np.random.default_rng(hidden)
    clique = np.array([[i, j] for i in range(12) for j in range(12) if i != j], np.int32)
    base = synthetic.make_batch("qm9", 60, seed=3)
    graphs = []
    for g in range(60):
        graphs.append((rng.uniform(-1, 1, (12, fin)).astype(np.float32), clique))
        x, e = base.graph(g)
        graphs.append((rng.uniform(-1, 1, (x.shape[0], fin)).astype(np.float32), e))
    batch = pack_graphs(graphs)
    maxn, maxdeg = int(np.diff(batch.node_ptr).max()), int(np.bincount(batch.coo[:, 1]).max())
    assert maxdeg == 11
    ref = O.forward_batched(model.spec(), canon(model), batch.x, batch.coo, batch.node_ptr, batch.edge_ptr)
    outs = {}
    names = ("pna_first", "pna_pagg", "sage_first_mean")
    try:
        for on in (1, 0):
            for n in names:
                runtime.set_option(n, on)
            cm = runtime.CompiledModel.from_model(model, batch.num_graphs, batch.num_nodes, batch.num_edges, max_graph_nodes=maxn)
            if conv == "pna":
                cm.set_max_degree(maxdeg)
            outs[on] = cm.forward(*to_dev(batch, dev)).cpu().numpy()
            cm.check()
    finally:
        for n in names:
            runtime.set_option(n, 1)
    scale = max(1.0, float(np.abs(ref).max()))
    assert np.abs(outs[1] - ref).max() < TOL * scale and np.abs(outs[0] - ref).max() < TOL * scale
    assert np.abs(outs[1] - outs[0]).max() < 3e-5 * scale and not (conv == "pna" and np.array_equal(outs[1], outs[0]))


@pytest.mark.parametrize("conv,layers,nbatch,promise", [("gin", 3, 4096, 47), ("gcn", 4, 1500, 47), ("gin", 2, 300, 29), ("gin", 3, 40, 47)])
def test_stack_workgroups_take_whole_stages_of_the_global_stage_list(dev, conv, layers, nbatch, promise):
    """`stage_cut` 1 (round 5, opt-in): graph prep plans the conv-stack kernel's workgroup runs as whole stages of the batch's
    global greedy stage list (k_stage_cut: binary lifting over the stage chain) instead of equal tile counts.  Same graphs, same
    per-graph arithmetic: against `stage_cut` 0 and the oracle on sampled graphs; batches with more and with fewer stages than
    workgroups, a large segment behind the stack."""
    model = make_model(conv, in_dim=9, hidden=128, layers=layers, out_dim=128, act="relu", pools=("add", "mean", "max"), task_out=2, seed=layers)
    batch = synthetic.make_batch("molhiv" if promise == 47 else "qm9", nbatch, seed=17)
    if promise == 29:
        rng = np.random.default_rng(1)
        batch = pack_graphs([(rng.uniform(-1, 1, (batch.graph(g)[0].shape[0], 9)).astype(np.float32), batch.graph(g)[1]) for g in range(nbatch)])
    assert int(np.diff(batch.node_ptr).max()) <= promise
    outs = {}
    try:
        for on in (1, 0):
            runtime.set_option("stage_cut", on)
            cm = runtime.CompiledModel.from_model(model, batch.num_graphs, batch.num_nodes, batch.num_edges, max_graph_nodes=promise)
            outs[on] = cm.forward(*to_dev(batch, dev)).cpu().numpy()
            cm.check()
            assert cm.last_path() == "stack"
            assert np.array_equal(outs[on], cm.forward(*to_dev(batch, dev)).cpu().numpy())
    finally:
        runtime.set_option("stage_cut", 0)
    idx = np.unique(np.concatenate([[0, nbatch - 1], np.random.default_rng(2).choice(nbatch, min(nbatch, 200), replace=False)]))
    sub = pack_graphs([batch.graph(int(g)) for g in idx])
    ref = O.forward_batched(model.spec(), canon(model), sub.x, sub.coo, sub.node_ptr, sub.edge_ptr)
    scale = max(1.0, float(np.abs(ref).max()))
    assert np.abs(outs[1][idx] - ref).max() < TOL * scale
    assert np.abs(outs[1] - outs[0]).max() < 2e-5 * scale


def test_pna_degree_promise_with_an_empty_batch(dev):
    """A PNA workspace with a max_degree promise and a batch WITHOUT nodes (graph prep allows it): no class tables are written
    for such a batch, so the forward must not take the class GEMM over stale tables (round-4 advisor finding).  Every graph's
    output is the head applied to a pooled row of zeros; the workspace then runs a real batch correctly."""
    model = make_model("pna", in_dim=11, hidden=128, layers=2, pools=("add", "mean", "max"), task_out=3, seed=5)
    real = synthetic.make_batch("qm9", 300, seed=2)
    empty = pack_graphs([(np.zeros((0, 11), np.float32), np.zeros((0, 2), np.int32))] * 5)
    cm = runtime.CompiledModel.from_model(model, real.num_graphs, real.num_nodes, real.num_edges)
    cm.set_max_degree(int(np.bincount(real.coo[:, 1]).max()))
    ref_real = O.forward_batched(model.spec(), canon(model), real.x, real.coo, real.node_ptr, real.edge_ptr)
    ref_empty = O.forward_batched(model.spec(), canon(model), empty.x, empty.coo, empty.node_ptr, empty.edge_ptr)
    for _ in range(2):
        got = cm.forward(*to_dev(real, dev)).cpu().numpy()
        cm.check()
        assert np.abs(got - ref_real).max() < TOL * max(1.0, float(np.abs(ref_real).max()))
        got0 = cm.forward_host(empty.x, empty.coo, empty.node_ptr, empty.edge_ptr)   # (host entry: checks the batch itself)
        assert got0.shape == (5, 3) and np.abs(got0 - ref_empty).max() < 1e-6


def test_full_size_config3_on_the_fused_gin_stack(dev):
    """BASELINE config 3 (GIN L3 d=128, 4096 molhiv-shaped graphs) the way bench.py runs it: with the largest graph of
    the batch as the promise the whole stack runs in the fused kernel.  256 sampled graphs (both ends, the largest
    graph) against the oracle; reversed order permutes the rows; same numbers as the layer-by-layer route."""
    model = make_model("gin", in_dim=9, hidden=128, layers=3, pools=("add",), task_out=1, seed=7)
    B = 4096
    batch = synthetic.make_batch("molhiv", B, seed=31)
    promise = int(np.diff(batch.node_ptr).max())
    assert promise <= 61
    cm = runtime.CompiledModel.from_model(model, batch.num_graphs, batch.num_nodes, batch.num_edges, max_graph_nodes=promise)
    args = to_dev(batch, dev)
    out_d = cm.forward(*args).cpu().numpy()
    cm.check()
    assert cm.gcn_stack_timed(args[0], 2) > 0.0
    idx = np.sort(np.random.default_rng(3).choice(B, 256, replace=False))
    idx[0], idx[-1], idx[1] = 0, B - 1, int(np.argmax(np.diff(batch.node_ptr)))
    sub = pack_graphs([batch.graph(int(g)) for g in idx])
    ref = O.forward_batched(model.spec(), canon(model), sub.x, sub.coo, sub.node_ptr, sub.edge_ptr)
    scale = max(1.0, float(np.abs(ref).max()))
    assert np.abs(out_d[idx] - ref).max() < TOL * scale
    rev = pack_graphs([batch.graph(int(g)) for g in range(B - 1, -1, -1)])
    out_rev = cm.forward(*to_dev(rev, dev)).cpu().numpy()
    assert np.abs(out_rev[::-1] - out_d).max() < 2e-5 * max(1.0, float(np.abs(out_d).max()))
    lw = runtime.CompiledModel.from_model(model, batch.num_graphs, batch.num_nodes, batch.num_edges)
    assert np.abs(lw.forward(*args).cpu().numpy() - out_d).max() < 5e-5 * max(1.0, float(np.abs(out_d).max()))


@pytest.mark.parametrize("conv,layers", [("gin", 3), ("gcn", 3), ("gin", 2)])
def test_opt_in_math_mode_never_leaves_the_stack(dev, conv, layers):
    """The bf16x6 math mode exists for the plain two-layer GCN stack and the GEMM kernels.  A GIN stack or a deeper GCN stack
    keeps its fp32 stack kernel when the mode is switched on (round 2 sent those models down the layer-by-layer path:
    math = 1 made config 3 slower)."""
    model = make_model(conv, in_dim=9, hidden=128, layers=layers, out_dim=128, act="relu", pools=("add",), task_out=1, seed=3)
    batch = synthetic.make_batch("molhiv", 300, seed=9)
    ref = O.forward_batched(model.spec(), canon(model), batch.x, batch.coo, batch.node_ptr, batch.edge_ptr)
    try:
        runtime.set_option("math", 1)
        cm = runtime.CompiledModel.from_model(model, batch.num_graphs, batch.num_nodes, batch.num_edges,
                                              max_graph_nodes=int(np.diff(batch.node_ptr).max()))
        out = cm.forward(*to_dev(batch, dev)).cpu().numpy()
        cm.check()
        assert cm.last_path() == "stack"
    finally:
        runtime.set_option("math", 0)
    assert np.abs(out - ref).max() < TOL * max(1.0, float(np.abs(ref).max()))


def _forward_with_large_segment(model, batch, limit, dev, **kw):
    """Order the batch (graphs beyond `limit` nodes last), run it with the large segment set, return the outputs in the
    caller's graph order plus the compiled model."""
    from gnnbuilder_amd.batching import order_large_last
    ordered, perm, (g0, n0, e0) = order_large_last(batch, limit)
    small_max = int(np.diff(ordered.node_ptr)[:g0].max()) if g0 else 1
    cm = runtime.CompiledModel.from_model(model, ordered.num_graphs, ordered.num_nodes, max(ordered.num_edges, 1),
                                          max_graph_nodes=small_max, **kw)
    if g0 < ordered.num_graphs:
        cm.set_large_segment(g0, n0, e0)
    out = cm.forward(*to_dev(ordered, dev)).cpu().numpy()
    cm.check()
    return out[np.argsort(perm)], cm, g0


@pytest.mark.parametrize("fork,math", [(2, 0), (1, 0), (0, 0), (2, 3), (0, 3)])
@pytest.mark.parametrize("conv,layers,limit", [("gin", 3, 57), ("gcn", 2, 40), ("gcn", 3, 57), ("gin", 2, 30)])
def test_large_segment_keeps_the_stack_for_the_rest_of_the_batch(dev, conv, layers, limit, fork, math):
    """Graphs beyond the stage capacity no longer demote the whole batch (reference: any graph up to MAX_NODES takes the
    same dataflow, model.cpp.jinja:5-22): ordered last and named as the large segment they run layer by layer, the rest
    stays in the LDS-resident stack, one pooled matrix, one readout.  Heavy-tailed molhiv-shaped batch + a 300-node graph
    + empty graphs on both sides of the boundary; against the oracle, and the path the workspace reports."""
    model = make_model(conv, in_dim=9, hidden=128, layers=layers, out_dim=128, act="relu", pools=("add", "max", "mean"), task_out=3, seed=5)
    b0 = synthetic.make_batch("molhiv_tail", 500, seed=11)
    rng = np.random.default_rng(limit)
    n_big = 300
    big_e = np.stack([rng.integers(0, n_big, 900), rng.integers(0, n_big, 900)], 1).astype(np.int32)
    empty = (np.zeros((0, 9), np.float32), np.zeros((0, 2), np.int32))
    graphs = [b0.graph(g) for g in range(250)] + [empty, (rng.uniform(-1, 1, (n_big, 9)).astype(np.float32), big_e), empty] + \
             [b0.graph(g) for g in range(250, 500)]
    batch = pack_graphs(graphs)
    assert int(np.diff(batch.node_ptr).max()) == n_big and (np.diff(batch.node_ptr) > limit).sum() >= 5
    ref = O.forward_batched(model.spec(), canon(model), batch.x, batch.coo, batch.node_ptr, batch.edge_ptr)
    try:  # 2: the segment's small kernels (k_conv_rows) behind the stack kernel (default); 1: on a forked stream; 0: the big kernels
        runtime.set_option("large_fork", fork)
        runtime.set_option("math", math)  # (3: the opt-in f16x3 forms of the stack kernels and of the large segment's big GEMMs)
        out, cm, g0 = _forward_with_large_segment(model, batch, limit, dev)
    finally:
        runtime.set_option("large_fork", 2)
        runtime.set_option("math", 0)
    assert g0 < batch.num_graphs
    assert cm.last_path() in ("stack+large_layerwise", "stack_zf+large_layerwise"), cm.last_path()
    scale = max(1.0, float(np.abs(ref).max()))
    assert np.abs(out - ref).max() < (2e-5 if math else TOL) * scale
    # the same workspace without the segment and with an honest promise: the whole batch goes layer by layer, same numbers
    cm.set_large_segment()
    cm.set_max_graph_nodes(n_big)
    from gnnbuilder_amd.batching import order_large_last
    ordered, perm, _ = order_large_last(batch, limit)
    lw = cm.forward(*to_dev(ordered, dev)).cpu().numpy()[np.argsort(perm)]
    assert cm.last_path() == "layerwise"
    assert np.abs(lw - out).max() < 5e-5 * scale


@pytest.mark.parametrize("conv,layers,hidden,degree", [("sage", 2, 256, False), ("sage", 3, 128, False), ("pna", 3, 128, True), ("pna", 2, 64, False)])
def test_large_segment_under_graphsage_and_pna(dev, conv, layers, hidden, degree):
    """Round-5 advisor finding (high): with a large segment set the max_graph_nodes promise covers graphs [0, promise_graphs)
    only -- graph prep validates nothing about the rest -- but GraphSAGE / PNA run the WHOLE batch layer by layer, and their
    stage kernels (k_sage_first_mean, k_pna_first, k_pna_pagg: whole graphs in a 56 / 64-row stage) tested the promise alone:
    the large graphs got clamped sources, unflagged.  Those launchers now refuse such a batch (the layer-by-layer kernels
    run); every graph against the oracle, the 300-node graph included."""
    model = make_model(conv, in_dim=9, hidden=hidden, layers=layers, out_dim=hidden, act="relu", pools=("add", "max", "mean"), task_out=3, seed=6)
    b0 = synthetic.make_batch("molhiv_tail", 300, seed=12)
    rng = np.random.default_rng(40)
    n_big = 300
    big_e = np.stack([rng.integers(0, n_big, 600), rng.integers(0, n_big, 600)], 1).astype(np.int32)
    graphs = [b0.graph(g) for g in range(150)] + [(rng.uniform(-1, 1, (n_big, 9)).astype(np.float32), big_e)] + [b0.graph(g) for g in range(150, 300)]
    batch = pack_graphs(graphs)
    limit = 40
    assert (np.diff(batch.node_ptr) > limit).sum() >= 3
    ref = O.forward_batched(model.spec(), canon(model), batch.x, batch.coo, batch.node_ptr, batch.edge_ptr)
    from gnnbuilder_amd.batching import order_large_last
    ordered, perm, (g0, n0, e0) = order_large_last(batch, limit)
    cm = runtime.CompiledModel.from_model(model, ordered.num_graphs, ordered.num_nodes, ordered.num_edges, max_graph_nodes=limit)
    if degree:  # (PNA's degree-class form: the promise holds for the whole batch)
        maxdeg = int(np.bincount(ordered.coo[:, 1], minlength=ordered.num_nodes).max())
        if maxdeg <= 15:
            cm.set_max_degree(maxdeg)
    cm.set_large_segment(g0, n0, e0)
    out = cm.forward(*to_dev(ordered, dev)).cpu().numpy()[np.argsort(perm)]
    cm.check()
    assert cm.last_path() == "layerwise"
    scale = max(1.0, float(np.abs(ref).max()))
    err = np.abs(out - ref).max(axis=1)
    assert err.max() < TOL * scale, (int(err.argmax()), float(err.max()))
    # the same batch without a segment and an honest promise: layer by layer too (other launch shapes: same numbers, not the same bits)
    cm2 = runtime.CompiledModel.from_model(model, ordered.num_graphs, ordered.num_nodes, ordered.num_edges, max_graph_nodes=n_big)
    if degree and maxdeg <= 15:
        cm2.set_max_degree(maxdeg)
    out2 = cm2.forward(*to_dev(ordered, dev)).cpu().numpy()[np.argsort(perm)]
    cm2.check()
    assert cm2.last_path() == "layerwise" and np.abs(out - out2).max() < 5e-5 * scale


def test_large_segment_edge_cases(dev):
    """All graphs large (segment starts at graph 0: plain layer-by-layer run), no graph large (segment empty: plain stack
    run), a segment outside the batch is refused, and a graph in FRONT of the segment that breaks the promise is still
    flagged by graph prep."""
    model = make_model("gin", in_dim=9, hidden=64, layers=2, out_dim=64, act="relu", pools=("add",), task_out=2)
    batch = synthetic.make_batch("molhiv_tail", 200, seed=3)
    ref = O.forward_batched(model.spec(), canon(model), batch.x, batch.coo, batch.node_ptr, batch.edge_ptr)
    scale = max(1.0, float(np.abs(ref).max()))
    out, cm, g0 = _forward_with_large_segment(model, batch, 2, dev)       # every graph has >= 3 nodes: all large
    assert g0 == 0 and cm.last_path() == "layerwise" and np.abs(out - ref).max() < TOL * scale
    out, cm, g0 = _forward_with_large_segment(model, batch, 250, dev)     # none large
    assert g0 == batch.num_graphs and cm.last_path() == "layerwise" or cm.last_path().startswith("stack")
    assert np.abs(out - ref).max() < TOL * scale
    with pytest.raises(runtime.GnnbError):
        cm.set_large_segment(batch.num_graphs + 5, batch.num_nodes, batch.num_edges)
        cm.forward(*to_dev(batch, dev))
    cm.set_large_segment()
    # promise broken in front of the segment
    sizes = np.diff(batch.node_ptr)
    cm2 = runtime.CompiledModel.from_model(model, batch.num_graphs, batch.num_nodes, batch.num_edges, max_graph_nodes=int(sizes[:100].max()) - 1)
    cm2.set_large_segment(100, int(batch.node_ptr[100]), int(batch.edge_ptr[100]))
    cm2.forward(*to_dev(batch, dev))
    with pytest.raises(runtime.GnnbError):
        cm2.check()


def test_stale_large_segment_triple_is_flagged(dev):
    """The large-segment triple is sticky workspace state (advisor, round 3): a triple left over from ANOTHER batch stays in
    range but names offsets that are not node_ptr / edge_ptr of its first graph -- the rows between the two boundaries
    would belong to neither half.  Graph prep checks the triple on the device: the batch is flagged (GNNB_ERR_GRAPH), a
    triple that matches passes, and so does the same workspace once the segment is re-stated for the new batch."""
    from gnnbuilder_amd.batching import order_large_last
    model = make_model("gin", in_dim=9, hidden=64, layers=2, out_dim=64, act="relu", pools=("add",), task_out=2)
    a = synthetic.make_batch("molhiv_tail", 300, seed=5)
    b = synthetic.make_batch("molhiv_tail", 300, seed=6)
    oa, _, (ga, na, ea) = order_large_last(a, 40)
    ob, permb, (gb, nb_, eb) = order_large_last(b, 40)
    assert 0 < ga < a.num_graphs and 0 < gb < b.num_graphs and (na, ea) != (int(ob.node_ptr[ga]), int(ob.edge_ptr[ga]))
    cap = max(oa.num_graphs, ob.num_graphs), max(oa.num_nodes, ob.num_nodes), max(oa.num_edges, ob.num_edges)
    cm = runtime.CompiledModel.from_model(model, *cap, max_graph_nodes=40)
    cm.set_large_segment(ga, na, ea)
    cm.forward(*to_dev(oa, dev))
    cm.check()  # the triple describes this batch
    cm.forward(*to_dev(ob, dev))  # ... and is stale for this one (still inside the batch: the host check cannot see it)
    with pytest.raises(runtime.GnnbError):
        cm.check()
    cm.set_large_segment(gb, nb_, eb)
    out = cm.forward(*to_dev(ob, dev)).cpu().numpy()[np.argsort(permb)]
    cm.check()
    ref = O.forward_batched(model.spec(), canon(model), b.x, b.coo, b.node_ptr, b.edge_ptr)
    assert np.abs(out - ref).max() < TOL * max(1.0, float(np.abs(ref).max()))


def test_full_size_config3_with_the_heavy_tail(dev):
    """BASELINE config 3 at full size on a batch with the real data set's heavy tail (synthetic 'molhiv_tail': log-normal
    sizes up to 222 nodes; about 1 graph in 100 beyond the 57-node stage limit): fused GIN stack for the bulk, layer by
    layer for the large segment.  256 sampled graphs incl. both ends, the largest graph and EVERY large graph's neighbours
    in the order, against the oracle; reversed order permutes the rows."""
    model = make_model("gin", in_dim=9, hidden=128, layers=3, pools=("add",), task_out=1, seed=7)
    B = 4096
    batch = synthetic.make_batch("molhiv_tail", B, seed=31)
    sizes = np.diff(batch.node_ptr)
    assert sizes.max() > 61 and (sizes > 57).sum() >= 20
    out_d, cm, g0 = _forward_with_large_segment(model, batch, 57, dev)
    assert cm.last_path() == "stack+large_layerwise"
    large = np.flatnonzero(sizes > 57)
    idx = np.unique(np.concatenate([np.random.default_rng(3).choice(B, 200, replace=False), [0, B - 1, int(np.argmax(sizes))],
                                    large[:40], np.clip(large[:8] + 1, 0, B - 1)]))
    sub = pack_graphs([batch.graph(int(g)) for g in idx])
    ref = O.forward_batched(model.spec(), canon(model), sub.x, sub.coo, sub.node_ptr, sub.edge_ptr)
    scale = max(1.0, float(np.abs(ref).max()))
    assert np.abs(out_d[idx] - ref).max() < TOL * scale
    rev = pack_graphs([batch.graph(int(g)) for g in range(B - 1, -1, -1)])
    out_rev, _, _ = _forward_with_large_segment(model, rev, 57, dev)
    assert np.abs(out_rev[::-1] - out_d).max() < 2e-5 * max(1.0, float(np.abs(out_d).max()))


# --------------------------------------------------------------------------- malformed batches stay inside the buffers
@pytest.mark.parametrize("conv", ["gcn", "gin", "sage", "pna"])
def test_malformed_batches_are_contained(dev, conv):
    """Whatever the ptr arrays and the edge list hold, graph prep leaves tables that the compute kernels can follow
    without leaving the buffers (advisor finding: rejected graphs used to leave rows uninitialised).  The whole
    forward is run on each malformed batch -- incl. one with a graph of more than 256 nodes (scan path of the prep
    kernel) -- the error flag is raised, and the SAME workspace then gives correct results on a good batch."""
    fin = 9
    model = make_model(conv, in_dim=fin, hidden=32, layers=2, task_out=1)
    good = synthetic.make_batch("molhiv", 96, seed=2)
    rng = np.random.default_rng(7)
    n_big = 300
    big_e = np.stack([rng.integers(0, n_big, 900), rng.integers(0, n_big, 900)], 1).astype(np.int32)
    withbig = pack_graphs([good.graph(g) for g in range(40)] + [(rng.uniform(-1, 1, (n_big, fin)).astype(np.float32), big_e)] +
                          [good.graph(g) for g in range(40, 96)])
    cases = []
    b = withbig
    nonmono = b.node_ptr.copy(); nonmono[30], nonmono[31] = nonmono[31], nonmono[30]
    cases.append(("node_ptr not monotone", b.coo, nonmono, b.edge_ptr))
    shifted = b.node_ptr.copy(); shifted[0] = 5
    cases.append(("node_ptr[0] != 0", b.coo, shifted, b.edge_ptr))
    short = b.node_ptr.copy(); short[-1] -= 40
    cases.append(("node_ptr[B] != N", b.coo, short, b.edge_ptr))
    huge = b.node_ptr.copy(); huge[50] = 2_000_000_000
    cases.append(("node_ptr entry out of range", b.coo, huge, b.edge_ptr))
    neg = b.edge_ptr.copy(); neg[41] = -7
    cases.append(("edge_ptr negative", b.coo, b.node_ptr, neg))
    ebig = b.edge_ptr.copy(); ebig[10] = b.num_edges + 1000
    cases.append(("edge_ptr beyond E", b.coo, b.node_ptr, ebig))
    wild = b.coo.copy(); wild[::17, 0] = rng.integers(-5, b.num_nodes + 5, wild[::17].shape[0]); wild[5::29, 1] = 2_000_000_000
    cases.append(("edges leave their graphs", wild, b.node_ptr, b.edge_ptr))
    cm = runtime.CompiledModel.from_model(model, b.num_graphs, b.num_nodes, b.num_edges)
    xd = torch.from_numpy(b.x).to(dev)
    for promise in (0, 64):      # 64: the 64-node prep variant + (GCN) the promise-sized stages see the same garbage
        cm.set_max_graph_nodes(promise)
        for what, coo, nptr, eptr in cases:
            args = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (coo, nptr, eptr)]
            out = cm.forward(xd, *args)
            torch.cuda.synchronize()                               # a fault would surface here
            with pytest.raises(runtime.GnnbError, match="malformed batch"):
                cm.check()
            assert out.shape == (b.num_graphs, 1), what
    cm.set_max_graph_nodes(0)
    out = cm.forward(*to_dev(withbig, dev)).cpu().numpy()
    cm.check()
    ref = O.forward_batched(model.spec(), canon(model), b.x, b.coo, b.node_ptr, b.edge_ptr)
    assert np.abs(out - ref).max() < TOL
