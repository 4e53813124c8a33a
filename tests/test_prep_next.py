"""gnnb_forward_prepared_prep_next (ABI 104): the forward of one batch + the graph prep of the stream's NEXT batch in one call.

Reference: every graph's forward starts with compute_degree_tables + compute_neighbor_tables (gnn_builder_lib.h:1051-1124,
model.cpp.jinja:737-765); batched, that is one graph-prep launch in front of the conv stack.  Where the forward runs the 2-layer
GCN stack kernel (k_gcn2_zf) the prep of the following batch runs INSIDE that kernel, on another workspace's tables.  What
must hold: outputs and tables bit-identical to gnnb_forward_batched per batch -- whatever the two batches' sizes --, the
device-side validation of the next batch still works, and every model / workspace the in-kernel form does not cover gives
the same results through the ordinary launch.  Needs a real MI355X (``-m gpu``).
"""
import numpy as np
import pytest
import torch

from gnnbuilder_amd import runtime, synthetic
from helpers import make_model, to_dev

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    runtime.load_library(require_gpu=True)  # fails loudly: no fallback
    return torch.device("cuda:0")


def _model(conv="gcn", hidden=128, layers=2, act="relu", in_dim=11, seed=3):
    return make_model(conv, in_dim=in_dim, hidden=hidden, layers=layers, out_dim=hidden, act=act, pools=("add", "mean", "max"), task_out=19, seed=seed)


def _compile(model, cap, promise):
    return runtime.CompiledModel.from_model(model, cap[0], cap[1], cap[2], max_graph_nodes=promise)


def _caps(batches):
    return (max(b.num_graphs for b in batches), max(b.num_nodes for b in batches), max(b.num_edges for b in batches))


def _tables(cm):
    row_ptr, col, in_deg = cm.tables_to_host()
    return row_ptr.copy(), col.copy(), in_deg.copy(), cm.edge_index_table_to_host().copy()


def _run_pipeline(model, batches, dev, promise, tables_of=None):
    """batches through two alternating workspaces; returns the outputs (and the tables every batch was forwarded with)."""
    cap = _caps(batches)
    pair = [_compile(model, cap, promise), _compile(model, cap, promise)]
    args = [to_dev(b, dev) for b in batches]
    outs, paths, tabs = [], [], []
    pair[0].graph_prep(args[0][1], args[0][2], args[0][3], batches[0].num_nodes)
    for i, b in enumerate(batches):
        cur, nxt = pair[i & 1], pair[(i + 1) & 1]
        if tables_of is not None:
            cur.check()
            tabs.append(_tables(cur))
        if i + 1 < len(batches):
            a = args[i + 1]
            outs.append(cur.forward_prepared_prep_next(args[i][0], nxt, a[1], a[2], a[3], batches[i + 1].num_nodes).cpu().numpy())
        else:
            outs.append(cur.forward_prepared(args[i][0]).cpu().numpy())
        paths.append(cur.last_path())
    for cm in pair:
        cm.check()
    return outs, paths, tabs


def _run_plain(model, batches, dev, promise, want_tables=False):
    cap = _caps(batches)
    cm = _compile(model, cap, promise)
    outs, tabs = [], []
    for b in batches:
        outs.append(cm.forward(*to_dev(b, dev)).cpu().numpy())
        cm.check()
        if want_tables:
            tabs.append(_tables(cm))
    return outs, tabs


def _qm9_batches(counts, seed=0):
    return [synthetic.make_batch("qm9", c, seed=seed + i) for i, c in enumerate(counts)]


@pytest.mark.parametrize("counts", [(1024, 1024, 1024, 1024), (300, 5000, 7, 2100, 1), (64, 3, 4100)], ids=["c2", "ragged", "few_then_many"])
def test_pipeline_is_bit_identical_to_forward_batched(dev, counts):
    model = _model()
    batches = _qm9_batches(counts)
    promise = max(int(np.diff(b.node_ptr).max()) for b in batches)
    assert promise <= 64
    want, want_tabs = _run_plain(model, batches, dev, promise, want_tables=True)
    got, paths, tabs = _run_pipeline(model, batches, dev, promise, tables_of=True)
    assert all(p == "stack_zf" for p in paths)
    for i in range(len(batches)):
        for a, b in zip(tabs[i], want_tabs[i]):
            assert np.array_equal(a, b), f"batch {i}: tables differ"
        assert np.array_equal(got[i], want[i]), f"batch {i}: outputs differ"


@pytest.mark.parametrize("hidden,act", [(64, "relu"), (32, "relu"), (128, "tanh"), (128, "sigmoid"), (64, "gelu")])
def test_other_widths_and_activations(dev, hidden, act):
    model = _model(hidden=hidden, act=act, seed=5)
    batches = _qm9_batches((500, 777, 256), seed=11)
    promise = max(int(np.diff(b.node_ptr).max()) for b in batches)
    want, _ = _run_plain(model, batches, dev, promise)
    got, paths, _ = _run_pipeline(model, batches, dev, promise)
    assert all(p == "stack_zf" for p in paths)
    for a, b in zip(got, want):
        assert np.array_equal(a, b)


def _random_multigraph_batch(seed, count):
    """Graphs the molecule generator never makes: odd edge counts, self loops, duplicate edges, empty graphs, isolated nodes,
    up to 64 nodes with 0 .. 150 directed edges (<= 64: the staged molecule path; more: the general path with its own fetches)."""
    from gnnbuilder_amd.batching import pack_graphs
    rng = np.random.default_rng(seed)
    graphs = []
    for i in range(count):
        n = int(rng.choice([0, 1, 2, 3, 17, 31, 32, 33, 63, 64])) if i % 3 == 0 else int(rng.integers(1, 40))
        e = 0 if n == 0 else int(rng.choice([0, 1, 2, 3, 5, 63, 64, 65, 150])) if i % 4 == 1 else int(rng.integers(0, 64))
        coo = np.stack([rng.integers(0, max(n, 1), e), rng.integers(0, max(n, 1), e)], 1).astype(np.int32).reshape(-1, 2)
        graphs.append((rng.uniform(-1, 1, (n, 11)).astype(np.float32), coo))
    return pack_graphs(graphs)


@pytest.mark.parametrize("counts", [(700, 900, 500), (5, 2600, 1)], ids=["mid", "one_wave_many_graphs"])
def test_random_multigraphs(dev, counts):
    model = _model(seed=4)
    batches = [_random_multigraph_batch(90 + i, c) for i, c in enumerate(counts)]
    want, want_tabs = _run_plain(model, batches, dev, 64, want_tables=True)
    got, paths, tabs = _run_pipeline(model, batches, dev, 64, tables_of=True)
    assert all(p == "stack_zf" for p in paths)
    for i in range(len(batches)):
        for a, b in zip(tabs[i], want_tabs[i]):
            assert np.array_equal(a, b), f"batch {i}: tables differ"
        assert np.array_equal(got[i], want[i]), f"batch {i}: outputs differ"


def test_in_kernel_form_really_ran_and_the_option_turns_it_off(dev):
    """The same pipeline with guest_prep = 0 (prep as a launch behind the forward): same bits.  That the in-kernel form is
    what ran by default is visible on the tables: with the option off AND the separate launch suppressed they stay stale --
    here simply: both settings give fresh tables for a batch the workspace has never seen."""
    model = _model(seed=8)
    batches = _qm9_batches((900, 1100, 1000), seed=20)
    promise = max(int(np.diff(b.node_ptr).max()) for b in batches)
    want, want_tabs = _run_plain(model, batches, dev, promise, want_tables=True)
    try:
        for setting in (1, 0):
            runtime.set_option("guest_prep", setting)
            got, _, tabs = _run_pipeline(model, batches, dev, promise, tables_of=True)
            for i in range(len(batches)):
                assert np.array_equal(got[i], want[i])
                for a, b in zip(tabs[i], want_tabs[i]):
                    assert np.array_equal(a, b)
    finally:
        runtime.set_option("guest_prep", 1)


@pytest.mark.parametrize("conv,layers,promise_on", [("gin", 3, True), ("gcn", 3, True), ("gcn", 2, False), ("sage", 2, True), ("pna", 2, True)])
def test_models_without_the_in_kernel_form_take_the_ordinary_launch(dev, conv, layers, promise_on):
    model = _model(conv=conv, layers=layers, hidden=64, seed=13)
    batches = _qm9_batches((200, 333, 150), seed=30)
    promise = max(int(np.diff(b.node_ptr).max()) for b in batches) if promise_on else 0
    want, _ = _run_plain(model, batches, dev, promise)
    got, _, _ = _run_pipeline(model, batches, dev, promise)
    for a, b in zip(got, want):
        assert np.array_equal(a, b)


def test_large_graphs_next_batch_keeps_the_ordinary_launch(dev):
    """A promise above 64 nodes (molhiv-sized graphs): k_gcn2_zf still runs the forward where a tile fits its stage, the next batch's
    prep is launched on its own (the in-kernel form is the 64-node molecule path only)."""
    model = _model(in_dim=9, seed=17)
    batches = [synthetic.make_batch("molhiv_tail", c, seed=40 + i) for i, c in enumerate((1200, 2000, 900))]
    promise = max(int(np.diff(b.node_ptr).max()) for b in batches)
    assert promise > 64
    want, _ = _run_plain(model, batches, dev, promise)
    got, _, _ = _run_pipeline(model, batches, dev, promise)
    for a, b in zip(got, want):
        assert np.array_equal(a, b)


def test_malformed_next_batch_is_flagged_on_its_own_workspace(dev):
    """An edge that leaves its graph in the NEXT batch: the forward that hosted the prep is untouched, the next workspace's check
    raises GNNB_ERR_GRAPH, and so does a promise broken by the next batch."""
    model = _model(seed=19)
    good, bad = _qm9_batches((600, 600), seed=50)
    promise = int(np.diff(good.node_ptr).max())
    want, _ = _run_plain(model, [good], dev, promise)
    cap = _caps([good, bad])
    a, b = _compile(model, cap, promise), _compile(model, cap, promise)
    ga = to_dev(good, dev)
    coo = bad.coo.copy()
    g = 5
    e = int(bad.edge_ptr[g])
    coo[e, 0] = int(bad.node_ptr[g + 2])  # source in another graph
    bad_coo = torch.from_numpy(coo).to(dev)
    _, _, bptr, beptr = to_dev(bad, dev)
    a.graph_prep(ga[1], ga[2], ga[3], good.num_nodes)
    out = a.forward_prepared_prep_next(ga[0], b, bad_coo, bptr, beptr, bad.num_nodes).cpu().numpy()
    assert a.last_path() == "stack_zf"
    a.check()
    assert np.array_equal(out, want[0])
    with pytest.raises(runtime.GnnbError):
        b.check()
    # a broken promise: the next batch holds a graph larger than the workspace promised
    small = _compile(model, cap, 8)
    a.graph_prep(ga[1], ga[2], ga[3], good.num_nodes)
    _, bcoo, _, _ = to_dev(bad, dev)
    out = a.forward_prepared_prep_next(ga[0], small, bcoo, bptr, beptr, bad.num_nodes).cpu().numpy()
    a.check()
    assert np.array_equal(out, want[0])
    with pytest.raises(runtime.GnnbError):
        small.check()


def test_argument_errors(dev):
    model = _model(hidden=32, seed=23)
    (batch,) = _qm9_batches((50,), seed=60)
    promise = int(np.diff(batch.node_ptr).max())
    cap = _caps([batch])
    a, b = _compile(model, cap, promise), _compile(model, (10, 100, 200), promise)
    x, coo, nptr, eptr = to_dev(batch, dev)
    a.graph_prep(coo, nptr, eptr, batch.num_nodes)
    with pytest.raises(runtime.GnnbError):  # the next batch needs its own workspace
        a.forward_prepared_prep_next(x, a, coo, nptr, eptr, batch.num_nodes)
    with pytest.raises(runtime.GnnbError):  # ... one that can hold it (nothing was enqueued: a's batch is still prepared)
        a.forward_prepared_prep_next(x, b, coo, nptr, eptr, batch.num_nodes)
    out = a.forward_prepared(x).cpu().numpy()
    want, _ = _run_plain(model, [batch], dev, promise)
    assert np.array_equal(out, want[0])
    fresh = _compile(model, cap, promise)
    with pytest.raises(runtime.GnnbError):  # no prepared batch to forward
        fresh.forward_prepared_prep_next(x, a, coo, nptr, eptr, batch.num_nodes)


def test_three_streams_of_alternating_workspaces(dev):
    """bench.py's shape: three streams, each a pipeline over its own pair of workspaces, batches dealt round robin."""
    model = _model(seed=29)
    batches = _qm9_batches((1024,) * 9, seed=70)
    promise = max(int(np.diff(b.node_ptr).max()) for b in batches)
    want, _ = _run_plain(model, batches, dev, promise)
    cap = _caps(batches)
    streams = [torch.cuda.Stream() for _ in range(3)]
    pairs = [[_compile(model, cap, promise), _compile(model, cap, promise)] for _ in range(3)]
    args = [to_dev(b, dev) for b in batches]
    outs = [None] * len(batches)
    torch.cuda.synchronize()
    for s in range(3):
        with torch.cuda.stream(streams[s]):
            pairs[s][0].graph_prep(args[s][1], args[s][2], args[s][3], batches[s].num_nodes, stream=streams[s])
    for i in range(len(batches)):
        s, k = i % 3, i // 3
        cur, nxt = pairs[s][k & 1], pairs[s][(k + 1) & 1]
        with torch.cuda.stream(streams[s]):
            if i + 3 < len(batches):
                a = args[i + 3]
                outs[i] = cur.forward_prepared_prep_next(args[i][0], nxt, a[1], a[2], a[3], batches[i + 3].num_nodes, stream=streams[s])
            else:
                outs[i] = cur.forward_prepared(args[i][0], stream=streams[s])
    torch.cuda.synchronize()
    for p in pairs:
        for cm in p:
            cm.check()
    for i in range(len(batches)):
        assert np.array_equal(outs[i].cpu().numpy(), want[i]), f"batch {i}"


def test_both_forms_of_the_readout_and_of_the_prep_give_the_same_bits(dev):
    """Options head_pairs (k_head_small's operands in pairs / four at once) and prep_group (graphs per prep wave) change register
    counts and schedules, never results: the settings for one stream of forwards against the pipeline defaults."""
    model = _model(seed=31)
    batches = _qm9_batches((2500, 300), seed=80)
    promise = max(int(np.diff(b.node_ptr).max()) for b in batches)
    want, want_tabs = _run_plain(model, batches, dev, promise, want_tables=True)
    try:
        runtime.set_option("head_pairs", 0)
        runtime.set_option("prep_group", 1)
        got, got_tabs = _run_plain(model, batches, dev, promise, want_tables=True)
    finally:
        runtime.set_option("head_pairs", 1)
        runtime.set_option("prep_group", 4)
    for i in range(len(batches)):
        assert np.array_equal(got[i], want[i])
        for a, b in zip(got_tabs[i], want_tabs[i]):
            assert np.array_equal(a, b)
