"""Host logic: packing graphs into the batched layout, slicing, node-balanced sharding."""
import numpy as np
import pytest

from gnnbuilder_amd import synthetic
from gnnbuilder_amd.batching import pack_graphs, shard_batch, shard_bounds


def test_pack_roundtrip_and_validate():
    rng = np.random.default_rng(0)
    graphs = []
    for n in (3, 1, 0, 7):
        e = rng.integers(0, max(n, 1), size=(2 * n, 2)) if n else np.zeros((0, 2), np.int64)
        graphs.append((rng.uniform(-1, 1, (n, 5)).astype(np.float32), e))
    b = pack_graphs(graphs)
    b.validate()
    assert b.num_graphs == 4 and b.num_nodes == 11 and b.num_edges == sum(g[1].shape[0] for g in graphs)
    for g, (x, e) in enumerate(graphs):
        xg, cg = b.graph(g)
        assert np.array_equal(xg, x) and np.array_equal(cg, np.asarray(e, np.int32).reshape(-1, 2))


def test_pack_accepts_edge_index_layout():
    x = np.zeros((4, 2), np.float32)
    ei = np.array([[0, 1, 2], [1, 2, 3]])  # PyG [2, E]
    want = np.array([[0, 1], [1, 2], [2, 3]], np.int32)
    assert np.array_equal(pack_graphs([(x, ei)], layout="edge_index").coo, want)
    assert np.array_equal(pack_graphs([(x, ei)], layout="auto").coo, want)
    with pytest.raises(ValueError):                       # the default is the package's own [e, 2] layout
        pack_graphs([(x, ei)])
    # [2, 2] is two (src, dst) rows OR a two-edge edge_index: never guessed
    two = np.array([[0, 1], [2, 3]])
    with pytest.raises(ValueError, match="ambiguous"):
        pack_graphs([(x, two)], layout="auto")
    assert np.array_equal(pack_graphs([(x, two)], layout="coo").coo, [[0, 1], [2, 3]])
    assert np.array_equal(pack_graphs([(x, two)], layout="edge_index").coo, [[0, 2], [1, 3]])


def test_pack_rejects_bad_input():
    x = np.zeros((2, 3), np.float32)
    with pytest.raises(ValueError):
        pack_graphs([(x, np.array([[0, 5]]))])
    with pytest.raises(ValueError):
        pack_graphs([(x, np.zeros((0, 2))), (np.zeros((2, 4), np.float32), np.zeros((0, 2)))])
    with pytest.raises(ValueError):
        pack_graphs([])


def test_validate_catches_cross_graph_edge():
    b = synthetic.make_batch("qm9", 5, 0)
    b.coo[0, 0] = b.num_nodes - 1
    with pytest.raises(ValueError):
        b.validate()


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_shards_partition_the_batch_and_balance_nodes(world):
    b = synthetic.make_batch("molhiv", 257, 3)
    bounds = shard_bounds(b.node_ptr, world)
    assert bounds[0][0] == 0 and bounds[-1][1] == b.num_graphs
    assert all(bounds[i][1] == bounds[i + 1][0] for i in range(world - 1))
    nodes = [int(b.node_ptr[g1] - b.node_ptr[g0]) for g0, g1 in bounds]
    assert sum(nodes) == b.num_nodes
    assert max(nodes) - min(nodes) <= 2 * 222  # within a graph or two of each other
    parts = [shard_batch(b, world, r) for r in range(world)]
    for p in parts:
        p.validate()
    assert np.array_equal(np.concatenate([p.x for p in parts]), b.x)
    assert sum(p.num_edges for p in parts) == b.num_edges


def test_synthetic_shapes_follow_the_survey_recipe():
    b = synthetic.make_batch("qm9", 2000, 0)
    b.validate()
    sizes = np.diff(b.node_ptr)
    assert 3 <= sizes.min() and sizes.max() <= 29 and abs(sizes.mean() - 18) < 0.5
    assert abs(b.num_edges / b.num_nodes - 2.15) < 0.15
    assert b.x.shape[1] == 11 and b.x.dtype == np.float32 and np.abs(b.x).max() <= 1.0
    # both directions stored, no self loops
    assert not np.any(b.coo[:, 0] == b.coo[:, 1])
    fwd = set(map(tuple, b.coo.tolist()))
    assert all((t, s) in fwd for s, t in list(fwd)[:500])
    assert np.array_equal(synthetic.make_batch("qm9", 50, 7).coo, synthetic.make_batch("qm9", 50, 7).coo)


def test_from_pyg_batch_groups_edges_stably_and_matches_pack_graphs():
    """A PyG ``Batch`` (x, edge_index with global ids, batch vector or ptr) becomes the same GraphBatch
    that packing the graphs one by one gives -- also when the edges arrive shuffled across graphs
    (the stable grouping keeps each graph's own edge order, i.e. the aggregation's sum order)."""
    from gnnbuilder_amd.batching import from_pyg_batch

    ref = synthetic.make_batch("esol", 40, seed=4)
    batch_vec = np.repeat(np.arange(ref.num_graphs), np.diff(ref.node_ptr))
    got = from_pyg_batch(ref.x, ref.coo.T, batch=batch_vec)
    for a, b in ((got.x, ref.x), (got.coo, ref.coo), (got.node_ptr, ref.node_ptr), (got.edge_ptr, ref.edge_ptr)):
        assert np.array_equal(a, b)
    got.validate()
    # interleave the edges of different graphs, keeping every graph's internal order
    rng = np.random.default_rng(0)
    keys = np.repeat(np.arange(ref.num_graphs), np.diff(ref.edge_ptr)) + rng.uniform(0, 5, ref.num_edges)
    perm = np.argsort(keys, kind="stable")
    within = np.repeat(np.arange(ref.num_graphs), np.diff(ref.edge_ptr))[perm]
    assert np.any(np.diff(within) < 0)  # really interleaved
    again = from_pyg_batch(ref.x, ref.coo[perm].T, ptr=ref.node_ptr)
    assert np.array_equal(again.edge_ptr, ref.edge_ptr)
    for g in range(ref.num_graphs):  # same edges per graph, each graph's relative order preserved
        assert np.array_equal(again.graph(g)[1], ref.coo[perm][within == g] - ref.node_ptr[g])
    # trailing empty graphs need num_graphs; ptr carries them by itself
    tail = from_pyg_batch(ref.x, ref.coo.T, batch=batch_vec, num_graphs=ref.num_graphs + 2)
    assert tail.num_graphs == ref.num_graphs + 2 and tail.node_ptr[-1] == ref.num_nodes
    with pytest.raises(ValueError):
        bad = ref.coo.copy()
        bad[0, 0] = ref.num_nodes - 1  # joins graph 0 and the last graph
        from_pyg_batch(ref.x, bad.T, batch=batch_vec)
    with pytest.raises(ValueError):
        from_pyg_batch(ref.x, ref.coo.T, batch=batch_vec[::-1])


def test_order_large_last_is_a_stable_partition():
    """Graphs beyond the limit move to the end (order kept inside both groups), edges follow their graphs with renumbered
    endpoints, and the returned triple names the first large graph / node / edge."""
    from gnnbuilder_amd import synthetic
    from gnnbuilder_amd.batching import order_large_last

    b = synthetic.make_batch("molhiv_tail", 300, seed=4)
    sizes = np.diff(b.node_ptr)
    o, perm, (g0, n0, e0) = order_large_last(b, 40)
    o.validate()
    assert sorted(perm.tolist()) == list(range(b.num_graphs))
    assert g0 == int((sizes <= 40).sum()) and n0 == int(o.node_ptr[g0]) and e0 == int(o.edge_ptr[g0])
    assert np.all(np.diff(perm[:g0]) > 0) and np.all(np.diff(perm[g0:]) > 0)          # stable
    assert np.all(np.diff(o.node_ptr)[:g0] <= 40) and np.all(np.diff(o.node_ptr)[g0:] > 40)
    for i in range(o.num_graphs):
        xa, ca = o.graph(i)
        xb, cb = b.graph(int(perm[i]))
        assert np.array_equal(xa, xb) and np.array_equal(ca, cb)
    same, perm2, seg = order_large_last(b, 10 ** 6)                                   # nothing large: the batch itself
    assert same is b and seg == (b.num_graphs, b.num_nodes, b.num_edges) and perm2.tolist() == list(range(b.num_graphs))


def test_run_cuts_in_32_bits_equal_the_64_bit_quotients():
    """csrc/gnnb_device.h run_cuts(): every persistent kernel's workgroup b of G takes units [floor(b T / G), floor((b + 1) T / G))
    -- computed as b q + floor(b r / G) with T = q G + r so that no 64-bit division is needed (round 6).  The identity, and
    that its intermediates fit 32 bits, over the launch shapes the kernels see and the corners."""
    import numpy as np
    rng = np.random.default_rng(0)
    cases = [(1, 1), (256, 0), (256, 1), (256, 255), (256, 256), (512, 9216), (256, 26112), (512, 31744), (4096, 1 << 21), (65535, (1 << 22) - 1)]
    cases += [(int(rng.integers(1, 4097)), int(rng.integers(0, 1 << 22))) for _ in range(2000)]
    for G, T in cases:
        b = np.unique(np.concatenate([np.arange(min(G, 4)), [G // 2, G - 1], rng.integers(0, G, 8)])).astype(np.uint64)
        q, r = T // G, T % G
        for bb in (b, b + 1):
            assert int((bb * np.uint64(r)).max()) < 2 ** 32 and int((bb * np.uint64(q)).max() + (bb * np.uint64(r) // np.uint64(G)).max()) < 2 ** 32
            assert np.array_equal(bb * np.uint64(q) + (bb * np.uint64(r)) // np.uint64(G), (bb * np.uint64(T)) // np.uint64(G))
