"""Packing many small graphs into one batched COO/ptr layout, and sharding batches across GPUs.

The reference processes one graph per ``<name>_top`` call (model_tb.cpp.jinja:189-201).  The
MI355X path packs thousands of independent graphs so one kernel launch covers them all:

    x        [N_tot, F]   fp32, graphs concatenated
    coo      [E_tot, 2]   int32 (src, dst) with batch-global node ids, edges grouped by graph
    node_ptr [B+1]        int32
    edge_ptr [B+1]        int32

This is host logic (numpy): it runs on CPU and is covered by the ``-m "not gpu"`` tests.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Sequence, Tuple

import numpy as np


@dataclass
class GraphBatch:
    x: np.ndarray
    coo: np.ndarray
    node_ptr: np.ndarray
    edge_ptr: np.ndarray

    @property
    def num_graphs(self) -> int:
        return int(self.node_ptr.shape[0]) - 1

    @property
    def num_nodes(self) -> int:
        return int(self.node_ptr[-1])

    @property
    def num_edges(self) -> int:
        return int(self.edge_ptr[-1])

    def graph(self, g: int) -> Tuple[np.ndarray, np.ndarray]:
        """(x, coo with graph-local ids) of graph g -- what ``<name>_top`` takes."""
        n0, n1 = int(self.node_ptr[g]), int(self.node_ptr[g + 1])
        e0, e1 = int(self.edge_ptr[g]), int(self.edge_ptr[g + 1])
        return self.x[n0:n1], self.coo[e0:e1] - n0

    def slice(self, g0: int, g1: int) -> "GraphBatch":
        n0, n1 = int(self.node_ptr[g0]), int(self.node_ptr[g1])
        e0, e1 = int(self.edge_ptr[g0]), int(self.edge_ptr[g1])
        return GraphBatch(
            x=np.ascontiguousarray(self.x[n0:n1]),
            coo=np.ascontiguousarray(self.coo[e0:e1] - n0),
            node_ptr=(self.node_ptr[g0:g1 + 1] - n0).astype(np.int32),
            edge_ptr=(self.edge_ptr[g0:g1 + 1] - e0).astype(np.int32),
        )

    def validate(self) -> None:
        B = self.num_graphs
        if self.node_ptr[0] != 0 or self.edge_ptr[0] != 0:
            raise ValueError("ptr arrays must start at 0")
        if np.any(np.diff(self.node_ptr) < 0) or np.any(np.diff(self.edge_ptr) < 0):
            raise ValueError("ptr arrays must be non-decreasing")
        if self.x.shape[0] != self.num_nodes or self.coo.shape[0] != self.num_edges:
            raise ValueError("x / coo sizes disagree with the ptr arrays")
        if self.num_edges:
            g_of_edge = np.repeat(np.arange(B), np.diff(self.edge_ptr))
            lo, hi = self.node_ptr[g_of_edge], self.node_ptr[g_of_edge + 1]
            if np.any(self.coo < lo[:, None]) or np.any(self.coo >= hi[:, None]):
                raise ValueError("an edge leaves its graph")


def pack_graphs(graphs: Sequence[Tuple[np.ndarray, np.ndarray]], layout: str = "coo") -> GraphBatch:
    """``graphs``: sequence of (x [n, F], edges); ids are graph-local.  ``layout``: ``"coo"`` (default) = edges are
    ``[e, 2]`` (src, dst) rows -- the tb_data ``*_coo.bin`` layout, reference code_gen.py:262, and what
    ``GraphBatch.graph`` returns; ``"edge_index"`` = PyG-style ``[2, e]``; ``"auto"`` accepts either and refuses the
    one ambiguous shape, ``[2, 2]``."""
    if layout not in ("auto", "coo", "edge_index"):
        raise ValueError("layout must be 'auto', 'coo' or 'edge_index'")
    xs, coos, nptr, eptr = [], [], [0], [0]
    feat = None
    for x, ei in graphs:
        x = np.asarray(x, dtype=np.float32)
        if x.ndim != 2:
            raise ValueError("node features must be [n, F]")
        feat = x.shape[1] if feat is None else feat
        if x.shape[1] != feat:
            raise ValueError("all graphs must share the feature width")
        ei = np.asarray(ei)
        if ei.size == 0:
            ei = np.zeros((0, 2), dtype=np.int32)
        elif layout == "edge_index":
            if ei.ndim != 2 or ei.shape[0] != 2:
                raise ValueError("edge_index must be [2, e]")
            ei = ei.T
        elif layout == "coo":
            if ei.ndim != 2 or ei.shape[1] != 2:
                raise ValueError("coo edges must be [e, 2]")
        elif ei.ndim == 2 and ei.shape == (2, 2):
            raise ValueError("a [2, 2] edge array is ambiguous: pass layout='coo' (two (src, dst) rows) or "
                             "layout='edge_index' (PyG [2, e])")
        elif ei.ndim == 2 and ei.shape[0] == 2 and ei.shape[1] != 2:
            ei = ei.T
        elif ei.ndim != 2 or ei.shape[1] != 2:
            raise ValueError("edges must be [e, 2] or [2, e]")
        ei = ei.astype(np.int32)
        if ei.size and (ei.min() < 0 or ei.max() >= x.shape[0]):
            raise ValueError("edge endpoint outside its graph")
        xs.append(x)
        coos.append(ei + nptr[-1])
        nptr.append(nptr[-1] + x.shape[0])
        eptr.append(eptr[-1] + ei.shape[0])
    if feat is None:
        raise ValueError("empty batch")
    return GraphBatch(
        x=np.ascontiguousarray(np.concatenate(xs, axis=0)) if xs else np.zeros((0, feat), np.float32),
        coo=np.ascontiguousarray(np.concatenate(coos, axis=0)).reshape(-1, 2).astype(np.int32),
        node_ptr=np.asarray(nptr, dtype=np.int32),
        edge_ptr=np.asarray(eptr, dtype=np.int32),
    )


def from_pyg_batch(x, edge_index, batch=None, ptr=None, num_graphs=None) -> GraphBatch:
    """Adapter for a PyG-style mini-batch (``torch_geometric.data.Batch`` attributes, passed as arrays
    or tensors so that PyG itself is not needed): ``x`` [N, F], ``edge_index`` [2, E] with batch-global
    node ids, and either ``batch`` [N] (graph id of every node, non-decreasing -- what ``Batch.batch``
    holds) or ``ptr`` [B+1] (``Batch.ptr``).  Edges may come in any order; they are grouped by graph
    with a STABLE sort, so the per-destination neighbour order -- hence the floating-point sum order
    of the aggregation -- is the one the reference's per-graph ``edge_index`` would give
    (reference code_gen.py:262 writes ``edge_index.T`` per graph)."""
    x = np.ascontiguousarray(np.asarray(x, dtype=np.float32))
    ei = np.asarray(edge_index)
    if ei.ndim != 2 or ei.shape[0] != 2:
        raise ValueError("edge_index must be [2, E]")
    ei = ei.astype(np.int64)
    N = x.shape[0]
    if ptr is not None:
        node_ptr = np.asarray(ptr, dtype=np.int64)
        if node_ptr.ndim != 1 or node_ptr.size < 1 or node_ptr[0] != 0 or node_ptr[-1] != N or np.any(np.diff(node_ptr) < 0):
            raise ValueError("ptr must be non-decreasing, start at 0 and end at the number of nodes")
    else:
        if batch is None:
            batch = np.zeros(N, dtype=np.int64)
        batch = np.asarray(batch, dtype=np.int64)
        if batch.shape != (N,):
            raise ValueError("batch must hold one graph id per node")
        if N and (np.any(np.diff(batch) < 0) or batch[0] < 0):
            raise ValueError("batch must be non-decreasing (nodes grouped by graph)")
        B = int(num_graphs) if num_graphs is not None else (int(batch[-1]) + 1 if N else 0)
        if N and int(batch[-1]) >= B:
            raise ValueError("num_graphs is smaller than the largest graph id")
        node_ptr = np.zeros(B + 1, dtype=np.int64)
        np.add.at(node_ptr, batch + 1, 1)
        np.cumsum(node_ptr, out=node_ptr)
    B = node_ptr.size - 1
    if ei.size and (ei.min() < 0 or ei.max() >= N):
        raise ValueError("edge endpoint outside the batch")
    g_src = np.searchsorted(node_ptr, ei[0], side="right") - 1
    g_dst = np.searchsorted(node_ptr, ei[1], side="right") - 1
    if np.any(g_src != g_dst):
        raise ValueError("an edge joins two different graphs")
    order = np.argsort(g_dst, kind="stable")
    coo = np.ascontiguousarray(ei[:, order].T).astype(np.int32)
    edge_ptr = np.zeros(B + 1, dtype=np.int64)
    np.add.at(edge_ptr, g_dst + 1, 1)
    np.cumsum(edge_ptr, out=edge_ptr)
    return GraphBatch(x=x, coo=coo.reshape(-1, 2), node_ptr=node_ptr.astype(np.int32), edge_ptr=edge_ptr.astype(np.int32))


def order_large_last(batch: GraphBatch, max_graph_nodes: int):
    """Reorder a batch so that the graphs with more than ``max_graph_nodes`` nodes come last (stable inside both
    groups): the layout ``gnnb_workspace_set_large_segment`` asks for.  Returns ``(ordered batch, perm, (first_graph,
    first_node, first_edge))`` with ``ordered.graph(i) == batch.graph(perm[i])``; outputs of the ordered batch go back to the
    caller's order with ``out[np.argsort(perm)]``.  ``first_graph == ordered.num_graphs`` when no graph is large."""
    sizes = np.diff(batch.node_ptr)
    large = sizes > int(max_graph_nodes)
    perm = np.concatenate([np.flatnonzero(~large), np.flatnonzero(large)]).astype(np.int64)
    first_graph = int((~large).sum())
    if not large.any():
        return batch, perm, (batch.num_graphs, batch.num_nodes, batch.num_edges)
    nptr, eptr = batch.node_ptr.astype(np.int64), batch.edge_ptr.astype(np.int64)
    nsz, esz = sizes[perm].astype(np.int64), np.diff(eptr)[perm]
    new_nptr = np.concatenate([[0], np.cumsum(nsz)])
    new_eptr = np.concatenate([[0], np.cumsum(esz)])
    # rows / edges of graph perm[i] move to [new_nptr[i], new_nptr[i+1]) / [new_eptr[i], new_eptr[i+1])
    node_src = np.concatenate([np.arange(nptr[g], nptr[g + 1]) for g in perm]) if batch.num_nodes else np.zeros(0, np.int64)
    edge_src = np.concatenate([np.arange(eptr[g], eptr[g + 1]) for g in perm]) if batch.num_edges else np.zeros(0, np.int64)
    shift = np.repeat(new_nptr[:-1] - nptr[perm], esz)          # per moved edge: new minus old node offset of its graph
    coo = (batch.coo[edge_src].astype(np.int64) + shift[:, None]).astype(np.int32) if batch.num_edges else batch.coo.copy()
    ordered = GraphBatch(x=np.ascontiguousarray(batch.x[node_src]), coo=np.ascontiguousarray(coo).reshape(-1, 2),
                         node_ptr=new_nptr.astype(np.int32), edge_ptr=new_eptr.astype(np.int32))
    return ordered, perm, (first_graph, int(new_nptr[first_graph]), int(new_eptr[first_graph]))


def shard_bounds(node_ptr: np.ndarray, world_size: int) -> List[Tuple[int, int]]:
    """Contiguous graph ranges per rank, cut on the cumulative NODE count so every GPU gets
    ~N_tot/world_size nodes; graphs are never split."""
    B = int(node_ptr.shape[0]) - 1
    total = int(node_ptr[-1])
    cuts = [0]
    for r in range(1, world_size):
        target = total * r / world_size
        g = int(np.searchsorted(node_ptr, target, side="left"))
        g = min(max(g, cuts[-1]), B)
        cuts.append(g)
    cuts.append(B)
    return [(cuts[r], cuts[r + 1]) for r in range(world_size)]


def shard_batch(batch: GraphBatch, world_size: int, rank: int) -> GraphBatch:
    g0, g1 = shard_bounds(batch.node_ptr, world_size)[rank]
    return batch.slice(g0, g1)
