"""gnnbuilder_amd -- MI355X (gfx950) backend for the GNNBuilder workflow.

Same public surface as the reference's ``gnnbuilder`` package (``gnnbuilder/__init__.py``):
the PyTorch model API and the ``Project`` compiler driver; ``Project`` emits a thin C-ABI host
shim over hand-written HIP kernels instead of Vitis-HLS C++.  The directory is named
``gnn-builder_amd``; import it as ``gnnbuilder_amd`` (alias package at the repo root).
"""
from .models import (
    MLP,
    GATConv_GNNB,
    GCNConv_GNNB,
    GIN_MLP,
    GINConv_GNNB,
    GlobalPooling,
    GNNModel,
    PNAConv_GNNB,
    SAGEConv_GNNB,
)
from .utils import (
    compute_average_degree,
    compute_average_nodes_and_edges,
    compute_max_nodes_and_edges,
    compute_median_degree,
    compute_median_nodes_and_edges,
)
from .code_gen import FPX, Project
from .batching import GraphBatch, from_pyg_batch, pack_graphs, shard_batch, shard_bounds

__all__ = [
    "Project", "FPX",
    "MLP", "GATConv_GNNB", "GCNConv_GNNB", "GIN_MLP", "GINConv_GNNB", "GlobalPooling", "GNNModel",
    "PNAConv_GNNB", "SAGEConv_GNNB",
    "compute_average_degree", "compute_average_nodes_and_edges", "compute_max_nodes_and_edges",
    "compute_median_degree", "compute_median_nodes_and_edges",
    "GraphBatch", "from_pyg_batch", "pack_graphs", "shard_batch", "shard_bounds",
]
