"""gnnbuilder_amd -- MI355X (gfx950) backend for the GNNBuilder workflow.

Same public surface as the reference's ``gnnbuilder`` package (``gnnbuilder/__init__.py``):
the PyTorch model API and the ``Project`` compiler driver; ``Project`` emits a thin C-ABI host
shim over hand-written HIP kernels instead of Vitis-HLS C++.

This file is the package's only ``__init__``.  The sources (Python modules, ``csrc/``,
``templates/``, the built ``libgnnb_hip.so``) live in the repository directory ``gnn-builder_amd/``,
whose name is not a Python identifier: the package's search path points there, so
``gnnbuilder_amd.runtime`` is ``gnn-builder_amd/runtime.py`` -- one module object per file.
"""
from pathlib import Path as _Path

__path__ = [str(_Path(__file__).resolve().parent.parent / "gnn-builder_amd")]

from .models import (
    MLP,
    GATConv_GNNB,
    GCNConv_GNNB,
    GIN_MLP,
    GINConv_GNNB,
    GINEConv_GNNB,
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
from .batching import GraphBatch, from_pyg_batch, order_large_last, pack_graphs, shard_batch, shard_bounds

__all__ = [
    "Project", "FPX",
    "MLP", "GATConv_GNNB", "GCNConv_GNNB", "GIN_MLP", "GINConv_GNNB", "GINEConv_GNNB", "GlobalPooling", "GNNModel",
    "PNAConv_GNNB", "SAGEConv_GNNB",
    "compute_average_degree", "compute_average_nodes_and_edges", "compute_max_nodes_and_edges",
    "compute_median_degree", "compute_median_nodes_and_edges",
    "GraphBatch", "from_pyg_batch", "order_large_last", "pack_graphs", "shard_batch", "shard_bounds",
]
