"""Model API of the MI355X backend -- counterpart of the reference's ``gnnbuilder/models.py``.

Same classes, constructor arguments, attribute names and ``state_dict`` parameter names as the
reference, so a user script (or a checkpoint) written against ``gnnbuilder.models`` keeps working.
The reference builds its conv layers on ``torch_geometric``; this package has no such dependency:
each conv is written here directly in PyTorch tensor ops (``index_add_`` / ``scatter_reduce``)
with the module layout PyG uses, so parameter names match
(``conv.lin.weight``, ``mlp.linear_0.weight``, ``conv.lin_l.weight``, ``conv.pre_nns.0.0.weight`` ...).

Role split (same as the reference): ``GNNModel.forward`` is the *model definition* in PyTorch --
what a user trains, and what ``Project.gen_testbench_data`` records as the golden output
(reference ``code_gen.py:279-285``).  The accelerated path is ``Project`` / ``runtime.CompiledModel``
(hand-written HIP behind the C ABI); it never dispatches back to this file.

Semantics follow SURVEY.md Appendix A, each checked against the reference's PyG golden vectors
in ``tests/test_models_torch.py``.
"""
from __future__ import annotations

import math
from typing import Callable, List, Optional

import torch
import torch.nn as nn
from torch import Tensor

from .utils import layer_param_name_combiner

TorchModuleArg = Callable[..., torch.nn.Module]
TorchModuleArgOptional = Optional[Callable[..., torch.nn.Module]]


def _in_degree(edge_index: Tensor, num_nodes: int) -> Tensor:
    deg = torch.zeros(num_nodes, dtype=torch.long, device=edge_index.device)
    if edge_index.numel():
        deg.index_add_(0, edge_index[1], torch.ones_like(edge_index[1]))
    return deg


def _glorot(w: Tensor) -> None:
    a = math.sqrt(6.0 / (w.size(-2) + w.size(-1)))
    with torch.no_grad():
        w.uniform_(-a, a)


# --------------------------------------------------------------------------------------- GCN
class _GCNConv(nn.Module):
    """``D^-1/2 (A+I) D^-1/2 X W^T + b`` with ``d_i = 1 + indeg(i)`` (reference models.py:41;
    native restatement gnn_builder_lib.h:1213-1387).  Parameters: ``bias``, ``lin.weight``."""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.lin = nn.Linear(in_channels, out_channels, bias=False)
        self.bias = nn.Parameter(torch.zeros(out_channels))
        _glorot(self.lin.weight)

    def forward(self, x: Tensor, edge_index: Tensor) -> Tensor:
        n = x.size(0)
        # PyG gcn_norm -> add_remaining_self_loops: the input's explicit self loops are REPLACED by exactly one
        # self loop per node, so an edge (v, v) neither adds a message nor raises the degree
        if edge_index.numel():
            edge_index = edge_index[:, edge_index[0] != edge_index[1]]
        src, dst = edge_index[0], edge_index[1]
        dinv = (_in_degree(edge_index, n).to(x.dtype) + 1.0).pow(-0.5)
        h = self.lin(x)
        out = h * (dinv * dinv).unsqueeze(-1)
        if src.numel():
            out = out.index_add(0, dst, h[src] * (dinv[src] * dinv[dst]).unsqueeze(-1))
        return out + self.bias


class GCNConv_GNNB(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, p_in: int = 1, p_out: int = 1):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.p_in = p_in
        self.p_out = p_out
        self.conv = _GCNConv(in_channels, out_channels)

    def forward(self, x: Tensor, edge_index: Tensor) -> Tensor:
        return self.conv(x, edge_index)


# --------------------------------------------------------------------------------------- GIN
class GIN_MLP(nn.Module):
    """Linear -> ReLU -> Linear (reference models.py:47-67)."""

    def __init__(self, in_dim: int, out_dim: int, hidden_dim: Optional[int] = None):
        super().__init__()
        self.in_dim = in_dim
        self.out_dim = out_dim
        self.hidden_dim = out_dim if hidden_dim is None else hidden_dim
        self.linear_0 = nn.Linear(self.in_dim, self.hidden_dim)
        self.linear_1 = nn.Linear(self.hidden_dim, self.out_dim)
        self.relu = nn.ReLU()
        self.in_features = self.in_dim

    def forward(self, x: Tensor) -> Tensor:
        return self.linear_1(self.relu(self.linear_0(x)))


class _GINConv(nn.Module):
    """``nn((1+eps) x_i + sum_j x_j)`` with a fixed eps (reference models.py:91)."""

    def __init__(self, mlp: nn.Module, eps: float = 0.0):
        super().__init__()
        self.nn = mlp
        self.register_buffer("eps", torch.tensor([float(eps)]))

    def forward(self, x: Tensor, edge_index: Tensor) -> Tensor:
        src, dst = edge_index[0], edge_index[1]
        agg = torch.zeros_like(x)
        if src.numel():
            agg = agg.index_add(0, dst, x[src])
        return self.nn(agg + (1.0 + self.eps) * x)


class GINConv_GNNB(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, hidden_dim: Optional[int] = None,
                 eps: float = 0.0, p_in: int = 1, p_out: int = 1):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        # The reference stores the argument but always builds the MLP with hidden = out_channels
        # (models.py:84,90; SURVEY finding 6).  The emitter here reads ``mlp.hidden_dim``.
        self.hidden_dim = hidden_dim
        self.eps = eps
        self.p_in = p_in
        self.p_out = p_out
        self.mlp = GIN_MLP(in_channels, out_channels, None)
        self.conv = _GINConv(self.mlp, eps=eps)

    def forward(self, x: Tensor, edge_index: Tensor) -> Tensor:
        return self.conv(x, edge_index)


# --------------------------------------------------------------------------------------- SAGE
class _SAGEConv(nn.Module):
    """``W_l mean_j x_j + b_l + W_r x_i`` (reference models.py:259; lib:2161-2341)."""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.lin_l = nn.Linear(in_channels, out_channels, bias=True)
        self.lin_r = nn.Linear(in_channels, out_channels, bias=False)

    def forward(self, x: Tensor, edge_index: Tensor) -> Tensor:
        n = x.size(0)
        src, dst = edge_index[0], edge_index[1]
        agg = torch.zeros_like(x)
        if src.numel():
            agg = agg.index_add(0, dst, x[src])
        deg = _in_degree(edge_index, n).clamp(min=1).to(x.dtype)
        return self.lin_l(agg / deg.unsqueeze(-1)) + self.lin_r(x)


class SAGEConv_GNNB(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, p_in: int = 1, p_out: int = 1):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.p_in = p_in
        self.p_out = p_out
        self.conv = _SAGEConv(in_channels, out_channels)

    def forward(self, x: Tensor, edge_index: Tensor) -> Tensor:
        return self.conv(x, edge_index)


# --------------------------------------------------------------------------------------- PNA
class _DegreeScalerAggregation(nn.Module):
    def __init__(self, avg_deg_log: float):
        super().__init__()
        self.avg_deg_log = torch.Tensor([avg_deg_log])


class _PNAConv(nn.Module):
    """PNA with aggregators [max, min, mean, std] x scalers [identity, amplification,
    attenuation], towers=1, one pre and one post layer (reference models.py:227-234).
    ``std`` is PyG's: ``sqrt(clamp(E[h^2]-E[h]^2, 1e-5))`` zeroed where ``<= sqrt(1e-5)``
    (SURVEY finding 5, pinned by ``tb_pna_output.bin``)."""

    def __init__(self, in_channels: int, out_channels: int, avg_deg_log: float):
        super().__init__()
        self.aggr_module = _DegreeScalerAggregation(avg_deg_log)
        self.pre_nns = nn.ModuleList([nn.Sequential(nn.Linear(2 * in_channels, in_channels))])
        self.post_nns = nn.ModuleList([nn.Sequential(nn.Linear(13 * in_channels, out_channels))])
        self.lin = nn.Linear(out_channels, out_channels)

    def forward(self, x: Tensor, edge_index: Tensor) -> Tensor:
        n, f = x.size(0), x.size(1)
        src, dst = edge_index[0], edge_index[1]
        h = self.pre_nns[0](torch.cat([x[dst], x[src]], dim=-1))  # destination first
        idx = dst.unsqueeze(-1).expand(-1, f)
        zeros = x.new_zeros(n, f)
        deg = _in_degree(edge_index, n)
        cnt = deg.clamp(min=1).to(x.dtype).unsqueeze(-1)
        if src.numel():
            mx = zeros.scatter_reduce(0, idx, h, "amax", include_self=False)
            mn = zeros.scatter_reduce(0, idx, h, "amin", include_self=False)
            s1 = zeros.index_add(0, dst, h)
            s2 = zeros.index_add(0, dst, h * h)
        else:
            mx = mn = s1 = s2 = zeros
        mean = s1 / cnt
        var = s2 / cnt - mean * mean
        std = var.clamp(min=1e-5).sqrt()
        std = std.masked_fill(std <= math.sqrt(1e-5), 0.0)
        has = (deg > 0).unsqueeze(-1)
        std = torch.where(has, std, zeros)
        agg = torch.cat([mx, mn, mean, std], dim=-1)
        delta = self.aggr_module.avg_deg_log.to(x.device, x.dtype)
        logd = torch.log(deg.clamp(min=1).to(x.dtype) + 1.0).unsqueeze(-1)
        out = torch.cat([x, agg, agg * (logd / delta), agg * (delta / logd)], dim=-1)
        return self.lin(self.post_nns[0](out))


class PNAConv_GNNB(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, delta: float = 1.0, p_in: int = 1,
                 p_out: int = 1):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.delta = delta
        self.p_in = p_in
        self.p_out = p_out
        self.aggregators = ["max", "min", "mean", "std"]
        self.scalers = ["identity", "amplification", "attenuation"]
        # `delta` is used verbatim as avg_deg_log (reference models.py:236-237)
        self.conv = _PNAConv(in_channels, out_channels, float(delta))
        self.delta_scaler = self.conv.aggr_module.avg_deg_log.item()

    def forward(self, x: Tensor, edge_index: Tensor) -> Tensor:
        return self.conv(x, edge_index)


# --------------------------------------------------------------------------------------- GINE
class _GINEConv(nn.Module):
    """``nn((1+eps) x_i + sum_j relu(x_j + lin(e_ij)))`` (PyG GINEConv with edge_dim; reference models.py:97-123,
    native gine_conv gnn_builder_lib.h:1555-1742).  Parameters: ``nn.*``, ``lin.weight`` [in, edge_dim], ``lin.bias``."""

    def __init__(self, mlp: nn.Module, in_channels: int, edge_dim: int, eps: float = 0.0):
        super().__init__()
        self.nn = mlp
        self.lin = nn.Linear(edge_dim, in_channels)
        self.register_buffer("eps", torch.tensor([float(eps)]))

    def forward(self, x: Tensor, edge_index: Tensor, edge_attr: Tensor) -> Tensor:
        src, dst = edge_index[0], edge_index[1]
        agg = torch.zeros_like(x)
        if src.numel():
            agg = agg.index_add(0, dst, torch.relu(x[src] + self.lin(edge_attr)))
        return self.nn(agg + (1.0 + self.eps) * x)


class GINEConv_GNNB(nn.Module):
    """Reference models.py:97-123.  Not in SUPPORTED_GNN_CONVS there (the emitter has a TODO for it,
    templates/model.cpp.jinja:143-144), so ``GNNModel`` does not stack it; the native layer is reachable through
    ``runtime.CompiledModel.gine_conv`` (C ABI ``gnnb_aggregate_edges`` + ``gnnb_linear``)."""

    def __init__(self, in_channels: int, out_channels: int, edge_dim: int, hidden_dim: Optional[int] = None,
                 eps: float = 0.0, p_in: int = 1, p_out: int = 1):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.edge_dim = edge_dim
        self.hidden_dim = hidden_dim
        self.eps = eps
        self.p_in = p_in
        self.p_out = p_out
        self.mlp = GIN_MLP(in_channels, out_channels, hidden_dim)
        self.conv = _GINEConv(self.mlp, in_channels, edge_dim, eps=eps)

    def forward(self, x: Tensor, edge_index: Tensor, edge_attr: Tensor) -> Tensor:
        return self.conv(x, edge_index, edge_attr)


class GATConv_GNNB(nn.Module):
    """Listed by the reference's SUPPORTED_GNN_CONVS but it has no native kernel for it
    (gnn_builder_lib.h:2343 ``// TODO: GAT layer``; template TODO model.cpp.jinja:141-142);
    out of scope here as well."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        raise NotImplementedError(
            "GATConv_GNNB has no native path in the reference (gnn_builder_lib.h:2343) and none here")


# --------------------------------------------------------------------------------------- pooling / MLP
SUPPORTED_GLOBAL_POOLING_AGGRS = {"add": "SumAggregation", "max": "MaxAggregation", "mean": "MeanAggregation"}
SUPPORTED_GLOBAL_POOLING_MODE = ["cat"]


class GlobalPooling(nn.Module):
    """Concatenation of add / mean / max readouts in the order of ``aggrs``
    (reference models.py:326-359)."""

    def __init__(self, aggrs: List[str], mode: str = "cat"):
        super().__init__()
        self.aggrs = aggrs
        self.mode = mode
        if aggrs == []:
            raise ValueError("GlobalPooling needs at least one aggregation (add / mean / max)")
        for a in self.aggrs:
            if a not in SUPPORTED_GLOBAL_POOLING_AGGRS:
                raise NotImplementedError(
                    f"unknown pooling aggregation {a!r}; choose from {SUPPORTED_GLOBAL_POOLING_AGGRS}")
        if self.mode not in SUPPORTED_GLOBAL_POOLING_MODE:
            raise NotImplementedError(
                f"unknown pooling mode {self.mode!r}; choose from {SUPPORTED_GLOBAL_POOLING_MODE}")

    def forward(self, x: Tensor, index: Optional[Tensor] = None, dim_size: Optional[int] = None) -> Tensor:
        if index is None:  # one graph: [N, d] -> [1, k*d]
            index = torch.zeros(x.size(0), dtype=torch.long, device=x.device)
            dim_size = 1
        if dim_size is None:
            dim_size = int(index.max().item()) + 1 if index.numel() else 0
        d = x.size(1)
        idx = index.unsqueeze(-1).expand(-1, d)
        zeros = x.new_zeros(dim_size, d)
        outs = []
        for a in self.aggrs:
            if a == "add":
                outs.append(zeros.index_add(0, index, x))
            elif a == "mean":
                cnt = torch.zeros(dim_size, dtype=x.dtype, device=x.device).index_add(
                    0, index, torch.ones_like(index, dtype=x.dtype)).clamp(min=1)
                outs.append(zeros.index_add(0, index, x) / cnt.unsqueeze(-1))
            else:
                outs.append(zeros.scatter_reduce(0, idx, x, "amax", include_self=False))
        return torch.cat(outs, dim=-1)

    @property
    def num_of_aggrs(self) -> int:
        return len(self.aggrs)

    def __repr__(self) -> str:
        return f"{self.__class__.__name__}({self.aggrs}, mode={self.mode})"


SUPPORTED_ACTIVATIONS = [nn.ReLU, nn.GELU, nn.Sigmoid, nn.Tanh]
_ACT_NAME = {nn.ReLU: "relu", nn.GELU: "gelu", nn.Sigmoid: "sigmoid", nn.Tanh: "tanh"}


class MLP(nn.Module):
    """Linear -> act, repeated ``hidden_layers`` times, then Linear (reference models.py:365-450)."""

    def __init__(self, in_dim: int, out_dim: int, hidden_dim: int = 64, hidden_layers: int = 2,
                 activation: TorchModuleArg = nn.ReLU, norm_layer: TorchModuleArgOptional = None,
                 p_in: int = 1, p_hidden: int = 1, p_out: int = 1):
        super().__init__()
        self.in_dim = in_dim
        self.out_dim = out_dim
        self.hidden_dim = hidden_dim
        self.hidden_layers = hidden_layers
        self.activation = activation
        self.norm_layer = norm_layer
        if self.activation not in SUPPORTED_ACTIVATIONS:
            raise ValueError(f"MLP activation must be one of {SUPPORTED_ACTIVATIONS}, got {activation}")
        if self.norm_layer is not None:
            raise NotImplementedError("MLP norm_layer: no normalisation layer has a native path (the reference has none either)")
        if hidden_layers < 0:
            raise ValueError(f"MLP hidden_layers cannot be negative (got {hidden_layers})")
        self.p_in = p_in
        self.p_hidden = p_hidden
        self.p_out = p_out

        self.linear_layers = nn.ModuleList()
        self.activations = nn.ModuleList()
        self.norm_layers = nn.ModuleList()
        dims = [in_dim] + [hidden_dim] * hidden_layers + [out_dim]
        for i in range(len(dims) - 1):
            self.linear_layers.append(nn.Linear(dims[i], dims[i + 1]))
            if i < len(dims) - 2:
                self.activations.append(self.activation())
        self.layer_list = []
        for i, lin in enumerate(self.linear_layers):
            self.layer_list.append(lin)
            if i < len(self.linear_layers) - 1:
                self.layer_list.append(self.activations[i])
        self.mlp = nn.Sequential(*self.layer_list)

    def forward(self, x: Tensor) -> Tensor:
        return self.mlp(x)

    @property
    def p_factors(self):
        if self.hidden_layers == 0:
            return [(self.p_in, self.p_out)]
        f = [(self.p_in if i == 0 else self.p_hidden, self.p_hidden) for i in range(self.hidden_layers)]
        f.append((self.p_hidden, self.p_out))
        return f

    @property
    def num_of_layers(self) -> int:
        return len(self.linear_layers)


SUPPORTED_GNN_CONVS = [GCNConv_GNNB, GINConv_GNNB, GATConv_GNNB, PNAConv_GNNB, SAGEConv_GNNB]
_CONV_NAME = {GCNConv_GNNB: "gcn", GINConv_GNNB: "gin", SAGEConv_GNNB: "sage", PNAConv_GNNB: "pna"}


class GNNModel(nn.Module):
    """Conv stack (+skip on middle layers, activation after every conv) -> global pooling ->
    MLP head (reference models.py:462-575)."""

    def __init__(self, graph_input_feature_dim: int, graph_input_edge_dim: Optional[int],
                 gnn_hidden_dim: int, gnn_num_layers: int, gnn_output_dim: int,
                 gnn_conv: TorchModuleArg, gnn_activation: TorchModuleArg, gnn_skip_connection: bool,
                 global_pooling: GlobalPooling, mlp_head: MLP,
                 output_activation: TorchModuleArgOptional, gnn_p_in: int = 1, gnn_p_hidden: int = 1,
                 gnn_p_out: int = 1) -> None:
        super().__init__()
        self.graph_input_feature_dim = graph_input_feature_dim
        self.graph_input_edge_dim = graph_input_edge_dim
        self.gnn_hidden_dim = gnn_hidden_dim
        self.gnn_num_layers = gnn_num_layers
        self.gnn_output_dim = gnn_output_dim
        self.gnn_conv = gnn_conv
        if self.gnn_conv not in SUPPORTED_GNN_CONVS:
            raise ValueError(f"gnn_conv {gnn_conv} is not a *_GNNB conv class of this package: {SUPPORTED_GNN_CONVS}")
        self.gnn_activation = gnn_activation
        if self.gnn_activation not in SUPPORTED_ACTIVATIONS:
            raise ValueError(f"gnn_activation {gnn_activation} has no native kernel; choose from {SUPPORTED_ACTIVATIONS}")
        self.gnn_skip_connection = gnn_skip_connection

        self.global_pooling = global_pooling
        self.mlp_head = mlp_head  # registered before gnn_convs: parameter order mlp_head_*, gnn_convs_*
        self.output_activation = output_activation
        self.output_activation_module = None
        if self.output_activation is not None:
            self.output_activation_module = self.output_activation(dim=-1)

        self.gnn_p_in = gnn_p_in
        self.gnn_p_hidden = gnn_p_hidden
        self.gnn_p_out = gnn_p_out

        self.gnn_convs = nn.ModuleList()
        self.gnn_activations = nn.ModuleList()
        L = self.gnn_num_layers
        if L == 0 and self.graph_input_feature_dim != self.gnn_output_dim:
            raise ValueError(
                f"a model without conv layers passes the node features straight to the pooling: gnn_output_dim "
                f"({self.gnn_output_dim}) must equal graph_input_feature_dim ({self.graph_input_feature_dim})")
        for i in range(L):
            if L == 1:
                dims = (self.graph_input_feature_dim, self.gnn_output_dim, self.gnn_p_in, self.gnn_p_out)
            elif i == 0:
                dims = (self.graph_input_feature_dim, self.gnn_hidden_dim, self.gnn_p_in, self.gnn_p_hidden)
            elif i == L - 1:
                dims = (self.gnn_hidden_dim, self.gnn_output_dim, self.gnn_p_hidden, self.gnn_p_out)
            else:
                dims = (self.gnn_hidden_dim, self.gnn_hidden_dim, self.gnn_p_hidden, self.gnn_p_hidden)
            self.gnn_convs.append(self.gnn_conv(dims[0], dims[1], p_in=dims[2], p_out=dims[3]))
            self.gnn_activations.append(self.gnn_activation())

    def forward(self, x: Tensor, edge_index: Tensor, batch: Optional[Tensor] = None) -> Tensor:
        """``batch=None``: one graph -> ``[1, out]``, exactly the reference.  With ``batch``
        (node -> graph index, PyG ``Batch`` convention) every graph is pooled separately ->
        ``[num_graphs, out]``; the reference ignores ``batch`` and pools all nodes into one row
        (models.py:551-553,569; SURVEY finding 3), which has no meaning for independent graphs."""
        h = x
        for i, (conv, act) in enumerate(zip(self.gnn_convs, self.gnn_activations)):
            h_in = h
            h = conv(h, edge_index)
            if self.gnn_skip_connection and i != 0 and i != self.gnn_num_layers - 1:
                h = h + h_in
            h = act(h)
        pooled = self.global_pooling(h) if batch is None else self.global_pooling(h, batch)
        out = self.mlp_head(pooled)
        if self.output_activation_module is not None:
            out = self.output_activation_module(out)
        return out

    # ------------------------------------------------------------------ introspection (reference models.py:577-634)
    @property
    def input_node_features_dim(self):
        return self.graph_input_feature_dim

    @property
    def input_edge_features_dim(self):
        return self.graph_input_edge_dim

    @property
    def output_features_dim(self):
        return self.mlp_head.out_dim

    @property
    def gnn_layer_sizes(self):
        return [(c.in_channels, c.out_channels) for c in self.gnn_convs]

    @property
    def layers(self):
        return dict(self.named_children())

    @property
    def layer_names(self):
        return {k: f"{k}" for k in self.layers}

    @property
    def layer_parameters(self):
        return {k: list(v.named_parameters()) for k, v in self.layers.items()}

    @property
    def layer_parameters_flat(self):
        return [p for l in self.layer_parameters.values() for p in l]

    @property
    def layer_parameter_names(self):
        return {k: [layer_param_name_combiner(self.layer_names[k], p[0]) for p in v]
                for k, v in self.layer_parameters.items()}

    @property
    def layer_parameter_names_flat(self):
        return [p for l in self.layer_parameter_names.values() for p in l]

    @property
    def layer_parameter_shapes(self):
        return {k: [list(p[1].size()) for p in v] for k, v in self.layer_parameters.items()}

    @property
    def layer_parameter_shapes_flat(self):
        return [p for l in self.layer_parameter_shapes.values() for p in l]

    # ------------------------------------------------------------------ what the MI355X emitter consumes
    def spec(self) -> dict:
        """Plain description of the architecture = the fields of ``gnnb_model_desc``
        (include/gnnb_hip.h)."""
        if self.gnn_conv not in _CONV_NAME:
            raise NotImplementedError(f"{self.gnn_conv.__name__} has no native path")
        out_act = None
        if self.output_activation is not None:
            # the reference builds output_activation(dim=-1) (models.py:500-502): a softmax-like module
            out_act = {nn.Softmax: "softmax", nn.LogSoftmax: "log_softmax"}.get(self.output_activation)
            if out_act is None:
                raise NotImplementedError(f"output_activation {self.output_activation} has no native path "
                                          "(nn.Softmax and nn.LogSoftmax do)")
        conv0 = self.gnn_convs[0] if len(self.gnn_convs) else None
        return {
            "conv": _CONV_NAME[self.gnn_conv],
            "num_layers": self.gnn_num_layers,
            "in_dim": self.graph_input_feature_dim,
            "hidden_dim": self.gnn_hidden_dim,
            "out_dim": self.gnn_output_dim,
            "activation": _ACT_NAME[self.gnn_activation],
            "skip": bool(self.gnn_skip_connection),
            "pools": list(self.global_pooling.aggrs),
            "mlp_hidden_layers": self.mlp_head.hidden_layers,
            "mlp_hidden": self.mlp_head.hidden_dim,
            "mlp_out": self.mlp_head.out_dim,
            "mlp_activation": _ACT_NAME[self.mlp_head.activation],
            "gin_eps": float(conv0.eps) if isinstance(conv0, GINConv_GNNB) else 0.0,
            "pna_delta": float(conv0.delta_scaler) if isinstance(conv0, PNAConv_GNNB) else 1.0,
            "output_activation": out_act,
        }

    def canonical_param_names(self) -> List[str]:
        """Flat parameter names (``layer_parameter_names_flat`` style) in the order the C ABI's
        ``gnnb_model_create`` expects: conv layers first, then the head."""
        per_conv = {
            "gcn": ["conv_lin_weight", "conv_bias"],
            "gin": ["mlp_linear_0_weight", "mlp_linear_0_bias", "mlp_linear_1_weight", "mlp_linear_1_bias"],
            "sage": ["conv_lin_l_weight", "conv_lin_l_bias", "conv_lin_r_weight"],
            "pna": ["conv_pre_nns_0_0_weight", "conv_pre_nns_0_0_bias", "conv_post_nns_0_0_weight",
                    "conv_post_nns_0_0_bias", "conv_lin_weight", "conv_lin_bias"],
        }[self.spec()["conv"]]
        names = [f"gnn_convs_{l}_{p}" for l in range(self.gnn_num_layers) for p in per_conv]
        for i in range(self.mlp_head.num_of_layers):
            names += [f"mlp_head_linear_layers_{i}_weight", f"mlp_head_linear_layers_{i}_bias"]
        return names

    def canonical_params(self) -> List[Tensor]:
        by_name = dict(zip(self.layer_parameter_names_flat, [p[1] for p in self.layer_parameters_flat]))
        return [by_name[n].detach() for n in self.canonical_param_names()]
