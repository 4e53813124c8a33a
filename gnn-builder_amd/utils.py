"""Helpers shared by the model API and the emitter -- counterpart of the reference's
``gnnbuilder/utils.py`` (dataset statistics :9-96, parameter naming :99-100, raw tensor
serialisation :113-115).  The Vitis csynth XML parser (:118-173) has no GPU analogue."""
from __future__ import annotations

from pathlib import Path

import numpy as np
import torch


def compute_max_nodes_and_edges(dataset):
    max_node, max_edge = 0, 0
    for data in dataset:
        max_node = max(max_node, int(data.num_nodes))
        max_edge = max(max_edge, int(data.num_edges))
    return max_node, max_edge


def compute_average_nodes_and_edges(dataset, round_val: bool = True):
    nodes = [int(d.num_nodes) for d in dataset]
    edges = [int(d.num_edges) for d in dataset]
    avg_nodes, avg_edges = float(np.mean(nodes)), float(np.mean(edges))
    if round_val:
        return int(round(avg_nodes)), int(round(avg_edges))
    return avg_nodes, avg_edges


def compute_median_nodes_and_edges(dataset, round_val: bool = True):
    nodes = [int(d.num_nodes) for d in dataset]
    edges = [int(d.num_edges) for d in dataset]
    return int(np.median(nodes)), int(np.median(edges))


def compute_degree(graph):
    ei = torch.as_tensor(graph.edge_index)
    n = int(graph.num_nodes)
    in_degree = torch.bincount(ei[1], minlength=n).float() if ei.numel() else torch.zeros(n)
    out_degree = torch.bincount(ei[0], minlength=n).float() if ei.numel() else torch.zeros(n)
    return in_degree.tolist(), out_degree.tolist()


def compute_average_degree(dataset, round_val=True):
    per_graph = [float(np.mean(compute_degree(d)[0])) for d in dataset]
    avg = float(np.mean(per_graph))
    return int(np.ceil(avg)) if round_val else avg


def compute_median_degree(dataset):
    per_graph = [float(np.median(compute_degree(d)[0])) for d in dataset]
    return int(np.ceil(np.median(per_graph)))


def layer_param_name_combiner(layer_name, param_name):
    return f"{layer_name}_{param_name.replace('.', '_')}"


def read_file(file_path):
    with open(file_path, "r") as f:
        return f.read()


def write_file(file_path, content):
    with open(file_path, "w") as f:
        f.write(content)


def serialize_tensor(param: torch.Tensor, fp: Path, np_type=np.float32):
    """Raw little-endian dump, no header (the tb_data format)."""
    np.ascontiguousarray(param.detach().cpu().numpy().astype(np_type)).tofile(fp)
