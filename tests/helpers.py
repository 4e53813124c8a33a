"""Shared builders for the test-suite (models with the reference's default PyTorch init,
as gen_test_data.py:217 does, and seeded synthetic batches)."""
import numpy as np
import torch

import gnnbuilder_amd as gnnb
from gnnbuilder_amd import synthetic

CONVS = {"gcn": gnnb.GCNConv_GNNB, "gin": gnnb.GINConv_GNNB, "sage": gnnb.SAGEConv_GNNB, "pna": gnnb.PNAConv_GNNB}
ACTS = {"relu": torch.nn.ReLU, "gelu": torch.nn.GELU, "sigmoid": torch.nn.Sigmoid, "tanh": torch.nn.Tanh}


def make_model(conv="gcn", in_dim=11, hidden=128, layers=2, out_dim=None, act="relu", skip=True,
               pools=("add", "mean", "max"), mlp_hidden=64, mlp_layers=2, task_out=19, mlp_act="relu", seed=0):
    torch.manual_seed(seed)
    out_dim = hidden if out_dim is None else out_dim
    gw = in_dim if layers == 0 else out_dim
    model = gnnb.GNNModel(in_dim, None, hidden, layers, out_dim, CONVS[conv], ACTS[act], skip,
                          gnnb.GlobalPooling(list(pools)), gnnb.MLP(len(pools) * gw, task_out, mlp_hidden, mlp_layers,
                                                                   activation=ACTS[mlp_act]), None)
    # PyG initialises GCN's bias to zero; give every bias a value so it is exercised
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("bias"):
                p.uniform_(-0.1, 0.1)
    return model.eval()


def canon(model):
    return [p.numpy() for p in model.canonical_params()]


def to_dev(batch, dev):
    return (torch.from_numpy(batch.x).to(dev), torch.from_numpy(batch.coo).to(dev),
            torch.from_numpy(batch.node_ptr).to(dev), torch.from_numpy(batch.edge_ptr).to(dev))


def batch_vector(batch):
    return np.repeat(np.arange(batch.num_graphs), np.diff(batch.node_ptr)).astype(np.int64)
