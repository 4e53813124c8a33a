"""Seeded synthetic molecule-shaped graphs (there is no network for QM9 / MoleculeNet / OGB).

Recipe from SURVEY.md 8(d): n ~ clip(round(N(mu, mu/5)), 3, n_max); random recursive tree
(parent of v uniform in [0, v)) plus floor(n/6) ring-closure edges, de-duplicated, both
directions stored, no self loops => about 2.15 directed edges per node; features U(-1, 1) fp32
(as the reference's gen_test_data.py:91-93).
"""
from __future__ import annotations

from typing import Dict

import numpy as np

from .batching import GraphBatch

SHAPES: Dict[str, dict] = {
    # name: mean nodes, max nodes, F_in, task output width
    "qm9": dict(mu=18.0, n_max=29, f_in=11, out=19),
    "molhiv": dict(mu=25.5, n_max=222, f_in=9, out=1),
    "esol": dict(mu=13.3, n_max=55, f_in=9, out=1),
    # ogbg-molhiv with the heavy tail the real set has (the SURVEY recipe's normal sizes never pass 46 nodes, the data set's
    # largest molecule has 222): same mean, log-normal sizes clipped at 222 -- about 1 graph in 100 beyond 57 nodes
    "molhiv_tail": dict(mu=25.5, n_max=222, f_in=9, out=1, dist="lognormal", sigma=0.42),
}


def molecule_edges(rng: np.random.Generator, n: int) -> np.ndarray:
    """Undirected tree + ring closures as a directed [e, 2] (src, dst) array, both directions."""
    pairs = set()
    for v in range(1, n):
        u = int(rng.integers(0, v))
        pairs.add((u, v))
    for _ in range(n // 6):
        a, b = int(rng.integers(0, n)), int(rng.integers(0, n))
        if a != b:
            pairs.add((min(a, b), max(a, b)))
    und = sorted(pairs)
    if not und:
        return np.zeros((0, 2), dtype=np.int32)
    fwd = np.asarray(und, dtype=np.int32)
    both = np.concatenate([fwd, fwd[:, ::-1]], axis=0)
    return both[rng.permutation(both.shape[0])]  # COO order is arbitrary in real datasets


def make_batch(shape: str, num_graphs: int, seed: int = 0) -> GraphBatch:
    cfg = SHAPES[shape]
    rng = np.random.default_rng(seed)
    if cfg.get("dist") == "lognormal":  # mean mu: E[exp(N(m, s^2))] = exp(m + s^2 / 2)
        sg = float(cfg["sigma"])
        raw = rng.lognormal(np.log(cfg["mu"]) - 0.5 * sg * sg, sg, size=num_graphs)
    else:
        raw = rng.normal(cfg["mu"], cfg["mu"] / 5.0, size=num_graphs)
    sizes = np.clip(np.rint(raw), 3, cfg["n_max"]).astype(np.int64)
    node_ptr = np.zeros(num_graphs + 1, dtype=np.int64)
    np.cumsum(sizes, out=node_ptr[1:])
    coos, eptr = [], [0]
    for g in range(num_graphs):
        e = molecule_edges(rng, int(sizes[g]))
        coos.append(e + node_ptr[g])
        eptr.append(eptr[-1] + e.shape[0])
    x = rng.uniform(-1.0, 1.0, size=(int(node_ptr[-1]), cfg["f_in"])).astype(np.float32)
    coo = np.concatenate(coos, axis=0).astype(np.int32) if coos else np.zeros((0, 2), np.int32)
    return GraphBatch(x=x, coo=np.ascontiguousarray(coo), node_ptr=node_ptr.astype(np.int32),
                      edge_ptr=np.asarray(eptr, dtype=np.int32))
