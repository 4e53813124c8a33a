#!/usr/bin/env python3
"""Whole-model golden vectors from the REFERENCE'S OWN generated C++ (build container only).

For every case below this script
  1. renders /root/reference/gnnbuilder/templates/model.{h,cpp}.jinja IN PLACE (jinja2 loader pointed at the
     reference tree) with a duck-typed model object that carries exactly the attributes the templates read
     (conv.__class__.__name__, p_in/p_out, eps, delta_scaler, gnn_layer_sizes, gnn_activation,
     gnn_skip_connection, global_pooling.aggrs, mlp_head.linear_layers / p_factors / activation ...);
  2. compiles the rendered model.cpp with g++ against the reference's gnn_builder_lib.h where it lies
     (-I/root/reference/gnnbuilder/gnn_builder_lib, -O2 -ffp-contract=off), one .so per case in a temp dir;
  3. calls `<name>_top` through ctypes graph by graph on seeded synthetic molecule graphs, as the reference's
     testbench does (model_tb.cpp.jinja:189-205);
  4. writes tests/golden/ref_models/<case>.npz: inputs (x, coo, node_ptr, edge_ptr), every weight under the
     reference's parameter name, the per-graph outputs, and the architecture as JSON.

Nothing from /root/reference is stored: the rendered sources and the .so live in a temp dir and are deleted;
only numbers are committed.  The weights and graphs come from numpy's seeded generator (no torch, no package
code): the fixtures are independent of gnnbuilder_amd.

What these fixtures pin that the reference's own tb_data cannot: the generated whole-model glue --
layer dimensions, skip placement (model.cpp.jinja:264-311), activation after every conv (:313-322), pooling
concat order (:440-448), the MLP head (:454-530), parameter naming / top signature (:686-766).

Caveats kept on purpose (SURVEY findings): PNA is the library's std flavour (sqrt(var + 1e-5)): compare with the
oracle's std="hls" only; GELU is not generated (the HLS emitter maps nn.GELU to the tanh approximation);
GIN needs conv.hidden_dim = out_channels set on the object (the reference leaves it None, finding 6).

Usage: python tools/gen_ref_model_fixtures.py        (needs /root/reference; ~1 min)
"""
from __future__ import annotations

import ctypes as C
import json
import shutil
import subprocess
import sys
import tempfile
from pathlib import Path
from types import SimpleNamespace

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
REF = Path("/root/reference/gnnbuilder")
OUT_DIR = ROOT / "tests" / "golden" / "ref_models"

MAX_NODES, MAX_EDGES = 64, 160


def _act_class(name):
    # the templates read `.__name__` of the activation CLASS (model.cpp.jinja:164-175, :460-471)
    return type({"relu": "ReLU", "sigmoid": "Sigmoid", "tanh": "Tanh"}[name], (), {})


def _conv_class(kind):
    return type({"gcn": "GCNConv_GNNB", "gin": "GINConv_GNNB", "sage": "SAGEConv_GNNB", "pna": "PNAConv_GNNB"}[kind], (), {})


def layer_dims(in_dim, hidden, out_dim, L):
    # reference models.py:519-549
    if L == 1:
        return [(in_dim, out_dim)]
    return [(in_dim if i == 0 else hidden, out_dim if i == L - 1 else hidden) for i in range(L)]


def conv_param_shapes(kind, fi, fo):
    """(reference parameter suffix, shape) in the order PyG registers them is irrelevant here: the top's
    signature is rendered from the same list the call is built from."""
    if kind == "gcn":
        return [("conv_bias", (fo,)), ("conv_lin_weight", (fo, fi))]
    if kind == "gin":
        return [("mlp_linear_0_weight", (fo, fi)), ("mlp_linear_0_bias", (fo,)),
                ("mlp_linear_1_weight", (fo, fo)), ("mlp_linear_1_bias", (fo,))]
    if kind == "sage":
        return [("conv_lin_l_weight", (fo, fi)), ("conv_lin_l_bias", (fo,)), ("conv_lin_r_weight", (fo, fi))]
    if kind == "pna":
        return [("conv_pre_nns_0_0_weight", (fi, 2 * fi)), ("conv_pre_nns_0_0_bias", (fi,)),
                ("conv_post_nns_0_0_weight", (fo, 13 * fi)), ("conv_post_nns_0_0_bias", (fo,)),
                ("conv_lin_weight", (fo, fo)), ("conv_lin_bias", (fo,))]
    raise ValueError(kind)


def build_case(case, rng):
    kind, L = case["conv"], case["layers"]
    in_dim, hidden, out_dim = case["in_dim"], case["hidden"], case["out_dim"]
    dims = layer_dims(in_dim, hidden, out_dim, L)
    convs = []
    for fi, fo in dims:
        c = _conv_class(kind)()
        c.in_channels, c.out_channels, c.p_in, c.p_out = fi, fo, 1, 1
        c.hidden_dim = fo                      # finding 6: the reference leaves None; its MLP uses out_channels
        c.eps = case.get("gin_eps", 0.0)
        c.delta_scaler = case.get("pna_delta", 1.0)
        convs.append(c)
    pools = list(case["pools"])
    gw = out_dim if L > 0 else in_dim
    mlp_in, mlp_hidden, mlp_out, hl = len(pools) * gw, case["mlp_hidden"], case["mlp_out"], case["mlp_hidden_layers"]
    lin_dims = [(mlp_in, mlp_out)] if hl == 0 else \
        [(mlp_in, mlp_hidden)] + [(mlp_hidden, mlp_hidden)] * (hl - 1) + [(mlp_hidden, mlp_out)]
    mlp = SimpleNamespace(in_dim=mlp_in, out_dim=mlp_out, hidden_dim=mlp_hidden, hidden_layers=hl,
                          num_of_layers=len(lin_dims), activation=_act_class(case["mlp_act"]),
                          linear_layers=[SimpleNamespace(in_features=a, out_features=b) for a, b in lin_dims],
                          p_factors=[(1, 1)] * len(lin_dims), p_in=1, p_hidden=1, p_out=1)
    model = SimpleNamespace(gnn_convs=convs, gnn_num_layers=L, gnn_layer_sizes=dims, gnn_hidden_dim=hidden,
                            gnn_output_dim=gw, gnn_activation=_act_class(case["act"]),
                            gnn_skip_connection=bool(case["skip"]),
                            global_pooling=SimpleNamespace(aggrs=pools, num_of_aggrs=len(pools)),
                            mlp_head=mlp, output_activation=None,
                            input_node_features_dim=in_dim, output_features_dim=mlp_out)
    # parameters: mlp_head first, then the convs (reference models.py:497,510)
    params = []
    for i, (a, b) in enumerate(lin_dims):
        params += [(f"mlp_head_linear_layers_{i}_weight", (b, a)), (f"mlp_head_linear_layers_{i}_bias", (b,))]
    for l, (fi, fo) in enumerate(dims):
        params += [(f"gnn_convs_{l}_{s}", shp) for s, shp in conv_param_shapes(kind, fi, fo)]
    weights = {}
    for name, shp in params:
        fan_in = shp[-1] if len(shp) == 2 else None
        bound = 1.0 / np.sqrt(fan_in) if fan_in else 0.1       # torch.nn.Linear-like scale; biases non-zero
        weights[name] = rng.uniform(-bound, bound, size=shp).astype(np.float32)
    return model, params, weights


def molecule(rng, n):
    pairs = set()
    for v in range(1, n):
        pairs.add((int(rng.integers(0, v)), v))
    for _ in range(n // 6):
        a, b = int(rng.integers(0, n)), int(rng.integers(0, n))
        if a != b:
            pairs.add((min(a, b), max(a, b)))
    und = np.asarray(sorted(pairs), dtype=np.int32).reshape(-1, 2)
    both = np.concatenate([und, und[:, ::-1]], axis=0)
    return both[rng.permutation(both.shape[0])]


def make_graphs(case, rng):
    graphs = []
    for n in case["sizes"]:
        coo = molecule(rng, n)
        if case.get("directed_extras"):
            # a few one-directional edges: in-degree != out-degree, duplicate edges allowed
            extra = rng.integers(0, n, size=(max(1, n // 5), 2)).astype(np.int32)
            extra = extra[extra[:, 0] != extra[:, 1]]
            coo = np.concatenate([coo, extra], axis=0)
        graphs.append((rng.uniform(-1, 1, size=(n, case["in_dim"])).astype(np.float32), np.ascontiguousarray(coo)))
    if case.get("isolated"):
        # a graph with an isolated node (in-degree 0) and one with no edges at all (never PNA: the library's
        # Welford finalize divides 0/0 for an empty neighbourhood, gnn_builder_lib.h:702)
        n = 7
        coo = molecule(rng, n - 1)
        graphs.append((rng.uniform(-1, 1, size=(n, case["in_dim"])).astype(np.float32), np.ascontiguousarray(coo)))
        graphs.append((rng.uniform(-1, 1, size=(3, case["in_dim"])).astype(np.float32), np.zeros((0, 2), np.int32)))
    return graphs


def render_and_build(name, model, params, tmp):
    import jinja2

    env = jinja2.Environment(loader=jinja2.FileSystemLoader(str(REF / "templates")))  # rendered where they lie
    ctx = dict(model_top_name=name, max_nodes=MAX_NODES, max_edges=MAX_EDGES, num_nodes_guess=20, num_edges_guess=44,
               degree_guess=3, input_node_features_dim=model.input_node_features_dim,
               output_features_dim=model.output_features_dim, model=model, float_or_fixed="float",
               fpx=SimpleNamespace(W=32, I=16),
               model_parameters=[dict(name=n, shape=list(s), shape_len=len(s), size=int(np.prod(s))) for n, s in params])
    d = Path(tmp) / name
    d.mkdir()
    (d / "model.h").write_text(env.get_template("model.h.jinja").render(**ctx))
    (d / "model.cpp").write_text(env.get_template("model.cpp.jinja").render(**ctx))
    so = d / f"lib{name}.so"
    cmd = ["g++", "-O2", "-ffp-contract=off", "-std=c++14", "-w", "-shared", "-fPIC", f"-I{REF / 'gnn_builder_lib'}",
           "-o", str(so), str(d / "model.cpp")]
    p = subprocess.run(cmd, capture_output=True, text=True)
    if p.returncode != 0:
        raise RuntimeError(f"g++ failed for {name}:\n{p.stderr[-3000:]}")
    return so


def run_case(name, case, tmp):
    rng = np.random.default_rng(case["seed"])
    model, params, weights = build_case(case, rng)
    graphs = make_graphs(case, rng)
    so = render_and_build(name, model, params, tmp)
    lib = C.CDLL(str(so))                      # a unique path per case (SURVEY App. C-7 pitfall)
    top = getattr(lib, f"{name}_top")
    top.restype = None
    wptrs = [weights[n].ctypes.data_as(C.c_void_p) for n, _ in params]
    outs = []
    xbuf = np.zeros((MAX_NODES, case["in_dim"]), np.float32)
    ebuf = np.zeros((MAX_EDGES, 2), np.int32)
    first = True
    for x, coo in graphs:
        n, e = x.shape[0], coo.shape[0]
        assert n <= MAX_NODES and e <= MAX_EDGES
        xbuf[:] = 0
        ebuf[:] = 0
        xbuf[:n] = x
        ebuf[:e] = coo
        out = np.zeros(case["mlp_out"], np.float32)
        top(xbuf.ctypes.data_as(C.c_void_p), ebuf.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p),
            C.c_int(n), C.c_int(e), C.c_int(1 if first else 0), *wptrs)
        first = False
        outs.append(out)
    node_ptr = np.cumsum([0] + [g[0].shape[0] for g in graphs]).astype(np.int32)
    edge_ptr = np.cumsum([0] + [g[1].shape[0] for g in graphs]).astype(np.int32)
    x_all = np.concatenate([g[0] for g in graphs], axis=0)
    coo_all = np.concatenate([g[1] + node_ptr[i] for i, g in enumerate(graphs)], axis=0).astype(np.int32)
    spec = {"conv": case["conv"], "num_layers": case["layers"], "in_dim": case["in_dim"], "hidden_dim": case["hidden"],
            "out_dim": case["out_dim"] if case["layers"] > 0 else case["in_dim"], "activation": case["act"],
            "skip": bool(case["skip"]), "pools": list(case["pools"]), "mlp_hidden_layers": case["mlp_hidden_layers"],
            "mlp_hidden": case["mlp_hidden"], "mlp_out": case["mlp_out"], "mlp_activation": case["mlp_act"],
            "gin_eps": case.get("gin_eps", 0.0), "pna_delta": case.get("pna_delta", 1.0)}
    OUT_DIR.mkdir(parents=True, exist_ok=True)
    np.savez(OUT_DIR / f"{name}.npz", x=x_all, coo=coo_all, node_ptr=node_ptr, edge_ptr=edge_ptr,
             out=np.stack(outs), spec=np.array(json.dumps(spec)),
             param_order=np.array(json.dumps([n for n, _ in params])),
             **{f"w__{n}": w for n, w in weights.items()})
    return np.stack(outs)


def C_(conv, layers, act="relu", skip=True, pools=("add", "mean", "max"), in_dim=9, hidden=16, out_dim=16, mlp_hidden=16,
       mlp_hidden_layers=2, mlp_out=3, mlp_act=None, sizes=(12, 18, 5, 23, 9), seed=0, **kw):
    return dict(conv=conv, layers=layers, act=act, skip=skip, pools=pools, in_dim=in_dim, hidden=hidden, out_dim=out_dim,
                mlp_hidden=mlp_hidden, mlp_hidden_layers=mlp_hidden_layers, mlp_out=mlp_out, mlp_act=mlp_act or act,
                sizes=sizes, seed=seed, **kw)


CASES = {
    # GCN: depth, skip on/off, pool orders, activations, head depths, degenerate graphs
    "gcn_l1_relu": C_("gcn", 1, in_dim=11, out_dim=24, pools=("max",), mlp_hidden_layers=0, isolated=True, seed=1),
    "gcn_l2_relu_b": C_("gcn", 2, in_dim=11, hidden=128, out_dim=128, mlp_hidden=64, mlp_out=19, isolated=True,
                        sizes=(18, 21, 3, 29, 14, 17), seed=2),          # BASELINE config 2 dims
    "gcn_l4_skip_tanh": C_("gcn", 4, act="tanh", pools=("mean", "add"), directed_extras=True, seed=3),
    "gcn_l4_noskip_sigmoid": C_("gcn", 4, act="sigmoid", skip=False, pools=("max", "add", "mean"), hidden=24, out_dim=8, seed=4),
    "gcn_l0": C_("gcn", 0, in_dim=9, out_dim=9, pools=("add", "max"), mlp_hidden_layers=1, seed=5),
    # GIN (eps != 0, hidden_dim patched)
    "gin_l1_relu": C_("gin", 1, pools=("add",), gin_eps=0.2, mlp_out=1, isolated=True, seed=6),
    "gin_l3_skip_relu_b": C_("gin", 3, hidden=32, out_dim=32, pools=("add",), mlp_hidden=64, mlp_out=1,
                             sizes=(25, 31, 12, 40, 22), seed=7),          # BASELINE config 3 shape, narrower
    "gin_l4_skip_tanh": C_("gin", 4, act="tanh", mlp_act="relu", pools=("mean", "max"), gin_eps=0.1, directed_extras=True, seed=8),
    # SAGE
    "sage_l1_sigmoid": C_("sage", 1, act="sigmoid", pools=("mean",), isolated=True, seed=9),
    "sage_l2_relu_b": C_("sage", 2, hidden=64, out_dim=64, mlp_hidden=64, mlp_out=1, isolated=True,
                         sizes=(25, 30, 8, 44, 19), seed=10),              # BASELINE config 5 shape, narrower
    "sage_l4_skip_relu": C_("sage", 4, pools=("max", "add", "mean"), hidden=24, out_dim=12, directed_extras=True, seed=11),
    "sage_l3_noskip_tanh": C_("sage", 3, act="tanh", skip=False, pools=("add", "mean"), mlp_hidden_layers=3, seed=12),
    # PNA (library std flavour; delta != 1 too)
    "pna_l1_relu": C_("pna", 1, in_dim=11, out_dim=16, mlp_out=19, seed=13),
    "pna_l2_relu": C_("pna", 2, in_dim=11, hidden=16, out_dim=16, mlp_out=19, pna_delta=1.0, seed=14),
    "pna_l4_skip_tanh": C_("pna", 4, act="tanh", in_dim=8, hidden=12, out_dim=12, pools=("max", "mean"), pna_delta=2.5,
                           directed_extras=True, seed=15),
    "pna_l3_skip_sigmoid": C_("pna", 3, act="sigmoid", in_dim=11, hidden=16, out_dim=8, pools=("mean", "add", "max"), seed=16),
    # the reference's one published benchmark model (experiments/build_base_benchmarks.py:61-81): SIX conv layers with skip
    # connections, hidden != out (128 / 64 there; 32 / 16 here -- what these vectors pin is the composition: layer dimensions,
    # skip placement over four middle layers, the last layer narrowing, a head of FOUR hidden layers), QM9-shaped inputs
    "ref6_gcn": C_("gcn", 6, in_dim=11, hidden=32, out_dim=16, mlp_hidden=16, mlp_hidden_layers=4, mlp_out=19,
                   sizes=(18, 21, 9, 29, 14), isolated=True, seed=21),
    "ref6_gin": C_("gin", 6, in_dim=11, hidden=32, out_dim=16, mlp_hidden=16, mlp_hidden_layers=4, mlp_out=19,
                   sizes=(18, 21, 9, 29, 14), seed=22),
    "ref6_sage": C_("sage", 6, in_dim=11, hidden=32, out_dim=16, mlp_hidden=16, mlp_hidden_layers=4, mlp_out=19,
                    sizes=(18, 21, 9, 29, 14), isolated=True, seed=23),
    "ref6_pna": C_("pna", 6, in_dim=11, hidden=16, out_dim=8, mlp_hidden=16, mlp_hidden_layers=4, mlp_out=19,
                   sizes=(18, 21, 9, 29, 14), seed=24),
}


def main():
    if not (REF / "templates" / "model.cpp.jinja").exists():
        sys.exit("needs the reference tree at /root/reference (build container only)")
    tmp = tempfile.mkdtemp(prefix="gnnb_ref_models_")
    try:
        only = set(sys.argv[1:])  # (optional: regenerate only the named cases)
        for name, case in CASES.items():
            if only and name not in only:
                continue
            out = run_case(name, case, tmp)
            print(f"{name:28s} graphs {out.shape[0]:2d}  out {out.shape[1]:3d}  |out|max {np.abs(out).max():.4f}")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
