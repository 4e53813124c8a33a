"""ctypes front-end for the CPU checker -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Loads ``oracle/libgnnb_oracle.so`` (the plain-C restatement, ``gnnb_oracle.c``) and,
when present, ``oracle/_ref/libgnnb_ref.so`` (the reference's own C++ kernel library
compiled in place, ``ref_driver.cpp``).  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ORACLE_SO = Path(os.environ.get("GNNB_ORACLE_SO", HERE / "libgnnb_oracle.so"))  # (override: the sanitizer build of the tests)
REF_SO = HERE / "_ref" / "libgnnb_ref.so"

CONV = {"gcn": 0, "gin": 1, "sage": 2, "pna": 3}
ACT = {"relu": 0, "gelu": 1, "sigmoid": 2, "tanh": 3, "none": 4}
POOL = {"add": 0, "mean": 1, "max": 2}
STD = {"pyg": 0, "hls": 1}
SELF_LOOPS = {"pyg": 0, "hls": 1}


class Desc(C.Structure):
    _fields_ = [
        ("conv_type", C.c_int32),
        ("num_layers", C.c_int32),
        ("in_dim", C.c_int32),
        ("hidden_dim", C.c_int32),
        ("out_dim", C.c_int32),
        ("activation", C.c_int32),
        ("skip", C.c_int32),
        ("num_pools", C.c_int32),
        ("pools", C.c_int32 * 3),
        ("mlp_num_linear", C.c_int32),
        ("mlp_hidden", C.c_int32),
        ("mlp_out", C.c_int32),
        ("mlp_activation", C.c_int32),
        ("gin_eps", C.c_float),
        ("pna_delta", C.c_float),
        ("pna_std_mode", C.c_int32),
        ("gcn_self_loop_mode", C.c_int32),
        ("output_activation", C.c_int32),
        ("fpx_w", C.c_int32),
        ("fpx_i", C.c_int32),
    ]


def build(ref: bool = True) -> None:
    """Compile the checker libraries (gcc/g++; seconds)."""
    subprocess.run(["make", "-C", str(HERE), "oracle"], check=True, capture_output=True)
    if ref:
        subprocess.run(["make", "-C", str(HERE), "ref"], check=True, capture_output=True)


_lib = None
_ref = None

_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not ORACLE_SO.exists() or ORACLE_SO.stat().st_mtime < (HERE / "gnnb_oracle.c").stat().st_mtime:
            build(ref=False)
        _lib = C.CDLL(str(ORACLE_SO))
        _lib.gnnb_oracle_forward.restype = C.c_int
        _lib.gnnb_oracle_forward_batched.restype = C.c_int
        _lib.gnnb_oracle_num_params.restype = C.c_int
    return _lib


def have_ref() -> bool:
    return REF_SO.exists()


def ref() -> C.CDLL:
    global _ref
    if _ref is None:
        if not REF_SO.exists():
            raise FileNotFoundError(f"{REF_SO} not built (needs /root/reference at build time)")
        _ref = C.CDLL(str(REF_SO))
    return _ref


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def _i32(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.int32))


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


# ----------------------------------------------------------------------------- graph prep
def tables(coo, n: int):
    """in_deg, out_deg, offsets, neighbors of ONE graph (graph-local ids)."""
    coo = _i32(coo).reshape(-1, 2)
    e = coo.shape[0]
    in_deg = np.zeros(max(n, 1), np.int32)
    out_deg = np.zeros(max(n, 1), np.int32)
    offsets = np.zeros(max(n, 1), np.int32)
    nbrs = np.zeros(max(e, 1), np.int32)
    L = lib()
    L.gnnb_oracle_degree_tables(_p(coo), n, e, _p(in_deg), _p(out_deg))
    L.gnnb_oracle_neighbor_tables(_p(coo), _p(in_deg), n, e, _p(offsets), _p(nbrs))
    return in_deg[:n], out_deg[:n], offsets[:n], nbrs[:e]


def ref_tables(coo, n: int):
    coo = _i32(coo).reshape(-1, 2)
    e = coo.shape[0]
    in_deg = np.zeros(max(n, 1), np.int32)
    out_deg = np.zeros(max(n, 1), np.int32)
    offsets = np.zeros(max(n, 1), np.int32)
    nbrs = np.zeros(max(e, 1), np.int32)
    rc = ref().gnnb_ref_tables(_p(coo), n, e, _p(in_deg), _p(out_deg), _p(offsets), _p(nbrs))
    if rc != 0:
        raise ValueError("graph exceeds the reference build's MAX_NODES/MAX_EDGES")
    return in_deg[:n], out_deg[:n], offsets[:n], nbrs[:e]


# ----------------------------------------------------------------------------- single convs
def _prep(x, coo):
    x = _f32(x)
    n = x.shape[0]
    coo = _i32(coo).reshape(-1, 2)
    in_deg, out_deg, offsets, nbrs = tables(coo, n)
    return x, n, coo, _i32(in_deg), _i32(out_deg), _i32(offsets), _i32(nbrs)


def conv(kind: str, x, coo, weights, *, eps: float = 0.0, delta: float = 1.0, std: str = "pyg",
         use_ref: bool = False):
    """One conv layer on ONE graph.  ``weights`` in canonical slot order
    (gcn: W,b | gin: W0,b0,W1,b1 | sage: Wl,bl,Wr | pna: Wpre,bpre,Wpost,bpost,Wlin,blin |
    simple/lg: none)."""
    x, n, coo, in_deg, out_deg, offsets, nbrs = _prep(x, coo)
    e = coo.shape[0]
    fin = x.shape[1]
    w = [_f32(t) for t in weights]
    if kind == "gcn":
        fout = w[0].shape[0]
    elif kind == "gin":
        fout = w[2].shape[0]
    elif kind == "sage":
        fout = w[0].shape[0]
    elif kind == "pna":
        fout = w[4].shape[0]
    else:
        fout = fin
    out = np.zeros((n, fout), np.float32)
    if use_ref:
        R = ref()
        common = (n, e, _p(x), _p(out), _p(coo), _p(offsets), _p(nbrs), _p(in_deg), _p(out_deg))
        if kind == "gcn":
            rc = R.gnnb_ref_gcn_conv(*common, _p(w[0]), _p(w[1]), fin, fout)
        elif kind == "gin":
            rc = R.gnnb_ref_gin_conv(*common, _p(w[0]), _p(w[1]), _p(w[2]), _p(w[3]), C.c_float(eps), fin, fout)
        elif kind == "sage":
            rc = R.gnnb_ref_sage_conv(*common, _p(w[0]), _p(w[1]), _p(w[2]), fin, fout)
        elif kind == "pna":
            rc = R.gnnb_ref_pna_conv(*common, _p(w[0]), _p(w[1]), _p(w[2]), _p(w[3]), _p(w[4]), _p(w[5]),
                                     C.c_float(delta), fin, fout)
        else:
            raise ValueError(kind)
        if rc != 0:
            raise ValueError(f"reference build has no instantiation for {kind} {fin}->{fout} (n={n}, e={e})")
        return out
    L = lib()
    common = (n, _p(x), _p(out), _p(offsets), _p(nbrs), _p(in_deg))
    if kind == "gcn":
        L.gnnb_oracle_gcn_conv(*common, _p(w[0]), _p(w[1]), fin, fout)
    elif kind == "gin":
        hidden = w[0].shape[0]
        L.gnnb_oracle_gin_conv(*common, _p(w[0]), _p(w[1]), _p(w[2]), _p(w[3]), C.c_float(eps), fin, hidden, fout)
    elif kind == "sage":
        L.gnnb_oracle_sage_conv(*common, _p(w[0]), _p(w[1]), _p(w[2]), fin, fout)
    elif kind == "pna":
        L.gnnb_oracle_pna_conv(*common, _p(w[0]), _p(w[1]), _p(w[2]), _p(w[3]), _p(w[4]), _p(w[5]),
                               C.c_float(delta), STD[std], fin, fout)
    elif kind == "simple":
        L.gnnb_oracle_simple_conv(*common, fin)
    elif kind == "lg":
        L.gnnb_oracle_lg_conv(*common, fin)
    else:
        raise ValueError(kind)
    return out


def edge_tables(coo, n: int):
    """in_deg, offsets, neighbors, edge_index_table of ONE graph (gnn_builder_lib.h:1126-1166)."""
    coo = _i32(coo).reshape(-1, 2)
    e = coo.shape[0]
    in_deg, out_deg, _, _ = tables(coo, n)
    in_deg = _i32(in_deg)
    offsets = np.zeros(max(n, 1), np.int32)
    nbrs = np.zeros(max(e, 1), np.int32)
    eidx = np.zeros(max(e, 1), np.int32)
    lib().gnnb_oracle_neighbor_edge_tables(_p(coo), _p(in_deg), n, e, _p(offsets), _p(nbrs), _p(eidx))
    return in_deg[:n], offsets[:n], nbrs[:e], eidx[:e]


def gine_conv(x, coo, edge_attr, weights, eps: float = 0.0):
    """One GINE layer on ONE graph; ``weights`` = [We, be, W0, b0, W1, b1]."""
    x = _f32(x)
    n, fin = x.shape
    coo = _i32(coo).reshape(-1, 2)
    ea = _f32(edge_attr)
    if ea.ndim != 2:
        ea = ea.reshape(coo.shape[0], -1)
    in_deg, offsets, nbrs, eidx = edge_tables(coo, n)
    w = [_f32(t) for t in weights]
    hidden, fout = w[2].shape[0], w[4].shape[0]
    out = np.zeros((n, fout), np.float32)
    in_deg, offsets, nbrs, eidx = _i32(in_deg), _i32(offsets), _i32(nbrs), _i32(eidx)
    lib().gnnb_oracle_gine_conv(n, _p(x), _p(ea), _p(out), _p(offsets), _p(nbrs), _p(eidx), _p(in_deg), _p(w[0]), _p(w[1]),
                                _p(w[2]), _p(w[3]), _p(w[4]), _p(w[5]), C.c_float(eps), fin, ea.shape[1], hidden, fout)
    return out


def linear(x, W, b, use_ref: bool = False):
    x, W = _f32(x), _f32(W)
    b = _f32(b) if b is not None else np.zeros(W.shape[0], np.float32)
    y = np.zeros(W.shape[0], np.float32)
    if use_ref:
        rc = ref().gnnb_ref_linear(_p(x), _p(y), _p(W), _p(b), W.shape[1], W.shape[0])
        if rc != 0:
            raise ValueError("reference build has no linear instantiation for this size")
    else:
        lib().gnnb_oracle_linear(_p(x), _p(y), _p(W), _p(b), W.shape[1], W.shape[0])
    return y


def activation(x, kind: str, use_ref: bool = False):
    y = _f32(x).copy()
    if use_ref:
        ref().gnnb_ref_activation(_p(y), C.c_long(y.size), ACT[kind])
    else:
        lib().gnnb_oracle_activation(_p(y), C.c_int64(y.size), ACT[kind])
    return y


def global_pool(x, kind: str, use_ref: bool = False):
    x = _f32(x)
    out = np.zeros(x.shape[1], np.float32)
    if use_ref:
        rc = ref().gnnb_ref_global_pool(_p(x), x.shape[0], x.shape[1], POOL[kind], _p(out))
        if rc != 0:
            raise ValueError("reference build has no pool instantiation for this width")
    else:
        lib().gnnb_oracle_global_pool(_p(x), x.shape[0], x.shape[1], POOL[kind], _p(out))
    return out


# ----------------------------------------------------------------------------- whole model
def make_desc(spec: dict, std: str = "pyg", self_loops: str = None) -> Desc:
    """``spec`` is the plain-dict model description produced by
    ``gnnbuilder_amd.models.GNNModel.spec()`` (conv, num_layers, dims, activation, ...)."""
    d = Desc()
    d.conv_type = CONV[spec["conv"]]
    d.num_layers = spec["num_layers"]
    d.in_dim = spec["in_dim"]
    d.hidden_dim = spec["hidden_dim"]
    d.out_dim = spec["out_dim"]
    d.activation = ACT[spec["activation"]]
    d.skip = int(bool(spec["skip"]))
    d.num_pools = len(spec["pools"])
    for i, p in enumerate(spec["pools"]):
        d.pools[i] = POOL[p]
    d.mlp_num_linear = spec["mlp_hidden_layers"] + 1
    d.mlp_hidden = spec["mlp_hidden"]
    d.mlp_out = spec["mlp_out"]
    d.mlp_activation = ACT[spec["mlp_activation"]]
    d.gin_eps = spec.get("gin_eps", 0.0)
    d.pna_delta = spec.get("pna_delta", 1.0)
    d.pna_std_mode = STD[std]
    # explicit self-loop edges under GCN: PyG drops them (parity target); the reference library counts them.
    # std="hls" selects the reference library's flavour as a whole unless told otherwise.
    d.gcn_self_loop_mode = SELF_LOOPS[self_loops if self_loops is not None else std]
    d.output_activation = {None: 0, "none": 0, "softmax": 1, "log_softmax": 2}[spec.get("output_activation")]
    fpx = spec.get("fpx") or (0, 0)
    d.fpx_w, d.fpx_i = int(fpx[0]), int(fpx[1])
    return d


def _param_array(params):
    keep = [_f32(p) for p in params]
    arr = (C.c_void_p * len(keep))(*[p.ctypes.data for p in keep])
    return keep, arr


def forward(spec: dict, params, x, coo, std: str = "pyg", self_loops: str = None) -> np.ndarray:
    """Whole model on ONE graph; ``params`` in canonical order (see gnnb_oracle.h)."""
    d = make_desc(spec, std, self_loops)
    keep, arr = _param_array(params)
    assert len(keep) == lib().gnnb_oracle_num_params(C.byref(d)), "wrong number of parameters"
    x = _f32(x).reshape(-1, spec["in_dim"])
    coo = _i32(coo).reshape(-1, 2)
    out = np.zeros(spec["mlp_out"], np.float32)
    rc = lib().gnnb_oracle_forward(C.byref(d), arr, _p(x), _p(coo), x.shape[0], coo.shape[0], _p(out))
    if rc != 0:
        raise RuntimeError(f"oracle forward failed rc={rc}")
    return out


def forward_batched(spec: dict, params, x, coo, node_ptr, edge_ptr, std: str = "pyg",
                    self_loops: str = None) -> np.ndarray:
    d = make_desc(spec, std, self_loops)
    keep, arr = _param_array(params)
    assert len(keep) == lib().gnnb_oracle_num_params(C.byref(d)), "wrong number of parameters"
    x = _f32(x).reshape(-1, spec["in_dim"])
    coo = _i32(coo).reshape(-1, 2)
    node_ptr, edge_ptr = _i32(node_ptr), _i32(edge_ptr)
    B = node_ptr.shape[0] - 1
    out = np.zeros((B, spec["mlp_out"]), np.float32)
    rc = lib().gnnb_oracle_forward_batched(C.byref(d), arr, _p(x), _p(coo), _p(node_ptr), _p(edge_ptr), B, _p(out))
    if rc != 0:
        raise RuntimeError(f"oracle batched forward failed rc={rc}")
    return out


def _layer_dims(spec: dict):
    L = spec["num_layers"]
    if L == 1:
        return [(spec["in_dim"], spec["out_dim"])]
    dims = []
    for i in range(L):
        fin = spec["in_dim"] if i == 0 else spec["hidden_dim"]
        fout = spec["out_dim"] if i == L - 1 else spec["hidden_dim"]
        dims.append((fin, fout))
    return dims


def ref_forward(spec: dict, params, x, coo) -> np.ndarray:
    """Whole model on ONE graph composed from the REFERENCE's compiled kernels, sequenced
    as templates/model.cpp.jinja:151-530 does (conv, skip on middle layers, activation,
    pooling concat, MLP head).  PNA std is the library's (HLS) flavour."""
    x = _f32(x).reshape(-1, spec["in_dim"])
    coo = _i32(coo).reshape(-1, 2)
    slots = {"gcn": 2, "gin": 4, "sage": 3, "pna": 6}[spec["conv"]]
    p = [_f32(t) for t in params]
    cur = x
    L = spec["num_layers"]
    for l, (fin, fout) in enumerate(_layer_dims(spec) if L > 0 else []):
        w = p[l * slots:(l + 1) * slots]
        y = conv(spec["conv"], cur, coo, w, eps=spec.get("gin_eps", 0.0), delta=spec.get("pna_delta", 1.0),
                 use_ref=True)
        if spec["skip"] and l != 0 and l != L - 1:
            y = y + cur
        cur = activation(y, spec["activation"], use_ref=True).reshape(y.shape)
    pooled = np.concatenate([global_pool(cur, k, use_ref=True) for k in spec["pools"]])
    h = pooled
    head = p[L * slots:]
    nl = spec["mlp_hidden_layers"] + 1
    for i in range(nl):
        h = linear(h, head[2 * i], head[2 * i + 1], use_ref=True)
        if i != nl - 1:
            h = activation(h, spec["mlp_activation"], use_ref=True)
    return h


def ref_forward_batched(spec: dict, params, x, coo, node_ptr, edge_ptr) -> np.ndarray:
    """The reference's per-graph loop over a packed batch, entirely inside the compiled
    reference library (no Python per graph): bench.py's cpu_baseline of kind "reference"."""
    d = make_desc(spec, "hls")
    keep, arr = _param_array(params)
    x = _f32(x).reshape(-1, spec["in_dim"])
    coo = _i32(coo).reshape(-1, 2)
    node_ptr, edge_ptr = _i32(node_ptr), _i32(edge_ptr)
    B = node_ptr.shape[0] - 1
    out = np.zeros((B, spec["mlp_out"]), np.float32)
    rc = ref().gnnb_ref_forward_batched(C.byref(d), arr, _p(x), _p(coo), _p(node_ptr), _p(edge_ptr), B, _p(out))
    if rc != 0:
        raise ValueError("reference build has no instantiation for this model / graph size")
    return out
