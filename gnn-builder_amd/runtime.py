"""Python binding of the C ABI in ``include/gnnb_hip.h`` (``libgnnb_hip.so``).

This is the accelerated product path.  It has NO fallback: if the HIP library is not built, or no
MI355X is visible, every entry point raises ``GnnbUnavailable`` -- it never routes to PyTorch or
to the CPU oracle.

PyTorch is used here only as plumbing: device tensors give us HBM allocations and the current
HIP stream; the kernels themselves are the hand-written ones in ``csrc/``.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path
from typing import Optional, Sequence

import numpy as np

PKG_DIR = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("GNNB_HIP_LIB", PKG_DIR / "libgnnb_hip.so"))  # override: diagnostic builds only
CSRC_DIR = PKG_DIR / "csrc"

CONV = {"gcn": 0, "gin": 1, "sage": 2, "pna": 3}
ACT = {"relu": 0, "gelu": 1, "sigmoid": 2, "tanh": 3, "none": 4}
POOL = {"add": 0, "mean": 1, "max": 2}
OUT_ACT = {None: 0, "none": 0, "softmax": 1, "log_softmax": 2}
AGG = {"gcn": 0, "sum": 1, "mean": 2, "pna": 3, "lg": 4, "simple": 5, "copy": 6}

GNNB_OK = 0


class GnnbError(RuntimeError):
    pass


class GnnbUnavailable(GnnbError):
    """The HIP extension or the GPU is missing: the product path cannot run (no fallback)."""


class ModelDesc(C.Structure):
    # field-for-field gnnb_model_desc (include/gnnb_hip.h)
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
        ("output_activation", C.c_int32),
        ("fpx_w", C.c_int32),
        ("fpx_i", C.c_int32),
        ("math", C.c_int32),
    ]


MATH_MODES = {"fp32": 0, "bf16x6": 1, "bf16x3": 2, "f16x3": 3}  # gnnb_model_desc::math (include/gnnb_hip.h)
GNNB_ERR_RANGE = -6


class GnnbRangeError(GnnbError):
    """A reduced-precision math mode (bf16x3 / f16x3) produced a non-finite value (``GNNB_ERR_RANGE``): fp16's range was
    exceeded by an activation or a weight.  The flagged forward's results are unspecified; run the model with math="fp32"."""


class GemmSeg(C.Structure):
    _fields_ = [("a_dev", C.c_void_p), ("rowscale_dev", C.c_void_p), ("lda", C.c_int32), ("k", C.c_int32)]


# every symbol include/gnnb_hip.h declares (tests check the .so exports each one)
PATH_NAMES = {0: "none", 1: "layerwise", 2: "stack", 3: "stack_zf"}  # gnnb_hip.h GNNB_PATH_*

EXPORTED_SYMBOLS = [
    "gnnb_version", "gnnb_last_error", "gnnb_device_count", "gnnb_stream_sync",
    "gnnb_model_num_params", "gnnb_model_create", "gnnb_model_destroy", "gnnb_model_get_desc",
    "gnnb_workspace_create", "gnnb_workspace_destroy", "gnnb_workspace_bytes", "gnnb_workspace_set_max_graph_nodes", "gnnb_workspace_set_max_degree",
    "gnnb_workspace_last_path", "gnnb_workspace_set_large_segment",
    "gnnb_forward_batched", "gnnb_forward_prepared", "gnnb_forward_prepared_prep_next", "gnnb_forward_batched_host", "gnnb_workspace_check",
    "gnnb_graph_prep", "gnnb_graph_tables_to_host", "gnnb_aggregate", "gnnb_linear", "gnnb_global_pool",
    "gnnb_event_create", "gnnb_event_record", "gnnb_event_elapsed_ms", "gnnb_event_destroy",
    "gnnb_malloc", "gnnb_free", "gnnb_memcpy_h2d", "gnnb_memcpy_d2h", "gnnb_set_option",
    "gnnb_aggregate_timed", "gnnb_linear_timed", "gnnb_gcn_stack_timed",
    "gnnb_aggregate_edges", "gnnb_edge_index_table_to_host", "gnnb_debug_stream_k_guard", "gnnb_pna_product_aggregate",
]


def build_library(force: bool = False) -> Path:
    """Compile csrc/*.hip for gfx950 with hipcc (cross-compiles without a GPU)."""
    if force and LIB_PATH.exists():
        LIB_PATH.unlink()
    proc = subprocess.run(["make", "-C", str(CSRC_DIR)], capture_output=True, text=True)
    if proc.returncode != 0 or not LIB_PATH.exists():
        raise GnnbError(f"hipcc build of libgnnb_hip.so failed:\n{proc.stdout}\n{proc.stderr}")
    return LIB_PATH


_lib: Optional[C.CDLL] = None


def load_library(require_gpu: bool = True) -> C.CDLL:
    """dlopen libgnnb_hip.so.  torch is imported first on purpose: its bundled libamdhip64 has the
    same SONAME as the one the library was linked against, so both share ONE HIP runtime and
    device pointers / streams can cross the boundary."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise GnnbUnavailable(
                f"{LIB_PATH} is not built; run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C gnn-builder_amd/csrc`.  There is no CPU fallback.")
        import torch  # noqa: F401  (HIP runtime first, see docstring)
        lib = C.CDLL(str(LIB_PATH))
        lib.gnnb_last_error.restype = C.c_char_p
        lib.gnnb_workspace_bytes.restype = C.c_size_t
        lib.gnnb_workspace_bytes.argtypes = [C.c_void_p]
        lib.gnnb_model_destroy.argtypes = [C.c_void_p]
        lib.gnnb_model_destroy.restype = None
        lib.gnnb_workspace_destroy.argtypes = [C.c_void_p]
        lib.gnnb_workspace_destroy.restype = None
        lib.gnnb_event_destroy.argtypes = [C.c_void_p]
        lib.gnnb_event_destroy.restype = None
        lib.gnnb_free.argtypes = [C.c_void_p]
        lib.gnnb_free.restype = None
        lib.gnnb_forward_batched.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        lib.gnnb_forward_prepared.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.gnnb_forward_prepared_prep_next.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                        C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        lib.gnnb_graph_prep.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                        C.c_int, C.c_float, C.c_void_p]
        lib.gnnb_aggregate.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                       C.c_float, C.c_void_p]
        lib.gnnb_linear.argtypes = [C.POINTER(GemmSeg), C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        lib.gnnb_global_pool.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int32), C.c_int,
                                         C.c_void_p, C.c_void_p]
        lib.gnnb_graph_tables_to_host.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.gnnb_workspace_check.argtypes = [C.c_void_p, C.c_void_p]
        lib.gnnb_event_record.argtypes = [C.c_void_p, C.c_void_p]
        lib.gnnb_event_elapsed_ms.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]
        lib.gnnb_stream_sync.argtypes = [C.c_void_p]
        lib.gnnb_set_option.argtypes = [C.c_char_p, C.c_int]
        lib.gnnb_workspace_set_max_graph_nodes.argtypes = [C.c_void_p, C.c_int]
        lib.gnnb_workspace_set_max_degree.argtypes = [C.c_void_p, C.c_int]
        lib.gnnb_workspace_last_path.argtypes = [C.c_void_p]
        lib.gnnb_workspace_set_large_segment.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        lib.gnnb_aggregate_timed.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                             C.c_int, C.c_float, C.c_int, C.c_void_p, C.POINTER(C.c_float)]
        lib.gnnb_linear_timed.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                          C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                          C.POINTER(C.c_float)]
        lib.gnnb_aggregate_edges.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p]
        lib.gnnb_edge_index_table_to_host.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        lib.gnnb_debug_stream_k_guard.argtypes = [C.c_void_p, C.c_void_p]
        lib.gnnb_pna_product_aggregate.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        lib.gnnb_gcn_stack_timed.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                             C.POINTER(C.c_float)]
        _lib = lib
    if require_gpu and _lib.gnnb_device_count() <= 0:
        raise GnnbUnavailable("libgnnb_hip.so loaded but no MI355X (HIP device) is visible; "
                              "the product path has no CPU fallback")
    return _lib


def _check(rc: int) -> None:
    if rc != GNNB_OK:
        msg = load_library(require_gpu=False).gnnb_last_error()
        raise (GnnbRangeError if rc == GNNB_ERR_RANGE else GnnbError)(f"libgnnb_hip error {rc}: {msg.decode() if msg else '?'}")


def set_option(name: str, value: int) -> None:
    _check(load_library(require_gpu=False).gnnb_set_option(name.encode(), int(value)))


def make_desc(spec: dict) -> ModelDesc:
    d = ModelDesc()
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
    d.output_activation = OUT_ACT[spec.get("output_activation")]
    fpx = spec.get("fpx") or (0, 0)            # (W, I) of the reference's FPX, or None for float
    d.fpx_w, d.fpx_i = int(fpx[0]), int(fpx[1])
    # the design's arithmetic (reference: Project(float_or_fixed, fpx) baked into the generated design): a mode name or
    # number fixes it for this model whatever set_option("math") says later; None = -1 = every launch follows the
    # process-wide option as it stands at that launch (A/B measurements and the tests that toggle it)
    m = spec.get("math")
    d.math = -1 if m is None else (MATH_MODES[m] if isinstance(m, str) else int(m))
    return d


def _stream_ptr(stream=None) -> int:
    import torch
    s = stream if stream is not None else torch.cuda.current_stream()
    return int(s.cuda_stream)


def _dptr(t) -> int:
    return int(t.data_ptr())


def _require(t, name: str, dtype, ndim: int, last: Optional[int] = None):
    """A raw pointer crosses the C ABI: a tensor of another dtype / layout / device would be silently reinterpreted
    (a PyG edge_index, int64 [2, E], read as int32 [E, 2] is garbage edges) -- refuse it here."""
    import torch
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise GnnbError(f"{name} must be a CUDA (HIP) tensor")
    if t.dtype != dtype:
        raise GnnbError(f"{name} must be {dtype}, got {t.dtype}")
    if t.dim() != ndim or (last is not None and t.shape[-1] != last):
        raise GnnbError(f"{name} has shape {tuple(t.shape)}; expected {ndim} dimensions" +
                        (f" with last dimension {last}" if last is not None else ""))
    if not t.is_contiguous():
        raise GnnbError(f"{name} must be contiguous")


class CompiledModel:
    """Device-resident model: weights uploaded once (the reference's copy_parameters_flag=1 call),
    plus one workspace sized for the largest batch it will see."""

    def __init__(self, spec: dict, params: Sequence, max_graphs: int, max_nodes: int, max_edges: int,
                 max_graph_nodes: int = 0):
        self.lib = load_library(require_gpu=True)
        self._N = self._E = self._B = 0  # sizes of the prepared batch (graph_prep / forward)
        self.spec = dict(spec)
        self.desc = make_desc(spec)
        host = [np.ascontiguousarray(np.asarray(p.detach().cpu().numpy() if hasattr(p, "detach") else p,
                                                dtype=np.float32)) for p in params]
        n_expect = self.lib.gnnb_model_num_params(C.byref(self.desc))
        if n_expect < 0:
            _check(n_expect)
        if len(host) != n_expect:
            raise GnnbError(f"model needs {n_expect} parameter tensors, got {len(host)}")
        arr = (C.c_void_p * len(host))(*[h.ctypes.data for h in host])
        self._model = C.c_void_p()
        _check(self.lib.gnnb_model_create(C.byref(self.desc), arr, len(host), C.byref(self._model)))
        self._ws = C.c_void_p()
        rc = self.lib.gnnb_workspace_create(self._model, int(max_graphs), int(max_nodes), int(max_edges),
                                            C.byref(self._ws))
        if rc != GNNB_OK:
            self.lib.gnnb_model_destroy(self._model)
            self._model = C.c_void_p()
            _check(rc)
        self.max_graphs, self.max_nodes, self.max_edges = int(max_graphs), int(max_nodes), int(max_edges)
        if max_graph_nodes:
            self.set_max_graph_nodes(max_graph_nodes)

    @classmethod
    def from_model(cls, model, max_graphs: int, max_nodes: int, max_edges: int,
                   max_graph_nodes: int = 0, fpx=None, math=None) -> "CompiledModel":
        """``model``: a ``gnnbuilder_amd.models.GNNModel``.  ``max_graph_nodes``: promise on the largest
        graph (0 = none); small molecules enable the LDS-resident fused kernels, and the promise is
        validated on the device (``check()`` raises if a batch breaks it).  ``fpx``: ``(W, I)`` or a
        ``code_gen.FPX`` = layer-boundary emulation of the reference's ``ap_fixed<W, I>`` build (None: float).
        ``math``: "fp32" | "bf16x6" | "bf16x3" | "f16x3" (or 0..3) = the model's own arithmetic, captured at creation
        (``gnnb_model_desc::math``); None = follow ``set_option("math", ...)`` at every launch.  In the reduced modes
        ``check()`` raises ``GnnbRangeError`` when a kernel produced a non-finite value (fp16's range)."""
        spec = model.spec()
        if math is not None:
            spec["math"] = math
        if fpx is not None:
            spec["fpx"] = (int(fpx.W), int(fpx.I)) if hasattr(fpx, "W") else (int(fpx[0]), int(fpx[1]))
        return cls(spec, model.canonical_params(), max_graphs, max_nodes, max_edges, max_graph_nodes)

    def set_max_graph_nodes(self, n: int) -> None:
        _check(self.lib.gnnb_workspace_set_max_graph_nodes(self._ws, int(n)))

    def set_max_degree(self, d: int) -> None:
        """Promise on the largest in-degree of the following batches (0 = none; a bound, where the reference's ``degree_guess`` is a hint): PNA models
        then run their post-NN products in the degree-class form (d <= 15).  Validated on the device (``check()``)."""
        _check(self.lib.gnnb_workspace_set_max_degree(self._ws, int(d)))

    def stream_k_guard(self, stream=None) -> None:
        """Diagnostics (``gnnb_debug_stream_k_guard``): raises unless this workspace's stream-K scratch has every arrival
        counter at zero and an untouched guard region behind the counters.  Synchronises the stream."""
        _check(self.lib.gnnb_debug_stream_k_guard(self._ws, _stream_ptr(stream)))

    def last_path(self) -> str:
        """Which kernels the last forward on this workspace ran: "layerwise", "stack" (k_gcn2_fused) or "stack_zf"
        (k_gcn2_zf); "none" before the first forward.  Diagnostics only."""
        v = int(self.lib.gnnb_workspace_last_path(self._ws))
        return PATH_NAMES.get(v & 15, "?") + ("+large_layerwise" if v & 16 else "")

    def set_large_segment(self, first_graph: int = -1, first_node: int = -1, first_edge: int = -1) -> None:
        """Graphs [first_graph, B) of the following batches are exempt from the max_graph_nodes promise and run layer by
        layer while the rest takes the LDS-resident stack (``gnnb_workspace_set_large_segment``); the caller orders the
        batch so that they come last (``batching.order_large_last``).  No arguments: no large segment."""
        _check(self.lib.gnnb_workspace_set_large_segment(self._ws, int(first_graph), int(first_node), int(first_edge)))

    @property
    def out_dim(self) -> int:
        return int(self.desc.mlp_out)

    @property
    def workspace_bytes(self) -> int:
        return int(self.lib.gnnb_workspace_bytes(self._ws))

    def close(self) -> None:
        if getattr(self, "_ws", None):
            self.lib.gnnb_workspace_destroy(self._ws)
            self._ws = C.c_void_p()
        if getattr(self, "_model", None):
            self.lib.gnnb_model_destroy(self._model)
            self._model = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ whole forward
    def forward(self, x, coo, node_ptr, edge_ptr, out=None, stream=None):
        """All arguments are torch CUDA tensors (fp32 / int32, contiguous; ``coo`` is [E, 2] (src, dst) rows, NOT a PyG
        ``edge_index`` [2, E] -- transpose it); returns ``out`` [B, mlp_out].  Asynchronous on the current torch stream.
        With a ``max_graph_nodes`` promise set, call ``check()`` on the batch: a broken promise is flagged there, and
        the results of a flagged batch are unspecified.  Without a ``check()`` the flag still surfaces: the next call
        on this workspace after a flagged batch has run raises ``GnnbError`` ("an earlier batch ...", no synchronisation,
        best effort)."""
        import torch
        self._check_batch(x, coo, node_ptr, edge_ptr)
        B = int(node_ptr.numel()) - 1
        N, E = int(x.shape[0]), int(coo.shape[0])
        if out is None:
            out = torch.empty((B, self.out_dim), dtype=torch.float32, device=x.device)
        else:
            _require(out, "out", torch.float32, 2, self.out_dim)
            if out.shape[0] != B or out.device != x.device:
                raise GnnbError(f"out must be [{B}, {self.out_dim}] on {x.device}, got {tuple(out.shape)} on {out.device}")
        _check(self.lib.gnnb_forward_batched(self._model, self._ws, _dptr(x), _dptr(coo), _dptr(node_ptr),
                                             _dptr(edge_ptr), B, N, E, _dptr(out), _stream_ptr(stream)))
        self._keep = (coo, node_ptr, edge_ptr)
        self._N, self._E, self._B = N, E, B  # (the batch is prepared now: the stage-level entry points may follow)
        return out

    def _check_batch(self, x, coo, node_ptr, edge_ptr) -> None:
        import torch
        if x is not None:
            _require(x, "x", torch.float32, 2, int(self.desc.in_dim))
        _require(coo, "coo", torch.int32, 2, 2)
        _require(node_ptr, "node_ptr", torch.int32, 1)
        _require(edge_ptr, "edge_ptr", torch.int32, 1)
        if node_ptr.numel() != edge_ptr.numel() or node_ptr.numel() < 1:
            raise GnnbError("node_ptr and edge_ptr must both have num_graphs + 1 entries")

    def graph_prep(self, coo, node_ptr, edge_ptr, num_nodes: int, stream=None) -> None:
        self._check_batch(None, coo, node_ptr, edge_ptr)
        B = int(node_ptr.numel()) - 1
        self._keep = (coo, node_ptr, edge_ptr)
        _check(self.lib.gnnb_graph_prep(self._ws, _dptr(coo), _dptr(node_ptr), _dptr(edge_ptr), B,
                                        int(num_nodes), int(coo.shape[0]), float(self.desc.pna_delta),
                                        _stream_ptr(stream)))
        self._N, self._E, self._B = int(num_nodes), int(coo.shape[0]), B

    def forward_prepared(self, x, out=None, stream=None):
        import torch
        _require(x, "x", torch.float32, 2, int(self.desc.in_dim))
        if int(x.shape[0]) != self._N:
            raise GnnbError(f"x has {int(x.shape[0])} rows, the prepared batch has {self._N} nodes")
        if out is None:
            out = torch.empty((self._B, self.out_dim), dtype=torch.float32, device=x.device)
        else:
            _require(out, "out", torch.float32, 2, self.out_dim)
            if out.shape[0] != self._B or out.device != x.device:
                raise GnnbError(f"out must be [{self._B}, {self.out_dim}] on {x.device}, got {tuple(out.shape)} on {out.device}")
        _check(self.lib.gnnb_forward_prepared(self._model, self._ws, _dptr(x), _dptr(out), _stream_ptr(stream)))
        return out

    def forward_prepared_prep_next(self, x, nxt: "CompiledModel", coo, node_ptr, edge_ptr, num_nodes: int, out=None, stream=None):
        """``forward_prepared(x)`` on this object's prepared batch, then the graph prep of the NEXT batch on the workspace of
        ``nxt`` -- a second ``CompiledModel`` of the same design (two alternate along a stream of batches) -- in one call
        (``gnnb_forward_prepared_prep_next``): where the forward runs the 2-layer GCN stack kernel, the prep runs inside it.
        ``nxt.forward_prepared`` / ``nxt.forward_prepared_prep_next`` is then the next batch's forward."""
        import torch
        _require(x, "x", torch.float32, 2, int(self.desc.in_dim))
        if int(x.shape[0]) != self._N:
            raise GnnbError(f"x has {int(x.shape[0])} rows, the prepared batch has {self._N} nodes")
        self._check_batch(None, coo, node_ptr, edge_ptr)
        if out is None:
            out = torch.empty((self._B, self.out_dim), dtype=torch.float32, device=x.device)
        else:
            _require(out, "out", torch.float32, 2, self.out_dim)
            if out.shape[0] != self._B or out.device != x.device:
                raise GnnbError(f"out must be [{self._B}, {self.out_dim}] on {x.device}, got {tuple(out.shape)} on {out.device}")
        B = int(node_ptr.numel()) - 1
        _check(self.lib.gnnb_forward_prepared_prep_next(self._model, self._ws, _dptr(x), _dptr(out), nxt._ws, _dptr(coo), _dptr(node_ptr),
                                                        _dptr(edge_ptr), B, int(num_nodes), int(coo.shape[0]), _stream_ptr(stream)))
        nxt._keep = (coo, node_ptr, edge_ptr)
        nxt._N, nxt._E, nxt._B = int(num_nodes), int(coo.shape[0]), B
        return out

    def check(self, stream=None) -> None:
        _check(self.lib.gnnb_workspace_check(self._ws, _stream_ptr(stream)))

    def forward_host(self, x: np.ndarray, coo: np.ndarray, node_ptr: np.ndarray, edge_ptr: np.ndarray) -> np.ndarray:
        """Host-buffer entry (numpy in, numpy out; synchronous)."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        coo = np.ascontiguousarray(coo, dtype=np.int32).reshape(-1, 2)
        node_ptr = np.ascontiguousarray(node_ptr, dtype=np.int32)
        edge_ptr = np.ascontiguousarray(edge_ptr, dtype=np.int32)
        B = node_ptr.shape[0] - 1
        out = np.zeros((B, self.out_dim), np.float32)
        _check(self.lib.gnnb_forward_batched_host(
            self._model, self._ws, x.ctypes.data_as(C.c_void_p), coo.ctypes.data_as(C.c_void_p),
            node_ptr.ctypes.data_as(C.c_void_p), edge_ptr.ctypes.data_as(C.c_void_p), B, x.shape[0],
            coo.shape[0], out.ctypes.data_as(C.c_void_p)))
        return out

    # ------------------------------------------------------------------ stage-level entry points
    def tables_to_host(self, stream=None):
        row_ptr = np.zeros(self._N + 1, np.int32)
        col = np.zeros(max(self._E, 1), np.int32)
        in_deg = np.zeros(max(self._N, 1), np.int32)
        _check(self.lib.gnnb_graph_tables_to_host(self._ws, row_ptr.ctypes.data_as(C.c_void_p),
                                                  col.ctypes.data_as(C.c_void_p),
                                                  in_deg.ctypes.data_as(C.c_void_p), _stream_ptr(stream)))
        return row_ptr, col[:self._E], in_deg[:self._N]

    def aggregate(self, kind: str, x, self_term=None, eps: float = 0.0, out=None, stream=None):
        import torch
        _require(x, "x", torch.float32, 2)
        w = int(x.shape[1])
        # (raw pointers cross the C ABI: the kernel reads x[j * w] for every source j of the prepared batch)
        if int(x.shape[0]) != self._N:
            raise GnnbError(f"x has {int(x.shape[0])} rows, the prepared batch has {self._N} nodes")
        ow = 4 * w if kind == "pna" else w
        if self_term is not None:
            _require(self_term, "self_term", torch.float32, 2, w)
            if int(self_term.shape[0]) != self._N or self_term.device != x.device:
                raise GnnbError(f"self_term must be [{self._N}, {w}] on {x.device}")
        if out is None:
            out = torch.empty((x.shape[0], ow), dtype=torch.float32, device=x.device)
        else:
            _require(out, "out", torch.float32, 2, ow)
            if out.shape[0] != x.shape[0] or out.device != x.device:
                raise GnnbError(f"out must be [{int(x.shape[0])}, {ow}] on {x.device}")
        _check(self.lib.gnnb_aggregate(self._ws, AGG[kind], _dptr(x),
                                       _dptr(self_term) if self_term is not None else None, _dptr(out), w,
                                       float(eps), _stream_ptr(stream)))
        return out

    def pna_product_aggregate(self, x, wb, out=None, stream=None):
        """``max | min | mean | std`` over every node's sources of ``p_j = wb @ x_j`` in one kernel (``k_pna_pagg``); ``wb``:
        [width, ldw] view of the x_j half of the pre-NN weight (``W_pre[:, width:]``: a strided view is fine, the row stride is
        passed on).  Needs the workspace's ``max_graph_nodes`` promise (<= 57 with the default node tiles)."""
        import torch
        _require(x, "x", torch.float32, 2)
        w = int(x.shape[1])
        if int(x.shape[0]) != self._N:
            raise GnnbError(f"x has {int(x.shape[0])} rows, the prepared batch has {self._N} nodes")
        if wb.dtype != torch.float32 or wb.dim() != 2 or tuple(wb.shape) != (w, w) or wb.stride(1) != 1 or wb.device != x.device:
            raise GnnbError(f"wb must be a [{w}, {w}] float32 view with unit column stride on {x.device}")
        if out is None:
            out = torch.empty((x.shape[0], 4 * w), dtype=torch.float32, device=x.device)
        else:
            _require(out, "out", torch.float32, 2, 4 * w)
        _check(self.lib.gnnb_pna_product_aggregate(self._ws, _dptr(x), wb.data_ptr(), int(wb.stride(0)), _dptr(out), w, _stream_ptr(stream)))
        return out

    def edge_index_table_to_host(self, stream=None) -> np.ndarray:
        """The reference's edge_index_table of the prepared batch: COO row of every CSR slot."""
        eid = np.zeros(max(self._E, 1), np.int32)
        _check(self.lib.gnnb_edge_index_table_to_host(self._ws, eid.ctypes.data_as(C.c_void_p), _stream_ptr(stream)))
        return eid[:self._E]

    def aggregate_edges(self, x, edge_term, eps: float = 0.0, out=None, stream=None):
        """GINE aggregate: ``(1 + eps) x_i + sum_j relu(x_j + edge_term[e])``; ``edge_term`` [E, width] in COO order."""
        import torch
        # raw pointers cross the C ABI: the kernel reads edge_term[eid * width] for every CSR slot and x[j * width] for
        # every source -- a short, narrower, int64, CPU or strided tensor would be read out of bounds or reinterpreted
        _require(x, "x", torch.float32, 2)
        w = int(x.shape[1])
        if int(x.shape[0]) != self._N:
            raise GnnbError(f"x has {int(x.shape[0])} rows, the prepared batch has {self._N} nodes")
        _require(edge_term, "edge_term", torch.float32, 2, w)
        if int(edge_term.shape[0]) != self._E or edge_term.device != x.device:
            raise GnnbError(f"edge_term must be [{self._E}, {w}] on {x.device} (one row per COO edge of the prepared batch), "
                            f"got {tuple(edge_term.shape)} on {edge_term.device}")
        if out is None:
            out = torch.empty_like(x)
        else:
            _require(out, "out", torch.float32, 2, w)
            if out.shape[0] != x.shape[0] or out.device != x.device:
                raise GnnbError(f"out must be {tuple(x.shape)} on {x.device}")
        _check(self.lib.gnnb_aggregate_edges(self._ws, _dptr(x), _dptr(edge_term), _dptr(out), int(x.shape[1]),
                                             float(eps), _stream_ptr(stream)))
        return out

    def gine_conv(self, x, edge_attr, w_edge, b_edge, w0, b0, w1, b1, eps: float = 0.0, stream=None):
        """One GINEConv layer on the prepared batch (reference gine_conv, gnn_builder_lib.h:1640-1742):
        edge projection (GEMM) -> GINE aggregate -> Linear -> ReLU -> Linear."""
        pe = linear([(edge_attr, None)], w_edge, b_edge, stream=stream)
        z = self.aggregate_edges(x, pe, eps=eps, stream=stream)
        h = linear([(z, None)], w0, b0, act="relu", stream=stream)
        return linear([(h, None)], w1, b1, stream=stream)

    def aggregate_timed(self, kind: str, xs, outs, iters: int, self_term=None, eps: float = 0.0, stream=None) -> float:
        """Mean microseconds per launch over ``iters`` back-to-back launches issued from C,
        rotating over the (x, out) buffer pairs; HIP events on the launch stream."""
        n = len(xs)
        xa = (C.c_void_p * n)(*[_dptr(t) for t in xs])
        oa = (C.c_void_p * n)(*[_dptr(t) for t in outs])
        us = C.c_float()
        _check(self.lib.gnnb_aggregate_timed(self._ws, AGG[kind], xa, _dptr(self_term) if self_term is not None else None,
                                             oa, n, int(xs[0].shape[1]), float(eps), int(iters), _stream_ptr(stream),
                                             C.byref(us)))
        return float(us.value)

    def gcn_stack_timed(self, x, iters: int, stream=None) -> float:
        """Mean microseconds per launch of the fused GCN stack + pooling kernel on the prepared
        batch (``graph_prep`` first); HIP events on the launch stream.  Raises if that path is not eligible."""
        us = C.c_float()
        _check(self.lib.gnnb_gcn_stack_timed(self._model, self._ws, _dptr(x), int(iters), _stream_ptr(stream),
                                             C.byref(us)))
        return float(us.value)

    def global_pool(self, x, pools: Sequence[str], out=None, stream=None):
        import torch
        d = int(x.shape[1])
        if out is None:
            out = torch.empty((self._B, len(pools) * d), dtype=torch.float32, device=x.device)
        arr = (C.c_int32 * len(pools))(*[POOL[p] for p in pools])
        _check(self.lib.gnnb_global_pool(self._ws, _dptr(x), d, arr, len(pools), _dptr(out), _stream_ptr(stream)))
        return out


def linear(segments, weight, bias=None, skip=None, act: str = "none", out=None, stream=None):
    """``segments``: list of (A [M, K_s] CUDA tensor, rowscale [M] or None).  weight [N, sum K_s]."""
    import torch
    lib = load_library(require_gpu=True)
    M = int(segments[0][0].shape[0])
    N = int(weight.shape[0])
    segs = (GemmSeg * len(segments))()
    for i, (a, rs) in enumerate(segments):
        segs[i].a_dev = _dptr(a)
        segs[i].rowscale_dev = _dptr(rs) if rs is not None else None
        segs[i].lda = int(a.stride(0))
        segs[i].k = int(a.shape[1])
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=weight.device)
    _check(lib.gnnb_linear(segs, len(segments), _dptr(weight), int(weight.stride(0)),
                           _dptr(bias) if bias is not None else None,
                           _dptr(skip) if skip is not None else None, _dptr(out), M, N, ACT[act],
                           _stream_ptr(stream)))
    return out


def stream_k_guard(stream=None) -> None:
    """Diagnostics: the standalone ``linear``'s stream-K scratch of (current device, stream) -- counters zero, guard whole."""
    _check(load_library().gnnb_debug_stream_k_guard(None, _stream_ptr(stream)))


def linear_timed(a, weight, bias, out, act: str, iters: int, stream=None) -> float:
    """Mean microseconds per launch of one ``gnnb_linear`` configuration, launched from C."""
    lib = load_library(require_gpu=True)
    us = C.c_float()
    _check(lib.gnnb_linear_timed(_dptr(a), int(a.stride(0)), int(a.shape[1]), _dptr(weight), int(weight.stride(0)),
                                 _dptr(bias) if bias is not None else None, _dptr(out), int(a.shape[0]),
                                 int(weight.shape[0]), ACT[act], int(iters), _stream_ptr(stream), C.byref(us)))
    return float(us.value)


class HipTimer:
    """hipEvent pair on an explicit stream (bench.py times kernels on the stream they run on)."""

    def __init__(self):
        self.lib = load_library(require_gpu=True)
        self.a, self.b = C.c_void_p(), C.c_void_p()
        _check(self.lib.gnnb_event_create(C.byref(self.a)))
        _check(self.lib.gnnb_event_create(C.byref(self.b)))

    def start(self, stream=None):
        _check(self.lib.gnnb_event_record(self.a, _stream_ptr(stream)))

    def stop(self, stream=None):
        _check(self.lib.gnnb_event_record(self.b, _stream_ptr(stream)))

    def elapsed_ms(self) -> float:
        ms = C.c_float()
        _check(self.lib.gnnb_event_elapsed_ms(self.a, self.b, C.byref(ms)))
        return float(ms.value)

    def __del__(self):
        try:
            self.lib.gnnb_event_destroy(self.a)
            self.lib.gnnb_event_destroy(self.b)
        except Exception:
            pass
