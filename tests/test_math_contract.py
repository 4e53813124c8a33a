"""The math mode is a property of the DESIGN, and the reduced modes have an overflow contract (round 6).

Reference: ``Project(float_or_fixed, fpx)`` (code_gen.py:63-82) is baked into the generated design (model.h.jinja:38-62:
F_TYPE / W_TYPE = float or ap_fixed<W, I, AP_TRN, AP_WRAP> -- overflow is DEFINED there).  Here:
* ``gnnb_model_desc::math`` is captured by ``gnnb_model_create``; two designs of different precision in one process do not
  change each other's arithmetic, whatever ``gnnb_set_option("math", ...)`` is called meanwhile;
* the reduced modes (bf16x3 / f16x3) look at what their kernels produce: a value beyond fp16's range sets flag 64 of the
  workspace, ``check()`` raises ``GnnbRangeError`` (GNNB_ERR_RANGE), and the same inputs run clean with math="fp32".
Every case is the route of one reduced kernel: k_gcn2_zf (bf16x3, f16x3), k_gcn2_fused<GIN> and its deep-GCN form,
k_linear_dma (GraphSAGE's K = 512 GEMM; PNA's row-class GEMM), k_pna_pagg.  Needs a real MI355X (``-m gpu``).
"""
import numpy as np
import pytest
import torch

from gnnbuilder_amd import runtime, synthetic
from helpers import canon, make_model, to_dev
from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    runtime.load_library(require_gpu=True)  # fails loudly: no fallback
    return torch.device("cuda:0")


# (conv, layers, hidden, reduced mode, path the workspace must report, degree promise?)
CASES = [
    ("gcn", 2, 128, "f16x3", "stack_zf", False),
    ("gcn", 2, 128, "bf16x3", "stack_zf", False),
    ("gin", 3, 128, "f16x3", "stack", False),
    ("gcn", 3, 128, "f16x3", "stack", False),
    ("sage", 2, 256, "f16x3", "layerwise", False),
    ("pna", 3, 128, "f16x3", "layerwise", True),
]
IDS = [f"{c[0]}{c[1]}_{c[3]}" for c in CASES]


def _setup(conv, layers, hidden, count=192, seed=4):
    model = make_model(conv, in_dim=11, hidden=hidden, layers=layers, out_dim=hidden, act="relu", pools=("add", "mean", "max"), task_out=7, seed=seed)
    batch = synthetic.make_batch("qm9", count, seed=seed + 1)
    return model, batch


def _compile(model, batch, math, degree):
    cm = runtime.CompiledModel.from_model(model, batch.num_graphs, batch.num_nodes, batch.num_edges,
                                          max_graph_nodes=int(np.diff(batch.node_ptr).max()), math=math)
    if degree:
        cm.set_max_degree(int(np.bincount(batch.coo[:, 1], minlength=batch.num_nodes).max()))
    return cm


@pytest.mark.parametrize("conv,layers,hidden,mode,path,degree", CASES, ids=IDS)
def test_two_designs_of_different_math_share_a_process(dev, conv, layers, hidden, mode, path, degree):
    model, batch = _setup(conv, layers, hidden)
    args = to_dev(batch, dev)
    ref = O.forward_batched(model.spec(), canon(model), batch.x, batch.coo, batch.node_ptr, batch.edge_ptr)
    scale = max(1.0, float(np.abs(ref).max()))
    exact = _compile(model, batch, "fp32", degree)
    reduced = _compile(model, batch, mode, degree)
    follows = _compile(model, batch, None, degree)  # math = -1: the process-wide option at every launch (the pre-103 behaviour)
    assert int(exact.desc.math) == 0 and int(reduced.desc.math) == runtime.MATH_MODES[mode] and int(follows.desc.math) == -1
    solo_exact = exact.forward(*args).cpu().numpy()
    exact.check()
    solo_reduced = reduced.forward(*args).cpu().numpy()
    reduced.check()  # (unit-scale inputs: nothing leaves fp16's range, no flag)
    assert reduced.last_path() == path and exact.last_path() == path
    assert np.abs(solo_exact - ref).max() < 1e-4 * scale and np.abs(solo_reduced - ref).max() < 1e-4 * scale
    assert not np.array_equal(solo_exact, solo_reduced)  # the modes are different arithmetic: the field is not ignored
    other = 2 if mode == "f16x3" else 3
    try:
        # interleaved forwards, the process-wide option flipped between them: each design keeps its own bits
        for flip in (other, 0, runtime.MATH_MODES[mode], 1, 0):
            runtime.set_option("math", flip)
            assert np.array_equal(reduced.forward(*args).cpu().numpy(), solo_reduced)
            assert np.array_equal(exact.forward(*args).cpu().numpy(), solo_exact)
        # ... and a model created with math = -1 follows it
        runtime.set_option("math", runtime.MATH_MODES[mode])
        assert np.array_equal(follows.forward(*args).cpu().numpy(), solo_reduced)
        runtime.set_option("math", 0)
        assert np.array_equal(follows.forward(*args).cpu().numpy(), solo_exact)
    finally:
        runtime.set_option("math", 0)
    for cm in (exact, reduced, follows):
        cm.check()


@pytest.mark.parametrize("conv,layers,hidden,mode,path,degree", CASES, ids=IDS)
def test_reduced_modes_flag_values_beyond_fp16_range(dev, conv, layers, hidden, mode, path, degree):
    """Hidden activations of ~1e6 (inputs scaled by 3e6): fp16 pieces overflow (bf16 pieces do not: bf16 has fp32's range).
    The f16x3 design reports GNNB_ERR_RANGE -- from check(), and lazily from the next forward --; the fp32 design of the same
    model runs the same inputs clean and matches the oracle."""
    model, batch = _setup(conv, layers, hidden, seed=9)
    big = batch.x * np.float32(3e6)
    args = list(to_dev(batch, dev))
    args[0] = torch.from_numpy(big).to(dev)
    ref = O.forward_batched(model.spec(), canon(model), big, batch.coo, batch.node_ptr, batch.edge_ptr)
    assert np.isfinite(ref).all()
    scale = float(np.abs(ref).max())
    reduced = _compile(model, batch, mode, degree)
    out = reduced.forward(*args).cpu().numpy()
    assert reduced.last_path() == path
    if mode == "bf16x3":
        reduced.check()  # bf16 pieces carry fp32's exponent range: nothing to flag, and the result is right
        assert np.abs(out - ref).max() < 1e-4 * scale
        return
    with pytest.raises(runtime.GnnbRangeError):
        reduced.check()
    reduced.check()  # (reported once: the flag is reset on read)
    # lazy detection for callers that never call check(): the forward AFTER a flagged one refuses
    reduced.forward(*args)
    torch.cuda.synchronize()
    with pytest.raises(runtime.GnnbRangeError):
        reduced.forward(*args)
    # the clean rerun: the same model as a native-fp32 design
    exact = _compile(model, batch, "fp32", degree)
    good = exact.forward(*args).cpu().numpy()
    exact.check()
    assert np.isfinite(good).all() and np.abs(good - ref).max() < 1e-4 * scale
    # unit-scale inputs on the workspace that was flagged: clean again
    clean = reduced.forward(*to_dev(batch, dev)).cpu().numpy()
    reduced.check()
    assert np.isfinite(clean).all()


def test_desc_rejects_unknown_math_modes(dev):
    model, batch = _setup("gcn", 2, 64, count=8)
    with pytest.raises(runtime.GnnbError):
        runtime.CompiledModel.from_model(model, batch.num_graphs, batch.num_nodes, batch.num_edges, math=4)
    with pytest.raises(runtime.GnnbError):
        runtime.CompiledModel.from_model(model, batch.num_graphs, batch.num_nodes, batch.num_edges, math=-2)


def test_a_thread_flipping_the_option_does_not_reach_a_design_with_its_own_math(dev):
    """The process-wide knobs are relaxed atomics and a model's math mode is its own: a second thread that flips
    gnnb_set_option("math", ...) as fast as it can while this thread runs forwards of an f16x3 design and of an fp32 design
    changes neither's bits (ctypes releases the GIL around every call: the two threads really run side by side)."""
    import threading
    model, batch = _setup("gcn", 2, 128, count=256, seed=21)
    args = to_dev(batch, dev)
    exact = _compile(model, batch, "fp32", False)
    reduced = _compile(model, batch, "f16x3", False)
    solo_exact = exact.forward(*args).cpu().numpy()
    solo_reduced = reduced.forward(*args).cpu().numpy()
    stop = threading.Event()

    def flip():
        i = 0
        while not stop.is_set():
            runtime.set_option("math", i & 3)
            i += 1

    t = threading.Thread(target=flip)
    t.start()
    try:
        for _ in range(40):
            assert np.array_equal(reduced.forward(*args).cpu().numpy(), solo_reduced)
            assert np.array_equal(exact.forward(*args).cpu().numpy(), solo_exact)
    finally:
        stop.set()
        t.join()
        runtime.set_option("math", 0)
    exact.check()
    reduced.check()
