"""The CPU restatement under AddressSanitizer + UndefinedBehaviorSanitizer (the reference keeps a
commented-out -fsanitize line, gnn_builder_lib_test/makefile:18; GPU sanitizers are not available on the pool, so the
CPU side is where memory errors of the shared index arithmetic can be caught).  The oracle is rebuilt with
-fsanitize=address,undefined into a temp dir and driven from a subprocess (ASan must be the first DSO loaded) over every
conv family, degenerate graphs, GINE, the fixed-point emulation and a softmax head."""
import os
import subprocess
import sys
import textwrap
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent

DRIVER = textwrap.dedent('''
    import sys
    import numpy as np
    sys.path.insert(0, "%(root)s"); sys.path.insert(0, "%(root)s/tests")
    from oracle import oracle as O
    from helpers import canon, make_model
    from gnnbuilder_amd import synthetic
    from gnnbuilder_amd.batching import pack_graphs
    rng = np.random.default_rng(0)
    graphs = [(rng.uniform(-1, 1, (1, 9)), np.zeros((0, 2), np.int32)),
              (rng.uniform(-1, 1, (5, 9)), np.array([[0, 1], [1, 0], [2, 2], [3, 1], [3, 1]])),
              (rng.uniform(-1, 1, (0, 9)), np.zeros((0, 2), np.int32)),
              (rng.uniform(-1, 1, (40, 9)), np.stack([rng.integers(0, 40, 300), rng.integers(0, 40, 300)], 1))]
    edge = pack_graphs([(np.asarray(x, np.float32), np.asarray(c, np.int32)) for x, c in graphs])
    mol = synthetic.make_batch("molhiv", 12, seed=1)
    for conv in ("gcn", "gin", "sage", "pna"):
        for layers in (0, 1, 3):
            if layers == 0 and conv != "gcn":
                continue
            model = make_model(conv, in_dim=9, hidden=9 if layers == 0 else 12, layers=layers, task_out=3)
            for batch in (edge, mol):
                for extra in ({}, {"fpx": (16, 8)}, {"output_activation": "softmax"}):
                    out = O.forward_batched(dict(model.spec(), **extra), canon(model), batch.x, batch.coo, batch.node_ptr, batch.edge_ptr)
                    assert np.isfinite(out).all()
                O.forward_batched(model.spec(), canon(model), batch.x, batch.coo, batch.node_ptr, batch.edge_ptr, std="hls")
    x, coo = mol.graph(0)
    ea = rng.uniform(-1, 1, (coo.shape[0], 4)).astype(np.float32)
    ws = [rng.uniform(-1, 1, s).astype(np.float32) for s in ((9, 4), (9,), (7, 9), (7,), (5, 7), (5,))]
    O.gine_conv(x, coo, ea, ws, eps=0.1)
    for kind in ("simple", "lg"):
        O.conv(kind, x, coo, [])
    print("sanitized oracle ok")
''')


def _libasan():
    p = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True)
    path = p.stdout.strip()
    return path if p.returncode == 0 and os.path.isabs(path) and os.path.exists(path) else None


@pytest.mark.skipif(_libasan() is None, reason="gcc's libasan is not installed")
def test_oracle_is_clean_under_asan_and_ubsan(tmp_path):
    so = tmp_path / "libgnnb_oracle.so"
    build = subprocess.run(["gcc", "-O1", "-g", "-fPIC", "-shared", "-std=c99", "-ffp-contract=off", "-fsanitize=address,undefined",
                            "-fno-omit-frame-pointer", "-o", str(so), str(ROOT / "oracle" / "gnnb_oracle.c"), "-lm"],
                           capture_output=True, text=True)
    assert build.returncode == 0, build.stderr
    env = dict(os.environ, LD_PRELOAD=_libasan(), ASAN_OPTIONS="detect_leaks=0:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1",
               GNNB_ORACLE_SO=str(so))
    run = subprocess.run([sys.executable, "-c", DRIVER % {"root": str(ROOT)}], capture_output=True, text=True, env=env, timeout=600)
    assert run.returncode == 0 and "sanitized oracle ok" in run.stdout, (run.stdout[-2000:], run.stderr[-4000:])
    assert "ERROR: AddressSanitizer" not in run.stderr and "runtime error" not in run.stderr, run.stderr[-4000:]
