#!/usr/bin/env python3
"""Randomised check of gnnb_linear's large-K path (k_linear_dma: tail slices / stream-K tail, narrow N, segments, row scalers, skip, activations,
the fp32 / bf16x6 / f16x3 math modes) against a float64 product on sampled rows.   python tools/fuzz_gemm.py [cases] [seed]"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from gnnbuilder_amd import runtime  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda:0")
acts = {"relu": torch.relu, "tanh": torch.tanh, "sigmoid": torch.sigmoid, "none": lambda v: v,
        "gelu": lambda v: torch.nn.functional.gelu(v)}
worst = {0: 0.0, 1: 0.0, 3: 0.0}
for it in range(cases):
    M = int(rng.choice([rng.integers(1, 600), rng.integers(600, 40000), rng.integers(40000, 160000)]))
    N = int(rng.integers(33, 300))                      # (33 .. 64: the 32-column wave tiles)
    nseg = int(rng.integers(1, 5))
    ks = [32 * int(rng.integers(1, 17 if it % 3 == 0 else 9)) for _ in range(nseg)]   # (K >= 1024: the stream-K tail)
    if nseg == 1 and ks[0] <= 128:
        ks[0] = 160                                     # (K <= 128 with one segment takes the register-resident kernels)
    act = str(rng.choice(list(acts)))
    use_skip, use_bias = bool(rng.integers(0, 2)), bool(rng.integers(0, 4))
    g = torch.Generator().manual_seed(it)
    segs_h = [torch.rand(M, k, generator=g) - 0.5 for k in ks]
    rs_h = [(torch.rand(M, generator=g) + 0.5) if rng.integers(0, 3) == 0 else None for _ in ks]
    K = sum(ks)
    w = (torch.rand(N, K, generator=g) - 0.5) / K ** 0.5
    b = torch.rand(N, generator=g) if use_bias else None
    skip = (torch.rand(M, N, generator=g) - 0.5) if use_skip else None
    rows = torch.from_numpy(np.unique(np.concatenate([rng.integers(0, M, 400), np.arange(max(M - 300, 0), M), np.arange(min(M, 300))])))
    cat = torch.cat([(s if r is None else s * r[:, None])[rows] for s, r in zip(segs_h, rs_h)], 1).double()
    ref = cat @ w.double().T
    if b is not None:
        ref = ref + b.double()
    if skip is not None:
        ref = ref + skip[rows].double()
    ref = acts[act](ref)
    segs = [(s.to(dev), None if r is None else r.to(dev)) for s, r in zip(segs_h, rs_h)]
    wd = w.to(dev)
    bd = None if b is None else b.to(dev)
    sd = None if skip is None else skip.to(dev)
    for math in (0, 1, 3):  # (3: f16x3, reduced precision -- its bound is looser)
        runtime.set_option("math", math)
        got = runtime.linear(segs, wd, bd, skip=sd, act=act).cpu()
        err = float((got[rows].double() - ref).abs().max())
        worst[math] = max(worst[math], err)
        if not err < (2e-5 if math == 3 else 5e-6):
            print(f"FAIL case {it} math={math}: M={M} N={N} ks={ks} act={act} skip={use_skip} bias={use_bias}: err {err:.3e}")
            runtime.set_option("math", 0)
            sys.exit(1)
    runtime.set_option("math", 0)
    if it % 10 == 0:
        print(f"case {it}: M={M} N={N} ks={ks} act={act}: ok", flush=True)
print(f"{cases} cases, worst |error| fp32-MFMA {worst[0]:.2e}, bf16x6 {worst[1]:.2e}, f16x3 {worst[3]:.2e}")
