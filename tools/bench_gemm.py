#!/usr/bin/env python3
"""Time gnnb_linear shapes on one GPU (C-side launch loop, HIP events).  python tools/bench_gemm.py ['{"gemm_wlds":0}' ...]"""
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from gnnbuilder_amd import runtime  # noqa: E402

dev = torch.device("cuda:0")
shapes = [(73763, 128, 128), (104448, 128, 128), (147456, 128, 128), (73763, 64, 64), (73763, 128, 64), (73763, 128, 11), (4096, 64, 384),
          (4096, 64, 64), (208896, 256, 512), (208896, 256, 256), (147456, 128, 1664)]
sets = [json.loads(a) for a in sys.argv[1:]] or [{}, {"gemm_wlds": 0}]
for opts in sets:
    for k, v in {"gemm_wlds": 1, "gemm_variant": 0, "gemm_dma": 1, **opts}.items():
        runtime.set_option(k, v)
    for M, N, K in shapes:
        a = torch.rand(M, K, device=dev) - 0.5
        w = (torch.rand(N, K, device=dev) - 0.5) / K ** 0.5
        b = torch.rand(N, device=dev)
        y = torch.empty(M, N, device=dev)
        us = runtime.linear_timed(a, w, b, y, "relu", 50)
        fl = 2.0 * M * N * K
        print(f"{json.dumps(opts):24s} M={M:7d} N={N:4d} K={K:5d}: {us:8.2f} us  {fl / us / 1e6:7.2f} TFLOP/s  "
              f"({fl / us / 1e6 / 157.3 * 100:5.1f}% fp32 MFMA)  {(M * K + M * N) * 4 / us / 1e3:7.0f} GB/s", flush=True)
