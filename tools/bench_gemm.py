#!/usr/bin/env python3
"""Time gnnb_linear shapes on one GPU (C-side launch loop, HIP events)."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from gnnbuilder_amd import runtime  # noqa: E402

dev = torch.device("cuda:0")
shapes = [(73763, 128, 128), (73763, 128, 11), (4096, 64, 384), (4096, 64, 64), (4096, 19, 64), (104448, 128, 128),
          (208896, 256, 512), (147456, 128, 1664)]
import itertools
for variant, loader, cap in [(0, 0, 4), (0, 0, 2), (0, 0, 1), (1, 0, 4)]:
    runtime.set_option("gemm_variant", variant)
    runtime.set_option("gemm_max_wg_per_cu", cap)
    for M, N, K in (shapes if variant == 1 or (loader, cap) == (0, 4) else shapes[:2]):
        a = torch.rand(M, K, device=dev) - 0.5
        w = (torch.rand(N, K, device=dev) - 0.5) / K ** 0.5
        b = torch.rand(N, device=dev)
        y = torch.empty(M, N, device=dev)
        us = runtime.linear_timed(a, w, b, y, "relu", 50)
        fl = 2.0 * M * N * K
        print(f"variant {variant} loader {loader} cap {cap}  M={M:7d} N={N:4d} K={K:5d}: {us:8.2f} us  {fl / us / 1e6:7.2f} TFLOP/s  "
              f"({fl / us / 1e6 / 157.3 * 100:5.1f}% fp32 MFMA)  {(M * K + M * N) * 4 / us / 1e3:7.0f} GB/s")
