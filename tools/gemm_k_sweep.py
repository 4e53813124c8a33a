#!/usr/bin/env python3
"""k_linear_dma at whole rounds of tiles (M = 131072, N = 256: 2048 tiles = 4 rounds of 512 resident workgroups) over K:
time per launch, TFLOP/s, and the fit  t = c + a K  (c = what a tile costs beside its k loop: epilogue, pipeline turn-around).
usage: gemm_k_sweep.py [math]"""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from gnnbuilder_amd import runtime  # noqa: E402

runtime.load_library(require_gpu=True)
runtime.set_option("math", int(sys.argv[1]) if len(sys.argv) > 1 else 0)
dev = torch.device("cuda:0")
M, N = 131072, 256
ks, ts = [256, 512, 768, 1024, 1536, 2048], []
for K in ks:
    a = torch.rand(M, K, device=dev) - 0.5
    w = (torch.rand(N, K, device=dev) - 0.5) / K ** 0.5
    b = torch.rand(N, device=dev)
    y = torch.empty(M, N, device=dev)
    runtime.linear_timed(a, w, b, y, "relu", 3)
    us = min(runtime.linear_timed(a, w, b, y, "relu", 20) for _ in range(3))
    ts.append(us)
    print("K %5d: %8.1f us  %6.1f TFLOP/s (%.3f of the fp32 MFMA peak)" % (K, us, 2.0 * M * N * K / us / 1e6, 2.0 * M * N * K / us / 1e6 / 157.3))
A = np.stack([np.ones(len(ks)), np.array(ks, float)], 1)
c, a_ = np.linalg.lstsq(A, np.array(ts), rcond=None)[0]
print("fit: t = %.1f us + %.4f us x K   (in-loop rate %.1f TFLOP/s = %.3f; per tile and workgroup beside the loop: %.2f us)" % (
    c, a_, 2.0 * M * N / a_ / 1e6, 2.0 * M * N / a_ / 1e6 / 157.3, c / 4))
runtime.set_option("math", 0)
