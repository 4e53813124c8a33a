#!/usr/bin/env python3
"""Launch one gnnb_linear shape a few times (for rocprofv3 --pmc runs)."""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from gnnbuilder_amd import runtime  # noqa: E402
M, N, K = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (73763, 128, 128)))
dev = torch.device("cuda:0")
a = torch.rand(M, K, device=dev) - 0.5
w = (torch.rand(N, K, device=dev) - 0.5) / K ** 0.5
b = torch.rand(N, device=dev)
y = torch.empty(M, N, device=dev)
for _ in range(20):
    runtime.linear([(a, None)], w, b, act="relu", out=y)
torch.cuda.synchronize()
