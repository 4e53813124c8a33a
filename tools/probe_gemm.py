#!/usr/bin/env python3
"""Diagnostic: where k_linear_reg's compute waves spend their cycles (probe build)."""
import ctypes as C
import os
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("GNNB_HIP_LIB", str(ROOT / "gnn-builder_amd" / "libgnnb_hip_probe.so"))
from gnnbuilder_amd import runtime  # noqa: E402

M, N, K = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (73763, 128, 128)))
import json
for k_, v_ in (json.loads(sys.argv[4]) if len(sys.argv) > 4 else {}).items():
    runtime.set_option(k_, v_)
dev = torch.device("cuda:0")
a = torch.rand(M, K, device=dev) - 0.5
w = (torch.rand(N, K, device=dev) - 0.5) / K ** 0.5
b = torch.rand(N, device=dev)
y = torch.empty(M, N, device=dev)
for _ in range(20):
    runtime.linear([(a, None)], w, b, act="relu", out=y)
torch.cuda.synchronize()
lib = runtime.load_library()
n = 8 * 8192
buf = (C.c_ulonglong * n)()
lib.gnnb_probe_read(buf, n)
p = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 8).astype(np.float64)
p = p[p[:, 0] > 0]
wall = (p[:, 1] - p[:, 0]) / 100.0
span = (p[:, 1].max() - p[:, 0].min()) / 100.0
print(f"M={M} N={N} K={K}: workgroups={len(p)} stages/WG={p[:, 6].mean():.2f} (max {p[:, 6].max():.0f})")
print(f"kernel span {span:.2f} us; WG lifetime mean {wall.mean():.2f} us max {wall.max():.2f} us; last start +{(p[:, 0].max() - p[:, 0].min()) / 100:.2f} us")
tot = p[:, 5]
for name, col in (("barrier wait", 2), ("mfma loop", 3), ("epilogue", 4)):
    print(f"  {name:13s} {100 * (p[:, col] / tot).mean():5.1f}% of wave cycles  ({(p[:, col] / p[:, 6]).mean():8.0f} cycles per stage)")
print(f"  clock {np.median(tot / wall):.0f} MHz")
st = np.sort((p[:, 0] - p[:, 0].min()) / 100.0)
print("  start-time deciles (us):", " ".join(f"{st[int(q * (len(st) - 1))]:.1f}" for q in np.linspace(0, 1, 11)))
