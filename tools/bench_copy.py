#!/usr/bin/env python3
"""Calibration: what a plain streaming copy of the aggregate kernel's byte count achieves on
this GPU (torch's vectorised copy kernel), HBM regime (rotating buffers) vs cache-resident."""
import sys
import torch

N, W = int(sys.argv[1]) if len(sys.argv) > 1 else 73763, int(sys.argv[2]) if len(sys.argv) > 2 else 128
dev = torch.device("cuda:0")
nbuf = 10
a = [torch.rand(N, W, device=dev) for _ in range(nbuf)]
b = [torch.empty(N, W, device=dev) for _ in range(nbuf)]
for rotate in (True, False):
    for i in range(20):
        b[i % nbuf if rotate else 0].copy_(a[i % nbuf if rotate else 0])
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    iters = 300
    s.record()
    for i in range(iters):
        k = i % nbuf if rotate else 0
        b[k].copy_(a[k])
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / iters
    print(f"copy {N}x{W} fp32 rotate={rotate}: {us:.2f} us/launch  {2*4*N*W/us/1e3:.0f} GB/s (read+write)")
# empty-ish kernel cadence: tiny copy
t = torch.zeros(64, device=dev); u = torch.zeros(64, device=dev)
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); s.record()
for i in range(300):
    u.copy_(t)
e.record(); torch.cuda.synchronize()
print(f"tiny kernel cadence: {s.elapsed_time(e)*1e3/300:.2f} us/launch")
