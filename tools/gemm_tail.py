"""Development: k_linear_dma's time against the number of tile rounds (is the partial last round what the kernel loses?)."""
import sys, json
import torch
sys.path.insert(0, ".")
import importlib
rt = importlib.import_module("gnn-builder_amd.runtime")
dev = torch.device("cuda:0")
K, N = int(sys.argv[1]) if len(sys.argv) > 1 else 1664, int(sys.argv[2]) if len(sys.argv) > 2 else 128
for M in [65536, 131072, 139264, 147456, 163840, 196608, 262144]:
    a = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) * 0.05
    b = torch.randn(N, device=dev)
    y = torch.empty(M, N, device=dev)
    rt.linear_timed(a, w, b, y, "relu", 5)
    us = min(rt.linear_timed(a, w, b, y, "relu", 20) for _ in range(3))
    tiles = (M + 127) // 128 * ((N + 127) // 128)
    print(json.dumps(dict(M=M, K=K, N=N, tiles=tiles, rounds=tiles / 512, us=us, tflops=2.0 * M * K * N / us / 1e6, us_per_round=us / (tiles / 512))))
