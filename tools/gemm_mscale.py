import sys, os
sys.path.insert(0, "/root/repo")
from streamvoiceanon_amd import engine as E
for name, N, K, mode in (("wqkv", 2304, 768, 0), ("wo", 768, 768, 2), ("w13", 4608, 768, 8), ("w2", 768, 2304, 2)):
    row = []
    for M in (16, 32, 64, 128, 256):
        us = E.bench_gemm(1, M, N, K, 1, 1, mode, iters=50)
        row.append(f"M={M}: {us:6.1f}us {2.0*M*N*K/us/1e6:5.1f}TF")
    print(name, " | ".join(row), flush=True)
