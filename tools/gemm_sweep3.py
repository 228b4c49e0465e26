import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from streamvoiceanon_amd import engine as E
for M in (64, 128):
    for name, N, K, mode in (("wqkv", 2304, 768, 0), ("wo", 768, 768, 2), ("w13", 4608, 768, 8), ("w2", 768, 2304, 2)):
        us = E.bench_gemm(1, M, N, K, 1, 1, mode, iters=30)
        print(f"M={M} {name:5s} N={N} K={K}: {us:8.2f} us  {2.0*M*N*K/us/1e6:7.2f} TF/s", flush=True)
