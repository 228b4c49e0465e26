"""Intercept / slope of the small-M GEMM in K and N (fixed-cost vs streaming analysis).  Run on the GPU box."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from streamvoiceanon_amd import engine as E
for M in (16, 64):
    for N in (768, 2304):
        row = []
        for K in (256, 768, 2304, 4608):
            us = E.bench_gemm(1, M, N, K, 1, 1, 0, iters=50)
            row.append(f"K={K}: {us:6.1f}us")
        print(f"M={M} N={N}", " | ".join(row), flush=True)
