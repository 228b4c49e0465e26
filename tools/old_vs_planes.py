"""In-loop split kernels as the tuned table picks them (sva_bench_gemm through the dispatcher, no weight planes) against the best planes
variant (H3) on mid-size encoder shapes: where, below the 6144-row threshold, would the planes kernel pay?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from streamvoiceanon_amd import engine as E

rng = np.random.default_rng(5)
for B in (16, 24, 32, 48):
    for (T, N, K, mode, what) in ((170, 1536, 384, 1, "pw1 384"), (170, 384, 1536, 2, "pw2 384"), (170, 2048, 512, 1, "pw1 512"), (170, 512, 2048, 2, "pw2 512"),
                                  (128, 3072, 512, 8, "w13"), (128, 512, 1536, 2, "w2"), (128, 1536, 512, 0, "wqkv"), (128, 512, 512, 2, "wo"),
                                  (170, 1024, 256, 1, "pw1 256"), (170, 256, 1024, 2, "pw2 256")):
        M = B * T
        old = E.bench_gemm(B, T, N, K, mode=mode, iters=30)
        A = rng.standard_normal((M, K)).astype(np.float32)
        W = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
        best = None
        for v in (0, 1, 2, 3, 6, 7):
            if (v in (0, 1, 7) and M < 128) or (v == 6 and M < 256) or (v in (0, 2, 6, 7) and N < 128):
                continue
            _, us = E.test_gemm_planes(A, W, mode=1, variant=v, iters=20)
            if best is None or us < best[0]:
                best = (us, v)
        print(f"streams {B:3d} {what:8s} M {M:5d} N {N:4d} K {K:4d}: dispatcher {old:6.1f} us | planes H3 best {best[0]:6.1f} us (v{best[1]})  ratio {old / best[0]:.2f}", flush=True)
