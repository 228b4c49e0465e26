"""The batch-scale GEMM shapes of the encoder (64 streams unless given) through the planes GEMM's tile variants: round 4's register-staged
forms (0: 128 x 128 / 4 waves, 6: 256 x 128 / 8 waves, 7: 128 x 128 / 8 waves) against the persistent LDS-DMA forms (9: 128 x 128 with one workgroup per CU and four stages,
10: two workgroups per CU and two stages), both operands as planes.  Prints us per launch and algorithmic TF/s (2 M N K; the matrix pipe does three times that in H3).
    python tools/planes_dma_bench.py [streams ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from streamvoiceanon_amd import engine as E

rng = np.random.default_rng(3)
# (rows per stream, N, K, epilogue): pwconv1 / pwconv2 of the three widths, wqkv, wo, w13, w2
SHAPES = ((170, 1536, 384, "gelu_cp"), (170, 384, 1536, "gamma_res"), (170, 2048, 512, "gelu_cp"), (170, 512, 2048, "gamma_res"),
          (170, 1024, 256, "gelu_cp"), (170, 256, 1024, "gamma_res"), (128, 1536, 512, ""), (128, 512, 512, "gamma_res"),
          (128, 3072, 512, "swiglu_cp"), (128, 512, 1536, "gamma_res"))
streams = [int(a) for a in sys.argv[1:]] or [64]
for B in streams:
    for (T, N, K, epi) in SHAPES:
        M = B * T
        A = rng.standard_normal((M, K)).astype(np.float32)
        W = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
        kw = dict(gelu="gelu" in epi, c_planes="cp" in epi, swiglu="swiglu" in epi, gamma_res="gamma_res" in epi)
        row = []
        for v in (7, 9, 10, 11, 12):
            us = min(E.test_gemm_planes(A, W, mode=1, variant=v, a_planes=True, iters=20, **kw)[1] for _ in range(2))
            row.append((v, us))
        best_old = min(u for v, u in row if v < 8)
        best_new = min(u for v, u in row if v >= 8)
        print(f"B {B:3d} M {M:6d} N {N:5d} K {K:5d} {epi:10s} " + " ".join(f"v{v} {u:7.1f}" for v, u in row) +
              f"  | best old {best_old:7.1f} new {best_new:7.1f} us  x{best_old / best_new:4.2f}  {2e-6 * M * N * K / best_new:6.1f} TF/s", flush=True)
