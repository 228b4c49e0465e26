"""Offline tile-variant table of the planes GEMM (csrc/gemm_planes.hip, H3 format) for the encoder's batch-scale shapes: every variant timed
per (streams, shape), the winner written to csrc/planes_table.inc (looked up by planes_variant in gemm.hip; any variant gives the same
bits -- the K loop is the same -- so the table is a speed choice only).     python tools/planes_tune.py > streamvoiceanon_amd/csrc/planes_table.inc"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from streamvoiceanon_amd import engine as E

rng = np.random.default_rng(5)
# (rows per stream, N, K, A handed over as planes)
# round 5: every batch-scale GEMM of the encoder takes its A operand as planes now (the producers write them), so the candidates include the
# persistent LDS-DMA forms 9 / 10 and their loader-wave forms 11 (128 x 128) / 12 (256 x 128); the three transition convs (LayerNorm rows -> conv k1) still arrive as fp32
SHAPES = ((170, 512, 128, True), (170, 128, 512, True), (170, 1024, 256, True), (170, 256, 1024, True), (170, 1536, 384, True), (170, 384, 1536, True),
          (170, 2048, 512, True), (170, 512, 2048, True), (88, 2048, 512, True), (88, 512, 2048, True), (47, 2048, 512, True), (47, 512, 2048, True),
          (128, 1536, 512, True), (128, 512, 512, True),
          (128, 3072, 512, True), (128, 512, 1536, True), (170, 256, 128, False), (170, 384, 256, False), (170, 512, 384, False))
print("// {M, N, K, variant}: measured winner of tools/planes_tune.py (MI355X, H3 format), sorted by (N, K, M)")
rows = []
for B in (12, 16, 20, 24, 32, 40, 48, 64, 96, 128):
    for (T, N, K, ap) in SHAPES:
        M = B * T
        if M < 1024:
            continue
        A = rng.standard_normal((M, K)).astype(np.float32)
        W = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
        best = None
        for v in (0, 1, 2, 3, 6, 7, 9, 10, 11, 12):
            if (v == 6 and M < 256) or (v in (0, 2, 6, 7) and N < 128) or (v >= 8 and (not ap or N % 128)):
                continue
            us = min(E.test_gemm_planes(A, W, mode=1, variant=v, a_planes=ap, iters=15)[1] for _ in range(2))
            if best is None or us < best[0]:
                best = (us, v)
        rows.append((N, K, M, best[1], best[0]))
        print(f"planes_tune: streams {B} M {M} N {N} K {K}: v{best[1]} {best[0]:.1f} us", file=sys.stderr, flush=True)
for N, K, M, v, us in sorted(rows):
    print(f"{{{M}, {N}, {K}, {v}}},")
