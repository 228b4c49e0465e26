"""Planes GEMM (csrc/gemm_planes.hip) against fp64 and against the in-loop split kernel (csrc/gemm_split.hip) on the encoder's hot shapes.

  python tools/planes_bench.py            accuracy on ragged shapes (every mode x variant x operand form), then timings

Accuracy is reported as max |C - C64| / max |C64| (fp32-grade modes: ~1e-6 for K ~ 1000; H1: ~1e-3)."""
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from streamvoiceanon_amd import engine as E

rng = np.random.default_rng(7)
MODES = {0: "S6", 1: "H3", 2: "H1"}


def accuracy():
    worst = {}
    for (M, N, K) in ((200, 192, 256), (515, 288, 128), (160, 320, 384), (1030, 132, 96)):
        A = rng.standard_normal((M, K)).astype(np.float32)
        W = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
        bias = rng.standard_normal(N).astype(np.float32)
        ref = A.astype(np.float64) @ W.astype(np.float64).T + bias
        for mode in MODES:
            for variant in range(8):
                if (variant in (0, 1, 4, 7) and M < 128) or (variant in (0, 2, 4, 6, 7) and N < 128) or (variant == 6 and M < 256):
                    continue
                for ap in (False, True):
                    for cp in (False, True):
                        out, _ = E.test_gemm_planes(A, W, bias=bias, mode=mode, variant=variant, a_planes=ap, c_planes=cp)
                        err = np.abs(out - ref).max() / np.abs(ref).max()
                        key = (MODES[mode], ap, cp)
                        worst[key] = max(worst.get(key, 0.0), err)
                        lim = 2e-3 if mode == 2 else (4e-6 if not cp else 4e-6)
                        if err > lim:
                            print("FAIL", M, N, K, MODES[mode], "variant", variant, "a_planes", ap, "c_planes", cp, "err", err)
    for k, v in sorted(worst.items()):
        print("accuracy", k, "max rel err %.3g" % v)
    # epilogues: GELU and SiLU-on-load
    A = rng.standard_normal((300, 256)).astype(np.float32)
    W = (rng.standard_normal((192, 256)) * 0.06).astype(np.float32)
    g64 = A.astype(np.float64) @ W.astype(np.float64).T
    from math import erf
    gel = 0.5 * g64 * (1.0 + np.vectorize(erf)(g64 / np.sqrt(2.0)))
    out, _ = E.test_gemm_planes(A, W, mode=1, variant=3, gelu=True)
    print("gelu epilogue H3 err %.3g" % (np.abs(out - gel).max() / np.abs(gel).max()))
    a64 = A.astype(np.float64)
    sil = (a64 / (1.0 + np.exp(-a64))) @ W.astype(np.float64).T
    out, _ = E.test_gemm_planes(A, W, mode=1, variant=3, silu=True)
    print("silu prologue H3 err %.3g" % (np.abs(out - sil).max() / np.abs(sil).max()))


def timings():
    if os.environ.get("SHAPES"):            # "M,N,K M,N,K ..."
        shapes = [tuple(int(x) for x in t.split(",")) for t in os.environ["SHAPES"].split()]
    else:
      shapes = [(8192, 3072, 512), (10880, 1536, 384), (10880, 384, 1536), (8192, 1536, 512), (8192, 512, 1536), (10880, 2048, 512), (10880, 512, 2048),
              (8192, 512, 512), (10880, 1024, 256), (10880, 256, 1024)]
    for (M, N, K) in shapes:
        A = rng.standard_normal((M, K)).astype(np.float32)
        W = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
        fl = 2.0 * M * N * K
        row = ["M %5d N %4d K %4d" % (M, N, K)]
        for mode in (0, 1, 2):
            best = None
            for variant in (0, 1, 2, 3, 4, 5, 6, 7):
                for ap in (False, True):
                    _, us = E.test_gemm_planes(A, W, mode=mode, variant=variant, a_planes=ap, iters=20)
                    if best is None or us < best[0]:
                        best = (us, variant, ap)
                    if os.environ.get("VERBOSE"):
                        print("   ", MODES[mode], "variant", variant, "a_planes", ap, "%.1f us %.0f TF/s" % (us, fl / us * 1e-6))
            row.append("%s %.1f us %.0f TF/s (v%d%s)" % (MODES[mode], best[0], fl / best[0] * 1e-6, best[1], " A-planes" if best[2] else ""))
        print(" | ".join(row), flush=True)


if __name__ == "__main__":
    if not os.environ.get("SHAPES"):
        accuracy()
    timings()
