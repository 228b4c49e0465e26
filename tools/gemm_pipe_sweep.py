"""Per-shape timing of the LDS-DMA ring GEMM variants against the older kernels (run once per SVA_PIPE_VARIANT / SVA_GEMM_PIPE)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from streamvoiceanon_amd import engine as E
shapes = [(1, 160, 1536, 384, 1, 1), (1, 160, 384, 1536, 1, 2), (1, 128, 1536, 512, 1, 0), (1, 128, 512, 1536, 1, 2), (1, 128, 512, 512, 1, 2),
          (1, 256, 128, 128 * 1, 11, 4), (1, 32, 256, 256, 11, 4), (1, 512, 64, 64, 11, 4), (64, 128, 1536, 512, 1, 0), (64, 160, 1536, 384, 1, 1)]
tag = os.environ.get("SVA_PIPE_VARIANT", "h") if os.environ.get("SVA_GEMM_PIPE", "1") != "0" else "old"
print(tag, " ".join(f"{E.bench_gemm(B, T, N, Cin, taps, 1, mode, iters=100):7.1f}" for (B, T, N, Cin, taps, mode) in shapes))
