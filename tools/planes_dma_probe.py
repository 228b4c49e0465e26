"""Leave-one-out timing of the persistent LDS-DMA planes GEMM (SVA_DEBUG planes_dbg: results are garbage, times are not).
    python tools/planes_dma_probe.py M N K variant"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from streamvoiceanon_amd import engine as E

M, N, K, variant = [int(x) for x in sys.argv[1:5]]
rng = np.random.default_rng(3)
A = rng.standard_normal((M, K)).astype(np.float32)
W = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
lib = E.load_library()
names = {0: "full", 1: "no DMA requests", 4: "no MFMAs", 8: "no epilogue", 16: "contiguous 1 KiB sources", 32: "no fragment reads", 24: "contiguous sources, no epilogue",
         36: "no MFMAs, no fragment reads (DMA + barriers + epilogue)", 44: "DMA + barriers only", 60: "contiguous DMA + barriers only", 5: "no DMA, no MFMAs (fragment reads + barriers + epilogue)",
         9: "no DMA, no epilogue (reads + MFMAs + barriers)", 41: "MFMAs + barriers only", 37: "barriers + epilogue only"}
for d in (0, 1, 4, 8, 16, 32, 24, 36, 44, 60, 5, 9, 41, 37):
    lib.sva_debug_configure(f"planes_dbg={d}".encode())
    _, us = E.test_gemm_planes(A, W, mode=1, variant=variant, a_planes=True, c_planes=True, iters=30)
    print(f"M {M} N {N} K {K} variant {variant}: {names[d]:60s} {us:8.1f} us", flush=True)
lib.sva_debug_configure(b"planes_dbg=0")
