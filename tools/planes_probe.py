"""Leave-one-out timing of the planes GEMM's K loop (SVA_DEBUG planes_dbg: results are garbage, times are not): which of global loads /
LDS stores / MFMAs / epilogue a tile's time is made of.   python tools/planes_probe.py M N K mode variant"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from streamvoiceanon_amd import engine as E

M, N, K, mode, variant = [int(x) for x in sys.argv[1:6]]
rng = np.random.default_rng(3)
A = rng.standard_normal((M, K)).astype(np.float32)
W = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
lib = E.load_library()
names = {0: "full", 1: "no global loads", 2: "no LDS stores", 4: "no MFMAs", 8: "no epilogue", 3: "no loads, no stores", 7: "K loop = barriers + fragment reads only", 15: "barriers + fragment reads, no epilogue",
         12: "no MFMAs, no epilogue", 9: "no loads, no epilogue", 6: "no stores, no MFMAs"}
for ap in (False, True):
    for d in (0, 1, 2, 4, 8, 3, 6, 7, 9, 12, 15):
        lib.sva_debug_configure(f"planes_dbg={d}".encode())
        _, us = E.test_gemm_planes(A, W, mode=mode, variant=variant, a_planes=ap, iters=30)
        print(f"M {M} N {N} K {K} mode {mode} variant {variant} a_planes {ap}: {names[d]:45s} {us:8.1f} us", flush=True)
lib.sva_debug_configure(b"planes_dbg=0")
