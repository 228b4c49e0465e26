"""A/B of the planes-DMA kernel's tile -> workgroup map (SVA_DEBUG planes_dbg bit 6): co-resident workgroups on neighbouring tiles or not."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from streamvoiceanon_amd import engine as E
rng = np.random.default_rng(3)
lib = E.load_library()
for (M, N, K, kw) in ((10880, 1536, 384, dict(gelu=True, c_planes=True)), (8192, 3072, 512, dict(swiglu=True, c_planes=True)), (8192, 1536, 512, {}),
                      (10880, 384, 1536, dict(gamma_res=True)), (10880, 2048, 512, dict(gelu=True, c_planes=True)), (10880, 512, 2048, dict(gamma_res=True))):
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
    row = []
    for rep in range(3):
        for d in (0, 64):
            lib.sva_debug_configure(f"planes_dbg={d}".encode())
            row.append((d, E.test_gemm_planes(A, W, mode=1, variant=10, a_planes=True, iters=30, **kw)[1]))
    lib.sva_debug_configure(b"planes_dbg=0")
    a = min(u for d, u in row if d == 0); b = min(u for d, u in row if d == 64)
    print(f"M {M} N {N} K {K}: default map {a:.1f} us, neighbour map {b:.1f} us", flush=True)
