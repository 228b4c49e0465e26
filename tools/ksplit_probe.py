"""Does the grid-level K split (fence-free granule hand-off) pay?  Times the table-free tuned choice with and without split candidates
for the long-K, few-row shapes of the B = 1 step and the M = 64 AR GEMMs:  SVA_KSPLIT=0/1 python tools/ksplit_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SVA_AUTOTUNE"] = "1"; os.environ["SVA_TUNE_TABLE"] = "0"; os.environ["SVA_TUNE_LOG"] = "1"
from streamvoiceanon_amd import engine as E
shapes = [  # B, T, N, Cin, taps, dil, mode
    (1, 170, 384, 1536, 1, 1, 2), (1, 128, 512, 1536, 1, 1, 2), (1, 170, 512, 2048, 1, 1, 2), (1, 170, 256, 1024, 1, 1, 2),
    (1, 32, 256, 256, 11, 1, 4), (1, 256, 128, 128, 11, 1, 4), (1, 512, 64, 64, 11, 1, 4), (1, 4, 512, 512, 13, 1, 0),
    (1, 64, 768, 768, 1, 1, 2), (1, 64, 768, 2304, 1, 1, 2), (1, 64, 2304, 768, 1, 1, 0), (1, 64, 4608, 768, 1, 1, 8),
]
for s in shapes:
    us = E.bench_gemm(*s, iters=200)
    print("RESULT", s, f"{us:.2f} us", flush=True)
