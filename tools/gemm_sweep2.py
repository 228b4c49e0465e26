"""tiled conv-GEMM microbenchmark at the B=64 head-pass shapes, epilogue variants"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from streamvoiceanon_amd import engine as E
B = 64
for name, T, N, Cin, taps in [("pw1 C=512", 160, 2048, 512, 1), ("pw2 C=512", 160, 512, 2048, 1), ("pw1 C=384", 160, 1536, 384, 1), ("pw2 C=384", 160, 384, 1536, 1),
                              ("pw1 C=256", 160, 1024, 256, 1), ("pw1 C=128", 160, 512, 128, 1), ("tr wqkv", 128, 1536, 512, 1), ("tr w13", 128, 3072, 512, 1),
                              ("big square", 128, 4096, 4096, 1)]:
    for mode, mn in ((0, "bias"), (1, "gelu"), (2, "gamma+res")):
        us = E.bench_gemm(B, T, N, Cin, taps, 1, mode, iters=20)
        fl = 2.0 * B * T * N * Cin * taps
        print(f"{name:12s} {mn:10s} M={B*T:6d} N={N:5d} K={Cin*taps:5d}  {us:9.2f} us  {fl/us/1e6:8.2f} TF/s", flush=True)
