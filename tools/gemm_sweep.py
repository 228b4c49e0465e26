"""conv-GEMM microbenchmark over the shapes of one chunk-step (B streams)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from streamvoiceanon_amd import engine as E

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
shapes = []
# encoder (T=512 positions per stream): (name, T, N, Cin, taps, mode)
for C in (128, 256, 384, 512):
    shapes += [(f"enc pw1 C={C}", 512, 4 * C, C, 1, 1), (f"enc pw2 C={C}", 512, C, 4 * C, 1, 2)]
shapes += [("enc mel", 512, 160, 1088, 1, 0), ("enc stem k7", 512, 128, 160, 7, 0), ("enc ds k2", 256, 512, 512, 2, 0),
           ("tr wqkv", 128, 1536, 512, 1, 0), ("tr wo", 128, 512, 512, 1, 2), ("tr w13", 128, 3072, 512, 1, 8), ("tr w2", 128, 512, 1536, 1, 2)]
# AR decode (M = 2B slow / B fast)
shapes += [("ar wqkv (2 tok)", 2, 2304, 768, 1, 0), ("ar wo", 2, 768, 768, 1, 2), ("ar w13", 2, 4608, 768, 1, 8), ("ar w2", 2, 768, 2304, 1, 2),
           ("ar out 8192", 1, 8192, 768, 1, 0), ("ar fast_out", 1, 1000, 768, 1, 0)]
# vocoder per code frame
shapes += [("voc up0", 1, 1024, 512, 1, 0), ("voc cnx pw1", 4, 2048, 512, 1, 1), ("voc conv_pre k13", 4, 512, 512, 13, 0),
           ("voc ups0", 4, 2048, 512, 2, 4), ("voc res0 k11", 32, 256, 256, 11, 4), ("voc ups1", 32, 1024, 256, 2, 4),
           ("voc res1 k11", 256, 128, 128, 11, 4), ("voc res1 k3", 256, 128, 128, 3, 4), ("voc ups2", 256, 128, 128, 2, 4),
           ("voc res2 k11", 512, 64, 64, 11, 4), ("voc ups3", 512, 64, 64, 2, 4), ("voc res3 k11", 1024, 32, 32, 11, 4),
           ("voc res4 k11", 2048, 16, 16, 11, 4), ("voc res4 k3", 2048, 16, 16, 3, 4)]
tot = 0
for name, T, N, Cin, taps, mode in shapes:
    us = E.bench_gemm(B, T, N, Cin, taps, 1, mode, iters=30)
    fl = 2.0 * B * T * N * Cin * taps
    print(f"{name:22s} M={B*T:6d} N={N:5d} K={Cin*taps:5d}  {us:9.2f} us  {fl/us/1e6:8.2f} TF/s", flush=True)
