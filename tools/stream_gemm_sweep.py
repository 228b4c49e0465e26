"""Weight-streaming GEMM (csrc/gemm_stream.hip) against the dispatcher's current choice on the decode-sized shapes of the AR chain (64 / 128 rows)
and the single-stream encoder / vocoder shapes, weights cold (rotating copies > MALL) and hot.

  python tools/stream_gemm_sweep.py [--quick] [--shapes ar|enc|voc|all] > gpurun_out/stream_sweep.txt

Per shape: dispatcher time (eager / graph), the stream kernel's best configuration per weight mode, the probes of one configuration
(launch floor, weight loads only, weight + activation loads), and every configuration's line with VERBOSE=1."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from streamvoiceanon_amd import engine as E

# (name, B, T, N, Cin, taps, dil, mode)   mode bits: 1 GELU, 2 residual + gamma, 4 SiLU on load, 8 SwiGLU, 16 fused RMSNorm
AR = [
    ("fast w13   64x4608x768 ", 64, 1, 4608, 768, 1, 1, 8 | 16),
    ("fast wqkv  64x2304x768 ", 64, 1, 2304, 768, 1, 1, 16),
    ("fast w2    64x768x2304 ", 64, 1, 768, 2304, 1, 1, 2),
    ("fast wo    64x768x768  ", 64, 1, 768, 768, 1, 1, 2),
    ("head       64x1000x768 ", 64, 1, 1000, 768, 1, 1, 16),
    ("slow w13  128x4608x768 ", 128, 1, 4608, 768, 1, 1, 8 | 16),
    ("slow wqkv 128x2304x768 ", 128, 1, 2304, 768, 1, 1, 16),
    ("slow w2   128x768x2304 ", 128, 1, 768, 2304, 1, 1, 2),
    ("slow wo   128x768x768  ", 128, 1, 768, 768, 1, 1, 2),
]
ENC = [
    ("pw2   170x384x1536 ", 1, 170, 384, 1536, 1, 1, 2),
    ("pw1   170x1536x384 ", 1, 170, 1536, 384, 1, 1, 1),
    ("w2    128x512x1536 ", 1, 128, 512, 1536, 1, 1, 2),
    ("w13   128x3072x512 ", 1, 128, 3072, 512, 1, 1, 8),
    ("wqkv  128x1536x512 ", 1, 128, 1536, 512, 1, 1, 0),
    ("wo    128x512x512  ", 1, 128, 512, 512, 1, 1, 2),
    ("pw2b  170x512x2048 ", 1, 170, 512, 2048, 1, 1, 2),
    ("pw1b  170x2048x512 ", 1, 170, 2048, 512, 1, 1, 1),
]
# the HiFiGAN levels: K = taps * C with C = 256 / 128 / 64 / 32 and a few taps variants; kept out of the default run (taps need Cin % 16 == 0)
VOC = [
    ("voc k11 C256  32x256x2816", 1, 32, 256, 256, 11, 1, 4),
    ("voc k11 C128 256x128x1408", 1, 256, 128, 128, 11, 3, 4),
    ("voc k7  C256  32x256x1792", 1, 32, 256, 256, 7, 1, 4),
]

CFGS = [(1, 1, 4), (1, 1, 8), (1, 1, 16), (2, 1, 4), (2, 1, 8), (2, 1, 16), (4, 1, 4), (4, 1, 8),
        (1, 2, 4), (1, 2, 8), (1, 2, 16), (2, 2, 4), (2, 2, 8), (4, 2, 4), (4, 2, 8)]
WMODES = {0: "row-major", 2: "packed"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--shapes", default="all")
    ap.add_argument("--iters", type=int, default=40)
    args = ap.parse_args()
    shapes = {"ar": AR, "enc": ENC, "voc": VOC, "all": AR + ENC + VOC}[args.shapes]
    verbose = bool(os.environ.get("VERBOSE"))
    for (name, B, T, N, Cin, taps, dil, mode) in shapes:
        K = taps * Cin
        wbytes = 4.0 * N * K
        nrot_cold = int(max(2, min(256, -(-700e6 // wbytes))))
        M = B * T
        mt_total = (M + 15) // 16
        for temp, nrot in (("cold", nrot_cold), ("hot", 1)):
            if args.quick and temp == "hot":
                continue
            e, gph, _, mx = E.bench_gemm_choice(B, T, N, Cin, -1, taps=taps, dil=dil, mode=mode, nrot=nrot, iters=args.iters)
            print("%s %-4s dispatcher        eager %6.2f us  graph %6.2f us   (W %.1f MB -> %.2f TB/s, %d rotating copies)" %
                  (name, temp, e, gph, wbytes / 1e6, wbytes / gph * 1e-6, nrot), flush=True)
            best = {}
            for (mt, nt, kw) in CFGS:
                if (mode & 8) and nt != 2:
                    continue
                if mt > mt_total or (nt == 2 and N % 32):
                    continue
                for wm in WMODES:
                    try:
                        e, gph, err, mx = E.bench_gemm_choice(B, T, N, Cin, 6, a=mt + 16 * nt, b=kw, c=wm, taps=taps, dil=dil, mode=mode, nrot=nrot, iters=args.iters)
                    except Exception as ex:         # noqa: BLE001
                        print("   stream mt%d nt%d kw%-2d %-12s FAILED %s" % (mt, nt, kw, WMODES[wm], ex), flush=True)
                        continue
                    rel = err / max(mx, 1e-30)
                    flag = "" if rel < 2e-5 else "  MISMATCH rel %.2e" % rel
                    if verbose or flag:
                        print("   stream mt%d nt%d kw%-2d %-12s eager %6.2f  graph %6.2f%s" % (mt, nt, kw, WMODES[wm], e, gph, flag), flush=True)
                    if wm not in best or gph < best[wm][0]:
                        best[wm] = (gph, e, mt, nt, kw, rel)
            for wm, (gph, e, mt, nt, kw, rel) in sorted(best.items()):
                print("%s %-4s stream %-12s eager %6.2f us  graph %6.2f us   mt%d nt%d kw%-2d  (%.2f TB/s, rel err %.1e)" %
                      (name, temp, WMODES[wm], e, gph, mt, nt, kw, wbytes / gph * 1e-6, rel), flush=True)
            if temp == "cold" and best:
                wm = 2
                _, _, mt, nt, kw, _ = best[wm]
                for probe, what in ((1, "launch floor"), (2, "weight loads only"), (3, "weight + activation loads")):
                    try:
                        e, gph, _, _ = E.bench_gemm_choice(B, T, N, Cin, 6, a=mt + 16 * nt, b=kw, c=wm + 16 * probe, taps=taps, dil=dil, mode=mode & ~(16 | 4), nrot=nrot, iters=args.iters)
                        print("%s      probe %-26s eager %6.2f us  graph %6.2f us   (mt%d nt%d kw%d %s)" % (name, what, e, gph, mt, nt, kw, WMODES[wm]), flush=True)
                    except Exception as ex:         # noqa: BLE001
                        print("   probe %d failed: %s" % (probe, ex), flush=True)


if __name__ == "__main__":
    main()
