"""fp32 conv-GEMM on the bf16 pipes (gemm_split.hip) against the tuned fp32-MFMA dispatch on the large B = 64 shapes.
   python tools/split_probe.py            (runs itself three times with different environments)"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
shapes = [  # B, T, N, Cin, taps, dil, mode
    (64, 128, 3072, 512, 1, 1, 8), (64, 170, 1536, 384, 1, 1, 1), (64, 170, 384, 1536, 1, 1, 2), (64, 128, 1536, 512, 1, 1, 0),
    (64, 128, 512, 1536, 1, 1, 2), (64, 170, 2048, 512, 1, 1, 1), (64, 170, 512, 2048, 1, 1, 2), (64, 256, 128, 128, 11, 1, 6),
    (64, 32, 256, 256, 11, 1, 6), (64, 512, 64, 64, 11, 1, 6), (1, 170, 1536, 384, 1, 1, 1), (1, 170, 384, 1536, 1, 1, 2),
]
if len(sys.argv) > 1:
    from streamvoiceanon_amd import engine as E
    for s in shapes:
        us = E.bench_gemm(*s, iters=100)
        fl = 2.0 * s[0] * s[1] * s[2] * s[3] * s[4]
        print(f"{sys.argv[1]:>10} {s}: {us:8.2f} us  {fl / us / 1e6:7.1f} TF/s", flush=True)
else:
    for tag, env in (("tuned", {}), ("split128", {"SVA_TUNE_TABLE": "0", "SVA_SPLIT_VARIANT": "0"}), ("split_ws", {"SVA_TUNE_TABLE": "0", "SVA_SPLIT_VARIANT": "4"})):
        subprocess.run([sys.executable, __file__, tag], env=dict(os.environ, **env))
