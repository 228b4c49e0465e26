"""Sweep of the small-M GEMM kernel's (rows per workgroup, K-split) choices on the encoder's B=1 shapes.
Run on the GPU box:  python tools/gemm_sweep4.py   (prints the heuristic's pick and every override)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [  # (name, M, N, K, mode)   mode bits: 1 GELU, 2 gamma+residual, 8 SwiGLU
    ("tr.qkv", 128, 1536, 512, 0), ("tr.wo", 128, 512, 512, 2), ("tr.w13", 128, 3072, 512, 8), ("tr.w2", 128, 512, 1536, 2),
    ("s2.pw1", 160, 1536, 384, 1), ("s2.pw2", 160, 384, 1536, 2), ("s3.pw1", 160, 2048, 512, 1), ("s3.pw2", 160, 512, 2048, 2),
    ("s1.pw1", 160, 1024, 256, 1), ("s1.pw2", 160, 256, 1024, 2),
    ("st.pw1", 4, 1536, 384, 1), ("st.pw2", 4, 384, 1536, 2),
    ("ar64.w13", 64, 4608, 768, 8), ("ar64.w2", 64, 768, 2304, 2), ("ar128.qkv", 128, 2304, 768, 0),
]
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    from streamvoiceanon_amd import engine as E
    for name, M, N, K, mode in SHAPES:
        us = E.bench_gemm(1, M, N, K, 1, 1, mode, iters=50)
        print(f"{name} {us:.2f}", flush=True)
    sys.exit(0)
res = {}
cfgs = [("auto", {})] + [(f"mt{mt}kw{kw}", {"SVA_SKINNY_MT": str(mt), "SVA_SKINNY_KW": str(kw)}) for mt in (1, 2, 4) for kw in (4, 8, 16)]
for tag, env in cfgs:
    e = dict(os.environ); e.update(env)
    out = subprocess.run([sys.executable, __file__, "child"], env=e, capture_output=True, text=True).stdout
    for line in out.strip().splitlines():
        n, us = line.split()
        res.setdefault(n, {})[tag] = float(us)
print("%-10s" % "shape" + "".join("%9s" % t for t, _ in cfgs))
for name, M, N, K, mode in SHAPES:
    r = res.get(name, {})
    best = min(r.values()) if r else 0
    print("%-10s" % name + "".join(("%8.1f%s" % (r.get(t, float("nan")), "*" if r.get(t) == best else " ")) for t, _ in cfgs))
