"""Stress of gemm_f16w.hip over shapes / epilogues / seeds: float64 reference on the rounded weights + launch-to-launch bit equality.
Run it against the normal library and against the -amdgpu-waitcnt-forcezero build (SVA_LIB_PATH): both must be clean."""
import itertools
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from streamvoiceanon_amd import engine as E  # noqa: E402

bad = n = 0
for M, (N, K), mode, seed in itertools.product((1, 2, 7, 16, 17, 24, 32, 33, 48, 64, 100, 128, 153, 200, 256), ((768, 768), (2304, 768), (4608, 768), (768, 2304), (1000, 768), (8192, 768), (512, 1024)),
                                               ("", "rms", "res", "rms+swiglu", "bias"), (0, 1)):
    if "swiglu" in mode and N % 32:
        continue
    rng = np.random.default_rng(seed * 7919 + M * 31 + N + K)
    A = rng.standard_normal((M, K)).astype(np.float32) * 3.0
    W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    x = A.astype(np.float64)
    kw = {}
    if "rms" in mode:
        kw["rms_w"] = (1.0 + 0.1 * rng.standard_normal(K)).astype(np.float32)
        x = x * kw["rms_w"].astype(np.float64) / np.sqrt((x * x).mean(-1, keepdims=True) + 1e-5)
    ref = x @ W.astype(np.float16).astype(np.float64).T
    if "bias" in mode:
        kw["bias"] = rng.standard_normal(N).astype(np.float32); ref = ref + kw["bias"]
    if "res" in mode:
        kw["res"] = rng.standard_normal((M, N)).astype(np.float32); ref = ref + kw["res"]
    if "swiglu" in mode:
        r3 = ref.reshape(M, N // 32, 2, 16); ref = (r3[:, :, 0] / (1.0 + np.exp(-r3[:, :, 0])) * r3[:, :, 1]).reshape(M, N // 2); kw["swiglu"] = True
    outs = [E.test_gemm_f16w(A, W, **kw)[0] for _ in range(3)]
    err = np.abs(outs[0] - ref).max() / max(np.abs(ref).max(), 1.0)
    same = all(np.array_equal(outs[0], o) for o in outs[1:])
    n += 1
    if err > 2e-6 or not same:
        bad += 1
        print("BAD", (M, N, K, mode, seed), "err", err, "deterministic", same)
print(f"{n} cases, {bad} bad")
