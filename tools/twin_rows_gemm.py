"""Are a GEMM choice's results independent of the row's position?  A has every row of its upper half repeated in the lower half; the two halves of C must agree bit for bit."""
import sys
import numpy as np
sys.path.insert(0, ".")
from streamvoiceanon_amd import engine as E
rng = np.random.default_rng(3)
for (M, N, K) in ((64, 768, 768), (64, 2304, 768), (64, 768, 2304), (128, 768, 768), (128, 2304, 768), (32, 1000, 768), (256, 512, 1536)):
    h = M // 2
    A = rng.standard_normal((h, K)).astype(np.float32); A = np.concatenate([A, A])
    W = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
    ref = A.astype(np.float64) @ W.astype(np.float64).T
    choices = [(7, mt, kw, nt) for mt in (1, 2, 4) for kw in (4, 8, 16) for nt in (1, 2) if not (kw == 16 and mt * nt > 2)]
    choices += [(0, a, b, 0) for a in (1, 2, 3, 4) for b in (4, 8)] + [(1, a, 0, 0) for a in range(8)] + [(2, a, 0, 0) for a in range(7)] + [(4, a, 0, 0) for a in range(5)]
    for choice in choices:
        try:
            C = E.test_gemm_choice(A, W, choice)
        except Exception as ex:      # noqa: BLE001
            continue
        d = C[:h] != C[h:]
        print(f"{M}x{N}x{K} choice {choice}: twin rows {'identical' if not d.any() else 'DIFFER in %d elements (rows %s)' % (int(d.sum()), np.nonzero(d.any(axis=1))[0][:6].tolist())}; max err vs fp64 {np.abs(C - ref).max():.2e}")
