"""Microbenchmark + cross-check of the nucleus-sampler implementations (sva_test_sampler variants 1..5) on cuda:0."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from streamvoiceanon_amd import engine as E

rng = np.random.default_rng(0)
for V, rows in ((8192, 1), (8192, 64), (1000, 1), (1000, 64)):
    L = (rng.standard_normal((rows, V)) * rng.uniform(1.0, 6.0, (rows, 1))).astype(np.float32)
    Q = rng.exponential(1.0, (rows, V)).astype(np.float32) + 1e-9
    ref = None
    for var in (1, 2, 3, 4, 5, 6, 7):
        if V == 8192 and var == 5:
            continue
        tok, us = E.test_sampler(L, Q, var, iters=200)
        if ref is None:
            ref = tok
        print(f"V={V} rows={rows} variant={var}: {us:7.2f} us/launch  equal_to_v1={bool((tok == ref).all())}", flush=True)
