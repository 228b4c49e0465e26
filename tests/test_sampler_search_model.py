"""Host model of the sort-free nucleus sampler (csrc/kernels.hip sampler_bisect_kernel): the threshold search over the float bit
pattern of the probabilities, with plain bisection and with the interpolated, key-snapped probes, must select exactly the set the
reference's sort + inclusive cumulative sum + cut selects (modules/dual_ar_stream.py:1099-1132), ties resolved in id order.
CPU only: this checks the algorithm the kernel implements, the kernel itself is checked on the GPU (test_gpu_parity.py)."""
import numpy as np
import pytest
import torch


def _kept_by_sort(logits, top_p):
    """reference rule with the kernels' tie order (descending value, smaller id first)"""
    p = torch.softmax(torch.from_numpy(logits), dim=-1).numpy()
    order = np.lexsort((np.arange(p.size), -p.astype(np.float64)))
    cum = np.cumsum(p[order].astype(np.float64)).astype(np.float32)
    rm = cum > np.float32(top_p)
    rm[0] = False
    kept = np.zeros(p.size, bool)
    kept[order[~rm]] = True
    return kept, p


def _kept_by_search(p, top_p, interpolate):
    key = p.view(np.uint32).astype(np.uint64)
    pd = p.astype(np.float64)
    tp = np.float32(top_p)

    def mass(k):
        return pd[key >= k].sum()

    if not (np.float32(mass(0)) > tp):
        return np.ones(p.size, bool), 1
    lo, hi = 0, int(key.max()) + 2
    f_lo, f_hi, it = 1.0, 0.0, 0
    while hi - lo > 1:
        mid = lo + (hi - lo) // 2
        if interpolate and it % 3 != 2:
            t = min(1.0, max(0.0, (f_lo - float(tp)) / (f_lo - f_hi)))
            mid = min(hi - 1, max(lo + 1, lo + int((hi - lo) * t)))
        it += 1
        fm = mass(mid)
        if np.float32(fm) > tp:
            lo = int(key[key >= mid].min()) if interpolate else mid
            f_lo = fm
        else:
            hi = (int(key[key < mid].max()) + 1) if interpolate else mid
            f_hi = fm
    kb = lo
    base = pd[key > kb].sum()
    ties = np.nonzero(key == kb)[0]
    nk, run = 0, base
    for _ in ties:
        run += float(p[ties[0]])
        if np.float32(run) > tp:
            break
        nk += 1
    if base == 0.0 and nk == 0:
        nk = 1
    kept = key > kb
    kept[ties[:nk]] = True
    return kept, it


@pytest.mark.parametrize("V", [1000, 8192])
def test_threshold_search_selects_the_sorted_nucleus(V):
    rng = np.random.default_rng(V)
    rows = [(rng.standard_normal(V) * s).astype(np.float32) for s in (0.3, 1.0, 2.5, 6.0, 12.0)]
    tied = np.round(rng.standard_normal(V) * 2).astype(np.float32)
    flat = np.full(V, 0.25, np.float32)
    top3 = rows[1].copy(); top3[[5, 17, 911 % V]] = top3.max() + 3
    spread = np.full(V, -200.0, np.float32); spread[40:44] = 0.0
    probes = []
    for lg in rows + [tied, flat, top3, spread]:
        for tp in (0.05, 0.3, 0.7, 0.95, 1.0):
            want, p = _kept_by_sort(lg, tp)
            for interp in (False, True):
                got, it = _kept_by_search(p, tp, interp)
                assert (got == want).all(), (V, tp, interp, int((got != want).sum()))
                probes.append((interp, it))
    plain = np.mean([n for i, n in probes if not i])
    smart = np.mean([n for i, n in probes if i])
    assert smart < 0.6 * plain, (plain, smart)          # the interpolated, snapped search needs far fewer probes
