"""Feasibility of split-bf16 MFMA for the encoder GEMMs: emulate a_hi/a_mid/a_lo x b_hi/b_mid/b_lo products (fp32 accumulate) in the oracle's
encoder and compare BSQ codes / pre-sign u with the exact fp32 oracle on the fixture inputs."""
import sys, numpy as np, torch
import torch.nn.functional as F
sys.path.insert(0, "/root/repo")
from oracle import sva_oracle as O
from streamvoiceanon_amd import specs, synth_weights
from streamvoiceanon_amd.synth_audio import synth_utterance
torch.set_grad_enabled(False)
W = {k: torch.from_numpy(v) for k, v in synth_weights.generate_all(0, specs.all_specs()).items() if k.startswith("tok.")}

def split(x, n):
    parts, r = [], x
    for _ in range(n):
        p = r.bfloat16().float()
        parts.append(p); r = r - p
    return parts

def make(terms):
    # terms: list of (i, j) index pairs of the operand parts to multiply
    n = 1 + max(max(i, j) for i, j in terms)
    lin0, conv0 = F.linear, F.conv1d
    def lin(x, w, b=None):
        xs, ws = split(x, n), split(w, n)
        y = None
        for i, j in sorted(terms, key=lambda t: -(t[0] + t[1])):      # small terms first
            t = lin0(xs[i], ws[j])
            y = t if y is None else y + t
        return y if b is None else y + b
    def conv(x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
        if groups != 1:
            return conv0(x, w, b, stride, padding, dilation, groups)     # depthwise: VALU kernel, stays fp32
        xs, ws = split(x, n), split(w, n)
        y = None
        for i, j in sorted(terms, key=lambda t: -(t[0] + t[1])):
            t = conv0(xs[i], ws[j], None, stride, padding, dilation, groups)
            y = t if y is None else y + t
        return y if b is None else y + b.view(1, -1, 1)
    return lin, conv, lin0, conv0

x = torch.from_numpy(np.stack([synth_utterance(1000, 262144), synth_utterance(1001, 262144)]))
taps = {}
ref = O.encode_window(x, W, taps=taps)
u_ref = taps["u"].clone()
print("min |u| of the exact run:", float(u_ref.abs().min()))
for name, terms in (("3-term", [(0, 0), (0, 1), (1, 0)]), ("4-term", [(0, 0), (0, 1), (1, 0), (1, 1)]),
                    ("6-term", [(0, 0), (0, 1), (1, 0), (0, 2), (1, 1), (2, 0)])):
    lin, conv, lin0, conv0 = make(terms)
    F.linear, F.conv1d = lin, conv
    try:
        t2 = {}
        got = O.encode_window(x, W, taps=t2)
    finally:
        F.linear, F.conv1d = lin0, conv0
    du = (t2["u"] - u_ref).abs().max()
    print(f"{name}: code mismatches {int((got != ref).sum())} of {ref.numel()}, max |du| {float(du):.3e}")
