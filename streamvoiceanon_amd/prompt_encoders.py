"""Host side of the prompt path's two speaker-embedding encoders (SURVEY.md 8f N1 iii / iv) on the HIP engine:

  * ``StyleEncoder``  -- ``InferenceWrapper.calculate_style_vec`` (evaluations/infer_arvc.py:179-211): Kaldi fbank (80 bins,
    16 kHz, mean-subtracted) -> CAM++ (modules/campplus/DTDNN.py:50-137, layers.py) -> style vector [1, 192]
  * ``TimbreEncoder`` -- ``calculate_timbre_latent`` (:213-223) = ``SpeakerEncoder.tokenize_wav``
    (modules/bicodec_speaker_encoder/speaker_encoder.py:136-144): MelSpectrogram -> ECAPA-TDNN latent -> PerceiverResampler
    (32 latents) -> FSQ [4]^6 -> timbre latents [1, 32, 128]

This module owns the topology, the (folded) weights and the activation buffers; all arithmetic runs in libsva_hip.so through
the `sva_op_*` primitives of include/sva.h (csrc/prompt_ops.hip) on device arrays -- there is no CPU path.  Eval-mode BatchNorm
is folded into per-channel scale / shift at load time; Conv1d weights are re-laid out tap-major ([Cout][k][Cin]) for the
engine's conv-GEMM.  Once per utterance, batch 1 (the reference's call sites are batch 1 too).
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from . import engine as E

BN_EPS = 1e-5


def _np(x):
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    return np.ascontiguousarray(np.asarray(x), dtype=np.float32)


class _Plan:
    """The device arrays of one encoder call for ONE input length, kept between calls, and the hipGraph of the call's op sequence.
    Call 1 of a length records the allocations (and runs eagerly), call 2 replays them under stream capture (sva_ops_capture_*),
    later calls are: upload the wav, one graph launch, download the result."""

    def __init__(self):
        self.addrs, self.graph, self.out = [], None, None
        self.put_sums = []          # checksum of every constant uploaded while recording: a replay must present the same payloads


class _Dev:
    """Device arrays of one encoder call (freed together) + the persistent weight arrays of an encoder.
    plan + mode "record": allocations are real and remembered in the plan (not freed with the call); "replay": the i-th allocation returns
    the plan's i-th array and constants are NOT uploaded again (they are functions of the input LENGTH only and still there)."""

    def __init__(self, engine: E.Engine, plan: _Plan = None, mode: str = "eager"):
        self.engine, self.lib, self.h = engine, engine.lib, engine.h
        self.ptrs = []
        self.plan, self.mode, self.cursor, self.puts = plan, mode, 0, 0

    def alloc(self, n: int) -> int:
        if self.mode == "replay":
            addr = self.plan.addrs[self.cursor]
            self.cursor += 1
            return addr
        p = C.POINTER(C.c_float)()
        E._check(self.lib.sva_dev_alloc(self.h, int(n), C.byref(p)), "sva_dev_alloc")
        addr = C.cast(p, C.c_void_p).value
        (self.plan.addrs if self.mode == "record" else self.ptrs).append(addr)
        return addr

    def put(self, arr, is_input: bool = False) -> int:
        a = _np(arr).reshape(-1)
        addr = self.alloc(a.size)
        if self.mode != "replay":
            E._check(self.lib.sva_dev_upload(self.h, C.c_void_p(addr), E._ptr(a), a.size), "sva_dev_upload")
        if self.plan is not None and not is_input:      # (the call's input array is uploaded per call by _call / prepare)
            # constants are functions of the input LENGTH only, which is what lets a replay skip their upload; one that depended on the input DATA
            # would go stale silently -- so the replay checks every payload against what the recording uploaded
            import zlib

            key = (a.size, zlib.crc32(a.tobytes()))
            if self.mode == "record":
                self.plan.put_sums.append(key)
            elif self.puts >= len(self.plan.put_sums) or self.plan.put_sums[self.puts] != key:
                raise RuntimeError("prompt encoder replay: a constant differs from the recorded call (data-dependent put(): use alloc + upload per call)")
            self.puts += 1
        return addr

    def upload(self, addr: int, arr):
        a = _np(arr).reshape(-1)
        E._check(self.lib.sva_dev_upload(self.h, C.c_void_p(addr), E._ptr(a), a.size), "sva_dev_upload")

    def get(self, addr: int, shape) -> np.ndarray:
        out = np.empty(shape, np.float32)
        E._check(self.lib.sva_dev_download(self.h, E._ptr(out), C.c_void_p(addr), out.size), "sva_dev_download")
        return out

    def free(self):
        for a in self.ptrs:
            self.lib.sva_dev_free(self.h, C.c_void_p(a))
        self.ptrs = []


def _declare(lib):
    if getattr(lib, "_prompt_ops_declared", False):
        return
    vp, i32, lng, f32 = C.c_void_p, C.c_int, C.c_long, C.c_float
    lib.sva_dev_alloc.argtypes = [vp, lng, C.POINTER(C.POINTER(C.c_float))]
    lib.sva_dev_free.argtypes = [vp, vp]
    lib.sva_dev_upload.argtypes = [vp, vp, vp, lng]
    lib.sva_dev_download.argtypes = [vp, vp, vp, lng]
    lib.sva_op_conv.argtypes = [vp, vp, lng, i32, i32, i32, i32, i32, vp, vp, i32, vp, lng]
    lib.sva_op_affine.argtypes = [vp, vp, lng, i32, i32, vp, vp, i32, vp, lng]
    lib.sva_op_unary.argtypes = [vp, vp, lng, i32, f32]
    lib.sva_op_colstats.argtypes = [vp, vp, lng, i32, i32, vp, vp, i32]
    lib.sva_op_cam_context.argtypes = [vp, vp, lng, i32, i32, i32, vp, vp, lng]
    lib.sva_op_mul.argtypes = [vp, vp, lng, vp, lng, i32, i32, i32]
    lib.sva_op_add.argtypes = [vp, vp, lng, vp, lng, i32, i32]
    lib.sva_op_conv2d.argtypes = [vp, vp, i32, i32, i32, vp, i32, i32, i32, vp, vp, vp, i32, vp]
    lib.sva_op_cf_to_rows.argtypes = [vp, vp, i32, i32, vp, lng]
    lib.sva_op_fbank_power.argtypes = [vp, vp, lng, vp, vp, i32, C.POINTER(i32)]
    lib.sva_op_stft_mag.argtypes = [vp, vp, lng, i32, i32, i32, vp, vp, i32, C.POINTER(i32)]
    lib.sva_op_attention.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp, vp]
    lib.sva_op_geglu.argtypes = [vp, vp, lng, i32, i32, vp, lng]
    lib.sva_op_l2norm.argtypes = [vp, vp, i32, i32, vp, f32, vp]
    lib.sva_ops_capture_begin.argtypes = [vp]
    lib.sva_ops_capture_end.argtypes = [vp, C.POINTER(vp)]
    lib.sva_ops_graph_launch.argtypes = [vp, vp]
    lib.sva_ops_graph_free.argtypes = [vp, vp]
    lib._prompt_ops_declared = True


PROMPT_OP_SYMBOLS = ["sva_dev_alloc", "sva_dev_free", "sva_dev_upload", "sva_dev_download", "sva_op_conv", "sva_op_affine", "sva_op_unary",
                     "sva_op_colstats", "sva_op_cam_context", "sva_op_mul", "sva_op_add", "sva_op_conv2d", "sva_op_cf_to_rows", "sva_op_fbank_power",
                     "sva_op_stft_mag", "sva_op_attention", "sva_op_geglu", "sva_op_l2norm", "sva_ops_capture_begin", "sva_ops_capture_end",
                     "sva_ops_graph_launch", "sva_ops_graph_free"]


def kaldi_mel_banks(num_bins=80, padded=512, sr=16000.0, low=20.0, high=0.0) -> np.ndarray:
    """torchaudio.compliance.kaldi.get_mel_banks (vtln_warp = 1), float32 arithmetic like torchaudio: [num_bins, padded/2 + 1]"""
    import torch

    nyq = 0.5 * sr
    if high <= 0.0:
        high += nyq
    nfb, bw = padded // 2, sr / padded
    mel = lambda f: 1127.0 * math.log(1.0 + f / 700.0)          # noqa: E731
    mlo, mhi = mel(low), mel(high)
    delta = (mhi - mlo) / (num_bins + 1)
    b = torch.arange(num_bins).unsqueeze(1)
    left, center, right = mlo + b * delta, mlo + (b + 1.0) * delta, mlo + (b + 2.0) * delta
    m = (1127.0 * (1.0 + bw * torch.arange(nfb) / 700.0).log()).unsqueeze(0)
    banks = torch.max(torch.zeros(1), torch.min((m - left) / (center - left), (right - m) / (right - center)))
    return torch.nn.functional.pad(banks, (0, 1)).numpy().astype(np.float32)


class _Net:
    def __init__(self, engine: E.Engine, weights: dict, prefix: str):
        _declare(engine.lib)
        self.engine, self.lib, self.h = engine, engine.lib, engine.h
        self.wd = _Dev(engine)                       # persistent: weights
        self.W = {k[len(prefix):]: _np(v) for k, v in weights.items() if k.startswith(prefix)}
        self.dev = {}
        self.plans = {}                              # input length -> _Plan (insertion-ordered: the oldest is evicted)
        self.use_graphs = True                       # False: every call allocates, runs op by op and frees (the round-3 behaviour; tests A/B)
        self.max_plans = 4

    # ---- one call: eager / recording / captured / replayed (see _Plan) ---------------------------------------------
    def _call(self, wav) -> np.ndarray:
        with self.engine.ops_lock:          # every encoder of an engine records / captures / replays on the engine's one ops stream
            return self._call_locked(wav)

    def _call_locked(self, wav) -> np.ndarray:
        n = wav.shape[0]
        if not self.use_graphs:
            d = _Dev(self.engine)
            try:
                addr, shape = self._run(d, wav)
                return d.get(addr, shape)
            finally:
                d.free()
        plan = self.plans.get(n)
        if plan is None:
            while len(self.plans) >= self.max_plans:
                self._drop(next(iter(self.plans)))
            plan = _Plan()
            d = _Dev(self.engine, plan, "record")
            try:
                addr, shape = self._run(d, wav)
                out = d.get(addr, shape)
            except Exception:
                self._free_plan(plan)
                raise
            self.plans[n] = plan
            return out
        d = _Dev(self.engine, plan, "replay")
        d.upload(plan.addrs[0], wav)                 # (the input is the call's first array)
        if plan.graph is None:
            E._check(self.lib.sva_ops_capture_begin(self.h), "sva_ops_capture_begin")
            g = C.c_void_p()
            try:
                plan.out = self._run(d, wav)
            except Exception:
                # the capture must be ended whatever happened; its half-recorded graph is freed and the plan dropped (the next call records anew)
                self.lib.sva_ops_capture_end(self.h, C.byref(g))
                if g.value:
                    self.lib.sva_ops_graph_free(self.h, g)
                self.plans.pop(n, None)
                self._free_plan(plan)
                raise
            E._check(self.lib.sva_ops_capture_end(self.h, C.byref(g)), "sva_ops_capture_end")
            plan.graph = g
        E._check(self.lib.sva_ops_graph_launch(self.h, plan.graph), "sva_ops_graph_launch")
        return d.get(*plan.out)

    # ---- split call for overlap with other device work (InferenceWrapper.calculate_prompt): prepare() uploads the input (synchronises the
    # ops stream, so do it for EVERY encoder before the first launch()), launch() enqueues the graph and returns at once, finish() downloads.
    # Lengths without a captured graph yet (the first two calls) compute synchronously in prepare().
    def prepare(self, wave16k):
        wav = _np(wave16k).reshape(-1)
        plan = self.plans.get(wav.shape[0]) if self.use_graphs else None
        if plan is None or plan.graph is None:
            return ("done", self._call(wav))
        _Dev(self.engine, plan, "replay").upload(plan.addrs[0], wav)
        return ("ready", plan)

    def launch(self, handle):
        if handle[0] == "ready":
            E._check(self.lib.sva_ops_graph_launch(self.h, handle[1].graph), "sva_ops_graph_launch")
        return handle

    def finish(self, handle) -> np.ndarray:
        if handle[0] == "done":
            return handle[1]
        return _Dev(self.engine, handle[1], "replay").get(*handle[1].out)

    def _free_plan(self, plan):
        if plan.graph is not None:
            self.lib.sva_ops_graph_free(self.h, plan.graph)
        for a in plan.addrs:
            self.lib.sva_dev_free(self.h, C.c_void_p(a))
        plan.addrs, plan.graph = [], None

    def _drop(self, n):
        self._free_plan(self.plans.pop(n))

    # ---- weight preparation -----------------------------------------------------------------------------------
    def w_conv(self, key, pad_k_to=None):
        """Conv1d [Cout][Cin][k] / Linear [N][K] -> device [N][k][Cin] (tap-major); optional zero-padding of Cin"""
        if key in self.dev:
            return self.dev[key]
        w = self.W[key]
        if w.ndim == 2:
            w = w[:, :, None]
        w = np.transpose(w, (0, 2, 1))               # [N][k][Cin]
        if pad_k_to and w.shape[2] < pad_k_to:
            w = np.concatenate([w, np.zeros((w.shape[0], w.shape[1], pad_k_to - w.shape[2]), np.float32)], axis=2)
        self.dev[key] = self.wd.put(w)
        return self.dev[key]

    def w_raw(self, key):
        if key not in self.dev:
            self.dev[key] = self.wd.put(self.W[key])
        return self.dev[key]

    def bn(self, p, affine=True):
        """eval-mode BatchNorm folded to (scale, shift) device vectors"""
        key = p + "#bn"
        if key not in self.dev:
            inv = 1.0 / np.sqrt(self.W[p + "running_var"].astype(np.float64) + BN_EPS)
            g = self.W[p + "weight"].astype(np.float64) if affine else 1.0
            b = self.W[p + "bias"].astype(np.float64) if affine else 0.0
            scale = g * inv
            shift = b - self.W[p + "running_mean"].astype(np.float64) * scale
            self.dev[key] = (self.wd.put(scale.astype(np.float32)), self.wd.put(shift.astype(np.float32)))
        return self.dev[key]

    # ---- ops ---------------------------------------------------------------------------------------------------
    def conv(self, x, ldx, T, stride, dil, taps, Cin, w, bias, N, y, ldy):
        E._check(self.lib.sva_op_conv(self.h, x, ldx, T, stride, dil, taps, Cin, w, bias, N, y, ldy), "sva_op_conv")

    def affine(self, x, ldx, T, Cc, scale, shift, relu_mode, y, ldy):
        E._check(self.lib.sva_op_affine(self.h, x, ldx, T, Cc, scale, shift, relu_mode, y, ldy), "sva_op_affine")

    def close(self):
        for n in list(self.plans):
            self._drop(n)
        self.wd.free()


F4 = 4      # bytes per float (device pointer arithmetic is done on integers here)


class StyleEncoder(_Net):
    """CAM++ style vector on the device: wav (16 kHz float array) -> numpy [1, 192]."""

    def __init__(self, engine: E.Engine, weights: dict):
        super().__init__(engine, weights, "style.")
        banks = np.zeros((80, 272), np.float32)                  # K padded 257 -> 272 (multiple of 16) for the conv-GEMM
        banks[:, :257] = kaldi_mel_banks()
        self.banks = self.wd.put(banks)

    def __call__(self, wave16k) -> np.ndarray:
        return self._call(_np(wave16k).reshape(-1))

    def _run(self, d, wav):
        lib, h = self.lib, self.h
        n = wav.shape[0]
        assert n >= 400, "style encoder: reference audio shorter than one 25 ms frame"
        m = 1 + (n - 400) // 160
        dw = d.put(wav, is_input=True)
        frames, spec, feat = d.alloc(m * 512), d.alloc(m * 272), d.alloc(m * 80)
        mo = C.c_int()
        E._check(lib.sva_op_fbank_power(h, dw, n, frames, spec, 272, C.byref(mo)), "sva_op_fbank_power")
        assert mo.value == m
        self.conv(spec, 272, m, 1, 1, 1, 272, self.banks, None, 80, feat, 80)
        E._check(lib.sva_op_unary(h, feat, m * 80, 1, float(np.finfo(np.float32).eps)), "log")
        mean = d.alloc(80)
        E._check(lib.sva_op_colstats(h, feat, 80, m, 80, mean, None, 0), "colstats")
        E._check(lib.sva_op_unary(h, mean, 80, 4, 0.0), "negate")                       # feat - feat.mean(0)   (evaluations/infer_arvc.py:192), on the device:
        self.affine(feat, 80, m, 80, None, mean, 0, feat, 80)                             # no host round trip inside the op sequence (it is captured as a graph)
        # ---- FCM head (DTDNN.py:14-48), channel-first [C][F][T]; input x[0][f][t] = feat[t][f]
        T = m
        x0 = d.alloc(80 * T)
        E._check(lib.sva_op_cf_to_rows(h, feat, T, 80, x0, T), "transpose")      # rows [80][T] <- [T][80] read as [CF = T][T' = 80]
        hp = "head."

        def c2d(x, cin, Fq, wkey, bnp, k, sf, res, relu):
            Fo = (Fq + 2 * (k // 2) - k) // sf + 1
            y = d.alloc(32 * Fo * T)
            sc, sh = self.bn(bnp)
            E._check(lib.sva_op_conv2d(h, x, cin, Fq, T, self.w_raw(wkey), 32, k, sf, sc, sh, res, relu, y), "sva_op_conv2d")
            return y, Fo

        x, Fq = c2d(x0, 1, 80, hp + "conv1.weight", hp + "bn1.", 3, 1, None, 1)
        for layer in ("layer1.", "layer2."):
            for bi in range(2):
                q = hp + layer + f"{bi}."
                sf = 2 if bi == 0 else 1
                o1, Fo = c2d(x, 32, Fq, q + "conv1.weight", q + "bn1.", 3, sf, None, 1)
                sc = x
                if bi == 0:
                    sc, _ = c2d(x, 32, Fq, q + "shortcut.0.weight", q + "shortcut.1.", 1, sf, None, 0)
                x, Fq = c2d(o1, 32, Fo, q + "conv2.weight", q + "bn2.", 3, 1, sc, 1)
        x, Fq = c2d(x, 32, Fq, hp + "conv2.weight", hp + "bn2.", 3, 2, None, 1)
        CF = 32 * Fq                                              # 320
        # ---- xvector TDNN (stride 2, k 5, padding 2): rows with 2 zero rows on either side
        xr = d.alloc((T + 4) * CF)
        E._check(lib.sva_op_cf_to_rows(h, x, CF, T, xr + 2 * CF * F4, CF), "cf_to_rows")
        T2 = (T + 4 - 5) // 2 + 1
        xv = "xvector."
        CM = 1024                                                 # widest dense block (512 + 16 * 32)
        buf = d.alloc(T2 * CM)
        self.conv(xr, CF, T2, 2, 1, 5, CF, self.w_conv(xv + "tdnn.linear.weight"), None, 128, buf, CM)
        sc, sh = self.bn(xv + "tdnn.nonlinear.batchnorm.")
        self.affine(buf, CM, T2, 128, sc, sh, 1, buf, CM)
        ch = 128
        hn = d.alloc(T2 * CM)                                     # BN-ReLU of the dense input
        bott = d.alloc((T2 + 4) * 128)                            # bottleneck rows with dilation padding (<= 2) on either side
        ctx, ctx1, ctx2, mean = d.alloc(T2 * 128), d.alloc(T2 * 64), d.alloc(T2 * 32), d.alloc(128)
        for bi, (nl, dil) in enumerate(((12, 1), (24, 2), (16, 2))):
            for li in range(nl):
                q = xv + f"block{bi + 1}.tdnnd{li + 1}."
                cin = ch + 32 * li
                sc, sh = self.bn(q + "nonlinear1.batchnorm.")
                self.affine(buf, CM, T2, cin, sc, sh, 1, hn, CM)
                yb = bott + 2 * 128 * F4                         # live rows
                self.conv(hn, CM, T2, 1, 1, 1, cin, self.w_conv(q + "linear1.weight"), None, 128, yb, 128)
                sc, sh = self.bn(q + "nonlinear2.batchnorm.")
                self.affine(yb, 128, T2, 128, sc, sh, 1, yb, 128)
                out = buf + cin * F4                              # the layer's 32 channels are appended to the dense buffer
                self.conv(yb - dil * 128 * F4, 128, T2, 1, dil, 3, 128, self.w_conv(q + "cam_layer.linear_local.weight"), None, 32, out, CM)
                E._check(lib.sva_op_colstats(h, yb, 128, T2, 128, mean, None, 0), "colstats")
                E._check(lib.sva_op_cam_context(h, yb, 128, T2, 128, 100, mean, ctx, 128), "cam_context")
                self.conv(ctx, 128, T2, 1, 1, 1, 128, self.w_conv(q + "cam_layer.linear1.weight"), self.w_raw(q + "cam_layer.linear1.bias"), 64, ctx1, 64)
                self.affine(ctx1, 64, T2, 64, None, None, 1, ctx1, 64)
                self.conv(ctx1, 64, T2, 1, 1, 1, 64, self.w_conv(q + "cam_layer.linear2.weight"), self.w_raw(q + "cam_layer.linear2.bias"), 32, ctx2, 32)
                E._check(lib.sva_op_mul(h, out, CM, ctx2, 32, T2, 32, 1), "mul_sigmoid")
            ch += 32 * nl
            q = xv + f"transit{bi + 1}."
            sc, sh = self.bn(q + "nonlinear.batchnorm.")
            self.affine(buf, CM, T2, ch, sc, sh, 1, hn, CM)
            self.conv(hn, CM, T2, 1, 1, 1, ch, self.w_conv(q + "linear.weight"), None, ch // 2, buf, CM)
            ch //= 2
        sc, sh = self.bn(xv + "out_nonlinear.batchnorm.")
        self.affine(buf, CM, T2, ch, sc, sh, 1, buf, CM)
        # masked statistics pooling over the first feat_len // 2 frames (layers.py:33-43, infer_arvc.py:196-201), unbiased std
        Tl = min(T2, m // 2)
        stats = d.alloc(2 * ch)
        E._check(lib.sva_op_colstats(h, buf, CM, Tl, ch, stats, stats + ch * F4, 1), "stats pooling")
        emb = d.alloc(192)
        self.conv(stats, 2 * ch, 1, 1, 1, 1, 2 * ch, self.w_conv("dense.linear.weight"), None, 192, emb, 192)
        sc, sh = self.bn("dense.nonlinear.batchnorm.", affine=False)
        self.affine(emb, 192, 1, 192, sc, sh, 0, emb, 192)
        return emb, (1, 192)


class TimbreEncoder(_Net):
    """SparkTTS timbre latents on the device: wav (16 kHz float array) -> numpy [1, 32, 128]."""

    def __init__(self, engine: E.Engine, weights: dict):
        super().__init__(engine, weights, "timbre.")
        fb = E.slaney_mel_fb(n_freqs=513, f_min=10.0, f_max=8000.0, n_mels=128, sample_rate=16000)     # [513, 128]
        w = np.zeros((128, 528), np.float32)                     # K padded 513 -> 528
        w[:, :513] = fb.T
        self.fb = self.wd.put(w)

    def __call__(self, wave16k) -> np.ndarray:
        return self._call(_np(wave16k).reshape(-1))

    def _crb(self, d, x, ldx, T, taps, dil, Cin, p, N, y, ldy):
        """Conv1dReluBn (ecapa_tdnn.py:68-85): bn(relu(conv(x)))"""
        self.conv(x, ldx, T, 1, dil, taps, Cin, self.w_conv(p + "conv.weight"), self.w_raw(p + "conv.bias"), N, y, ldy)
        sc, sh = self.bn(p + "bn.")
        self.affine(y, ldy, T, N, sc, sh, 2, y, ldy)

    def _run(self, d, wav):
        lib, h = self.lib, self.h
        n = wav.shape[0]
        T = 1 + n // 320
        dw = d.put(wav, is_input=True)
        frames, spec = d.alloc(T * 1024), d.alloc(T * 528)
        mo = C.c_int()
        E._check(lib.sva_op_stft_mag(h, dw, n, 1024, 640, 320, frames, spec, 528, C.byref(mo)), "sva_op_stft_mag")
        assert mo.value == T
        PAD = 4                                                   # rows of zero padding on either side (k 5 -> 2, k 3 dil 4 -> 4)
        mel = d.alloc((T + 2 * PAD) * 128)
        self.conv(spec, 528, T, 1, 1, 1, 528, self.fb, None, 128, mel + PAD * 128 * F4, 128)
        se = "speaker_encoder."
        o1 = d.alloc(T * 512)
        self._crb(d, mel + (PAD - 2) * 128 * F4, 128, T, 5, 1, 128, se + "layer1.", 512, o1, 512)
        cat = d.alloc(T * 1536)                                   # [out2 | out3 | out4]
        y0 = d.alloc(T * 512)
        sp = d.alloc((T + 2 * PAD) * 64)                          # padded input rows of one Res2 branch conv
        r2 = d.alloc(T * 512)
        semean, se1, se2 = d.alloc(512), d.alloc(128), d.alloc(512)
        xin, ldin = o1, 512
        for li, dil in ((2, 2), (3, 3), (4, 4)):
            q = se + f"layer{li}.se_res2block."
            self._crb(d, xin, ldin, T, 1, 1, 512, q + "0.", 512, y0, 512)
            spl = sp + PAD * 64 * F4
            for i in range(7):                                    # Res2Conv1dReluBn (ecapa_tdnn.py:12-61), width 64, scale 8
                self.affine(y0 + i * 64 * F4, 512, T, 64, None, None, 0, spl, 64)        # sp = spx[i]
                if i >= 1:
                    E._check(lib.sva_op_add(h, spl, 64, r2 + (i - 1) * 64 * F4, 512, T, 64), "add")   # + previous branch output
                o = r2 + i * 64 * F4
                self.conv(spl - dil * 64 * F4, 64, T, 1, dil, 3, 64, self.w_conv(q + f"1.convs.{i}.weight"), self.w_raw(q + f"1.convs.{i}.bias"), 64, o, 512)
                sc, sh = self.bn(q + f"1.bns.{i}.")
                self.affine(o, 512, T, 64, sc, sh, 2, o, 512)
            self.affine(y0 + 7 * 64 * F4, 512, T, 64, None, None, 0, r2 + 7 * 64 * F4, 512)
            out = cat + (li - 2) * 512 * F4
            self._crb(d, r2, 512, T, 1, 1, 512, q + "2.", 512, out, 1536)
            # SE_Connect (ecapa_tdnn.py:93-108) + residual
            E._check(lib.sva_op_colstats(h, out, 1536, T, 512, semean, None, 0), "se mean")
            self.conv(semean, 512, 1, 1, 1, 1, 512, self.w_conv(q + "3.linear1.weight"), self.w_raw(q + "3.linear1.bias"), 128, se1, 128)
            self.affine(se1, 128, 1, 128, None, None, 1, se1, 128)
            self.conv(se1, 128, 1, 1, 1, 1, 128, self.w_conv(q + "3.linear2.weight"), self.w_raw(q + "3.linear2.bias"), 512, se2, 512)
            E._check(lib.sva_op_mul(h, out, 1536, se2, 0, T, 512, 1), "se scale")
            E._check(lib.sva_op_add(h, out, 1536, xin, ldin, T, 512), "residual")
            xin, ldin = out, 1536
        lat = d.alloc(T * 1536)
        self.conv(cat, 1536, T, 1, 1, 1, 1536, self.w_conv(se + "conv.weight"), self.w_raw(se + "conv.bias"), 1536, lat, 1536)
        self.affine(lat, 1536, T, 1536, None, None, 1, lat, 1536)
        # ---- PerceiverResampler (perceiver_encoder.py:287-341): keys = [32 latents | context frames], the first 32 + n // 320 valid
        ps = "perceiver_sampler."
        Lk = 32 + T
        kvin = d.alloc(Lk * 128)
        ctx = kvin + 32 * 128 * F4
        self.conv(lat, 1536, T, 1, 1, 1, 1536, self.w_conv(ps + "proj_context.weight"), self.w_raw(ps + "proj_context.bias"), 128, ctx, 128)
        lat0 = d.put(self.W[ps + "latents"])                      # the learned latents are updated in place below: work on a per-call copy
        latents = d.alloc(32 * 128)
        self.affine(lat0, 128, 32, 128, None, None, 0, latents, 128)
        qb, kv, att, tmp = d.alloc(32 * 512), d.alloc(Lk * 1024), d.alloc(32 * 512), d.alloc(32 * 128)
        scr = d.alloc(32 * 8 * Lk)
        hbuf, gbuf = d.alloc(32 * 688), d.alloc(32 * 352)
        n_valid = min(Lk, 32 + n // 320)
        for l in range(2):
            q = ps + f"layers.{l}."
            self.affine(latents, 128, 32, 128, None, None, 0, kvin, 128)                   # cross_attn_include_queries
            self.conv(latents, 128, 32, 1, 1, 1, 128, self.w_conv(q + "0.to_q.weight"), None, 512, qb, 512)
            self.conv(kvin, 128, Lk, 1, 1, 1, 128, self.w_conv(q + "0.to_kv.weight"), None, 1024, kv, 1024)
            E._check(lib.sva_op_attention(h, qb, kv, 32, Lk, n_valid, 8, att, scr), "sva_op_attention")
            self.conv(att, 512, 32, 1, 1, 1, 512, self.w_conv(q + "0.to_out.weight"), None, 128, tmp, 128)
            E._check(lib.sva_op_add(h, latents, 128, tmp, 128, 32, 128), "attn residual")
            self.conv(latents, 128, 32, 1, 1, 1, 128, self.w_conv(q + "1.0.weight"), self.w_raw(q + "1.0.bias"), 682, hbuf, 688)
            E._check(lib.sva_op_geglu(h, hbuf, 688, 32, 341, gbuf, 352), "sva_op_geglu")
            self.conv(gbuf, 352, 32, 1, 1, 1, 352, self.w_conv(q + "1.2.weight", pad_k_to=352), self.w_raw(q + "1.2.bias"), 128, tmp, 128)
            E._check(lib.sva_op_add(h, latents, 128, tmp, 128, 32, 128), "ff residual")
        xq = d.alloc(32 * 128)
        E._check(lib.sva_op_l2norm(h, latents, 32, 128, self.w_raw(ps + "norm.gamma"), float(128 ** 0.5), xq), "sva_op_l2norm")
        # ---- ResidualFSQ [4]^6, one quantizer (residual_fsq.py:160-230): project_in -> quantise -> project_out
        z6, zq = d.alloc(32 * 8), d.alloc(32 * 128)
        self.conv(xq, 128, 32, 1, 1, 1, 128, self.w_conv("quantizer.project_in.weight"), self.w_raw("quantizer.project_in.bias"), 6, z6, 6)
        E._check(lib.sva_op_unary(h, z6, 32 * 6, 2, 0.0), "fsq")
        self.conv(z6, 6, 32, 1, 1, 1, 6, self.w_conv("quantizer.project_out.weight"), self.w_raw("quantizer.project_out.bias"), 128, zq, 128)
        return zq, (1, 32, 128)
