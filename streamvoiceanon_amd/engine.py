"""ctypes binding of the C ABI in include/sva.h (libsva_hip.so, built by csrc/Makefile).

This is the ONLY compute path of the package: if the HIP library is missing or no MI355X is
visible the constructors raise -- there is no CPU fallback (the CPU oracle under oracle/ is
test infrastructure and is never imported from here).
"""
from __future__ import annotations

import ctypes as C
import math
import os
import threading
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# The engine overlaps its stages on four HIP streams (plus the process's default stream).  The runtime multiplexes streams
# onto GPU_MAX_HW_QUEUES hardware queues (default 4) and streams that share a queue serialise; a host application that wants
# the pipelined throughput mode at its best exports GPU_MAX_HW_QUEUES=8 BEFORE the HIP runtime starts (bench.py does; see
# INTEGRATION.md).  This module does not touch the process environment.  Measured: 382 -> 465 frames/s single-stream.
RECOMMENDED_ENV = {"GPU_MAX_HW_QUEUES": "8"}

LIB_PATH = os.environ.get("SVA_LIB_PATH") or os.path.join(_HERE, "libsva_hip.so")      # override: A/B builds of the kernels


class SvaConfig(C.Structure):
    _fields_ = [
        ("n_mels", C.c_int), ("enc_depths", C.c_int * 4), ("enc_dims", C.c_int * 4),
        ("tr_layers", C.c_int), ("tr_heads", C.c_int), ("tr_dim", C.c_int), ("tr_inter", C.c_int), ("bsq_bits", C.c_int),
        ("ar_dim", C.c_int), ("ar_heads", C.c_int), ("ar_layers", C.c_int), ("ar_fast_layers", C.c_int), ("ar_inter", C.c_int),
        ("ar_vocab", C.c_int), ("codebook_size", C.c_int), ("num_codebooks", C.c_int), ("max_delay", C.c_int),
        ("max_seq_len", C.c_int), ("timbre_dim", C.c_int), ("timbre_tokens", C.c_int), ("style_dim", C.c_int),
        ("voc_dim", C.c_int), ("ar_dtype", C.c_int), ("mm_mode", C.c_int), ("voc_dtype", C.c_int),
    ]


class SvaStreamParams(C.Structure):
    _fields_ = [
        ("n_streams", C.c_int), ("encode_window_frames", C.c_int), ("decode_window_frames", C.c_int),
        ("chunk_frames", C.c_int), ("delay", C.c_int), ("max_seq_frames", C.c_int), ("buffer_frames", C.c_int),
        ("max_prompt_frames", C.c_int), ("temperature", C.c_float), ("top_p", C.c_float),
        ("voc_max_frames", C.c_int), ("use_graph", C.c_int), ("skip_semantic", C.c_int), ("pipeline", C.c_int),
    ]


_lib = None


def load_library():
    """dlopen libsva_hip.so; raises RuntimeError (never falls back) if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). The engine has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    vp, i32, f32p = C.c_void_p, C.c_int, C.POINTER(C.c_float)
    lib.sva_last_error.restype = C.c_char_p
    lib.sva_config_default.argtypes = [C.POINTER(SvaConfig)]
    lib.sva_stream_params_default.argtypes = [C.POINTER(SvaStreamParams)]
    lib.sva_engine_create.argtypes = [C.POINTER(SvaConfig), i32, C.POINTER(vp)]
    lib.sva_engine_load_weight.argtypes = [vp, C.c_char_p, i32, C.POINTER(C.c_int64), vp]
    lib.sva_engine_finalize.argtypes = [vp]
    lib.sva_engine_destroy.argtypes = [vp]
    lib.sva_engine_destroy.restype = None
    lib.sva_batch_create.argtypes = [vp, C.POINTER(SvaStreamParams), C.POINTER(vp)]
    lib.sva_batch_destroy.argtypes = [vp]
    lib.sva_batch_destroy.restype = None
    lib.sva_prefill_prompt.argtypes = [vp, i32, vp, vp, i32, vp, vp, C.c_uint64]
    lib.sva_streams_begin.argtypes = [vp]
    lib.sva_step.argtypes = [vp, vp, vp, vp, vp]
    lib.sva_step_device.argtypes = [vp, vp, vp]
    lib.sva_step_device_on.argtypes = [vp, vp, vp, vp, C.c_int]
    lib.sva_join_stream.argtypes = [vp, vp]
    lib.sva_batch_uses_persistent_decode.argtypes = [vp]
    lib.sva_test_force_ar_timeout.argtypes = [vp]
    lib.sva_debug_configure.argtypes = [C.c_char_p]
    lib.sva_sync.argtypes = [vp]
    lib.sva_encode_window.argtypes = [vp, vp, vp, vp]
    lib.sva_vocode_window.argtypes = [vp, vp, i32, vp]
    lib.sva_vocode_stream.argtypes = [vp, vp, i32, vp]
    lib.sva_quantizer_decode.argtypes = [vp, vp, i32, vp]
    lib.sva_vocoder_head.argtypes = [vp, vp, i32, vp]
    lib.sva_vocode_reset.argtypes = [vp]
    lib.sva_ar_delay_fill.argtypes = [vp, vp]
    lib.sva_ar_decode_one.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.sva_firefly_encode.argtypes = [vp, vp, vp]
    lib.sva_stream_chunks.argtypes = [vp, vp, vp, i32]
    lib.sva_generate.argtypes = [vp, vp, vp, i32, vp, i32, vp, vp, C.c_uint64, vp, vp]
    lib.sva_get_tap.argtypes = [vp, C.c_char_p, vp, C.c_long]
    lib.sva_get_tap.restype = C.c_long
    lib.sva_get_timings.argtypes = [vp, f32p]
    lib.sva_get_gemm_stats.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_long)]
    lib.sva_get_gemm_bytes.argtypes = [vp, C.POINTER(C.c_double)]
    lib.sva_stream_codes.argtypes = [vp, C.c_int, C.c_int, vp, C.POINTER(C.c_long)]
    lib.sva_profile_gemm.argtypes = [vp, i32]
    lib.sva_get_gemm_profile.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_long)]
    lib.sva_get_gemm_profile_table.argtypes = [vp, vp, C.c_long]
    lib.sva_get_gemm_profile_table.restype = C.c_long
    lib.sva_bench_gemm.argtypes = [i32] * 9 + [f32p]
    lib.sva_bench_gemm_choice.argtypes = [i32] * 14 + [f32p]
    lib.sva_test_gemm.argtypes = [i32, i32, i32, i32, vp, vp, vp, vp]
    lib.sva_test_gemm_choice.argtypes = [i32, i32, i32, i32, vp, vp, vp, vp, i32, i32, i32, i32]
    lib.sva_set_sampler_edits.argtypes = [vp, vp, i32, C.c_float, vp, i32]
    lib.sva_test_prefill_attention.argtypes = [i32, i32, i32, i32, i32, vp, vp, vp, i32, vp, vp, i32, vp]
    lib.sva_test_pair_attention.argtypes = [i32, i32, i32, i32, i32, vp, vp, vp, i32, vp, vp, i32, vp]
    lib.sva_test_gemm_f16w.argtypes = [i32, i32, i32, i32, vp, vp, vp, vp, vp, i32, vp, i32, vp]
    lib.sva_test_gemm_planes.argtypes = [i32, i32, i32, i32, vp, vp, vp, vp, i32, i32, i32, i32, vp]
    lib.sva_host_launch_cost.argtypes = [i32, i32, f32p]
    lib.sva_test_sampler.argtypes = [i32, i32, i32, i32, vp, vp, C.c_float, C.c_float, vp, i32, f32p]
    _lib = lib
    return lib


EXPORTED_SYMBOLS = [
    "sva_last_error", "sva_config_default", "sva_stream_params_default", "sva_engine_create",
    "sva_engine_load_weight", "sva_engine_finalize", "sva_engine_destroy", "sva_batch_create", "sva_batch_destroy",
    "sva_prefill_prompt", "sva_streams_begin", "sva_step", "sva_step_device", "sva_step_device_on", "sva_join_stream", "sva_batch_uses_persistent_decode", "sva_test_force_ar_timeout", "sva_debug_configure", "sva_sync", "sva_stream_chunks", "sva_encode_window", "sva_firefly_encode",
    "sva_vocode_window", "sva_vocode_stream", "sva_vocode_reset", "sva_quantizer_decode", "sva_vocoder_head", "sva_ar_delay_fill", "sva_ar_decode_one", "sva_generate", "sva_get_tap", "sva_get_timings",
    "sva_dev_alloc", "sva_dev_free", "sva_dev_upload", "sva_dev_download", "sva_op_conv", "sva_op_affine", "sva_op_unary", "sva_op_colstats",
    "sva_op_cam_context", "sva_op_mul", "sva_op_add", "sva_op_conv2d", "sva_op_cf_to_rows", "sva_op_fbank_power", "sva_op_stft_mag", "sva_op_attention",
    "sva_op_geglu", "sva_op_l2norm", "sva_ops_capture_begin", "sva_ops_capture_end", "sva_ops_graph_launch", "sva_ops_graph_free",
    "sva_get_gemm_stats", "sva_get_gemm_bytes", "sva_stream_codes", "sva_profile_gemm", "sva_get_gemm_profile", "sva_get_gemm_profile_table", "sva_test_gemm", "sva_test_gemm_choice", "sva_test_gemm_f16w", "sva_test_gemm_planes", "sva_test_prefill_attention", "sva_test_pair_attention", "sva_set_sampler_edits", "sva_bench_gemm", "sva_bench_gemm_choice", "sva_test_sampler", "sva_host_launch_cost",
]


def _check(rc, what):
    if rc != 0:
        msg = load_library().sva_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed (rc={rc}): {msg}")


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


# --------------------------------------------------------------------------------------------
# derived (non-persistent) buffers of the reference modules, computed exactly as the reference does
# --------------------------------------------------------------------------------------------
def slaney_mel_fb(n_freqs=1025, f_min=0.0, f_max=22050.0, n_mels=160, sample_rate=44100) -> np.ndarray:
    """LogMelSpectrogram.fb (modules/vqgan/spectrogram.py:93-101) = torchaudio.functional.
    melscale_fbanks(norm="slaney", mel_scale="slaney") of torchaudio==2.4.0, restated with torch fp32
    ops in the same order (torchaudio is not a dependency of this package).  [n_freqs, n_mels]."""
    import torch

    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = math.log(6.4) / 27.0

    def hz2mel(f):
        return min_log_mel + math.log(f / min_log_hz) / logstep if f >= min_log_hz else f / f_sp

    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_pts = torch.linspace(hz2mel(f_min), hz2mel(f_max), n_mels + 2)
    f_pts = f_sp * m_pts
    is_log = m_pts >= min_log_mel
    f_pts[is_log] = min_log_hz * torch.exp(logstep * (m_pts[is_log] - min_log_mel))
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    fb = torch.clamp(torch.min(down, up), min=0.0)
    fb = fb * (2.0 / (f_pts[2: n_mels + 2] - f_pts[:n_mels])).unsqueeze(0)
    return fb.numpy().astype(np.float32)


def rope_table(seq_len: int, n_elem: int = 64, base: float = 10000.0) -> np.ndarray:
    """precompute_freqs_cis (modules/dual_ar_stream.py:993-1001): cos/sin rounded to bf16, returned as
    float32 [seq_len, n_elem/2, 2].  Computed with torch so the bf16 rounding matches the reference bit
    for bit on the same host."""
    import torch

    freqs = 1.0 / (base ** (torch.arange(0, n_elem, 2)[: n_elem // 2].float() / n_elem))
    ang = torch.outer(torch.arange(seq_len).float(), freqs)
    tab = torch.stack([torch.cos(ang), torch.sin(ang)], dim=-1)
    return tab.to(torch.bfloat16).float().numpy()


def derived_buffers(cfg: SvaConfig | None = None) -> dict:
    import torch

    out = {
        "tok.spec_transform.fb": slaney_mel_fb(),
        "tok.spec_transform.spectrogram.window": torch.hann_window(2048).numpy(),
        "tok.quantizer.pre_module.freqs_cis": rope_table(2048),
        "arvc.decoder.model.freqs_cis": rope_table(2048 if cfg is None else cfg.max_seq_len),
        "arvc.decoder.model.fast_freqs_cis": rope_table(8 if cfg is None else cfg.num_codebooks),
    }
    return out


class Engine:
    """Weights on one MI355X.  `weights`: dict name -> array-like (numpy / torch CPU tensor) keyed by the
    reference state-dict names prefixed with 'arvc.' / 'tok.' / 'voc.'."""

    def __init__(self, weights: dict, device: int = 0, ar_dtype: int = 0, mm_mode: int = None, voc_dtype: int = None):
        """mm_mode / voc_dtype: sva_config fields of the same names (None = the library's default): precision format of the batch-scale
        encoder / vocoder GEMMs (csrc/gemm_planes.hip) and the reference-precision (fp16 operand) vocoder."""
        self.lib = load_library()
        self.cfg = SvaConfig()
        _check(self.lib.sva_config_default(C.byref(self.cfg)), "sva_config_default")
        self.cfg.ar_dtype = ar_dtype
        if mm_mode is not None:
            self.cfg.mm_mode = mm_mode
        if voc_dtype is not None:
            self.cfg.voc_dtype = voc_dtype
        self.h = C.c_void_p()
        _check(self.lib.sva_engine_create(C.byref(self.cfg), device, C.byref(self.h)), "sva_engine_create")
        self.device = device
        allw = dict(derived_buffers(self.cfg))
        allw.update(weights)
        for name, arr in allw.items():
            a = np.ascontiguousarray(arr.detach().cpu().numpy() if hasattr(arr, "detach") else arr, dtype=np.float32)
            shape = (C.c_int64 * max(a.ndim, 1))(*(a.shape if a.ndim else (1,)))
            _check(self.lib.sva_engine_load_weight(self.h, name.encode(), max(a.ndim, 1), shape, _ptr(a)),
                   f"sva_engine_load_weight({name})")
        _check(self.lib.sva_engine_finalize(self.h), "sva_engine_finalize")
        self._batches = weakref.WeakSet()
        self.ops_lock = threading.RLock()      # serialises users of the engine's ops stream (prompt_encoders.py: recording / capture / replay)

    def close(self):
        if self.h:
            for b in list(getattr(self, "_batches", ())):      # a batch must not outlive its engine (its destroy touches the engine)
                b.close()
            self.lib.sva_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Batch:
    """B lock-step streams on one engine (InferenceWrapper.setup_stream_caches for B streams)."""

    def __init__(self, engine: Engine, n_streams=1, encode_window_frames=128, decode_window_frames=64, chunk_frames=1,
                 delay=2, max_seq_frames=768, buffer_frames=32, max_prompt_frames=256, temperature=0.7, top_p=0.7,
                 voc_max_frames=None, use_graph=False, skip_semantic=False, pipeline=False):
        self.engine = engine
        self.lib = engine.lib
        p = SvaStreamParams()
        _check(self.lib.sva_stream_params_default(C.byref(p)), "sva_stream_params_default")
        p.n_streams, p.encode_window_frames, p.decode_window_frames = n_streams, encode_window_frames, decode_window_frames
        p.chunk_frames, p.delay, p.max_seq_frames, p.buffer_frames = chunk_frames, delay, max_seq_frames, buffer_frames
        p.max_prompt_frames, p.temperature, p.top_p = max_prompt_frames, temperature, top_p
        p.voc_max_frames = voc_max_frames or chunk_frames
        p.use_graph, p.skip_semantic = int(use_graph), int(skip_semantic)
        p.pipeline = int(pipeline)
        self.p = p
        self.B, self.chunk = n_streams, chunk_frames
        self.h = C.c_void_p()
        _check(self.lib.sva_batch_create(engine.h, C.byref(p), C.byref(self.h)), "sva_batch_create")
        engine._batches.add(self)
        cfg = engine.cfg
        self.noise_stride = cfg.ar_vocab + cfg.num_codebooks * cfg.codebook_size

    def close(self):
        if self.h:
            self.lib.sva_batch_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def prefill_prompt(self, slot, ref_content_codes, ref_audio_codes, style, timbre, noise_seed=0):
        cc = np.ascontiguousarray(ref_content_codes, dtype=np.int64).reshape(-1)
        ac = np.ascontiguousarray(ref_audio_codes, dtype=np.int32).reshape(8, -1)
        assert ac.shape[1] == cc.shape[0]
        st = np.ascontiguousarray(style, dtype=np.float32).reshape(-1)
        tm = np.ascontiguousarray(timbre, dtype=np.float32)
        _check(self.lib.sva_prefill_prompt(self.h, slot, _ptr(cc), _ptr(ac), cc.shape[0], _ptr(st), _ptr(tm), int(noise_seed)),
               "sva_prefill_prompt")

    def begin(self):
        _check(self.lib.sva_streams_begin(self.h), "sva_streams_begin")

    def step(self, pcm_in, noise=None, forced_codes=None):
        x = np.ascontiguousarray(pcm_in, dtype=np.float32).reshape(self.B, 2048 * self.chunk)
        out = np.empty_like(x)
        nz = None if noise is None else np.ascontiguousarray(noise, dtype=np.float32).reshape(self.B, self.chunk, self.noise_stride)
        fc = None if forced_codes is None else np.ascontiguousarray(forced_codes, dtype=np.int32).reshape(self.B, 8, self.chunk)
        _check(self.lib.sva_step(self.h, _ptr(x), _ptr(out), _ptr(nz), _ptr(fc)), "sva_step")
        return out

    def encode_window(self, audio, return_u=False):
        a = np.ascontiguousarray(audio, dtype=np.float32).reshape(self.B, -1)
        W = self.p.encode_window_frames
        assert a.shape[1] == W * 2048
        codes = np.empty((self.B, W), dtype=np.int64)
        u = np.empty((self.B, W, self.engine.cfg.bsq_bits), dtype=np.float32) if return_u else None
        _check(self.lib.sva_encode_window(self.h, _ptr(a), _ptr(codes), _ptr(u)), "sva_encode_window")
        return (codes, u) if return_u else codes

    def firefly_encode(self, audio):
        """wav2target_fn (evaluations/infer_arvc.py:168-171): audio [B, W*2048] -> acoustic codes int32 [B, 8, W]."""
        a = np.ascontiguousarray(audio, dtype=np.float32).reshape(self.B, -1)
        W = self.p.encode_window_frames
        assert a.shape[1] == W * 2048
        codes = np.empty((self.B, 8, W), dtype=np.int32)
        _check(self.lib.sva_firefly_encode(self.h, _ptr(a), _ptr(codes)), "sva_firefly_encode")
        return codes

    def vocode_window(self, codes):
        c = np.ascontiguousarray(codes, dtype=np.int32).reshape(self.B, 8, -1)
        T = c.shape[2]
        out = np.empty((self.B, 2048 * T), dtype=np.float32)
        _check(self.lib.sva_vocode_window(self.h, _ptr(c), T, _ptr(out)), "sva_vocode_window")
        return out

    def quantizer_decode(self, codes):
        """firefly.quantizer.decode: codes [B, 8, T] -> z float32 [B, 512, 4T] (the reference's channel-first layout)."""
        c = np.ascontiguousarray(codes, dtype=np.int32).reshape(self.B, 8, -1)
        T = c.shape[2]
        z = np.empty((self.B, 4 * T, self.engine.cfg.voc_dim), dtype=np.float32)
        _check(self.lib.sva_quantizer_decode(self.h, _ptr(c), T, _ptr(z)), "sva_quantizer_decode")
        return np.ascontiguousarray(z.transpose(0, 2, 1))

    def vocoder_head(self, z):
        """firefly.head: z [B, 512, 4T] -> pcm float32 [B, 1, 2048 T]."""
        zz = np.ascontiguousarray(np.asarray(z, dtype=np.float32).reshape(self.B, self.engine.cfg.voc_dim, -1).transpose(0, 2, 1))
        T = zz.shape[1] // 4
        out = np.empty((self.B, 2048 * T), dtype=np.float32)
        _check(self.lib.sva_vocoder_head(self.h, _ptr(zz), T, _ptr(out)), "sva_vocoder_head")
        return out[:, None, :]

    def vocode_stream(self, codes):
        c = np.ascontiguousarray(codes, dtype=np.int32).reshape(self.B, 8, -1)
        T = c.shape[2]
        out = np.empty((self.B, 2048 * T), dtype=np.float32)
        _check(self.lib.sva_vocode_stream(self.h, _ptr(c), T, _ptr(out)), "sva_vocode_stream")
        return out

    def ar_delay_fill(self, codes):
        c = np.ascontiguousarray(codes, dtype=np.int64).reshape(self.B, self.p.delay)
        _check(self.lib.sva_ar_delay_fill(self.h, _ptr(c)), "sva_ar_delay_fill")

    def set_sampler_edits(self, previous_tokens=None, repetition_penalty=1.5, suppress_tokens=None):
        """decode_one_token_ar's previous_tokens [1 + num_codebooks, W] / repetition_penalty / suppress_tokens
        (modules/dual_ar_stream.py:1099-1117, 1175-1213); no arguments = no edits."""
        pt = None if previous_tokens is None else np.ascontiguousarray(_as_np(previous_tokens), dtype=np.int32)
        if pt is not None:
            assert pt.ndim == 2 and pt.shape[0] == 1 + self.engine.cfg.num_codebooks, "previous_tokens must be [1 + num_codebooks, W]"
        sp = None if suppress_tokens is None else np.ascontiguousarray(np.asarray(list(suppress_tokens)), dtype=np.int32).reshape(-1)
        W = 0 if pt is None else int(pt.shape[1])
        ns = 0 if sp is None else int(sp.size)
        _check(self.lib.sva_set_sampler_edits(self.h, _ptr(pt) if W else None, W, float(repetition_penalty), _ptr(sp) if ns else None, ns),
               "sva_set_sampler_edits")

    def ar_decode_one(self, code, noise=None, forced=None):
        c = np.ascontiguousarray(code, dtype=np.int64).reshape(self.B)
        nz = None if noise is None else np.ascontiguousarray(noise, dtype=np.float32).reshape(self.B, self.noise_stride)
        fc = None if forced is None else np.ascontiguousarray(forced, dtype=np.int32).reshape(self.B, 8)
        out = np.empty((self.B, 8), dtype=np.int32)
        pos = np.empty((self.B,), dtype=np.int32)
        _check(self.lib.sva_ar_decode_one(self.h, _ptr(c), _ptr(nz), _ptr(fc), _ptr(out), _ptr(pos)), "sva_ar_decode_one")
        return out, pos

    def generate(self, ref_content_codes, ref_audio_codes, src_content_codes, style, timbre, noise_seed=0, noise=None):
        """offline ARVCWrapper.generate -> codes int32 [8, S]"""
        cc = np.ascontiguousarray(ref_content_codes, dtype=np.int64).reshape(-1)
        ac = np.ascontiguousarray(ref_audio_codes, dtype=np.int32).reshape(8, -1)
        src = np.ascontiguousarray(src_content_codes, dtype=np.int64).reshape(-1)
        st = np.ascontiguousarray(style, dtype=np.float32).reshape(-1)
        tm = np.ascontiguousarray(timbre, dtype=np.float32)
        nz = None if noise is None else np.ascontiguousarray(noise, dtype=np.float32).reshape(src.shape[0], self.noise_stride)
        out = np.empty((8, src.shape[0]), dtype=np.int32)
        _check(self.lib.sva_generate(self.h, _ptr(cc), _ptr(ac), cc.shape[0], _ptr(src), src.shape[0], _ptr(st), _ptr(tm), int(noise_seed),
                                     _ptr(nz), _ptr(out)), "sva_generate")
        return out

    def vocode_reset(self):
        _check(self.lib.sva_vocode_reset(self.h), "sva_vocode_reset")

    def tap(self, what, shape, dtype=np.float32):
        out = np.empty(shape, dtype=dtype)
        n = self.lib.sva_get_tap(self.h, what.encode(), _ptr(out), out.nbytes)
        if n < 0:
            raise RuntimeError(f"sva_get_tap({what}): " + self.lib.sva_last_error().decode())
        assert n == out.nbytes, (what, n, out.nbytes)
        return out

    def timings(self):
        ms = (C.c_float * 4)()
        _check(self.lib.sva_get_timings(self.h, ms), "sva_get_timings")
        return dict(encoder=ms[0], ar=ms[1], vocoder=ms[2], total=ms[3])

    def profile_gemm(self, enable=True):
        _check(self.lib.sva_profile_gemm(self.h, int(enable)), "sva_profile_gemm")

    def gemm_profile(self):
        t, n = C.c_double(), C.c_long()
        _check(self.lib.sva_get_gemm_profile(self.h, C.byref(t), C.byref(n)), "sva_get_gemm_profile")
        return t.value, n.value

    def gemm_profile_table(self, max_rows=4096):
        out = np.zeros((max_rows, 6), dtype=np.float64)
        n = self.lib.sva_get_gemm_profile_table(self.h, _ptr(out), max_rows)
        return out[:max(n, 0)]

    def step_device(self, d_in_ptr, d_out_ptr):
        """sva_step_device: asynchronous chunk-step on device buffers.  The engine runs on streams of its own: the producer of d_in (a
        torch op on torch's stream, say) must have completed before the call, and d_out is valid after sync()."""
        _check(self.lib.sva_step_device(self.h, C.c_void_p(d_in_ptr), C.c_void_p(d_out_ptr)), "sva_step_device")

    def step_device_on(self, d_in_ptr, d_out_ptr, stream=None, join_output=True):
        """sva_step_device_on: the stream-ordered chunk-step.  `stream` = a hipStream_t handle as int (default: torch's current
        stream); no host synchronisation anywhere: the engine waits on the device for the stream's earlier work, and (join_output) the
        stream waits for the step's output."""
        if stream is None:
            import torch
            stream = torch.cuda.current_stream().cuda_stream
        _check(self.lib.sva_step_device_on(self.h, C.c_void_p(d_in_ptr), C.c_void_p(d_out_ptr), C.c_void_p(stream), int(bool(join_output))),
               "sva_step_device_on")

    def decode_path(self):
        """0 = multi-launch decode, 1 = persistent kernel (ar_decode.hip, small batches), 2 = batched persistent kernel (ar_batch.hip)."""
        return int(self.lib.sva_batch_uses_persistent_decode(self.h))

    def uses_persistent_decode(self):
        return self.decode_path() != 0

    def join_stream(self, stream=None):
        if stream is None:
            import torch
            stream = torch.cuda.current_stream().cuda_stream
        _check(self.lib.sva_join_stream(self.h, C.c_void_p(stream)), "sva_join_stream")

    def stream_chunks(self, pcm_in):
        """n consecutive chunk-steps over a host array [B, n_chunks * 2048 * chunk] in one call (device RNG; stages pipelined
        when the batch was created with pipeline=True) -> converted audio of the same shape."""
        x = np.ascontiguousarray(pcm_in, dtype=np.float32).reshape(self.B, -1)
        n = 2048 * self.chunk
        assert x.shape[1] % n == 0 and x.shape[1] >= n
        out = np.empty_like(x)
        _check(self.lib.sva_stream_chunks(self.h, _ptr(x), _ptr(out), x.shape[1] // n), "sva_stream_chunks")
        return out

    def sync(self):
        _check(self.lib.sva_sync(self.h), "sva_sync")

    def frames_decoded(self, slot=0):
        n = C.c_long()
        _check(self.lib.sva_stream_codes(self.h, slot, 0, None, C.byref(n)), "sva_stream_codes")
        return int(n.value)

    def pred_codes(self, slot=0, n=None):
        """`pred_codes[..., -n:]` of the slot's stream state (evaluations/infer_arvc.py:520-523): int32 [8, n]; n=None -> every
        frame decoded since begin() (at most the 4096-frame device ring)."""
        if n is None:
            n = min(self.frames_decoded(slot), 4096)
        out = np.empty((8, n), np.int32)
        _check(self.lib.sva_stream_codes(self.h, slot, n, _ptr(out) if n else None, None), "sva_stream_codes")
        return out

    def gemm_bytes(self):
        v = C.c_double()
        _check(self.lib.sva_get_gemm_bytes(self.h, C.byref(v)), "sva_get_gemm_bytes")
        return v.value

    def gemm_stats(self):
        f, n = C.c_double(), C.c_long()
        _check(self.lib.sva_get_gemm_stats(self.h, C.byref(f), C.byref(n)), "sva_get_gemm_stats")
        return f.value, n.value


def test_gemm(A, W, bias=None, device=0):
    lib = load_library()
    A = np.ascontiguousarray(A, dtype=np.float32)
    W = np.ascontiguousarray(W, dtype=np.float32)
    M, K = A.shape
    N = W.shape[0]
    out = np.empty((M, N), dtype=np.float32)
    b = None if bias is None else np.ascontiguousarray(bias, dtype=np.float32)
    _check(lib.sva_test_gemm(device, M, N, K, _ptr(A), _ptr(W), _ptr(b), _ptr(out)), "sva_test_gemm")
    return out


def _as_np(x):
    return x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)


def test_gemm_f16w(A, W, bias=None, rms_w=None, res=None, swiglu=False, iters=0, device=0):
    """The fp16-weight GEMM of the batched fp16 AR decode (csrc/gemm_f16w.hip): epi(norm(A) @ fp16(W).T).  Returns (C, us_per_launch)."""
    lib = load_library()
    A = np.ascontiguousarray(A, dtype=np.float32)
    W = np.ascontiguousarray(W, dtype=np.float32)
    M, K = A.shape
    N = W.shape[0]
    out = np.empty((M, N // 2 if swiglu else N), dtype=np.float32)
    cv = lambda x: None if x is None else np.ascontiguousarray(x, dtype=np.float32)
    b, nw, r = cv(bias), cv(rms_w), cv(res)
    us = np.zeros(1, dtype=np.float32)
    mode = (1 if nw is not None else 0) | (2 if r is not None else 0) | (4 if swiglu else 0)
    _check(lib.sva_test_gemm_f16w(device, M, N, K, _ptr(A), _ptr(W), _ptr(b), _ptr(nw), _ptr(r), mode, _ptr(out), int(iters), _ptr(us)), "sva_test_gemm_f16w")
    return out, float(us[0])


def test_gemm_planes(A, W, bias=None, mode=1, variant=0, a_planes=False, c_planes=False, gelu=False, silu=False, iters=0, device=0, range_check=False,
                     swiglu=False, gamma_res=False, skip_rows=False):
    """The planes GEMM (csrc/gemm_planes.hip): epi(A @ W.T) with both operands as pre-split 16-bit planes.  Returns (C, us_per_launch).
    swiglu: W rows interleave w1 | w3 in groups of 16, C is [M, N / 2]; gamma_res: C = gamma * (.) + residual with gamma[i] = 0.5 + 0.001 ((37 i) % 101),
    residual.flat[i] = 0.01 ((13 i) % 257) - 1; skip_rows (M = 170 b): rows 56..61 of every 170-row item are left at the marker -77."""
    lib = load_library()
    A = np.ascontiguousarray(A, dtype=np.float32)
    W = np.ascontiguousarray(W, dtype=np.float32)
    M, K = A.shape
    N = W.shape[0]
    out = np.empty((M, N // 2 if swiglu else N), dtype=np.float32)
    b = None if bias is None else np.ascontiguousarray(bias, dtype=np.float32)
    us = np.zeros(1, dtype=np.float32)
    flags = ((1 if a_planes else 0) | (2 if c_planes else 0) | (4 if gelu else 0) | (8 if silu else 0) | (16 if range_check else 0) | (32 if swiglu else 0) |
             (64 if gamma_res else 0) | (128 if skip_rows else 0))
    _check(lib.sva_test_gemm_planes(device, M, N, K, _ptr(A), _ptr(W), _ptr(b), _ptr(out), int(mode), int(variant), flags, int(iters), _ptr(us)),
           "sva_test_gemm_planes")
    return out, float(us[0])


def test_prefill_attention(q, keys, vals, pos0=0, S=2048, half_kv=False, iters=0, device=0):
    """q [M, H*64], keys / vals [pos0 + M, H*64] -> (per-row kernel output, MFMA flash kernel output, (us_ref, us_mfma))."""
    lib = load_library()
    q = np.ascontiguousarray(q, dtype=np.float32)
    keys = np.ascontiguousarray(keys, dtype=np.float32)
    vals = np.ascontiguousarray(vals, dtype=np.float32)
    M, D = q.shape
    assert keys.shape == (pos0 + M, D) and vals.shape == keys.shape and D % 64 == 0
    o1, o2 = np.empty((M, D), np.float32), np.empty((M, D), np.float32)
    us = np.zeros(2, np.float32)
    _check(lib.sva_test_prefill_attention(device, M, D // 64, int(pos0), int(S), _ptr(q), _ptr(keys), _ptr(vals), int(bool(half_kv)), _ptr(o1), _ptr(o2),
                                          int(iters), _ptr(us)), "sva_test_prefill_attention")
    return o1, o2, (float(us[0]), float(us[1]))


def test_pair_attention(q, keys, vals, pos0=0, S=2048, half_kv=False, iters=0, device=0):
    """as test_prefill_attention, the second output from the decode frame's PAIRED kernel (rows 2 i, 2 i + 1 = consecutive positions; M even)"""
    lib = load_library()
    q = np.ascontiguousarray(q, dtype=np.float32)
    keys = np.ascontiguousarray(keys, dtype=np.float32)
    vals = np.ascontiguousarray(vals, dtype=np.float32)
    M, D = q.shape
    assert M % 2 == 0 and keys.shape == (pos0 + M, D) and vals.shape == keys.shape and D % 64 == 0
    o1, o2 = np.empty((M, D), np.float32), np.empty((M, D), np.float32)
    us = np.zeros(2, np.float32)
    _check(lib.sva_test_pair_attention(device, M, D // 64, int(pos0), int(S), _ptr(q), _ptr(keys), _ptr(vals), int(bool(half_kv)), _ptr(o1), _ptr(o2),
                                       int(iters), _ptr(us)), "sva_test_pair_attention")
    return o1, o2, (float(us[0]), float(us[1]))


def test_gemm_choice(A, W, choice, bias=None, device=0):
    """C = A @ W.T (+bias) through ONE dispatch choice (kind, a, b, c) of the autotuned GEMM (see include/sva.h)."""
    lib = load_library()
    A = np.ascontiguousarray(A, dtype=np.float32)
    W = np.ascontiguousarray(W, dtype=np.float32)
    M, K = A.shape
    N = W.shape[0]
    out = np.empty((M, N), dtype=np.float32)
    b = None if bias is None else np.ascontiguousarray(bias, dtype=np.float32)
    _check(lib.sva_test_gemm_choice(device, M, N, K, _ptr(A), _ptr(W), _ptr(b), _ptr(out), *[int(x) for x in choice]), "sva_test_gemm_choice")
    return out


def host_launch_cost(device=0, iters=300):
    """microseconds the calling thread spends enqueueing one kernel (idle stream)"""
    us = (C.c_float * 1)()
    _check(load_library().sva_host_launch_cost(device, int(iters), us), "sva_host_launch_cost")
    return float(us[0])


def pin_enqueue_thread(device=0, group=8, iters=120):
    """Place the calling (enqueueing) thread on the group of `group` consecutive CPUs from which kernel launches to `device`
    are cheapest, among the CPUs the thread may run on now.  One streaming step is ~430 kernel launches; at 1-8 streams the
    enqueue rate of this thread is as tight a bound as the GPU's dependent-kernel chains, and it varies by ~30 % between
    core groups of a two-socket host (distance to the GPU's PCIe root).  Only the calling thread moves (sched_setaffinity on
    its tid); runtime helper threads that already exist keep their placement, so call this after the first batch has run.
    Returns (chosen cpu list, {first cpu of group: us per launch})."""
    import os
    allowed = sorted(os.sched_getaffinity(0))
    groups = {}
    for c in allowed:
        groups.setdefault(c // group, []).append(c)
    if len(groups) <= 1:
        return allowed, {}
    table = {}
    try:
        for g, cpus in sorted(groups.items()):
            os.sched_setaffinity(0, {cpus[0]})
            table[cpus[0]] = host_launch_cost(device, iters)   # (the call itself starts with 32 untimed launches: the thread has settled)
        best = min(table, key=table.get)
        chosen = groups[best // group]
    except Exception:
        os.sched_setaffinity(0, set(allowed))
        raise
    os.sched_setaffinity(0, set(chosen))
    return chosen, table


def test_sampler(logits, noise, variant, temperature=0.7, top_p=0.7, iters=0, device=0):
    """tokens [rows] of the nucleus sampler through ONE implementation (see include/sva.h); with iters > 0 also the average
    microseconds per launch -> (tokens, us)."""
    lib = load_library()
    L = np.ascontiguousarray(logits, dtype=np.float32)
    Q = np.ascontiguousarray(noise, dtype=np.float32)
    rows, V = L.shape
    assert Q.shape == L.shape
    out = np.empty(rows, dtype=np.int32)
    us = (C.c_float * 1)()
    _check(lib.sva_test_sampler(device, int(variant), rows, V, _ptr(L), _ptr(Q), float(temperature), float(top_p), _ptr(out), int(iters),
                                us if iters > 0 else None), "sva_test_sampler")
    return (out, float(us[0])) if iters > 0 else out


def bench_gemm_choice(B, T, N, Cin, kind, a=0, b=0, c=0, taps=1, dil=1, mode=0, nrot=1, iters=50, device=0):
    """one dispatch choice (csrc/testhooks.hip: sva_bench_gemm_choice) -> (us eager, us as a graph, max |C - C_dispatcher|, max |C_dispatcher|)"""
    lib = load_library()
    out = (C.c_float * 4)()
    _check(lib.sva_bench_gemm_choice(device, B, T, N, Cin, taps, dil, mode, kind, a, b, c, nrot, iters, out), "sva_bench_gemm_choice")
    return float(out[0]), float(out[1]), float(out[2]), float(out[3])


def bench_gemm(B, T, N, Cin, taps=1, dil=1, mode=0, iters=50, device=0):
    """average microseconds per conv-GEMM launch (device-resident random data)"""
    lib = load_library()
    out = (C.c_float * 1)()
    _check(lib.sva_bench_gemm(device, B, T, N, Cin, taps, dil, mode, iters, out), "sva_bench_gemm")
    return float(out[0])
