"""CPU-only checks of the host-side logic: deterministic generators (known answers pin the integer hash that the GPU
box must reproduce), the streaming left-pad rule, the spec table, and the wrapper surfaces."""
import inspect

import pytest

import numpy as np


def test_weight_generator_known_answers():
    from streamvoiceanon_amd import synth_weights as sw

    a = sw.generate(0, "arvc.embedding.weight", (8192, 768))
    assert a.dtype == np.float32 and a.shape == (8192, 768)
    np.testing.assert_allclose(a[0, :3], [0.2343647, 0.34477454, -0.4923129], rtol=0, atol=1e-7)
    assert abs(float(a.std()) - 0.46163636) < 1e-6
    assert sw.generate(0, "tok.head.anything.weight", (4, 4)) is None          # dead at inference: never generated
    g = sw.generate(3, "tok.backbone.stages.0.0.gamma", (128,))
    assert g.min() >= 0.1 and g.max() <= 0.5
    # same tensor, different seed -> different values; same seed -> identical
    assert not np.array_equal(sw.generate(1, "arvc.style_in.bias", (768,)), sw.generate(2, "arvc.style_in.bias", (768,)))
    np.testing.assert_array_equal(sw.generate(1, "arvc.style_in.bias", (768,)), sw.generate(1, "arvc.style_in.bias", (768,)))


def test_noise_key_is_pure_integer_and_stable():
    from streamvoiceanon_amd import synth_weights as sw

    assert sw.noise_key(1000, 0, 0) == sw.noise_key(1000, 0, 0)
    assert sw.noise_key(1000, 0, 0) != sw.noise_key(1000, 0, 1) != sw.noise_key(1000, 1, 0)
    k = sw.u24_from_key(sw.noise_key(1000, 0, 0), 8)
    assert k.dtype == np.uint32 and int(k.max()) < (1 << 24)
    nz = sw.exp1_noise(1000, 0, 0, 8192)
    np.testing.assert_allclose(nz[:4], [0.23771206, 0.00791767, 0.12068872, 0.21326408], rtol=1e-6)
    assert 0.9 < float(nz.mean()) < 1.1 and nz.min() > 0          # Exp(1)


def test_stream_left_pad_rule():
    from streamvoiceanon_amd.synth_audio import pad_to_chunks

    x = np.ones(2048 * 3 + 5, np.float32)
    assert pad_to_chunks(x, 1).shape[0] == 2048 * 4 and pad_to_chunks(x, 1)[:2043].sum() == 0
    y = np.ones(2048 * 3, np.float32)
    assert pad_to_chunks(y, 1).shape[0] == 2048 * 4          # evaluations/infer_arvc.py:648-649: a FULL extra chunk when aligned
    assert pad_to_chunks(y, 4).shape[0] == 2048 * 4


def test_spec_table_counts_match_reference_probe():
    from streamvoiceanon_amd import specs

    assert len(specs.arvc_specs()) == 126                    # 126 tensors, 149.5 M elements
    assert sum(int(np.prod(s)) for s in specs.arvc_specs().values()) == 149523456
    assert len(specs.all_specs(prompt_path=True)) == 853     # asserted against the reference state_dicts by tools/make_golden.py


def test_wrapper_surface_matches_reference_signatures():
    """Same method names / defaults as evaluations/infer_arvc.py:443-460, 492, 598-613 and modules/arvc_wrapper.py:25-126."""
    from streamvoiceanon_amd.arvc_wrapper import ARVCWrapper
    from streamvoiceanon_amd.infer_arvc import InferenceWrapper

    sig = inspect.signature(InferenceWrapper.setup_stream_caches)
    assert [sig.parameters[k].default for k in ("encode_window_frames", "decode_window_frames", "max_seq_frames", "buffer_frames",
                                                "decode_chunk_frames", "delay")] == [96, 64, 768, 32, 1, None]
    sig = inspect.signature(InferenceWrapper.stream_infer)
    assert [sig.parameters[k].default for k in ("encode_window_frames", "decode_window_frames", "max_prompt_frames", "max_seq_frames",
                                                "buffer_frames", "decode_chunk_frames")] == [128, 64, 256, 768, 32, 1]
    assert InferenceWrapper.SAMPLES_PER_FRAME == 2048 and InferenceWrapper.NUM_CODEBOOKS == 8
    for name in ("setup_caches", "set_delay", "compile_ar_decode_fn", "prefill_prompt", "prefill_src_condition4delay", "decode_one"):
        assert callable(getattr(ARVCWrapper, name))


def test_synth_audio_is_deterministic_and_bounded():
    from streamvoiceanon_amd.synth_audio import synth_prompt, synth_utterance

    x = synth_utterance(1000, 2048 * 4)
    np.testing.assert_array_equal(x, synth_utterance(1000, 2048 * 4))
    assert x.dtype == np.float32 and abs(float(np.abs(x).max()) - 0.5) < 1e-6
    ac, cc, style, timbre = synth_prompt(2000, 107)
    assert ac.shape == (8, 107) and ac.dtype == np.int32 and 0 <= ac.min() and ac.max() < 1000
    assert cc.shape == (107,) and cc.dtype == np.int64 and cc.max() < 8192
    assert style.shape == (192,) and timbre.shape == (32, 128)


def test_load_checkpoints_unwraps_like_the_reference(tmp_path):
    """InferenceWrapper.load_checkpoints mirrors evaluations/infer_arvc.py:67-94, 160-165: the main checkpoint is a plain state
    dict, the speech tokenizer's may be wrapped in {'net': ...} with 'module.'-prefixed keys (DDP), the vocoder's is plain
    (weight-norm pairs included, folded later by the engine); non-float entries are dropped."""
    import numpy as np
    import torch
    import yaml

    from streamvoiceanon_amd.infer_arvc import InferenceWrapper

    g = torch.Generator().manual_seed(0)
    arvc = {"embedding.weight": torch.randn(5, 3, generator=g), "decoder.model.norm.weight": torch.randn(3, generator=g)}
    tok = {"backbone.norm.weight": torch.randn(4, generator=g), "quantizer.pre_module.layers.0.attention.wo.weight": torch.randn(2, 2, generator=g)}
    voc = {"head.conv_pre.conv.parametrizations.weight.original0": torch.rand(3, 1, 1, generator=g),
           "head.conv_pre.conv.parametrizations.weight.original1": torch.randn(3, 2, 5, generator=g),
           "head.conv_pre.conv.bias": torch.randn(3, generator=g), "backbone.bn.num_batches_tracked": torch.tensor(7)}
    torch.save(arvc, tmp_path / "arvc.pth")
    torch.save({"net": {"module." + k: v for k, v in tok.items()}, "iters": 123}, tmp_path / "tok.pth")
    torch.save(voc, tmp_path / "voc.pth")
    cfg = {"speech_tokenizer": {"checkpoint_path": str(tmp_path / "tok.pth")}, "firefly": {"checkpoint_path": str(tmp_path / "voc.pth")}}
    yaml.safe_dump(cfg, open(tmp_path / "config.yaml", "w"))
    W = InferenceWrapper.load_checkpoints(str(tmp_path / "config.yaml"), str(tmp_path / "arvc.pth"))
    want = {**{"arvc." + k: v for k, v in arvc.items()}, **{"tok." + k: v for k, v in tok.items()},
            **{"voc." + k: v for k, v in voc.items() if v.dtype.is_floating_point}}
    assert set(W) == set(want)
    for k in want:
        np.testing.assert_array_equal(np.asarray(W[k]), want[k].numpy())
    # a tokenizer checkpoint that is already a bare state dict loads the same way
    torch.save(tok, tmp_path / "tok.pth")
    W2 = InferenceWrapper.load_checkpoints(str(tmp_path / "config.yaml"), str(tmp_path / "arvc.pth"))
    assert set(W2) == set(want)


def test_sampling_kwargs_are_validated_not_dropped():
    """sample()'s arguments (modules/dual_ar_stream.py:1081-1132, 1175-1213): temperature / top_p become batch parameters, the
    edit arguments travel under "edits" (a penalty without previous_tokens does nothing, as in the reference), unknown names raise."""
    from streamvoiceanon_amd.infer_arvc import check_sampling_kwargs

    assert check_sampling_kwargs({}) == {}
    assert check_sampling_kwargs({"temperature": 1, "top_p": 0.9, "repetition_penalty": 1.2, "previous_tokens": None}) == {"temperature": 1.0, "top_p": 0.9}
    got = check_sampling_kwargs({"previous_tokens": [[1]] * 9, "repetition_penalty": 1.3, "suppress_tokens": [5]})
    assert got == {"edits": {"previous_tokens": [[1]] * 9, "repetition_penalty": 1.3, "suppress_tokens": [5]}}
    assert check_sampling_kwargs({"suppress_tokens": [7]}) == {"edits": {"suppress_tokens": [7]}}
    with pytest.raises(TypeError):
        check_sampling_kwargs({"top_k": 5})


def test_xcd_tile_remap_is_a_bijection():
    """The workgroup-id remap of the tiled GEMM kernels (csrc/sva_common.h: xcd_tile) must visit every tile exactly once for ANY tile
    count -- the simple `(id % 8) * ceil(T / 8) + id / 8` is not a bijection when T % 8 != 0 -- and must hand XCD k (ids = k mod 8) a
    contiguous range of the tile sequence."""
    def remap(L, T):
        xcd, idx, q, r = L & 7, L >> 3, T >> 3, T & 7
        return (xcd * (q + 1) if xcd < r else r * (q + 1) + (xcd - r) * q) + idx

    for T in list(range(1, 300)) + [765, 1020, 1536, 4097]:
        V = [remap(L, T) for L in range(T)]
        assert sorted(V) == list(range(T)), T
        for k in range(min(8, T)):
            mine = sorted(V[L] for L in range(k, T, 8))
            assert mine == list(range(mine[0], mine[0] + len(mine))), (T, k)


def test_compiled_reference_restatement_matches_oracle():
    """bench.py's torch.compile baseline times a static-KV-cache restatement of decode_one_token_ar (the reference's
    compile-friendly form, dual_ar_stream.py:312-356): run eagerly on the CPU it must reproduce the oracle's slow-AR hidden state."""
    import torch

    import bench
    from oracle import sva_oracle as O
    from streamvoiceanon_amd import specs, synth_weights
    from streamvoiceanon_amd.synth_audio import synth_prompt

    torch.set_grad_enabled(False)
    W = {k: torch.from_numpy(v) for k, v in synth_weights.generate_all(0, specs.all_specs()).items() if k.startswith("arvc.")}
    ac, cc, style, timbre = synth_prompt(2000, 6)
    ar = O.DualAR(W)
    ar.prefill_prompt(torch.from_numpy(cc), torch.from_numpy(ac), torch.from_numpy(style), torch.from_numpy(timbre), 2)
    ref = bench.CompiledReference(W, torch.device("cpu"), S=ar.k[0].shape[1] if hasattr(ar, "k") else 2048)
    for l in range(ar.cfg.n_layer):
        ref.k[l].copy_(ar.k[l]); ref.v[l].copy_(ar.v[l])
    x = torch.randn(2, 768, generator=torch.Generator().manual_seed(3)) * 0.1
    pos = torch.arange(2) + ar.last_pos + 1
    hidden_o, logits_o = ar.slow_forward(x.clone(), pos)
    h = x.clone()
    for l in range(ar.cfg.n_layer):
        h = ref._block(h, f"arvc.decoder.model.layers.{l}.", ref.tab, ref.k[l], ref.v[l], pos, ref.S)
    assert (h[-1] - hidden_o).abs().max() <= 2e-4 * max(1.0, float(hidden_o.abs().max()))
    for l in range(ar.cfg.n_layer):          # the static cache received the same K / V rows
        assert torch.allclose(ref.k[l][:, pos], ar.k[l][:, pos], atol=1e-5)


def test_stream_server_protocol_with_a_stub_model():
    """stream_server.py's wire protocol and its lazy re-prefill bookkeeping (real-time-gui.py:32-49) against a stub model set: no GPU."""
    import threading

    import numpy as np

    from streamvoiceanon_amd import stream_server as S

    class Stub:
        def __init__(self):
            self.prefills, self.setups = [], []

        def prefill_prompt(self, ref, max_prompt_frames=64, delay=2, alpha=1.0):
            self.prefills.append((float(np.asarray(ref).sum()), delay, alpha))

        def setup_stream_caches(self, **kw):
            self.setups.append(kw["decode_chunk_frames"])

        def process_one_chunk(self, block):
            return np.asarray(block) * 0.5

    stub, ready = Stub(), threading.Event()
    th = threading.Thread(target=S.serve, args=(stub, "127.0.0.1", 0, 1, ready), daemon=True)
    th.start()
    assert ready.wait(10)
    c = S.Client(port=ready.port)
    x = np.arange(4096, dtype=np.float32)
    try:
        c.convert(x[:2048])
        raise AssertionError("a block before any reference must be refused")
    except RuntimeError as ex:
        assert "reference" in str(ex)
    c.set_reference("r1", np.ones(2048 * 4, np.float32))
    c.configure(alpha=0.5, block_frame=1, n_frame_delay=3)
    np.testing.assert_array_equal(c.convert(x[:2048]), x[:2048] * 0.5)
    np.testing.assert_array_equal(c.convert(x[2048:]), x[2048:] * 0.5)
    assert len(stub.prefills) == 1 and stub.prefills[0][1:] == (3, 0.5) and stub.setups == [1]
    c.configure(alpha=0.5, block_frame=2, n_frame_delay=3)
    np.testing.assert_array_equal(c.convert(x), x * 0.5)                 # new block size: caches rebuilt
    c.set_reference("r2", np.full(2048 * 4, 2.0, np.float32))
    c.convert(x)
    assert len(stub.prefills) == 3 and stub.setups == [1, 2, 2]
    c.close()
    th.join(10)
    assert not th.is_alive()


def test_stream_server_survives_malformed_frames():
    """Wire-level hardening: a short CONF payload is answered with an error on the open connection; an oversize or unknown
    frame is refused before its payload is buffered and only that connection is dropped -- the server accepts the next client."""
    import socket
    import struct
    import threading

    import numpy as np

    from streamvoiceanon_amd import stream_server as S

    class Stub:
        def prefill_prompt(self, ref, max_prompt_frames=64, delay=2, alpha=1.0):
            pass

        def setup_stream_caches(self, **kw):
            pass

        def process_one_chunk(self, block):
            return np.asarray(block)

    ready = threading.Event()
    th = threading.Thread(target=S.serve, args=(Stub(), "127.0.0.1", 0, 3, ready), daemon=True)
    th.start()
    assert ready.wait(10)

    def read_frame(sock):
        k, n = struct.unpack("<II", S._recv_exact(sock, 8))
        return k, S._recv_exact(sock, n) if n else b""

    # 1: a CONF frame of the wrong length -> struct.error is reported, the connection keeps working
    c = S.Client(port=ready.port)
    S._send(c.sock, S.KIND_CONF, b"\x00" * 5)
    k, body = read_frame(c.sock)
    assert k == S.KIND_ERR and b"error" in body.lower()
    c.set_reference("r", np.ones(2048 * 4, np.float32))
    c.configure(block_frame=1)
    assert c.convert(np.zeros(2048, np.float32)).shape == (2048,)
    try:
        c.configure(block_frame=S.MAX_BLOCK_FRAMES + 1)
        raise AssertionError("an oversize block_frame must be refused")
    except RuntimeError:
        pass
    c.close()
    # 2: a 3 GiB BLOCK header: refused without reading the payload, connection closed by the server
    s2 = socket.create_connection(("127.0.0.1", ready.port))
    s2.sendall(struct.pack("<II", S.KIND_BLOCK, 3 << 30))
    k, body = read_frame(s2)
    assert k == S.KIND_ERR and b"refused" in body
    assert s2.recv(1) == b""
    s2.close()
    # 3: an unknown kind, then a well-behaved client is still served
    s3 = socket.create_connection(("127.0.0.1", ready.port))
    s3.sendall(struct.pack("<II", 77, 0))
    k, body = read_frame(s3)
    assert k == S.KIND_ERR
    s3.close()
    th.join(10)
    assert not th.is_alive()


def test_planes_variant_table_is_well_formed():
    """csrc/planes_table.inc (tools/planes_tune.py): rows {M, N, K, variant} that csrc/gemm.hip includes verbatim -- one row per
    (M, N, K), variants the launcher knows, shapes the planes kernel accepts (K in whole 32-wide tiles, N in 16-byte rows)."""
    import os
    import re

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "streamvoiceanon_amd", "csrc", "planes_table.inc")
    rows = [tuple(int(x) for x in m.groups()) for m in re.finditer(r"^\{(\d+), (\d+), (\d+), (\d+)\},$", open(path).read(), re.M)]
    assert len(rows) >= 100
    assert len({r[:3] for r in rows}) == len(rows)
    for M, N, K, v in rows:
        assert v in (0, 1, 2, 3, 6, 7, 9, 10, 11, 12) and K % 32 == 0 and N % 4 == 0 and M >= 1024          # (9 .. 12: the persistent LDS-DMA forms, whole 128-column tiles; 11 / 12 with loader waves)
        assert not (v in (0, 2, 6, 7) and N < 128) and not (v == 6 and M < 256) and not (v >= 9 and N % 128)
