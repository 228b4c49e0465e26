"""Shapes of every tensor the hot path touches, keyed by the reference's state-dict names
(the checkpoint wire format) with a network prefix (``arvc.``/``tok.``/``voc.``).

The architecture constants come from the reference's Hydra YAMLs
(configs/hydra_arcs/{vc/firefly_arvc_bsq_8192_delay0_8, speech_tokenizers/causal-encoder-lfq-8192,
vocoders/firefly_gan_vq}.yaml); tools/make_golden.py asserts this table against the
reference modules' own ``state_dict()``.
"""
from __future__ import annotations

from dataclasses import dataclass, field


@dataclass(frozen=True)
class ModelConfig:
    # content encoder (speech tokenizer)
    n_mels: int = 160
    enc_depths: tuple = (3, 3, 9, 3)
    enc_dims: tuple = (128, 256, 384, 512)
    tr_layers: int = 8
    tr_heads: int = 8
    tr_dim: int = 512
    tr_inter: int = 1536
    bsq_bits: int = 13
    # dual AR
    ar_dim: int = 768
    ar_heads: int = 12
    ar_layers: int = 12
    ar_fast_layers: int = 4
    ar_inter: int = 2304
    ar_vocab: int = 8192
    codebook_size: int = 1000
    num_codebooks: int = 8
    max_delay: int = 8
    max_seq_len: int = 2048
    timbre_dim: int = 128
    timbre_tokens: int = 32
    style_dim: int = 192
    # vocoder
    voc_dim: int = 512
    fsq_levels: tuple = (8, 5, 5, 5)
    ups: tuple = ((16, 8), (16, 8), (4, 2), (4, 2), (4, 2))       # (kernel, stride)
    res_kernels: tuple = (3, 7, 11)
    res_dilations: tuple = (1, 3, 5)
    pre_kernel: int = 13
    post_kernel: int = 13


def _convnext(p, dim, out):
    out[p + "gamma"] = (dim,)
    out[p + "dwconv.conv.weight"] = (dim, 1, 7)
    out[p + "dwconv.conv.bias"] = (dim,)
    out[p + "norm.weight"] = (dim,)
    out[p + "norm.bias"] = (dim,)
    out[p + "pwconv1.weight"] = (4 * dim, dim)
    out[p + "pwconv1.bias"] = (4 * dim,)
    out[p + "pwconv2.weight"] = (dim, 4 * dim)
    out[p + "pwconv2.bias"] = (dim,)


def _convnext_encoder(p, in_ch, depths, dims, out):
    out[p + "downsample_layers.0.0.conv.weight"] = (dims[0], in_ch, 7)
    out[p + "downsample_layers.0.0.conv.bias"] = (dims[0],)
    out[p + "downsample_layers.0.1.weight"] = (dims[0],)
    out[p + "downsample_layers.0.1.bias"] = (dims[0],)
    for i in range(1, len(dims)):
        out[p + f"downsample_layers.{i}.0.weight"] = (dims[i - 1],)
        out[p + f"downsample_layers.{i}.0.bias"] = (dims[i - 1],)
        out[p + f"downsample_layers.{i}.1.weight"] = (dims[i], dims[i - 1], 1)
        out[p + f"downsample_layers.{i}.1.bias"] = (dims[i],)
    for i, (dep, dim) in enumerate(zip(depths, dims)):
        for j in range(dep):
            _convnext(p + f"stages.{i}.{j}.", dim, out)
    out[p + "norm.weight"] = (dims[-1],)
    out[p + "norm.bias"] = (dims[-1],)


def _llama_layer(p, dim, inter, out):
    out[p + "attention.wqkv.weight"] = (3 * dim, dim)
    out[p + "attention.wo.weight"] = (dim, dim)
    out[p + "feed_forward.w1.weight"] = (inter, dim)
    out[p + "feed_forward.w3.weight"] = (inter, dim)
    out[p + "feed_forward.w2.weight"] = (dim, inter)
    out[p + "ffn_norm.weight"] = (dim,)
    out[p + "attention_norm.weight"] = (dim,)


def tokenizer_specs(c: ModelConfig = ModelConfig()) -> dict:
    out = {}
    _convnext_encoder("tok.backbone.", c.n_mels, c.enc_depths, c.enc_dims, out)
    d = c.enc_dims[-1]
    for i in range(2):
        out[f"tok.quantizer.downsample.{i}.0.conv.weight"] = (d, d, 2)
        out[f"tok.quantizer.downsample.{i}.0.conv.bias"] = (d,)
        _convnext(f"tok.quantizer.downsample.{i}.1.", d, out)
    for l in range(c.tr_layers):
        p = f"tok.quantizer.pre_module.layers.{l}."
        _llama_layer(p, c.tr_dim, c.tr_inter, out)
        out[p + "attention_layer_scale.gamma"] = (c.tr_dim,)
        out[p + "ffn_layer_scale.gamma"] = (c.tr_dim,)
    out["tok.quantizer.pre_module.norm.weight"] = (c.tr_dim,)
    out["tok.quantizer.residual_bsq.rvqs.0.project_in.weight"] = (c.bsq_bits, d)
    out["tok.quantizer.residual_bsq.rvqs.0.project_in.bias"] = (c.bsq_bits,)
    return out


def arvc_specs(c: ModelConfig = ModelConfig()) -> dict:
    out = {}
    D = c.ar_dim
    out["arvc.embedding.weight"] = (c.ar_vocab, D)
    m = "arvc.decoder.model."
    out[m + "embeddings.weight"] = (c.ar_vocab, D)
    out[m + "codebook_embeddings.weight"] = (c.codebook_size * c.num_codebooks, D)
    for l in range(c.ar_layers):
        _llama_layer(m + f"layers.{l}.", D, c.ar_inter, out)
    out[m + "norm.weight"] = (D,)
    out[m + "output.weight"] = (c.ar_vocab, D)
    out[m + "fast_embeddings.weight"] = (c.codebook_size, D)
    for l in range(c.ar_fast_layers):
        _llama_layer(m + f"fast_layers.{l}.", D, c.ar_inter, out)
    out[m + "fast_norm.weight"] = (D,)
    out[m + "fast_output.weight"] = (c.codebook_size, D)
    out["arvc.decoder.wait4start_embedding.weight"] = (c.max_delay, D)
    out["arvc.decoder.wait4end_embedding.weight"] = (c.max_delay, D)
    out["arvc.context_in.weight"] = (D, c.timbre_dim)
    out["arvc.context_in.bias"] = (D,)
    out["arvc.style_in.weight"] = (D, c.style_dim)
    out["arvc.style_in.bias"] = (D,)
    return out


def vocoder_specs(c: ModelConfig = ModelConfig(), prompt_path: bool = False) -> dict:
    out = {}
    V = c.voc_dim
    g_dim = V // c.num_codebooks
    for g in range(c.num_codebooks):
        out[f"voc.quantizer.residual_fsq.rvqs.{g}.project_out.weight"] = (g_dim, len(c.fsq_levels))
        out[f"voc.quantizer.residual_fsq.rvqs.{g}.project_out.bias"] = (g_dim,)
    for i in range(2):
        out[f"voc.quantizer.upsample.{i}.0.conv.weight"] = (V, V, 2)
        out[f"voc.quantizer.upsample.{i}.0.conv.bias"] = (V,)
        _convnext(f"voc.quantizer.upsample.{i}.1.", V, out)
    h = "voc.head."
    out[h + "conv_pre.conv.weight"] = (V, V, c.pre_kernel)
    out[h + "conv_pre.conv.bias"] = (V,)
    ch = V
    for i, (k, s) in enumerate(c.ups):
        out[h + f"ups.{i}.conv.weight"] = (ch, ch // 2, k)
        out[h + f"ups.{i}.conv.bias"] = (ch // 2,)
        ch //= 2
        for b, rk in enumerate(c.res_kernels):
            for j in range(len(c.res_dilations)):
                for cs in ("convs1", "convs2"):
                    out[h + f"resblocks.{i}.blocks.{b}.{cs}.{j}.conv.weight"] = (ch, ch, rk)
                    out[h + f"resblocks.{i}.blocks.{b}.{cs}.{j}.conv.bias"] = (ch,)
    out[h + "conv_post.conv.weight"] = (1, ch, c.post_kernel)
    out[h + "conv_post.conv.bias"] = (1,)
    if prompt_path:      # firefly.encode of the reference prompt
        _convnext_encoder("voc.backbone.", c.n_mels, c.enc_depths, c.enc_dims, out)
        for i in range(2):
            out[f"voc.quantizer.downsample.{i}.0.conv.weight"] = (V, V, 2)
            out[f"voc.quantizer.downsample.{i}.0.conv.bias"] = (V,)
            _convnext(f"voc.quantizer.downsample.{i}.1.", V, out)
        for g in range(c.num_codebooks):
            out[f"voc.quantizer.residual_fsq.rvqs.{g}.project_in.weight"] = (len(c.fsq_levels), g_dim)
            out[f"voc.quantizer.residual_fsq.rvqs.{g}.project_in.bias"] = (len(c.fsq_levels),)
    return out


def _bn_spec(p, n, out, affine=True):
    if affine:
        out[p + "weight"] = (n,)
        out[p + "bias"] = (n,)
    out[p + "running_mean"] = (n,)
    out[p + "running_var"] = (n,)


def style_specs() -> dict:
    """CAMPPlus(feat_dim=80, embedding_size=192) (modules/campplus/DTDNN.py:50-117, configs/hydra_arcs/sv/campplus.yaml): the
    style encoder of the prompt path (SURVEY.md 8f N1 iii), keys prefixed 'style.'"""
    out = {}
    h = "style.head."
    out[h + "conv1.weight"] = (32, 1, 3, 3)
    _bn_spec(h + "bn1.", 32, out)
    for layer in ("layer1.", "layer2."):
        for b in range(2):
            q = h + layer + f"{b}."
            out[q + "conv1.weight"] = (32, 32, 3, 3)
            _bn_spec(q + "bn1.", 32, out)
            out[q + "conv2.weight"] = (32, 32, 3, 3)
            _bn_spec(q + "bn2.", 32, out)
            if b == 0:                                  # stride 2: projection shortcut
                out[q + "shortcut.0.weight"] = (32, 32, 1, 1)
                _bn_spec(q + "shortcut.1.", 32, out)
    out[h + "conv2.weight"] = (32, 32, 3, 3)
    _bn_spec(h + "bn2.", 32, out)
    xv = "style.xvector."
    out[xv + "tdnn.linear.weight"] = (128, 320, 5)
    _bn_spec(xv + "tdnn.nonlinear.batchnorm.", 128, out)
    ch = 128
    for bi, nl in enumerate((12, 24, 16)):
        for li in range(nl):
            q = xv + f"block{bi + 1}.tdnnd{li + 1}."
            cin = ch + 32 * li
            _bn_spec(q + "nonlinear1.batchnorm.", cin, out)
            out[q + "linear1.weight"] = (128, cin, 1)
            _bn_spec(q + "nonlinear2.batchnorm.", 128, out)
            out[q + "cam_layer.linear_local.weight"] = (32, 128, 3)
            out[q + "cam_layer.linear1.weight"] = (64, 128, 1)
            out[q + "cam_layer.linear1.bias"] = (64,)
            out[q + "cam_layer.linear2.weight"] = (32, 64, 1)
            out[q + "cam_layer.linear2.bias"] = (32,)
        ch += 32 * nl
        q = xv + f"transit{bi + 1}."
        _bn_spec(q + "nonlinear.batchnorm.", ch, out)
        out[q + "linear.weight"] = (ch // 2, ch, 1)
        ch //= 2
    _bn_spec(xv + "out_nonlinear.batchnorm.", ch, out)
    out["style.dense.linear.weight"] = (192, 2 * ch, 1)
    _bn_spec("style.dense.nonlinear.batchnorm.", 192, out, affine=False)
    return out


def timbre_specs() -> dict:
    """SpeakerEncoder(input_dim=128, out_dim=1024, latent_dim=128, token_num=32, fsq_levels=[4]*6)
    (modules/bicodec_speaker_encoder/speaker_encoder.py:36-63, configs/hydra_arcs/sv/sparktts_speaker_encoder.yaml): the tensors
    tokenize_wav touches (:136-144 -- the x-vector head, `project` and the pooling layer are not on the path), prefixed 'timbre.'"""
    out = {}
    se = "timbre.speaker_encoder."
    out[se + "layer1.conv.weight"] = (512, 128, 5)
    out[se + "layer1.conv.bias"] = (512,)
    _bn_spec(se + "layer1.bn.", 512, out)
    for li in (2, 3, 4):
        q = se + f"layer{li}.se_res2block."
        for j in (0, 2):
            out[q + f"{j}.conv.weight"] = (512, 512, 1)
            out[q + f"{j}.conv.bias"] = (512,)
            _bn_spec(q + f"{j}.bn.", 512, out)
        for i in range(7):
            out[q + f"1.convs.{i}.weight"] = (64, 64, 3)
            out[q + f"1.convs.{i}.bias"] = (64,)
            _bn_spec(q + f"1.bns.{i}.", 64, out)
        out[q + "3.linear1.weight"] = (128, 512)
        out[q + "3.linear1.bias"] = (128,)
        out[q + "3.linear2.weight"] = (512, 128)
        out[q + "3.linear2.bias"] = (512,)
    out[se + "conv.weight"] = (1536, 1536, 1)
    out[se + "conv.bias"] = (1536,)
    ps = "timbre.perceiver_sampler."
    out[ps + "latents"] = (32, 128)
    out[ps + "proj_context.weight"] = (128, 1536)
    out[ps + "proj_context.bias"] = (128,)
    for l in range(2):
        q = ps + f"layers.{l}."
        out[q + "0.to_q.weight"] = (512, 128)
        out[q + "0.to_kv.weight"] = (1024, 128)
        out[q + "0.to_out.weight"] = (128, 512)
        out[q + "1.0.weight"] = (682, 128)
        out[q + "1.0.bias"] = (682,)
        out[q + "1.2.weight"] = (128, 341)
        out[q + "1.2.bias"] = (128,)
    out[ps + "norm.gamma"] = (128,)
    out["timbre.quantizer.project_in.weight"] = (6, 128)
    out["timbre.quantizer.project_in.bias"] = (6,)
    out["timbre.quantizer.project_out.weight"] = (128, 6)
    out["timbre.quantizer.project_out.bias"] = (128,)
    return out


def prompt_encoder_specs() -> dict:
    out = style_specs()
    out.update(timbre_specs())
    return out


def all_specs(c: ModelConfig = ModelConfig(), prompt_path: bool = False) -> dict:
    out = {}
    out.update(arvc_specs(c))
    out.update(tokenizer_specs(c))
    out.update(vocoder_specs(c, prompt_path))
    return out
