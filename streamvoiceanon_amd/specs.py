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


def all_specs(c: ModelConfig = ModelConfig(), prompt_path: bool = False) -> dict:
    out = {}
    out.update(arvc_specs(c))
    out.update(tokenizer_specs(c))
    out.update(vocoder_specs(c, prompt_path))
    return out
