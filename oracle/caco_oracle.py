"""CPU oracle for the Cacophony inference hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain fp32 restatement (NumPy; optionally torch CPU ops through the same code) of the
reference's algorithm for: waveform -> log-mel -> 16x16 patches, the AudioMAE-ViT audio encoder
and attention pooler, the causal RoBERTa text encoder, pooler and projection, L2 normalisation,
the contrastive logits, and the AudioMAE stage-1 decoder forward.  It does not import the
reference.  Each function cites the reference file:line it follows (paths relative to the
reference repo root).

Who may use it: only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg, and
only as the checker / reported baseline.  The product path (`cacophony_amd`) never imports this
module and fails loudly when the HIP library is missing.

Pinning: the reference ships no tests, golden vectors or fixtures for this path ("parity unpinned"
by its own tests).  The oracle is therefore pinned against outputs of the
reference itself, produced in the build container by importing `/root/reference/src/caco_torch`
(and `src/eval/eval_caco_torch.py` pre-processing) on seeded inputs and committed under
`tests/golden/` by `tests/golden/make_golden.py`; `tests/test_oracle_golden.py` replays them.

Third-party arithmetic restated here because it is absent from the reference tree:
`torchaudio.functional.melscale_fbanks` (torchaudio 2.5.1, requirements_torch.txt:51; call site
src/eval/eval_caco_torch.py:94-101)  --  HTK mel scale, triangular filters, no normalisation.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import numpy as np

NORM_EPS = 1e-10  # src/caco_torch/caco.py:14


# ----------------------------------------------------------------------------------------------
# array back-ends: the same oracle code runs on NumPy (the checker) or torch CPU (the timed
# `cpu_baseline` "port" of the reference's CPU torch path)
# ----------------------------------------------------------------------------------------------

class NumpyOps:
    name = "numpy"

    @staticmethod
    def f32(x):
        return np.asarray(x, dtype=np.float32)

    @staticmethod
    def i64(x):
        return np.asarray(x, dtype=np.int64)

    @staticmethod
    def to_numpy(x):
        return np.asarray(x)

    matmul = staticmethod(np.matmul)
    exp = staticmethod(np.exp)
    sin = staticmethod(np.sin)
    cos = staticmethod(np.cos)
    sqrt = staticmethod(np.sqrt)
    where = staticmethod(np.where)

    @staticmethod
    def cat(xs, axis):
        return np.concatenate(xs, axis=axis)

    @staticmethod
    def transpose(x, a, b):
        return np.swapaxes(x, a, b)

    @staticmethod
    def erf(x):
        from scipy.special import erf
        return erf(x).astype(np.float32)

    @staticmethod
    def arange(n, dtype=np.float32):
        return np.arange(n, dtype=dtype)

    @staticmethod
    def full_like(x, v):
        return np.full_like(x, v)

    @staticmethod
    def tril_bool(n):
        return np.tril(np.ones((n, n), dtype=bool))

    @staticmethod
    def amax(x, axis):
        return np.max(x, axis=axis, keepdims=True)

    @staticmethod
    def sum(x, axis):
        return np.sum(x, axis=axis, keepdims=True)

    @staticmethod
    def mean(x, axis):
        return np.mean(x, axis=axis, keepdims=True)

    @staticmethod
    def take_rows(table, idx):
        return table[idx]


class TorchOps:
    name = "torch"

    def __init__(self):
        import torch
        self.t = torch

    def f32(self, x):
        return self.t.as_tensor(np.asarray(x) if not self.t.is_tensor(x) else x, dtype=self.t.float32)

    def i64(self, x):
        return self.t.as_tensor(np.asarray(x) if not self.t.is_tensor(x) else x, dtype=self.t.int64)

    def to_numpy(self, x):
        return x.detach().cpu().numpy() if self.t.is_tensor(x) else np.asarray(x)

    def matmul(self, a, b):
        return self.t.matmul(a, b)

    def exp(self, x):
        return self.t.exp(x)

    def sin(self, x):
        return self.t.sin(x)

    def cos(self, x):
        return self.t.cos(x)

    def sqrt(self, x):
        return self.t.sqrt(x)

    def where(self, c, a, b):
        return self.t.where(c, a, b)

    def cat(self, xs, axis):
        return self.t.cat(list(xs), dim=axis)

    def transpose(self, x, a, b):
        return x.transpose(a, b)

    def erf(self, x):
        return self.t.erf(x)

    def arange(self, n, dtype=np.float32):
        return self.t.arange(n, dtype=self.t.float32 if dtype == np.float32 else self.t.int64)

    def full_like(self, x, v):
        return self.t.full_like(x, v)

    def tril_bool(self, n):
        return self.t.tril(self.t.ones(n, n, dtype=self.t.bool))

    def amax(self, x, axis):
        return self.t.amax(x, dim=axis, keepdim=True)

    def sum(self, x, axis):
        return self.t.sum(x, dim=axis, keepdim=True)

    def mean(self, x, axis):
        return self.t.mean(x, dim=axis, keepdim=True)

    def take_rows(self, table, idx):
        return table[idx]


def get_ops(backend: str = "numpy"):
    return TorchOps() if backend == "torch" else NumpyOps()


# ----------------------------------------------------------------------------------------------
# primitive layers
# ----------------------------------------------------------------------------------------------

def linear(ops, x, w, b=None):
    """nn.Linear: y = x @ W^T + b with W stored [out, in]."""
    y = ops.matmul(x, ops.transpose(w, -1, -2))
    return y if b is None else y + b


def layer_norm(ops, x, g, b, eps):
    """nn.LayerNorm over the last axis, biased variance."""
    mu = ops.mean(x, -1)
    xc = x - mu
    var = ops.mean(xc * xc, -1)
    return xc / ops.sqrt(var + eps) * g + b


def softmax_last(ops, x):
    m = ops.amax(x, -1)
    e = ops.exp(x - m)
    return e / ops.sum(e, -1)


def silu(ops, x):
    return x / (1.0 + ops.exp(-x))


def gelu_erf(ops, x):
    """F.gelu default (exact erf form), src/caco_torch/text_models/roberta.py:157."""
    return 0.5 * x * (1.0 + ops.erf(x * (1.0 / math.sqrt(2.0))))


def l2_normalize(ops, x):
    """x / ||x + 1e-10||_2  --  eps is added to the vector, not the norm (caco.py:144-146,171-173)."""
    y = x + NORM_EPS
    return x / ops.sqrt(ops.sum(y * y, -1))


# ----------------------------------------------------------------------------------------------
# front end: waveform -> log-mel -> patches
# ----------------------------------------------------------------------------------------------

def hann_window_periodic(n: int) -> np.ndarray:
    """torch.hann_window(n) (periodic=True): 0.5 - 0.5 cos(2 pi k / n)."""
    k = np.arange(n, dtype=np.float64)
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * k / n)).astype(np.float32)


def melscale_fbanks_htk(n_freqs: int, f_min: float, f_max: float, n_mels: int, sample_rate: int) -> np.ndarray:
    """torchaudio.functional.melscale_fbanks(..., norm=None, mel_scale='htk') -> [n_freqs, n_mels].

    Published algorithm (torchaudio 2.5.1 functional.py `melscale_fbanks` / `_create_triangular_filterbank`):
    bin centres linspace(0, sr//2, n_freqs); n_mels+2 points equally spaced on the HTK mel scale
    m = 2595 log10(1 + f/700); filter = max(0, min(rising slope, falling slope)).
    Call site: src/eval/eval_caco_torch.py:94-101.
    """
    all_freqs = np.linspace(0.0, sample_rate // 2, n_freqs)
    m_min = 2595.0 * np.log10(1.0 + f_min / 700.0)
    m_max = 2595.0 * np.log10(1.0 + f_max / 700.0)
    m_pts = np.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts[None, :] - all_freqs[:, None]
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return np.maximum(0.0, np.minimum(down, up)).astype(np.float32)


def num_frames(audio_len: int, hop_length: int = 160) -> int:
    """ceil(audio_len / hop) frames (src/eval/eval_caco_torch.py:66-72)."""
    return (audio_len + hop_length - 1) // hop_length


def compute_mel_spectrogram(audio, sr: int = 16000, hop_length: int = 160, win_length: int = 400,
                            n_fft: int = 512, n_mels: int = 128, scale: float = 0.2, bias: float = 0.9,
                            backend: str = "numpy") -> np.ndarray:
    """src/eval/eval_caco_torch.py:41-105.

    Zero-pad to (ceil(L/hop)-1)*hop + n_fft (:66-78); torch.stft(center=False) with a periodic
    Hann(win_length) that torch centres inside the n_fft frame (:81-89), i.e. frame k is
    samples [k*hop, k*hop+n_fft) times the window zero-padded by (n_fft-win_length)//2 on the left;
    magnitude (:91); HTK mel filterbank (:94-103); log(x + 1e-5)*scale + bias (:104).
    """
    audio = np.asarray(audio, dtype=np.float32).reshape(-1)
    audio_len = audio.shape[0]
    frames = num_frames(audio_len, hop_length)
    required = (frames - 1) * hop_length + n_fft
    if required > audio_len:
        audio = np.concatenate([audio, np.zeros(required - audio_len, dtype=np.float32)])
    n_stft = (audio.shape[0] - n_fft) // hop_length + 1
    left = (n_fft - win_length) // 2
    window = np.zeros(n_fft, dtype=np.float32)
    window[left:left + win_length] = hann_window_periodic(win_length)
    idx = np.arange(n_stft)[:, None] * hop_length + np.arange(n_fft)[None, :]
    framed = audio[idx] * window[None, :]
    fb = melscale_fbanks_htk(n_fft // 2 + 1, 0.0, sr / 2, n_mels, sr)
    if backend == "torch":
        import torch
        spec = torch.fft.rfft(torch.from_numpy(framed), dim=-1).abs()
        mel = spec @ torch.from_numpy(fb)
        return (torch.log(mel + 1e-5) * scale + bias).numpy()
    spec = np.abs(np.fft.rfft(framed.astype(np.float64), axis=-1)).astype(np.float32)
    mel = spec @ fb
    return (np.log(mel + np.float32(1e-5)) * np.float32(scale) + np.float32(bias)).astype(np.float32)


def spectrogram_to_patches(spectrogram: np.ndarray, time_patch_size: int = 16, freq_patch_size: int = 16,
                           max_patches: int = 512) -> Dict[str, np.ndarray]:
    """src/eval/eval_caco_torch.py:108-151: truncate, 16x16 patchify (time-patch major, then
    freq-patch; inside a patch 16 time x 16 mel, time major), pad / truncate to max_patches."""
    spectrogram = np.asarray(spectrogram, dtype=np.float32)
    n_t = spectrogram.shape[0] // time_patch_size
    n_f = spectrogram.shape[1] // freq_patch_size
    full = n_t * n_f
    x = spectrogram[: n_t * time_patch_size, : n_f * freq_patch_size]
    x = x.reshape(n_t, time_patch_size, n_f, freq_patch_size).transpose(0, 2, 1, 3)
    x = x.reshape(full, time_patch_size * freq_patch_size)
    pos = np.arange(max_patches)
    if full > max_patches:
        x = x[:max_patches]
        mask = np.ones(max_patches, dtype=np.float32)
        time_inds = pos // n_f
        freq_inds = pos % n_f
    else:
        mask = (pos < full).astype(np.float32)
        kept = (mask * pos).astype(np.int64)
        time_inds = kept // n_f
        freq_inds = kept % n_f
        x = np.concatenate([x, np.zeros((max_patches - full, x.shape[1]), dtype=np.float32)], axis=0)
    return {
        "audio_patches": x.astype(np.float32),
        "audio_time_inds": time_inds.astype(np.float32),
        "audio_freq_inds": freq_inds.astype(np.float32),
        "audio_mask": mask.astype(np.float32),
    }


def prepare_audio_batch(wav: np.ndarray, max_patches: int = 500, backend: str = "numpy") -> Dict[str, np.ndarray]:
    """Batched form of src/eval/eval_caco_torch.py:181-206 (the reference does one clip per call)."""
    wav = np.asarray(wav, dtype=np.float32)
    if wav.ndim == 1:
        wav = wav[None]
    outs = [spectrogram_to_patches(compute_mel_spectrogram(w, backend=backend), max_patches=max_patches) for w in wav]
    return {k: np.stack([o[k] for o in outs], axis=0) for k in outs[0]}


# ----------------------------------------------------------------------------------------------
# audio encoder (AudioMAE ViT) and pooler
# ----------------------------------------------------------------------------------------------

def sin_cos_pos_embed(ops, position_ids, embed_size: int):
    """get_sin_cos_pos_embed, src/caco_torch/audio_models/mae.py:102-109 (halves concatenated)."""
    half = embed_size // 2
    freqs = ops.exp(2.0 * ops.arange(half) * (-math.log(10000.0) / embed_size))
    ang = position_ids[..., None] * freqs
    return ops.cat([ops.sin(ang), ops.cos(ang)], -1)


def _mha_self(ops, h, w_in, b_in, w_out, b_out, num_heads: int, key_keep):
    """torch.nn.MultiheadAttention(batch_first=True)(h, h, h, key_padding_mask=~keep, need_weights=False)
    as called at src/caco_torch/audio_models/mae.py:92: packed in_proj rows [Wq; Wk; Wv], heads are
    contiguous head_dim slices, q scaled by head_dim**-0.5, masked keys -> -inf, softmax, out_proj."""
    B, S, H = h.shape
    hd = H // num_heads
    qkv = linear(ops, h, w_in, b_in)
    q = qkv[..., :H].reshape(B, S, num_heads, hd)
    k = qkv[..., H:2 * H].reshape(B, S, num_heads, hd)
    v = qkv[..., 2 * H:].reshape(B, S, num_heads, hd)
    q = ops.transpose(q, 1, 2) * (1.0 / math.sqrt(hd))
    k = ops.transpose(k, 1, 2)
    v = ops.transpose(v, 1, 2)
    scores = ops.matmul(q, ops.transpose(k, -1, -2))                      # [B, nh, S, S]
    scores = ops.where(key_keep[:, None, None, :], scores, ops.full_like(scores, -math.inf))
    p = softmax_last(ops, scores)
    o = ops.transpose(ops.matmul(p, v), 1, 2).reshape(B, S, H)
    return linear(ops, o, w_out, b_out)


def audio_encoder_layer(ops, x, key_keep, P, prefix: str, num_heads: int, eps: float):
    """AudioEncoderLayer.forward, mae.py:80-99 (pre-LN; DropPath/Dropout are identity at inference)."""
    h = layer_norm(ops, x, P[prefix + ".norm1.weight"], P[prefix + ".norm1.bias"], eps)
    h = _mha_self(ops, h, P[prefix + ".attn.in_proj_weight"], P[prefix + ".attn.in_proj_bias"],
                  P[prefix + ".attn.out_proj.weight"], P[prefix + ".attn.out_proj.bias"], num_heads, key_keep)
    x = x + h
    h = layer_norm(ops, x, P[prefix + ".norm2.weight"], P[prefix + ".norm2.bias"], eps)
    h = linear(ops, h, P[prefix + ".mlp.fc1.weight"], P[prefix + ".mlp.fc1.bias"])     # MLP.forward mae.py:55-61
    h = silu(ops, h)
    h = linear(ops, h, P[prefix + ".mlp.fc2.weight"], P[prefix + ".mlp.fc2.bias"])
    return x + h


def _pos_embeds(ops, P, prefix, time_inds, freq_inds, hidden):
    t = sin_cos_pos_embed(ops, time_inds, hidden)
    f = ops.take_rows(P[prefix + ".freq_positional_embedding"], ops.i64(ops.to_numpy(freq_inds).astype(np.int64)))
    return t, f


def audio_encoder(ops, P, cfg, patches, time_inds, freq_inds, mask, prefix: str = "audio_module",
                  probes: Optional[dict] = None):
    """AudioEncoder.forward, mae.py:125-148."""
    x = linear(ops, patches, P[prefix + ".input_proj.weight"], P[prefix + ".input_proj.bias"])
    t, f = _pos_embeds(ops, P, prefix, time_inds, freq_inds, cfg.hidden_size)
    x = x + t
    x = x + f
    keep = mask != 0
    for n in range(cfg.num_layers):
        x = audio_encoder_layer(ops, x, keep, P, f"{prefix}.layers.{n}", cfg.num_heads, cfg.layer_norm_eps)
        if probes is not None:
            probes[f"audio_layer{n}"] = ops.to_numpy(x)
    return layer_norm(ops, x, P[prefix + ".norm.weight"], P[prefix + ".norm.bias"], cfg.layer_norm_eps)


def audio_attention_pool(ops, P, hidden, mask, num_heads: int, prefix: str = "audio_attention_pool"):
    """AudioAttentionPooler.forward, src/caco_torch/caco.py:41-79."""
    B, S, H = hidden.shape
    hd = H // num_heads
    kv = linear(ops, hidden, P[prefix + ".kv_proj.weight"], P[prefix + ".kv_proj.bias"])
    k = kv[..., :H].reshape(B, S, num_heads, hd)
    v = kv[..., H:].reshape(B, S, num_heads, hd)
    q = P[prefix + ".query"].reshape(num_heads, hd) * (1.0 / math.sqrt(hd))
    # einsum('hd,bjhd->bhj')
    w = ops.matmul(ops.transpose(k, 1, 2), q[None, :, :, None])[..., 0]     # [B, nh, S]
    w = ops.where((mask != 0)[:, None, :], w, ops.full_like(w, -math.inf))
    w = softmax_last(ops, w)
    # einsum('bhj,bjhd->bhd')
    o = ops.matmul(w[:, :, None, :], ops.transpose(v, 1, 2))[:, :, 0, :]    # [B, nh, hd]
    o = o.reshape(B, H)
    return linear(ops, o, P[prefix + ".out_proj.weight"], P[prefix + ".out_proj.bias"])


# ----------------------------------------------------------------------------------------------
# text encoder (causal RoBERTa), pooler, projection
# ----------------------------------------------------------------------------------------------

def roberta_embeddings(ops, P, cfg, input_ids, position_ids, prefix: str):
    """RobertaEmbeddings.forward, src/caco_torch/text_models/roberta.py:35-53 (token_type_ids = 0)."""
    e = prefix + ".embeddings"
    x = ops.take_rows(P[e + ".word_embeddings.weight"], input_ids)
    x = x + ops.take_rows(P[e + ".position_embeddings.weight"], position_ids)
    x = x + P[e + ".token_type_embeddings.weight"][0]
    return layer_norm(ops, x, P[e + ".LayerNorm.weight"], P[e + ".LayerNorm.bias"], cfg.layer_norm_eps)


def roberta_layer(ops, P, cfg, x, bias, prefix: str):
    """RobertaLayer.forward (post-LN), roberta.py:67-104,114-124,155-158,168-178,191-215."""
    B, T, H = x.shape
    nh = cfg.num_attention_heads
    hd = H // nh
    a = prefix + ".attention.self"
    q = linear(ops, x, P[a + ".query.weight"], P[a + ".query.bias"]).reshape(B, T, nh, hd)
    k = linear(ops, x, P[a + ".key.weight"], P[a + ".key.bias"]).reshape(B, T, nh, hd)
    v = linear(ops, x, P[a + ".value.weight"], P[a + ".value.bias"]).reshape(B, T, nh, hd)
    q, k, v = ops.transpose(q, 1, 2), ops.transpose(k, 1, 2), ops.transpose(v, 1, 2)
    s = ops.matmul(q, ops.transpose(k, -1, -2)) / math.sqrt(hd) + bias
    p = softmax_last(ops, s)
    o = ops.transpose(ops.matmul(p, v), 1, 2).reshape(B, T, H)
    o = linear(ops, o, P[prefix + ".attention.output.dense.weight"], P[prefix + ".attention.output.dense.bias"])
    att = layer_norm(ops, o + x, P[prefix + ".attention.output.LayerNorm.weight"],
                     P[prefix + ".attention.output.LayerNorm.bias"], cfg.layer_norm_eps)
    m = gelu_erf(ops, linear(ops, att, P[prefix + ".intermediate.dense.weight"], P[prefix + ".intermediate.dense.bias"]))
    m = linear(ops, m, P[prefix + ".output.dense.weight"], P[prefix + ".output.dense.bias"])
    return layer_norm(ops, m + att, P[prefix + ".output.LayerNorm.weight"],
                      P[prefix + ".output.LayerNorm.bias"], cfg.layer_norm_eps)


def _roberta_attention(ops, P, cfg, x, kv, bias, prefix: str):
    """RobertaAttention.forward = RobertaSelfAttention (optionally over `key_value_states`) + RobertaSelfOutput,
    roberta.py:67-104, 114-124, 133-147.  x [B,T,H] supplies the queries and the residual; kv [B,S,H] keys / values."""
    B, T, H = x.shape
    S = kv.shape[1]
    nh = cfg.num_attention_heads
    hd = H // nh
    a = prefix + ".self"
    q = linear(ops, x, P[a + ".query.weight"], P[a + ".query.bias"]).reshape(B, T, nh, hd)
    k = linear(ops, kv, P[a + ".key.weight"], P[a + ".key.bias"]).reshape(B, S, nh, hd)
    v = linear(ops, kv, P[a + ".value.weight"], P[a + ".value.bias"]).reshape(B, S, nh, hd)
    q, k, v = ops.transpose(q, 1, 2), ops.transpose(k, 1, 2), ops.transpose(v, 1, 2)
    s_ = ops.matmul(q, ops.transpose(k, -1, -2)) / math.sqrt(hd) + bias
    p = softmax_last(ops, s_)
    o = ops.transpose(ops.matmul(p, v), 1, 2).reshape(B, T, H)
    o = linear(ops, o, P[prefix + ".output.dense.weight"], P[prefix + ".output.dense.bias"])
    return layer_norm(ops, o + x, P[prefix + ".output.LayerNorm.weight"], P[prefix + ".output.LayerNorm.bias"],
                      cfg.layer_norm_eps)


def roberta_decoder(ops, P, cfg, text_hidden, attention_mask, audio_hidden, audio_mask, prefix: str = "decoder_module"):
    """RobertaDecoder.forward, roberta.py:337-373: per layer (RobertaLayer with has_cross_attention, :191-215)
    self-attention under causal AND caption-padding mask (:347-356), cross-attention over the audio tokens with padded
    audio tokens at -inf (:358-362), GELU MLP, all post-LN; then decoder_proj (:371)."""
    B, T, H = text_hidden.shape
    allowed = ops.tril_bool(T)[None, None, :, :] & (attention_mask != 0)[:, None, None, :]
    zeros = ops.f32(np.zeros((B, 1, T, T), dtype=np.float32))
    self_bias = ops.where(allowed, zeros, ops.full_like(zeros, -math.inf))
    S = audio_hidden.shape[1]
    zc = ops.f32(np.zeros((B, 1, 1, S), dtype=np.float32))
    cross_bias = ops.where((audio_mask != 0)[:, None, None, :], zc, ops.full_like(zc, -math.inf))
    x = text_hidden
    for n in range(cfg.num_hidden_layers):
        p = f"{prefix}.encoder.layers.{n}"
        att = _roberta_attention(ops, P, cfg, x, x, self_bias, p + ".attention")
        att = _roberta_attention(ops, P, cfg, att, audio_hidden, cross_bias, p + ".crossattention")
        m = gelu_erf(ops, linear(ops, att, P[p + ".intermediate.dense.weight"], P[p + ".intermediate.dense.bias"]))
        m = linear(ops, m, P[p + ".output.dense.weight"], P[p + ".output.dense.bias"])
        x = layer_norm(ops, m + att, P[p + ".output.LayerNorm.weight"], P[p + ".output.LayerNorm.bias"], cfg.layer_norm_eps)
    return linear(ops, x, P[prefix + ".decoder_proj.weight"], P[prefix + ".decoder_proj.bias"])


def text_attention_pool(ops, P, hidden, mask, prefix: str):
    """AttentionPooler.forward, roberta.py:253-271."""
    H = hidden.shape[-1]
    key = linear(ops, hidden, P[prefix + ".key_proj.weight"], P[prefix + ".key_proj.bias"]) / math.sqrt(H)
    value = linear(ops, hidden, P[prefix + ".value_proj.weight"], P[prefix + ".value_proj.bias"])
    q = P[prefix + ".attention_pool_query"]                                    # [1, H]
    w = ops.matmul(key, ops.transpose(q, 0, 1))[..., 0]                      # [B, T]
    w = ops.where(mask != 0, w, ops.full_like(w, -math.inf))
    w = softmax_last(ops, w)
    return ops.matmul(w[:, None, :], value)[:, 0, :]


def roberta_model(ops, P, cfg, input_ids, attention_mask, position_ids=None, prefix: str = "text_module",
                  probes: Optional[dict] = None):
    """RobertaModel.forward, roberta.py:283-326: position_ids = arange(T) (:292-293), additive bias
    0 where (lower-triangular AND key not padded) else -inf (:297-310)."""
    B, T = input_ids.shape
    if position_ids is None:
        position_ids = ops.i64(np.broadcast_to(np.arange(T), (B, T)).copy())
    allowed = ops.tril_bool(T)[None, None, :, :] & (attention_mask != 0)[:, None, None, :]
    zeros = ops.f32(np.zeros((B, 1, T, T), dtype=np.float32))
    bias = ops.where(allowed, zeros, ops.full_like(zeros, -math.inf))
    x = roberta_embeddings(ops, P, cfg, input_ids, position_ids, prefix)
    if probes is not None:
        probes["text_embed"] = ops.to_numpy(x)
    for n in range(cfg.num_hidden_layers):
        x = roberta_layer(ops, P, cfg, x, bias, f"{prefix}.encoder.layers.{n}")
        if probes is not None:
            probes[f"text_layer{n}"] = ops.to_numpy(x)
    pooled = text_attention_pool(ops, P, x, attention_mask, prefix + ".pooler")
    return pooled, x


# ----------------------------------------------------------------------------------------------
# model API (CACO) and AudioMAE
# ----------------------------------------------------------------------------------------------

class CacoOracle:
    """Mirror of `CACO` (src/caco_torch/caco.py:82-261) over a reference-named state dict."""

    def __init__(self, state: Dict[str, np.ndarray], audio_cfg, text_cfg, caco_cfg, backend: str = "numpy",
                 decoder_cfg=None):
        self.ops = get_ops(backend)
        self.P = {k: self.ops.f32(v) for k, v in state.items()}
        self.audio_cfg, self.text_cfg, self.caco_cfg = audio_cfg, text_cfg, caco_cfg
        self.decoder_cfg = decoder_cfg
        self.logit_scale = float(np.asarray(state["logit_scale"]))

    def get_audio_embedding(self, audio_patches, audio_time_inds, audio_freq_inds, audio_mask,
                            return_hidden_state: bool = True, normalize: bool = False, probes=None):
        """caco.py:123-150"""
        o = self.ops
        mask = o.f32(audio_mask)
        hidden = audio_encoder(o, self.P, self.audio_cfg, o.f32(audio_patches), o.f32(audio_time_inds),
                               o.f32(audio_freq_inds), mask, probes=probes)
        emb = audio_attention_pool(o, self.P, hidden, mask, self.caco_cfg.num_attention_pool_heads)
        if normalize:
            emb = l2_normalize(o, emb)
        emb, hidden = o.to_numpy(emb), o.to_numpy(hidden)
        return (emb, hidden) if return_hidden_state else emb

    def get_text_embedding(self, text_input_ids, text_mask, position_ids=None,
                           return_hidden_state: bool = True, normalize: bool = False, probes=None):
        """caco.py:152-177"""
        o = self.ops
        pos = None if position_ids is None else o.i64(position_ids)
        pooled, hidden = roberta_model(o, self.P, self.text_cfg, o.i64(text_input_ids), o.i64(text_mask), pos,
                                       probes=probes)
        emb = linear(o, pooled, self.P["text_proj.weight"], self.P["text_proj.bias"])
        if normalize:
            emb = l2_normalize(o, emb)
        emb, hidden = o.to_numpy(emb), o.to_numpy(hidden)
        return (emb, hidden) if return_hidden_state else emb

    def get_contrastive_logits(self, audio_patches, audio_time_inds, audio_freq_inds, audio_mask,
                               text_input_ids, text_mask):
        """caco.py:179-212"""
        a = self.get_audio_embedding(audio_patches, audio_time_inds, audio_freq_inds, audio_mask,
                                     return_hidden_state=False, normalize=True)
        t = self.get_text_embedding(text_input_ids, text_mask, return_hidden_state=False, normalize=True)
        s = np.float32(math.exp(self.logit_scale))
        return (s * a) @ t.T, (s * t) @ a.T

    forward = get_contrastive_logits

    def get_decoder_logits(self, audio_hidden_state, audio_mask, text_input_ids, text_mask):
        """caco.py:212-240"""
        if self.decoder_cfg is None:
            raise ValueError("Decoder module not initialized")
        o = self.ops
        _, th = self.get_text_embedding(text_input_ids, text_mask)
        return o.to_numpy(roberta_decoder(o, self.P, self.decoder_cfg, o.f32(th), o.i64(text_mask),
                                          o.f32(audio_hidden_state), o.f32(audio_mask)))

    def greedy_decode(self, audio_hidden_state, audio_mask, max_decode_length: int, bos_id: int = 0, eos_id: int = 2,
                      pad_id: int = 1):
        """The decoding loop the reference intends (eval_caco_torch.py:412-461 with the call of caco.py:212-240; per-row
        stop as in the JAX loop src/caco/caco.py:180-199), arg-max instead of sampling.  Returns (ids [B, L], margins
        [B, L-1] = top-1 minus top-2 logit of every emitted token, for tests that compare against a bf16 path)."""
        b = np.asarray(audio_hidden_state).shape[0]
        gen = np.full((b, 1), bos_id, dtype=np.int64)
        alive = np.ones(b, dtype=bool)
        margins = []
        for _ in range(max_decode_length):
            lg = self.get_decoder_logits(audio_hidden_state, audio_mask, gen, np.ones_like(gen))[:, -1]
            top2 = np.sort(lg, axis=-1)[:, -2:]
            margins.append(top2[:, 1] - top2[:, 0])
            nxt = np.where(alive, lg.argmax(-1), pad_id)
            gen = np.concatenate([gen, nxt[:, None]], axis=1)
            alive &= nxt != eos_id
            if not alive.any():
                break
        return gen, np.stack(margins, axis=1)

    # convenience wrappers named in BASELINE.json north_star
    def encode_audio(self, wav, max_patches: int = 500):
        b = prepare_audio_batch(wav, max_patches, backend=self.ops.name)
        return self.get_audio_embedding(b["audio_patches"], b["audio_time_inds"], b["audio_freq_inds"],
                                        b["audio_mask"], return_hidden_state=False, normalize=True)

    def encode_text(self, ids, mask):
        return self.get_text_embedding(ids, mask, return_hidden_state=False, normalize=True)


def similarity(a: np.ndarray, t: np.ndarray, scale: float = 1.0) -> np.ndarray:
    """Unscaled cosine matrix of callers (src/eval/eval_caco_torch.py:398 uses T @ A^T; :330 scaled)."""
    return (np.float32(scale) * np.asarray(a, np.float32)) @ np.asarray(t, np.float32).T


class AudioMAEOracle:
    """`AudioMAE.forward` (src/caco_torch/audio_models/mae.py:217-247) = encoder on the visible
    patches, then `AudioDecoder.forward` (mae.py:166-207)."""

    def __init__(self, state, enc_cfg, dec_cfg, backend: str = "numpy"):
        self.ops = get_ops(backend)
        self.P = {k: self.ops.f32(v) for k, v in state.items()}
        self.enc_cfg, self.dec_cfg = enc_cfg, dec_cfg

    def forward(self, x, mask, time_inds, freq_inds, restore_time_inds, restore_freq_inds, restore_mask):
        o, P, dc = self.ops, self.P, self.dec_cfg
        x, mask = o.f32(x), o.f32(mask)
        time_inds, freq_inds = o.f32(time_inds), o.f32(freq_inds)
        rt, rf, rmask = o.f32(restore_time_inds), o.f32(restore_freq_inds), o.f32(restore_mask)
        h = audio_encoder(o, P, self.enc_cfg, x, time_inds, freq_inds, mask, prefix="encoder")
        h = linear(o, h, P["decoder.input_proj.weight"], P["decoder.input_proj.bias"])          # mae.py:177
        t, f = _pos_embeds(o, P, "decoder", time_inds, freq_inds, dc.hidden_size)
        h = h + t
        h = h + f                                                                               # :179-186
        rtp, rfp = _pos_embeds(o, P, "decoder", rt, rf, dc.hidden_size)
        r = P["decoder.restore_patch"][None, None, :] + rtp                                     # :188-190
        r = r + rfp                                                                             # :191-196
        h = o.cat([h, r], 1)                                                                    # :198
        m = o.cat([mask, rmask], 1)                                                             # :199
        keep = m != 0
        for n in range(dc.num_layers):
            h = audio_encoder_layer(o, h, keep, P, f"decoder.layers.{n}", dc.num_heads, dc.layer_norm_eps)
        h = layer_norm(o, h, P["decoder.norm.weight"], P["decoder.norm.bias"], dc.layer_norm_eps)
        return o.to_numpy(linear(o, h, P["decoder.output_proj.weight"], P["decoder.output_proj.bias"]))


# ----------------------------------------------------------------------------------------------
# retrieval scoring
# ----------------------------------------------------------------------------------------------
def argsort_desc(scores: np.ndarray, k: int = 10) -> np.ndarray:
    """First k columns of argsort(-scores, axis=-1): src/eval/eval_caco_torch.py:403,407 (`torch.argsort(-logits)`;
    only the first 10 columns are consumed, eval_utils.py:26).  Stable sort: ties keep ascending index order."""
    order = np.argsort(-np.asarray(scores, dtype=np.float64), axis=-1, kind="stable")
    return order[..., :k].astype(np.int32)


def jackknife_mean(data, confidence_level: float = 0.95):
    """astropy.stats.jackknife_stats(data, np.mean, 0.95) as called at src/eval/eval_utils.py:57-67 (astropy absent
    here; published algorithm: leave-one-out statistics, bias (n-1)(mean_jack - stat), normal-quantile interval)."""
    from scipy.special import erfinv
    x = np.asarray(data, dtype=np.float64)
    n = x.size
    stat = x.mean()
    jack = np.array([np.delete(x, i).mean() for i in range(n)])
    mean_jack = jack.mean()
    bias = (n - 1) * (mean_jack - stat)
    std_err = np.sqrt((n - 1) * np.mean((jack - mean_jack) ** 2))
    est = stat - bias
    z = np.sqrt(2.0) * erfinv(confidence_level)
    return float(est), float(bias), float(std_err), (float(est - z * std_err), float(est + z * std_err))


def retrieval_hits(indices, all_querys, all_keys, gt_query_key, retrieval_type="at"):
    """Per-query R1 / R5 / R10 / mAP10 lists of compute_retrieval_metric, src/eval/eval_utils.py:18-54."""
    R1, R5, R10, mAP10 = [], [], [], []
    for i, query in enumerate(all_querys):
        pred_keys = [all_keys[int(j)] for j in indices[i, :10]]            # :26
        if retrieval_type == "at":                                         # :28-39
            preds, taken = [], []
            for pred in pred_keys:
                ok = (pred not in taken) and (pred in gt_query_key[query])
                if ok:
                    taken.append(pred)
                preds.append(ok)
            preds = np.asarray(preds)
        else:                                                              # :41-42
            preds = np.asarray([gt_query_key[query] == pred for pred in pred_keys])
        R1.append(float(np.any(preds[:1])))                                # :45-47
        R5.append(float(np.any(preds[:5])))
        R10.append(float(np.any(preds[:10])))
        positions = np.arange(1, 11, dtype=float)[preds[:10] > 0]          # :49-54
        mAP10.append(float(np.mean(np.arange(1, len(positions) + 1, dtype=float) / positions)) if len(positions) else 0.0)
    return {"R1": R1, "R5": R5, "R10": R10, "mAP10": mAP10}


# ----------------------------------------------------------------------------------------------
# HEAR embeddings: src/eval/heareval/embeddings/audio_embedding/caco_embeddings.py
# ----------------------------------------------------------------------------------------------
def hear_event_embeddings(hidden: np.ndarray, audio_max_len: float = 10.0, group: int = 8):
    """caco_embeddings.py:118-124: tf.nn.avg_pool(hidden[b], ksize=8, strides=8, 'VALID') over the token axis
    ([B, S, H] -> [B, S // 8, H]) and timestamps = linspace(0, audio_max_len * 1000, S // 8) in ms.
    (The JAX reference is not importable here: parity for this function is UNPINNED by reference outputs; it is a
    direct restatement of two library calls whose semantics are unambiguous.)"""
    hidden = np.asarray(hidden, dtype=np.float32)
    b, s, h = hidden.shape
    n = s // group
    ev = hidden[:, :n * group].reshape(b, n, group, h).astype(np.float64).mean(axis=2).astype(np.float32)
    ts = np.linspace(0, audio_max_len * 1000, n)
    return ev, ts


def zs_topk_accuracy(audio_emb: np.ndarray, class_text_emb: np.ndarray, target_idx, logit_scale: float = 0.0, ks=(1,)):
    """src/eval/eval_caco_torch.py:326-340 for all clips: exp(logit_scale) * A @ T^T, argsort(-logits), target in first k."""
    logits = np.float32(math.exp(logit_scale)) * (np.asarray(audio_emb, np.float32) @ np.asarray(class_text_emb, np.float32).T)
    order = np.argsort(-logits, axis=-1, kind="stable")
    tgt = np.asarray(target_idx).reshape(-1)
    return {str(int(k)): float((order[:, :int(k)] == tgt[:, None]).any(axis=1).mean()) for k in ks}
