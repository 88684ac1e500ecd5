"""Deterministic synthetic weights and inputs for the Cacophony hot path.

No pretrained checkpoint or tokenizer vocabulary is available offline, so parity
is established on seeded random weights and synthetic clips / captions
(SURVEY.md section 8c/8d).  Everything here is a pure function of integer seeds
through a counter-based splitmix64 hash, so the same tensors are regenerated
bit-for-bit in this container (where the reference is imported to make the
golden fixtures) and on the GPU box (where only this generator travels).  It
does not depend on torch's or numpy's RNG streams.

State-dict key names and shapes are the reference's weight contract
(src/caco_torch/caco.py:100-121, audio_models/mae.py:116-123,
text_models/roberta.py:29-32,62-64,110-111,153,164-165,249-251).
"""
from __future__ import annotations

import zlib
from collections import OrderedDict
from typing import Dict, Optional, Tuple

import numpy as np

from .config import AudioTransformerConfig, CACOConfig, RobertaConfig

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = x + _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def _stream_key(name: str, seed: int) -> np.uint64:
    h = zlib.crc32(name.encode("utf-8")) & 0xFFFFFFFF
    return np.uint64(((int(seed) & 0xFFFFFFFF) << 32) | h)


def hash_uniform(name: str, n: int, seed: int = 0, lane: int = 0) -> np.ndarray:
    """n float64 uniforms in (0, 1), a pure function of (name, seed, lane, index)."""
    with np.errstate(over="ignore"):
        base = _splitmix64(np.array([_stream_key(name, seed) + np.uint64(lane) * _M1], dtype=np.uint64))[0]
        idx = np.arange(n, dtype=np.uint64)
        bits = _splitmix64(idx * _GOLDEN + base)
    return ((bits >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def hash_normal(name: str, n: int, seed: int = 0) -> np.ndarray:
    """n float64 standard normals (Box-Muller over two hashed uniform lanes)."""
    u1 = hash_uniform(name, n, seed, lane=1)
    u2 = hash_uniform(name, n, seed, lane=2)
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


def _uniform(name, shape, bound, seed):
    n = int(np.prod(shape)) if len(shape) else 1
    return ((hash_uniform(name, n, seed) * 2.0 - 1.0) * bound).astype(np.float32).reshape(shape)


def _normal(name, shape, std, seed, mean=0.0):
    n = int(np.prod(shape)) if len(shape) else 1
    return (hash_normal(name, n, seed) * std + mean).astype(np.float32).reshape(shape)


def _linear(sd, prefix, out_f, in_f, seed):
    bound = 1.0 / np.sqrt(in_f)
    sd[prefix + ".weight"] = _uniform(prefix + ".weight", (out_f, in_f), bound, seed)
    sd[prefix + ".bias"] = _uniform(prefix + ".bias", (out_f,), bound, seed)


def _layernorm(sd, prefix, h, seed):
    # non-trivial affine so that a dropped gamma/beta shows up in parity checks
    sd[prefix + ".weight"] = _normal(prefix + ".weight", (h,), 0.1, seed, mean=1.0)
    sd[prefix + ".bias"] = _normal(prefix + ".bias", (h,), 0.05, seed)


def _audio_layers(sd, prefix, cfg: AudioTransformerConfig, seed):
    h, i = cfg.hidden_size, cfg.intermediate_size
    for n in range(cfg.num_layers):
        p = f"{prefix}.layers.{n}"
        _layernorm(sd, p + ".norm1", h, seed)
        bound = 1.0 / np.sqrt(h)
        sd[p + ".attn.in_proj_weight"] = _uniform(p + ".attn.in_proj_weight", (3 * h, h), bound, seed)
        sd[p + ".attn.in_proj_bias"] = _uniform(p + ".attn.in_proj_bias", (3 * h,), bound, seed)
        _linear(sd, p + ".attn.out_proj", h, h, seed)
        _layernorm(sd, p + ".norm2", h, seed)
        _linear(sd, p + ".mlp.fc1", i, h, seed)
        _linear(sd, p + ".mlp.fc2", h, i, seed)


def make_audio_encoder_state(cfg: AudioTransformerConfig, prefix: str = "audio_module", seed: int = 0):
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()
    _linear(sd, prefix + ".input_proj", cfg.hidden_size, cfg.patch_size, seed)
    sd[prefix + ".freq_positional_embedding"] = _normal(
        prefix + ".freq_positional_embedding", (cfg.num_freq_patches, cfg.hidden_size), 0.02, seed)
    _audio_layers(sd, prefix, cfg, seed)
    _layernorm(sd, prefix + ".norm", cfg.hidden_size, seed)
    return sd


def make_audio_decoder_state(cfg: AudioTransformerConfig, prefix: str = "decoder", seed: int = 0):
    """AudioDecoder parameters (src/caco_torch/audio_models/mae.py:152-164)."""
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()
    _linear(sd, prefix + ".input_proj", cfg.hidden_size, cfg.hidden_size, seed)
    sd[prefix + ".freq_positional_embedding"] = _normal(
        prefix + ".freq_positional_embedding", (cfg.num_freq_patches, cfg.hidden_size), 0.02, seed)
    sd[prefix + ".restore_patch"] = _normal(prefix + ".restore_patch", (cfg.hidden_size,), 0.02, seed)
    _audio_layers(sd, prefix, cfg, seed)
    _layernorm(sd, prefix + ".norm", cfg.hidden_size, seed)
    _linear(sd, prefix + ".output_proj", cfg.patch_size, cfg.hidden_size, seed)
    return sd


def make_audiomae_state(enc_cfg: AudioTransformerConfig, dec_cfg: AudioTransformerConfig, seed: int = 0):
    sd = make_audio_encoder_state(enc_cfg, "encoder", seed)
    sd.update(make_audio_decoder_state(dec_cfg, "decoder", seed))
    return sd


def make_text_encoder_state(cfg: RobertaConfig, prefix: str = "text_module", seed: int = 0):
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()
    h, i = cfg.hidden_size, cfg.intermediate_size
    e = prefix + ".embeddings"
    sd[e + ".word_embeddings.weight"] = _normal(e + ".word_embeddings.weight", (cfg.vocab_size, h), 0.05, seed)
    sd[e + ".position_embeddings.weight"] = _normal(
        e + ".position_embeddings.weight", (cfg.max_position_embeddings, h), 0.05, seed)
    sd[e + ".token_type_embeddings.weight"] = _normal(
        e + ".token_type_embeddings.weight", (cfg.type_vocab_size, h), 0.05, seed)
    _layernorm(sd, e + ".LayerNorm", h, seed)
    for n in range(cfg.num_hidden_layers):
        p = f"{prefix}.encoder.layers.{n}"
        _linear(sd, p + ".attention.self.query", h, h, seed)
        _linear(sd, p + ".attention.self.key", h, h, seed)
        _linear(sd, p + ".attention.self.value", h, h, seed)
        _linear(sd, p + ".attention.output.dense", h, h, seed)
        _layernorm(sd, p + ".attention.output.LayerNorm", h, seed)
        _linear(sd, p + ".intermediate.dense", i, h, seed)
        _linear(sd, p + ".output.dense", h, i, seed)
        _layernorm(sd, p + ".output.LayerNorm", h, seed)
    pl = prefix + ".pooler"
    sd[pl + ".attention_pool_query"] = _normal(pl + ".attention_pool_query", (1, h), 0.5, seed)
    _linear(sd, pl + ".key_proj", h, h, seed)
    _linear(sd, pl + ".value_proj", h, h, seed)
    return sd


def make_text_decoder_state(cfg: RobertaConfig, prefix: str = "decoder_module", seed: int = 0):
    """`decoder_module.*` of a CACO state dict: RobertaDecoder, src/caco_torch/text_models/roberta.py:329-335."""
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()
    h, i = cfg.hidden_size, cfg.intermediate_size
    for n in range(cfg.num_hidden_layers):
        p = f"{prefix}.encoder.layers.{n}"
        for att in ("attention", "crossattention"):
            _linear(sd, f"{p}.{att}.self.query", h, h, seed)
            _linear(sd, f"{p}.{att}.self.key", h, h, seed)
            _linear(sd, f"{p}.{att}.self.value", h, h, seed)
            _linear(sd, f"{p}.{att}.output.dense", h, h, seed)
            _layernorm(sd, f"{p}.{att}.output.LayerNorm", h, seed)
        _linear(sd, p + ".intermediate.dense", i, h, seed)
        _linear(sd, p + ".output.dense", h, i, seed)
        _layernorm(sd, p + ".output.LayerNorm", h, seed)
    _linear(sd, prefix + ".decoder_proj", cfg.vocab_size, h, seed)
    return sd


def make_caco_state(audio_cfg: AudioTransformerConfig, text_cfg: RobertaConfig, caco_cfg: CACOConfig,
                    seed: int = 0, decoder_cfg: Optional[RobertaConfig] = None) -> "OrderedDict[str, np.ndarray]":
    """Full CACO state dict; `decoder_module.*` (the caption decoder) only when `decoder_cfg` is given."""
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()
    sd["logit_scale"] = np.array(caco_cfg.logit_scale_init_value, dtype=np.float32)
    sd.update(make_audio_encoder_state(audio_cfg, "audio_module", seed))
    h = audio_cfg.hidden_size
    p = "audio_attention_pool"
    # pool query std 0.5 (not the init's 0.02): a near-zero query makes the pooling softmax uniform
    # and the check blind to mask / scale errors.
    sd[p + ".query"] = _normal(p + ".query", (h,), 0.5, seed)
    _linear(sd, p + ".kv_proj", 2 * h, h, seed)
    _linear(sd, p + ".out_proj", caco_cfg.projection_size, h, seed)
    sd.update(make_text_encoder_state(text_cfg, "text_module", seed))
    _linear(sd, "text_proj", caco_cfg.projection_size, text_cfg.hidden_size, seed)
    if decoder_cfg is not None:
        sd.update(make_text_decoder_state(decoder_cfg, "decoder_module", seed))
    return sd


# ----------------------------------------------------------------------------------------------
# synthetic inputs
# ----------------------------------------------------------------------------------------------

def make_waveform(index: int, n_samples: int = 160000, sr: int = 16000) -> np.ndarray:
    """One structured clip: 3-6 enveloped sinusoids / linear chirps + coloured noise, peak 0.5.

    iid noise is avoided on purpose: with random-init weights it collapses all audio embeddings to
    one direction (pairwise cosine 0.99985) and makes cosine parity checks vacuous.
    """
    seed = 1000 + index
    u = hash_uniform("wave.params", 64, seed)
    dur = n_samples / sr
    t = np.arange(n_samples, dtype=np.float64) / sr
    n_comp = 3 + int(u[0] * 4)  # 3..6
    x = np.zeros(n_samples, dtype=np.float64)
    for c in range(n_comp):
        q = u[1 + 8 * c: 9 + 8 * c]
        f0 = 50.0 + q[0] * 7450.0
        f1 = f0 if q[1] < 0.5 else 50.0 + q[2] * 7450.0
        onset = q[3] * max(dur - 0.2, 0.0)
        length = 0.2 + q[4] * 2.8
        amp = 0.2 + 0.8 * q[5]
        ph = 2.0 * np.pi * q[6]
        tt = np.clip(t - onset, 0.0, length)
        k = (f1 - f0) / max(length, 1e-6)
        phase = 2.0 * np.pi * (f0 * tt + 0.5 * k * tt * tt) + ph
        env = np.where((t >= onset) & (t <= onset + length),
                       np.sin(np.pi * np.clip((t - onset) / length, 0.0, 1.0)) ** 2, 0.0)
        x += amp * env * np.sin(phase)
    noise = hash_normal("wave.noise", n_samples, seed)
    a = 0.5 + 0.45 * u[60]
    # one-pole low-pass for colour: y[n] = a*y[n-1] + (1-a)*x[n], float64
    from scipy.signal import lfilter
    col = lfilter([1.0 - a], [1.0, -a], noise)
    gain_db = -40.0 + 30.0 * u[61]
    col = col / (np.std(col) + 1e-12) * (10.0 ** (gain_db / 20.0))
    x = x + col
    x = 0.5 * x / (np.max(np.abs(x)) + 1e-12)
    return x.astype(np.float32)


def make_waveforms(batch: int, n_samples: int = 160000, start: int = 0) -> np.ndarray:
    return np.stack([make_waveform(start + i, n_samples) for i in range(batch)], axis=0)


def make_captions(batch: int, max_len: int = 32, vocab_size: int = 50265, start: int = 0,
                  min_len: int = 8) -> Tuple[np.ndarray, np.ndarray]:
    """Synthetic RoBERTa-style ids/mask: <s>=0 first, </s>=2 last valid, <pad>=1 tail.

    Row i is full length when i % 4 == 0 and padded otherwise, so both the pure-causal and the
    causal-and-padding mask paths (text_models/roberta.py:297-310) are exercised.
    """
    ids = np.ones((batch, max_len), dtype=np.int64)
    mask = np.zeros((batch, max_len), dtype=np.int64)
    for i in range(batch):
        seed = 2000 + start + i
        u = hash_uniform("caption", max_len + 1, seed)
        lo = min(min_len, max_len)
        length = max_len if (start + i) % 4 == 0 else lo + int(u[max_len] * (max_len - lo))
        body = 3 + (u[:max_len] * (vocab_size - 3)).astype(np.int64)
        ids[i, :length] = body[:length]
        ids[i, 0] = 0
        ids[i, length - 1] = 2
        mask[i, :length] = 1
    return ids, mask


def make_mae_split(batch: int, n_patches: int = 496, n_visible: int = 100, num_freq_patches: int = 8,
                   start: int = 0) -> Dict[str, np.ndarray]:
    """Visible / restore index sets for the AudioMAE stage-1 forward (mae.py:217-247).

    The reference holds no masking code (no training loop), so the 100 / 396 split is this build's
    stated assumption: a hashed permutation of the patch grid.
    """
    vis = np.zeros((batch, n_visible), dtype=np.int64)
    res = np.zeros((batch, n_patches - n_visible), dtype=np.int64)
    for i in range(batch):
        u = hash_uniform("mae.perm", n_patches, 3000 + start + i)
        perm = np.argsort(u, kind="stable")
        vis[i] = np.sort(perm[:n_visible])
        res[i] = np.sort(perm[n_visible:])
    return {
        "visible": vis, "restore": res,
        "time_inds": (vis // num_freq_patches).astype(np.float32),
        "freq_inds": (vis % num_freq_patches).astype(np.float32),
        "restore_time_inds": (res // num_freq_patches).astype(np.float32),
        "restore_freq_inds": (res % num_freq_patches).astype(np.float32),
    }


def make_retrieval_scenario(seed: int = 7, n_audio: int = 40, caps_per: int = 3, dim: int = 32, signal: float = 0.3):
    """Seeded synthetic retrieval set in the reference's bookkeeping (src/eval/eval_caco_torch.py:355-395): clip names,
    caption strings (one string occurs under two clips, one string twice under the same clip), the two ground-truth
    dicts, and L2-normalised audio / text embeddings whose similarity is informative but far from perfect."""
    rng = np.random.RandomState(seed)
    all_audio = [f"clip_{i:03d}.wav" for i in range(n_audio)]
    all_text, gt_audio_text, gt_text_audio = [], {a: [] for a in all_audio}, {}
    for i, a in enumerate(all_audio):
        for c in range(caps_per):
            s = f"caption {i}-{c}"
            if i in (5, 17) and c == 0:
                s = "a dog barks"                 # same string under two clips: the later clip owns it in gt_text_audio
            if i == 9 and c == 2:
                s = "caption 9-1"                 # duplicate string inside one clip's list
            gt_audio_text[a].append(s)
            gt_text_audio[s] = a
            all_text.append(s)
    A = rng.randn(n_audio, dim).astype(np.float32)
    T = np.repeat(A, caps_per, 0) * signal + rng.randn(n_audio * caps_per, dim).astype(np.float32)
    A /= np.linalg.norm(A, axis=1, keepdims=True)
    T /= np.linalg.norm(T, axis=1, keepdims=True)
    return all_audio, all_text, gt_audio_text, gt_text_audio, A, T
