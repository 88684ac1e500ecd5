"""Hyper-parameter dataclasses: the reference's config surface for the hot path.

Field names and defaults mirror `AudioTransformerConfig` (src/caco_torch/audio_models/mae.py:9-20),
`RobertaConfig` (src/caco_torch/text_models/roberta.py:11-23), `CACOConfig`
(src/caco_torch/caco.py:17-21) and `DatasetConfig` (src/eval/eval_caco_torch.py:30-38); the default
factories reproduce `create_caco_model()` (src/caco_torch/caco.py:264-317).
"""
from __future__ import annotations

from dataclasses import dataclass, replace


@dataclass
class AudioTransformerConfig:
    hidden_size: int = 768
    num_layers: int = 12
    num_heads: int = 8
    intermediate_size: int = 3072
    patch_size: int = 256
    max_time_ind: int = 512          # unused by the forward
    num_freq_patches: int = 8
    dropout_rate: float = 0.0
    drop_path_rate: float = 0.0
    layer_norm_eps: float = 1e-5     # torch nn.LayerNorm default


@dataclass
class AudioMAEConfig:
    encoder_config: AudioTransformerConfig
    decoder_config: AudioTransformerConfig


@dataclass
class RobertaConfig:
    vocab_size: int = 50265
    hidden_size: int = 768
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    intermediate_size: int = 3072
    hidden_dropout_prob: float = 0.1
    attention_probs_dropout_prob: float = 0.1
    max_position_embeddings: int = 514
    type_vocab_size: int = 1
    layer_norm_eps: float = 1e-5
    pad_token_id: int = 1


@dataclass
class CACOConfig:
    projection_size: int = 768
    num_attention_pool_heads: int = 2
    logit_scale_init_value: float = 2.6592


@dataclass
class DatasetConfig:
    batch_size: int = 1
    patches_seq_len: int = 512
    time_patch_size: int = 16
    freq_patch_size: int = 16
    max_text_len: int = 100
    synthetic_prob: float = 0.8


@dataclass
class MelConfig:
    """compute_mel_spectrogram keyword defaults (src/eval/eval_caco_torch.py:41-50)."""
    sr: int = 16000
    hop_length: int = 160
    win_length: int = 400
    n_fft: int = 512
    n_mels: int = 128
    scale: float = 0.2
    bias: float = 0.9
    log_eps: float = 1e-5


def default_audio_config() -> AudioTransformerConfig:
    return AudioTransformerConfig()


def default_text_config() -> RobertaConfig:
    return RobertaConfig()


def default_caco_config() -> CACOConfig:
    return CACOConfig()


def tiny_configs(layers: int = 2):
    """Full-width, few-layer configs used by layer-exact debugging goldens and fast tests."""
    a = replace(default_audio_config(), num_layers=layers)
    t = replace(default_text_config(), num_hidden_layers=layers, vocab_size=1024)
    return a, t, default_caco_config()
