"""HEAR-benchmark embedding wrapper: scene and timestamp ("event") embeddings of the audio tower.

Mirror of `Embedding` in src/eval/heareval/embeddings/audio_embedding/caco_embeddings.py:41-131 (the reference's copy
runs the JAX model under `pmap`; this one runs the MI355X path):

* scene embedding  = the L2-normalised pooled audio embedding (`get_audio_embedding(normalize=True)[0]`, `:129-131`);
* event embeddings = the encoder's hidden states averaged over the 8 frequency patches of each 160 ms time step
  (`tf.nn.avg_pool(hidden, ksize=8, strides=8, padding='VALID')`, `:118-124`) with
  `timestamps = linspace(0, audio_max_len * 1000, n_steps)` in milliseconds.

The reference decodes one wav file per call; here a whole batch of clips goes through `caco_mel_patches`,
`caco_audio_forward` and `caco_token_group_mean` on the device.  File decoding / resampling stays with the caller.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np
import torch

from . import _lib
from .frontend import mel_patches_device
from .model import CACO, _dev_tensor, _ptr, _stream

FREQ_PATCHES = 8          # spec_num_mels // freq_patch_size = 128 // 16 (caco_embeddings.py:73)


def token_group_mean(hidden: torch.Tensor, group: int = FREQ_PATCHES) -> torch.Tensor:
    """[B, S, H] fp32 -> [B, S // group, H]: mean over `group` consecutive tokens (C ABI: caco_token_group_mean)."""
    lib = _lib.load()
    if hidden.device.type != "cuda":
        raise RuntimeError("token_group_mean: hidden states must be on the GPU")
    if hidden.dim() != 3:
        raise ValueError(f"token_group_mean: expected [B, S, H], got {tuple(hidden.shape)}")
    if group < 1:
        raise ValueError("token_group_mean: group must be >= 1")
    hidden = hidden.to(torch.float32).contiguous()
    B, S, H = hidden.shape
    out = torch.empty(B, S // group, H, dtype=torch.float32, device=hidden.device)
    with torch.cuda.device(hidden.device):
        _lib.check(lib.caco_token_group_mean(_ptr(hidden), B, S, H, int(group), _ptr(out), _stream()), "token_group_mean")
    return out


class Embedding:
    """`Embedding(model_path, audio_max_len, batch_size, sample_rate)` of the reference, over an already-built model."""

    def __init__(self, model: CACO, audio_max_len: float = 10, sample_rate: int = 16000):
        self.model = model
        self.audio_max_len = audio_max_len
        self.sample_rate = sample_rate
        seg = int(audio_max_len * sample_rate)
        # maximum usable patches (caco_embeddings.py:72-73): (segment // hop // time_patch) * (mels // freq_patch)
        self.max_patches = (seg // 160 // 16) * FREQ_PATCHES
        self.segment_len = seg

    def _forward(self, wav) -> Tuple[torch.Tensor, torch.Tensor]:
        wav = _dev_tensor(wav, torch.float32, self.model.device, "wav")
        if wav.dim() == 1:
            wav = wav[None]
        if wav.shape[1] > self.segment_len:
            wav = wav[:, :self.segment_len].contiguous()
        batch = mel_patches_device(wav, self.max_patches)
        return self.model.get_audio_embedding(**batch, deterministic=True, return_hidden_state=True, normalize=True)

    def get_scene_embeddings(self, wav) -> torch.Tensor:
        """[B, n_samples] -> [B, projection_size], L2-normalised."""
        return self._forward(wav)[0]

    def get_timestamp_embeddings(self, wav) -> Tuple[torch.Tensor, torch.Tensor]:
        """[B, n_samples] -> (embeddings [B, n_steps, hidden], timestamps_ms [n_steps])."""
        _, hidden = self._forward(wav)
        ev = token_group_mean(hidden, FREQ_PATCHES)
        ts = torch.linspace(0, self.audio_max_len * 1000, ev.shape[1], dtype=torch.float64)
        return ev, ts

    def get_embedding_as_numpy(self, wav, embedding_type: Optional[str] = None):
        """The reference's entry point (`:97-131`), one clip: 'event' -> (emb [1, n_steps, H], [timestamps]), else emb [P]."""
        if embedding_type == "event":
            ev, ts = self.get_timestamp_embeddings(wav)
            return ev[:1].cpu().numpy(), [ts.numpy()]
        return self.get_scene_embeddings(wav)[0].cpu().numpy()
