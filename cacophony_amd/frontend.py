"""Pre-processing free functions of the reference (src/eval/eval_caco_torch.py:41-227) over the HIP front end.

`compute_mel_spectrogram` and `prepare_audio_batch` run the fused HIP kernel (no host round trip);
`spectrogram_to_patches` is the reference's pure host-side reshuffle for callers that already hold a
host spectrogram; `prepare_text_batch` forwards to the caller's tokenizer exactly like the reference.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib
from .config import DatasetConfig
from .model import _dev_tensor, _ptr, _stream

_FIXED = dict(sr=16000, hop_length=160, win_length=400, n_fft=512, n_mels=128)


def _check_geometry(**kw):
    bad = {k: v for k, v in kw.items() if _FIXED[k] != v}
    if bad:
        raise ValueError(f"the HIP front end is specialised for {_FIXED}; unsupported override {bad}")


def _device(device=None) -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError("cacophony_amd front end needs a ROCm GPU; there is no CPU path")
    return torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")


def mel_spectrogram_device(wav: torch.Tensor, scale: float = 0.2, bias: float = 0.9) -> torch.Tensor:
    """wav fp32 [B, n] on the GPU -> log-mel fp32 [B, ceil(n/160), 128] on the GPU."""
    lib = _lib.load()
    wav = _dev_tensor(wav, torch.float32, wav.device if torch.is_tensor(wav) and wav.is_cuda else _device(), "wav")
    if wav.dim() == 1:
        wav = wav[None]
    B, n = wav.shape
    if n == 0:
        raise ValueError("empty waveform")
    frames = int(lib.caco_mel_num_frames(n))
    mel = torch.empty(B, frames, 128, dtype=torch.float32, device=wav.device)
    with torch.cuda.device(wav.device):
        _lib.check(lib.caco_mel_spectrogram(_ptr(wav), B, n, scale, bias, _ptr(mel), _stream()), "compute_mel_spectrogram")
    return mel


def compute_mel_spectrogram(audio, sr: int = 16000, hop_length: int = 160, win_length: int = 400, n_fft: int = 512,
                            n_mels: int = 128, scale: float = 0.2, bias: float = 0.9) -> np.ndarray:
    """Same signature and return type as src/eval/eval_caco_torch.py:41-105: one clip -> np.ndarray [T, 128]."""
    _check_geometry(sr=sr, hop_length=hop_length, win_length=win_length, n_fft=n_fft, n_mels=n_mels)
    audio = audio if torch.is_tensor(audio) else torch.as_tensor(np.asarray(audio))
    audio = audio.squeeze()
    if audio.dim() != 1:
        raise ValueError(f"compute_mel_spectrogram takes one clip, got shape {tuple(audio.shape)}")
    dev = audio.device if audio.is_cuda else _device()
    return mel_spectrogram_device(audio.to(dev, torch.float32)[None], scale, bias)[0].cpu().numpy()


def spectrogram_to_patches(spectrogram: np.ndarray, time_patch_size: int = 16, freq_patch_size: int = 16,
                           max_patches: int = 512) -> Dict[str, np.ndarray]:
    """src/eval/eval_caco_torch.py:108-151 (host-side data movement, as in the reference)."""
    spectrogram = np.asarray(spectrogram, dtype=np.float32)
    n_t = spectrogram.shape[0] // time_patch_size
    n_f = spectrogram.shape[1] // freq_patch_size
    full = n_t * n_f
    x = spectrogram[: n_t * time_patch_size, : n_f * freq_patch_size]
    x = x.reshape(n_t, time_patch_size, n_f, freq_patch_size).transpose(0, 2, 1, 3)
    x = x.reshape(full, time_patch_size * freq_patch_size)
    pos = np.arange(max_patches)
    if full > max_patches:
        x, mask = x[:max_patches], np.ones(max_patches, dtype=np.float32)
        kept = pos
    else:
        mask = (pos < full).astype(np.float32)
        kept = (mask * pos).astype(np.int64)
        x = np.concatenate([x, np.zeros((max_patches - full, x.shape[1]), np.float32)], axis=0)
    return {"audio_patches": x.astype(np.float32), "audio_time_inds": (kept // n_f).astype(np.float32),
            "audio_freq_inds": (kept % n_f).astype(np.float32), "audio_mask": mask}


def mel_patches_device(wav: torch.Tensor, max_patches: int, dtype: torch.dtype = torch.float32, scale: float = 0.2,
                       bias: float = 0.9, lengths=None) -> Dict[str, torch.Tensor]:
    """Fused waveform -> patch batch on the GPU: wav [B, n] -> the four tensors get_audio_embedding takes.
    `lengths` int64 [B]: real samples per clip when the rows are zero-padded clips of different lengths; clip b then gets
    exactly the patches / indices / mask of prepare_audio_batch (eval_caco_torch.py:181-206) run on that clip alone."""
    lib = _lib.load()
    if dtype not in (torch.float32, torch.bfloat16):
        raise ValueError("patch dtype must be float32 or bfloat16")
    dev = wav.device if torch.is_tensor(wav) and wav.is_cuda else _device()
    wav = _dev_tensor(wav, torch.float32, dev, "wav")
    if wav.dim() == 1:
        wav = wav[None]
    B, n = wav.shape
    if n == 0 or max_patches <= 0:
        raise ValueError("empty waveform or non-positive max_patches")
    patches = torch.empty(B, max_patches, 256, dtype=dtype, device=dev)
    tinds = torch.empty(B, max_patches, dtype=torch.float32, device=dev)
    finds = torch.empty_like(tinds)
    mask = torch.empty_like(tinds)
    lens = None
    if lengths is not None:
        lens = _dev_tensor(lengths, torch.int64, dev, "lengths")
        if tuple(lens.shape) != (B,):
            raise ValueError(f"lengths must be [{B}], got {tuple(lens.shape)}")
    with torch.cuda.device(dev):
        _lib.check(lib.caco_mel_patches_lens(_ptr(wav), _ptr(lens), B, n, max_patches, scale, bias, _ptr(patches),
                                             _lib.DTYPE_BF16 if dtype == torch.bfloat16 else _lib.DTYPE_F32, _ptr(tinds),
                                             _ptr(finds), _ptr(mask), _stream()), "mel_patches")
    return {"audio_patches": patches, "audio_time_inds": tinds, "audio_freq_inds": finds, "audio_mask": mask}


def prepare_audio_batch(audio, datasetconfig: DatasetConfig, device=None, lengths=None) -> Dict[str, torch.Tensor]:
    """src/eval/eval_caco_torch.py:181-206 (one clip -> batch of one).  Also accepts a batch [B, n], or a LIST of clips
    of different lengths (padded here, each masked at its own length, as the reference does clip by clip); `lengths`
    gives the real sample counts of an already padded [B, n] batch."""
    if datasetconfig.time_patch_size != 16 or datasetconfig.freq_patch_size != 16:
        raise ValueError("the HIP front end is specialised for 16 x 16 patches")
    if isinstance(audio, (list, tuple)) and len(audio) > 0 and np.ndim(audio[0]) == 1:
        clips = [c if torch.is_tensor(c) else torch.as_tensor(np.asarray(c)) for c in audio]
        lengths = torch.tensor([int(c.numel()) for c in clips], dtype=torch.int64)
        n = int(lengths.max())
        audio = torch.zeros(len(clips), n, dtype=torch.float32)
        for i, c in enumerate(clips):
            audio[i, : c.numel()] = c.to(torch.float32).cpu()
    audio = audio if torch.is_tensor(audio) else torch.as_tensor(np.asarray(audio))
    if audio.dim() == 1:
        audio = audio[None]
    return mel_patches_device(audio.to(_device(device), torch.float32), datasetconfig.patches_seq_len, lengths=lengths)


def prepare_text_batch(text: str, tokenizer, max_text_len: int, device=None) -> Dict[str, torch.Tensor]:
    """src/eval/eval_caco_torch.py:209-227: the tokenizer is the caller's (RobertaTokenizerFast in the reference)."""
    tok = tokenizer([text], padding="max_length", truncation=True, max_length=max_text_len, return_tensors="pt")
    dev = _device(device)
    return {"text_input_ids": tok["input_ids"].to(dev), "text_mask": tok["attention_mask"].to(dev)}
