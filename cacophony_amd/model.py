"""Host-side mirror of the reference model API over libcaco_hip.so.

`CACO` has the method names, keyword arguments, return conventions and error behaviour of
`src/caco_torch/caco.py:82-261`; `AudioMAE` those of `src/caco_torch/audio_models/mae.py:210-247`.
Tensors in and out are torch CUDA tensors (PyTorch-ROCm is only the container / stream provider);
every FLOP runs in the HIP library on torch's current stream.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import replace
from typing import Dict, Mapping, Optional, Tuple, Union

import numpy as np
import torch

from . import _lib
from .config import (AudioMAEConfig, AudioTransformerConfig, CACOConfig, RobertaConfig, default_audio_config,
                     default_caco_config, default_text_config)

Tensor = torch.Tensor


def _stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t: Optional[Tensor]) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


def _dev_tensor(x, dtype: torch.dtype, device: torch.device, name: str) -> Tensor:
    if not torch.is_tensor(x):
        x = torch.as_tensor(np.asarray(x))
    if x.device != device:
        x = x.to(device, non_blocking=True)
    if x.dtype != dtype:
        x = x.to(dtype)
    return x.contiguous()


def _patch_tensor(x, device) -> Tuple[Tensor, int]:
    if torch.is_tensor(x) and x.dtype == torch.bfloat16:
        return _dev_tensor(x, torch.bfloat16, device, "audio_patches"), _lib.DTYPE_BF16
    return _dev_tensor(x, torch.float32, device, "audio_patches"), _lib.DTYPE_F32


class _HipModel:
    """Owns one caco_model handle."""

    def __init__(self, audio_config: Optional[AudioTransformerConfig], text_config: Optional[RobertaConfig],
                 caco_config: CACOConfig, mae_decoder_layers: int = 0, device: Union[str, torch.device, None] = None,
                 caption_decoder_layers: int = 0):
        if not torch.cuda.is_available():
            raise RuntimeError("cacophony_amd needs a ROCm GPU (torch.cuda.is_available() is False); there is no CPU path")
        self._lib = _lib.load()
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        if self.device.type != "cuda":
            raise RuntimeError(f"cacophony_amd runs on MI355X only, got device '{self.device}'")
        if self.device.index is None:
            # 'cuda' / torch.device('cuda') (the reference CLI's idiom) means the current device; tensors allocated on it
            # report an index, and torch.device('cuda') != torch.device('cuda:0'), so the index is resolved once, here
            self.device = torch.device("cuda", torch.cuda.current_device())
        cfg = _lib.CacoConfigC()
        self._lib.caco_default_config(C.byref(cfg))
        a, t = audio_config, text_config
        cfg.has_audio, cfg.has_text = int(a is not None), int(t is not None)
        if a is not None:
            cfg.audio_hidden, cfg.audio_layers, cfg.audio_heads = a.hidden_size, a.num_layers, a.num_heads
            cfg.audio_intermediate, cfg.patch_size, cfg.num_freq_patches = a.intermediate_size, a.patch_size, a.num_freq_patches
            cfg.audio_ln_eps = a.layer_norm_eps
        if t is not None:
            cfg.text_vocab, cfg.text_hidden, cfg.text_layers = t.vocab_size, t.hidden_size, t.num_hidden_layers
            cfg.text_heads, cfg.text_intermediate = t.num_attention_heads, t.intermediate_size
            cfg.text_max_pos, cfg.text_type_vocab, cfg.text_ln_eps = t.max_position_embeddings, t.type_vocab_size, t.layer_norm_eps
        cfg.projection_size, cfg.pool_heads = caco_config.projection_size, caco_config.num_attention_pool_heads
        cfg.logit_scale = caco_config.logit_scale_init_value
        cfg.mae_decoder_layers = mae_decoder_layers
        cfg.caption_decoder_layers = caption_decoder_layers
        self._cfg_c = cfg
        self._handle = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self._lib.caco_create(C.byref(cfg), C.byref(self._handle)), "caco_create")
        self._loaded = False

    def __del__(self):
        h = getattr(self, "_handle", None)
        if h is not None and h.value:
            self._lib.caco_destroy(h)
            self._handle = C.c_void_p()

    def _load_state(self, state_dict: Mapping[str, object]) -> None:
        for key in ("model_state_dict", "state_dict"):     # checkpoint wrappers, src/eval/eval_caco_torch.py:160-166
            if key in state_dict and isinstance(state_dict[key], Mapping):
                state_dict = state_dict[key]
                break
        if self._loaded:
            raise RuntimeError("weights already loaded; create a new model to load another state dict")
        for name, value in state_dict.items():
            arr = value.detach().cpu().numpy() if torch.is_tensor(value) else np.asarray(value)
            arr = np.ascontiguousarray(arr, dtype=np.float32)
            shape = (C.c_int64 * max(arr.ndim, 1))(*arr.shape)
            _lib.check(self._lib.caco_load_tensor(self._handle, name.encode(), arr.ctypes.data_as(C.c_void_p), shape,
                                                  arr.ndim), f"load_state_dict[{name}]")
        with torch.cuda.device(self.device):
            _lib.check(self._lib.caco_finalize_weights(self._handle), "load_state_dict")
        self._loaded = True

    # nn.Module-flavoured no-ops so reference call sites (`model.to(device); model.eval()`) keep working
    def eval(self):
        return self

    def to(self, device=None, *_, **__):
        if device is not None and torch.device(device).type != "cuda":
            raise RuntimeError("cacophony_amd models live on the GPU only")
        return self

    def set_ln_fold(self, mode: int) -> int:
        """LayerNorm folding of THIS model's audio stack: 0 = separate LayerNorm passes (default), 1 = folded into the
        neighbouring GEMM epilogues, -1 = folded when the batch fills the chip.  Returns the mode in force."""
        return int(self._lib.caco_model_set_ln_fold(self._handle, int(mode)))

    @property
    def workspace_bytes(self) -> int:
        return int(self._lib.caco_workspace_bytes(self._handle))


class CACO(_HipModel):
    """Drop-in for `CACO` (src/caco_torch/caco.py:82).  Inference only: `deterministic=False` is rejected."""

    def __init__(self, audio_config: AudioTransformerConfig, text_config: RobertaConfig, caco_config: CACOConfig,
                 decoder_config: Optional[RobertaConfig] = None, device=None):
        if decoder_config is not None:
            # RobertaDecoder (text_models/roberta.py:329-373): this build shares the text tower's kernels and shapes
            same = ("vocab_size", "hidden_size", "num_attention_heads", "intermediate_size", "layer_norm_eps")
            if any(getattr(decoder_config, k) != getattr(text_config, k) for k in same):
                raise ValueError("decoder_config must match text_config in " + ", ".join(same))
        super().__init__(audio_config, text_config, caco_config, 0, device,
                         caption_decoder_layers=decoder_config.num_hidden_layers if decoder_config is not None else 0)
        self.audio_config, self.text_config, self.caco_config = audio_config, text_config, caco_config
        self.decoder_config = decoder_config
        # truthy when the caption decoder is present (reference call sites test `model.decoder_module is None`)
        self.decoder_module = self._decoder_forward if decoder_config is not None else None
        self.logit_scale = torch.tensor(caco_config.logit_scale_init_value, dtype=torch.float32, device=self.device)

    def load_state_dict(self, state_dict, strict: bool = True):
        self._load_state(state_dict)
        self.logit_scale = torch.tensor(float(self._lib.caco_get_logit_scale(self._handle)), dtype=torch.float32,
                                        device=self.device)
        return self

    @staticmethod
    def _inference_only(deterministic: bool):
        if not deterministic:
            raise ValueError("cacophony_amd is inference-only: deterministic=False (dropout) is not supported")

    def get_audio_embedding(self, audio_patches, audio_time_inds, audio_freq_inds, audio_mask, deterministic: bool = True,
                            return_hidden_state: bool = True, normalize: bool = False):
        """caco.py:123-150"""
        self._inference_only(deterministic)
        patches, pdt = _patch_tensor(audio_patches, self.device)
        if patches.dim() != 3 or patches.shape[-1] != self.audio_config.patch_size:
            raise ValueError(f"audio_patches must be [B, S, {self.audio_config.patch_size}], got {tuple(patches.shape)}")
        B, S, _ = patches.shape
        tinds = _dev_tensor(audio_time_inds, torch.float32, self.device, "audio_time_inds")
        finds = _dev_tensor(audio_freq_inds, torch.float32, self.device, "audio_freq_inds")
        mask = _dev_tensor(audio_mask, torch.float32, self.device, "audio_mask")
        for n, t in (("audio_time_inds", tinds), ("audio_freq_inds", finds), ("audio_mask", mask)):
            if tuple(t.shape) != (B, S):
                raise ValueError(f"{n} must be [{B}, {S}], got {tuple(t.shape)}")
        emb = torch.empty(B, self.caco_config.projection_size, dtype=torch.float32, device=self.device)
        hidden = torch.empty(B, S, self.audio_config.hidden_size, dtype=torch.float32, device=self.device) \
            if return_hidden_state else None
        with torch.cuda.device(self.device):
            _lib.check(self._lib.caco_audio_forward(self._handle, _ptr(patches), pdt, _ptr(tinds), _ptr(finds), _ptr(mask),
                                                    B, S, int(normalize), _ptr(emb), _ptr(hidden), _stream()),
                       "get_audio_embedding")
        return (emb, hidden) if return_hidden_state else emb

    def _check_token_ids(self, ids: Tensor, pos: Optional[Tensor]) -> None:
        """nn.Embedding raises IndexError on an out-of-range id (roberta.py:44-47); the embedding kernel would clamp it
        and return a plausible but wrong embedding.  One device reduction + host sync: callers on a hot path that
        already trust their tokenizer pass check_ids=False."""
        lo, hi = int(ids.min()), int(ids.max())
        if lo < 0 or hi >= self.text_config.vocab_size:
            raise IndexError(f"text_input_ids out of range [0, {self.text_config.vocab_size}): min {lo}, max {hi}")
        if pos is not None:
            lo, hi = int(pos.min()), int(pos.max())
            if lo < 0 or hi >= self.text_config.max_position_embeddings:
                raise IndexError(f"position_ids out of range [0, {self.text_config.max_position_embeddings}): min {lo}, max {hi}")

    def get_text_embedding(self, text_input_ids, text_mask, position_ids=None, deterministic: bool = True,
                           return_hidden_state: bool = True, normalize: bool = False, check_ids: bool = True):
        """caco.py:152-177"""
        self._inference_only(deterministic)
        ids = _dev_tensor(text_input_ids, torch.int64, self.device, "text_input_ids")
        mask = _dev_tensor(text_mask, torch.int64, self.device, "text_mask")
        if ids.dim() != 2 or ids.shape != mask.shape:
            raise ValueError(f"text_input_ids / text_mask must both be [B, T], got {tuple(ids.shape)} / {tuple(mask.shape)}")
        B, T = ids.shape
        pos = None
        if position_ids is not None:
            pos = _dev_tensor(position_ids, torch.int64, self.device, "position_ids")
            if pos.shape != ids.shape:
                pos = pos.expand(B, T).contiguous()
        if check_ids:
            self._check_token_ids(ids, pos)
        emb = torch.empty(B, self.caco_config.projection_size, dtype=torch.float32, device=self.device)
        hidden = torch.empty(B, T, self.text_config.hidden_size, dtype=torch.float32, device=self.device) \
            if return_hidden_state else None
        with torch.cuda.device(self.device):
            _lib.check(self._lib.caco_text_forward(self._handle, _ptr(ids), _ptr(mask), _ptr(pos), B, T, int(normalize),
                                                   _ptr(emb), _ptr(hidden), _stream()), "get_text_embedding")
        return (emb, hidden) if return_hidden_state else emb

    def similarity(self, audio_embedding: Tensor, text_embedding: Tensor, scale: float = 1.0) -> Tensor:
        """scale * A @ T^T on the fp32 MFMA kernel (caco.py:208-210, eval_caco_torch.py:330,398)."""
        return similarity(audio_embedding, text_embedding, scale)

    def get_contrastive_logits(self, audio_patches, audio_time_inds, audio_freq_inds, audio_mask, text_input_ids,
                               text_mask, deterministic: bool = True):
        """caco.py:179-212.  The second matrix is the exact transpose of the first, returned as a view."""
        a = self.get_audio_embedding(audio_patches, audio_time_inds, audio_freq_inds, audio_mask, deterministic,
                                     return_hidden_state=False, normalize=True)
        t = self.get_text_embedding(text_input_ids, text_mask, None, deterministic, return_hidden_state=False,
                                    normalize=True)
        at = similarity(a, t, float(torch.exp(self.logit_scale)))
        return at, at.T

    def _decoder_forward(self, text_hidden_state, attention_mask, audio_hidden_state, audio_mask, deterministic: bool = True):
        """RobertaDecoder.forward (text_models/roberta.py:337-373): [B,T,H], [B,T], [B,S,H], [B,S] -> logits [B,T,vocab]."""
        self._inference_only(deterministic)
        th = _dev_tensor(text_hidden_state, torch.float32, self.device, "text_hidden_state")
        tm = _dev_tensor(attention_mask, torch.int64, self.device, "attention_mask")
        ah = _dev_tensor(audio_hidden_state, torch.float32, self.device, "audio_hidden_state")
        am = _dev_tensor(audio_mask, torch.float32, self.device, "audio_mask")
        H = self.text_config.hidden_size
        if th.dim() != 3 or ah.dim() != 3 or th.shape[2] != H or ah.shape[2] != H or th.shape[0] != ah.shape[0]:
            raise ValueError(f"decoder: hidden states must be [B, T, {H}] and [B, S, {H}], got {tuple(th.shape)} / {tuple(ah.shape)}")
        B, T, _ = th.shape
        S = ah.shape[1]
        if tuple(tm.shape) != (B, T) or tuple(am.shape) != (B, S):
            raise ValueError(f"decoder: masks must be [{B}, {T}] and [{B}, {S}], got {tuple(tm.shape)} / {tuple(am.shape)}")
        logits = torch.empty(B, T, self.text_config.vocab_size, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.caco_decoder_forward(self._handle, _ptr(th), _ptr(tm), _ptr(ah), _ptr(am), B, T, S,
                                                      _ptr(logits), _stream()), "decoder")
        return logits

    def get_decoder_logits(self, audio_hidden_state, audio_mask, text_input_ids, text_mask, deterministic: bool = True):
        """caco.py:212-240: caption-prefix hidden states from the text tower, then the cross-attending decoder."""
        if self.decoder_module is None:
            raise ValueError("Decoder module not initialized")      # caco.py:223-224
        _, text_hidden_state = self.get_text_embedding(text_input_ids=text_input_ids, text_mask=text_mask,
                                                       deterministic=deterministic)
        return self._decoder_forward(text_hidden_state, text_mask, audio_hidden_state, audio_mask, deterministic)

    def forward(self, audio_patches, audio_time_inds, audio_freq_inds, audio_mask, text_input_ids, text_mask,
                deterministic: bool = True):
        return self.get_contrastive_logits(audio_patches, audio_time_inds, audio_freq_inds, audio_mask, text_input_ids,
                                           text_mask, deterministic)

    __call__ = forward

    # ---- convenience wrappers named in BASELINE.json north_star ---------------------------------
    @staticmethod
    def _out_rows(out: Optional[Tensor], B: int, P: int, device, what: str) -> Tuple[Tensor, int]:
        """`out`: fp32 [B, P] whose rows may be strided (e.g. bank[:, 0, :] of a packed [B, 2, P] exchange buffer)."""
        if out is None:
            return torch.empty(B, P, dtype=torch.float32, device=device), 0
        if tuple(out.shape) != (B, P) or out.dtype != torch.float32 or out.stride(1) != 1 or out.stride(0) < P or out.device != device:
            raise ValueError(f"{what}: `out` must be fp32 [{B}, {P}] on {device} with unit column stride")
        return out, int(out.stride(0))

    def encode_audio(self, wav, max_patches: Optional[int] = None, out: Optional[Tensor] = None, lengths=None) -> Tensor:
        """wav fp32 [B, n_samples] (16 kHz) -> L2-normalised audio embeddings [B, projection_size].
        = prepare_audio_batch (eval_caco_torch.py:181-206) + get_audio_embedding(normalize=True), all on device.
        `lengths` int64 [B]: real samples per clip when the clips differ in length (rows zero-padded to n_samples); each
        clip is then masked exactly as the reference masks it when it pre-processes that clip alone."""
        wav = _dev_tensor(wav, torch.float32, self.device, "wav")
        if wav.dim() == 1:
            wav = wav[None]
        B, n = wav.shape
        if max_patches is None:
            max_patches = max(8, n * 8 // 160 // 16)    # patches_seq_len rule, eval_caco_torch.py:573,607-612
        lens = None
        if lengths is not None:
            lens = _dev_tensor(lengths, torch.int64, self.device, "lengths")
            if tuple(lens.shape) != (B,):
                raise ValueError(f"lengths must be [{B}], got {tuple(lens.shape)}")
        emb, ld = self._out_rows(out, B, self.caco_config.projection_size, self.device, "encode_audio")
        with torch.cuda.device(self.device):
            _lib.check(self._lib.caco_encode_audio_ex(self._handle, _ptr(wav), _ptr(lens), B, n, int(max_patches), _ptr(emb), ld,
                                                      _stream()), "encode_audio")
        return emb

    def encode_text(self, text_input_ids, text_mask, out: Optional[Tensor] = None, check_ids: bool = True) -> Tensor:
        """ids / mask int64 [B, T] -> L2-normalised text embeddings [B, projection_size] (get_text_embedding, normalize=True)."""
        ids = _dev_tensor(text_input_ids, torch.int64, self.device, "text_input_ids")
        mask = _dev_tensor(text_mask, torch.int64, self.device, "text_mask")
        if ids.dim() != 2 or ids.shape != mask.shape:
            raise ValueError(f"text_input_ids / text_mask must both be [B, T], got {tuple(ids.shape)} / {tuple(mask.shape)}")
        if check_ids:
            self._check_token_ids(ids, None)
        B, T = ids.shape
        emb, ld = self._out_rows(out, B, self.caco_config.projection_size, self.device, "encode_text")
        with torch.cuda.device(self.device):
            _lib.check(self._lib.caco_encode_text(self._handle, _ptr(ids), _ptr(mask), B, T, _ptr(emb), ld, _stream()), "encode_text")
        return emb

    def encode_pairs(self, wav, text_input_ids, text_mask, max_patches: Optional[int] = None, audio_streams: int = 1,
                     lengths=None, packed: bool = False, check_ids: bool = False):
        """Embed clips and captions CONCURRENTLY: the text tower runs on a side stream next to the audio tower, and
        the clip batch is split over `audio_streams` streams.  The library keeps one workspace per (tower, stream),
        so the launches of different streams interleave on the GPU: one stream's memory-bound kernels (LayerNorm,
        attention, GEMM epilogues) fill the HBM time another stream's MFMA-bound GEMM leaves idle.  Same results
        as encode_audio + encode_text.  Returns (audio_emb [B,P], text_emb [B,P]), valid on the current stream;
        packed=True: both towers write into ONE fp32 [B, 2, P] buffer (the all-gather payload of the data-parallel
        path, dist.gather_packed) and that buffer is returned instead.  Token ids are not range-checked here unless
        check_ids (a host sync per call): this is the throughput path."""
        wav = _dev_tensor(wav, torch.float32, self.device, "wav")
        if wav.dim() == 1:
            wav = wav[None]
        B = wav.shape[0]
        text_input_ids = _dev_tensor(text_input_ids, torch.int64, self.device, "text_input_ids")    # lists / arrays as well
        text_mask = _dev_tensor(text_mask, torch.int64, self.device, "text_mask")
        if text_input_ids.dim() != 2 or text_input_ids.shape != text_mask.shape:
            raise ValueError(f"text_input_ids / text_mask must both be [B, T], got {tuple(text_input_ids.shape)} / {tuple(text_mask.shape)}")
        Bt = int(text_input_ids.shape[0])
        P = self.caco_config.projection_size
        if packed and Bt != B:
            raise ValueError(f"encode_pairs(packed=True) needs as many captions as clips, got {Bt} vs {B}")
        lens = None if lengths is None else _dev_tensor(lengths, torch.int64, self.device, "lengths")
        if lens is not None and tuple(lens.shape) != (B,):
            raise ValueError(f"lengths must be [{B}], got {tuple(lens.shape)}")
        n_split = max(1, min(int(audio_streams), B))
        with torch.cuda.device(self.device):
            cur = torch.cuda.current_stream()
            pool = self._side_streams(n_split)            # n_split - 1 audio side streams + 1 text stream
            if packed:
                bank = torch.empty(B, 2, P, dtype=torch.float32, device=self.device)
                ea, et = bank[:, 0, :], bank[:, 1, :]
            else:
                bank = None
                ea = torch.empty(B, P, dtype=torch.float32, device=self.device)
                et = torch.empty(Bt, P, dtype=torch.float32, device=self.device)
            bounds = [B * i // n_split for i in range(n_split + 1)]
            for st in pool:
                st.wait_stream(cur)
            with torch.cuda.stream(pool[-1]):
                self.encode_text(text_input_ids, text_mask, out=et, check_ids=check_ids)
            for i in range(n_split):
                lo, hi = bounds[i], bounds[i + 1]
                ln = None if lens is None else lens[lo:hi]
                if i == 0:
                    self.encode_audio(wav[lo:hi], max_patches, out=ea[lo:hi], lengths=ln)
                else:
                    with torch.cuda.stream(pool[i - 1]):
                        self.encode_audio(wav[lo:hi], max_patches, out=ea[lo:hi], lengths=ln)
            for st in pool:
                cur.wait_stream(st)
        return bank if packed else (ea, et)

    def _side_streams(self, n_split: int):
        key = max(1, n_split)
        cache = self.__dict__.setdefault("_streams", {})
        if key not in cache:
            # (stream priorities were tried for the text stream: no measurable effect on the step)
            cache[key] = [torch.cuda.Stream(device=self.device) for _ in range(key)]
        return cache[key]


def _bank(x: Tensor, what: str) -> Tensor:
    """fp32 [N, D] with unit column stride and a row stride that is a multiple of 4 (a slice of a packed buffer is fine)."""
    if x.dtype != torch.float32:
        x = x.to(torch.float32)
    if x.dim() != 2:
        raise ValueError(f"similarity: {what} must be [N, D], got {tuple(x.shape)}")
    if x.stride(1) != 1 or x.stride(0) < x.shape[1] or x.stride(0) % 4 or x.data_ptr() % 16:
        x = x.contiguous()
    return x


def similarity(a: Tensor, t: Tensor, scale: float = 1.0, out: Optional[Tensor] = None) -> Tensor:
    """out[i, j] = scale * <a_i, t_j>, fp32 in / fp32 MFMA / fp32 out (C ABI: caco_similarity_ld).  Row-strided banks
    (views into a packed [N, 2, D] exchange buffer) are read in place."""
    lib = _lib.load()
    if a.device.type != "cuda" or t.device != a.device:
        raise RuntimeError("similarity: both embedding banks must be on the same GPU")
    a, t = _bank(a, "audio bank"), _bank(t, "text bank")
    if a.shape[1] != t.shape[1]:
        raise ValueError(f"similarity: expected [Na, D] and [Nt, D], got {tuple(a.shape)} and {tuple(t.shape)}")
    if out is None:
        out = torch.empty(a.shape[0], t.shape[0], dtype=torch.float32, device=a.device)
    with torch.cuda.device(a.device):
        _lib.check(lib.caco_similarity_ld(_ptr(a), a.shape[0], a.stride(0), _ptr(t), t.shape[0], t.stride(0), a.shape[1],
                                          float(scale), _ptr(out), out.stride(0), _stream()), "similarity")
    return out


def l2_normalize(x: Tensor) -> Tensor:
    lib = _lib.load()
    x = x.to(torch.float32).contiguous()
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _lib.check(lib.caco_l2_normalize(_ptr(x), x.shape[0], x.shape[1], _ptr(out), _stream()), "l2_normalize")
    return out


def create_caco_model(device=None, use_decoder: bool = False) -> CACO:
    """Default-configuration model, src/caco_torch/caco.py:264-317.  The reference always builds the 4-layer caption
    decoder; here it is opt-in (`use_decoder=True`): the embedding / scoring path never touches it, and a state dict's
    `decoder_module.*` tensors are skipped when the model was built without it."""
    dec = replace(default_text_config(), num_hidden_layers=4) if use_decoder else None     # caco.py:297-309
    return CACO(default_audio_config(), default_text_config(), default_caco_config(), decoder_config=dec, device=device)


class AudioMAE(_HipModel):
    """Drop-in for `AudioMAE` (src/caco_torch/audio_models/mae.py:210-247): encoder on visible patches + decoder."""

    def __init__(self, config: AudioMAEConfig, device=None):
        e, d = config.encoder_config, config.decoder_config
        if (d.hidden_size, d.num_heads, d.intermediate_size, d.patch_size, d.num_freq_patches) != \
                (e.hidden_size, e.num_heads, e.intermediate_size, e.patch_size, e.num_freq_patches):
            raise ValueError("AudioMAE: this build needs matching encoder / decoder widths (README.md:55-61 config)")
        super().__init__(e, None, default_caco_config(), d.num_layers, device)
        self.config = config

    def load_state_dict(self, state_dict, strict: bool = True):
        self._load_state(state_dict)
        return self

    def forward(self, x, mask, time_inds, freq_inds, restore_time_inds, restore_freq_inds, restore_mask,
                deterministic: bool = True) -> Tensor:
        CACO._inference_only(deterministic)
        patches, pdt = _patch_tensor(x, self.device)
        B, V, _ = patches.shape
        f = lambda t, n: _dev_tensor(t, torch.float32, self.device, n)
        mask, time_inds, freq_inds = f(mask, "mask"), f(time_inds, "time_inds"), f(freq_inds, "freq_inds")
        rt, rf, rm = f(restore_time_inds, "restore_time_inds"), f(restore_freq_inds, "restore_freq_inds"), f(restore_mask, "restore_mask")
        R = rt.shape[1]
        out = torch.empty(B, V + R, self.config.decoder_config.patch_size, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.caco_mae_forward(self._handle, _ptr(patches), pdt, _ptr(mask), _ptr(time_inds), _ptr(freq_inds),
                                                  _ptr(rt), _ptr(rf), _ptr(rm), B, V, R, _ptr(out), _stream()), "AudioMAE.forward")
        return out

    __call__ = forward
