"""Caption decoding loop over the caption decoder: mirror of `decode_caption` (src/eval/eval_caco_torch.py:412-461)
and of the JAX `decode` (src/caco/caco.py:154-230).

The torch reference's loop calls `model.decoder_module(text_input_ids=..., audio_hidden=...)`, which does not match
`RobertaDecoder.forward`; the call it intends - and the one made here - is
`model.get_decoder_logits(audio_hidden, audio_mask, generated, ones)` (src/caco_torch/caco.py:212-240): the whole
prefix is re-embedded by the text tower and re-decoded every step, exactly as the reference does (no KV cache).
Per-row bookkeeping follows the JAX loop: a row that has emitted EOS stops generating and is padded.

Everything numeric runs on the MI355X path (`caco_text_forward`, `caco_decoder_forward`, `caco_topk` for the greedy
arg-max); only temperature sampling draws with `torch.multinomial`, as the reference does.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

import ctypes as C

from . import _lib, retrieval
from .model import CACO, _dev_tensor, _ptr, _stream


class CaptionDecodeState:
    """Key / value caches of one clip batch (C ABI: caco_decode_begin / caco_decode_step / caco_decode_end): the JAX
    path's incremental `get_next_decoder_logits` loop (src/caco/caco.py:154-230).  `step(tokens)` feeds one token per
    clip and returns the logits of the next position, equal to `get_decoder_logits(prefix)[:, -1]`."""

    def __init__(self, model: CACO, audio_hidden: torch.Tensor, audio_mask: torch.Tensor, max_len: int):
        if model.decoder_module is None:
            raise ValueError("Model does not have a decoder module. Load with use_decoder=True.")
        self.model, self._lib = model, _lib.load()
        ah = _dev_tensor(audio_hidden, torch.float32, model.device, "audio_hidden")
        am = _dev_tensor(audio_mask, torch.float32, model.device, "audio_mask")
        if ah.dim() != 3 or tuple(am.shape) != tuple(ah.shape[:2]):
            raise ValueError(f"CaptionDecodeState: expected [B, S, H] and [B, S], got {tuple(ah.shape)} / {tuple(am.shape)}")
        self.batch, self.vocab, self.max_len, self.pos = ah.shape[0], model.text_config.vocab_size, int(max_len), 0
        self._handle = C.c_void_p()
        with torch.cuda.device(model.device):
            _lib.check(self._lib.caco_decode_begin(model._handle, _ptr(ah), _ptr(am), ah.shape[0], ah.shape[1], int(max_len),
                                                   C.byref(self._handle), _stream()), "decode_begin")

    def step(self, tokens: torch.Tensor) -> torch.Tensor:
        tok = _dev_tensor(tokens, torch.int64, self.model.device, "tokens").reshape(-1)
        if tok.shape[0] != self.batch:
            raise ValueError(f"CaptionDecodeState.step: {tok.shape[0]} tokens for a batch of {self.batch}")
        logits = torch.empty(self.batch, self.vocab, dtype=torch.float32, device=self.model.device)
        with torch.cuda.device(self.model.device):
            _lib.check(self._lib.caco_decode_step(self._handle, _ptr(tok), _ptr(logits), _stream()), "decode_step")
        self.pos += 1
        return logits

    def close(self):
        h = getattr(self, "_handle", None)
        if h is not None and h.value:
            torch.cuda.synchronize(self.model.device)
            self._lib.caco_decode_end(h)
            self._handle = C.c_void_p()

    __del__ = close


@torch.no_grad()
def decode_caption_ids(model: CACO, audio_batch: Dict[str, torch.Tensor], max_decode_length: int = 100,
                       temperature: float = 0.1, bos_id: int = 0, eos_id: int = 2, pad_id: int = 1, greedy: bool = False,
                       generator: Optional[torch.Generator] = None, use_cache: bool = True) -> torch.Tensor:
    """audio_batch = the four tensors of `prepare_audio_batch`; returns int64 [B, L <= max_decode_length + 1] starting
    with BOS.  `greedy=True` takes the arg-max (what temperature -> 0 converges to; deterministic).  `use_cache=True`
    decodes incrementally with key / value caches (the JAX loop); False re-runs the whole prefix every step (the torch
    reference's loop) - same logits, O(T^2) work."""
    if model.decoder_module is None:
        raise ValueError("Model does not have a decoder module. Load with use_decoder=True.")     # eval_caco_torch.py:420-421
    _, audio_hidden = model.get_audio_embedding(audio_batch["audio_patches"], audio_batch["audio_time_inds"],
                                                audio_batch["audio_freq_inds"], audio_batch["audio_mask"],
                                                deterministic=True, return_hidden_state=True, normalize=False)
    B = audio_hidden.shape[0]
    dev = audio_hidden.device
    generated = torch.full((B, 1), bos_id, dtype=torch.long, device=dev)
    generating = torch.ones(B, dtype=torch.bool, device=dev)
    state = CaptionDecodeState(model, audio_hidden, audio_batch["audio_mask"], max_decode_length) if use_cache else None
    for _ in range(max_decode_length):
        if state is not None:
            last = state.step(generated[:, -1])                                                                 # [B, V]
        else:
            text_mask = torch.ones(generated.shape, dtype=torch.long, device=dev)
            last = model.get_decoder_logits(audio_hidden, audio_batch["audio_mask"], generated, text_mask)[:, -1, :]
        if greedy or temperature <= 0:
            nxt = retrieval.topk(last, 1)[0][:, 0].long()            # device arg-max (caco_topk, ties -> lowest id)
        else:
            probs = torch.softmax(last / temperature, dim=-1)          # eval_caco_torch.py:459-461
            nxt = torch.multinomial(probs, num_samples=1, generator=generator)[:, 0]
        nxt = torch.where(generating, nxt, torch.full_like(nxt, pad_id))
        generated = torch.cat([generated, nxt[:, None]], dim=1)
        generating = generating & (nxt != eos_id)
        if not bool(generating.any()):
            break
    if state is not None:
        state.close()
    return generated


@torch.no_grad()
def decode_caption(model: CACO, tokenizer, audio_batch: Dict[str, torch.Tensor], max_decode_length: int = 100,
                   temperature: float = 0.1, greedy: bool = False) -> str:
    """Same signature as the reference's `decode_caption`: the first clip's caption as a string."""
    ids = decode_caption_ids(model, audio_batch, max_decode_length, temperature, tokenizer.bos_token_id,
                             tokenizer.eos_token_id, tokenizer.pad_token_id, greedy)
    return tokenizer.batch_decode(ids, skip_special_tokens=True)[0].strip()
