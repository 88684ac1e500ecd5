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

from . import retrieval
from .model import CACO


@torch.no_grad()
def decode_caption_ids(model: CACO, audio_batch: Dict[str, torch.Tensor], max_decode_length: int = 100,
                       temperature: float = 0.1, bos_id: int = 0, eos_id: int = 2, pad_id: int = 1, greedy: bool = False,
                       generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """audio_batch = the four tensors of `prepare_audio_batch`; returns int64 [B, L <= max_decode_length + 1] starting
    with BOS.  `greedy=True` takes the arg-max (what temperature -> 0 converges to; deterministic)."""
    if model.decoder_module is None:
        raise ValueError("Model does not have a decoder module. Load with use_decoder=True.")     # eval_caco_torch.py:420-421
    _, audio_hidden = model.get_audio_embedding(audio_batch["audio_patches"], audio_batch["audio_time_inds"],
                                                audio_batch["audio_freq_inds"], audio_batch["audio_mask"],
                                                deterministic=True, return_hidden_state=True, normalize=False)
    B = audio_hidden.shape[0]
    dev = audio_hidden.device
    generated = torch.full((B, 1), bos_id, dtype=torch.long, device=dev)
    generating = torch.ones(B, dtype=torch.bool, device=dev)
    for _ in range(max_decode_length):
        text_mask = torch.ones(generated.shape, dtype=torch.long, device=dev)
        logits = model.get_decoder_logits(audio_hidden, audio_batch["audio_mask"], generated, text_mask)   # [B, T, V]
        last = logits[:, -1, :]
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
    return generated


@torch.no_grad()
def decode_caption(model: CACO, tokenizer, audio_batch: Dict[str, torch.Tensor], max_decode_length: int = 100,
                   temperature: float = 0.1, greedy: bool = False) -> str:
    """Same signature as the reference's `decode_caption`: the first clip's caption as a string."""
    ids = decode_caption_ids(model, audio_batch, max_decode_length, temperature, tokenizer.bos_token_id,
                             tokenizer.eos_token_id, tokenizer.pad_token_id, greedy)
    return tokenizer.batch_decode(ids, skip_special_tokens=True)[0].strip()
