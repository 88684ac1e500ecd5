"""Data-parallel scoring: shard clips / captions over ranks, one all-gather, local similarity row block.

Each clip and caption is embedded independently (no cross-sample op anywhere in the encoders), so the
only exchange on the path is the CLIP-style gather of both L2-normalised embedding banks before the
global similarity matrix.  One process per GPU; `torch.distributed` backend
"nccl" is RCCL over xGMI on ROCm; the same host logic runs on "gloo" for the CPU tests.

The reference has no multi-GPU torch path; its precedent is the JAX `pmap('dp')` with replicated
parameters (src/eval/eval_caco.py:53-64) and the host-side concat before `T @ A.T`
(src/eval/eval_caco_torch.py:394-398).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n_global: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of rank `rank`; earlier ranks take the remainder."""
    base, rem = divmod(n_global, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pack_banks(audio_emb: torch.Tensor, text_emb: torch.Tensor) -> torch.Tensor:
    """[B, D] x 2 -> one send buffer [B, 2, D] fp32 (1.57 MB per rank at B = 256, D = 768).  Only for callers that hold
    two separate banks: CACO.encode_pairs(packed=True) lets the towers write the packed buffer directly."""
    if audio_emb.shape != text_emb.shape:
        raise ValueError(f"bank shapes differ: {tuple(audio_emb.shape)} vs {tuple(text_emb.shape)}")
    return torch.stack([audio_emb.float(), text_emb.float()], dim=1).contiguous()


def _check_equal_shards(b: int, device, group) -> None:
    """all_gather_into_tensor needs the same B on every rank (shard_range hands earlier ranks one more item when the
    global count does not divide): a mismatch would hang or corrupt, so it is checked with one tiny all-reduce pair."""
    t = torch.tensor([b, -b], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    if int(t[0]) != b or int(-t[1]) != b:
        raise ValueError(f"gather: this rank holds {b} rows but ranks range from {int(-t[1])} to {int(t[0])}; pad the shards to one size")


def gather_packed(bank: torch.Tensor, group: Optional[dist.ProcessGroup] = None, always_communicate: bool = False,
                  check_sizes: bool = False) -> torch.Tensor:
    """ONE all-gather of the packed banks: [B, 2, D] per rank -> [W*B, 2, D] in rank order (no other device work: the
    similarity kernel reads the two banks through their row stride).  A single-rank job returns `bank` itself unless
    `always_communicate` (used by the 1-GPU test of the RCCL call)."""
    if bank.dim() != 3 or bank.shape[1] != 2 or not bank.is_contiguous() or bank.dtype != torch.float32:
        raise ValueError(f"gather_packed: expected a contiguous fp32 [B, 2, D] buffer, got {tuple(bank.shape)} {bank.dtype}")
    if not dist.is_available() or not dist.is_initialized() or (dist.get_world_size(group) == 1 and not always_communicate):
        return bank
    world = dist.get_world_size(group)
    if check_sizes:
        _check_equal_shards(bank.shape[0], bank.device, group)
    recv = torch.empty((world * bank.shape[0], 2, bank.shape[2]), dtype=bank.dtype, device=bank.device)   # rank-major
    dist.all_gather_into_tensor(recv, bank, group=group)
    return recv


def gather_embedding_banks(audio_emb: torch.Tensor, text_emb: torch.Tensor, group: Optional[dist.ProcessGroup] = None,
                           always_communicate: bool = False, check_sizes: bool = True) -> Tuple[torch.Tensor, torch.Tensor]:
    """Two separate banks -> (A_all [W*B, D], T_all [W*B, D]) in rank order, through ONE all-gather; the results are
    strided views of the receive buffer (no copies).  Every rank must contribute the same B (checked)."""
    if not dist.is_available() or not dist.is_initialized() or (dist.get_world_size(group) == 1 and not always_communicate):
        return audio_emb.float(), text_emb.float()
    recv = gather_packed(pack_banks(audio_emb, text_emb), group, always_communicate, check_sizes)
    return recv[:, 0, :], recv[:, 1, :]


def sharded_similarity(audio_emb: torch.Tensor, text_emb: torch.Tensor, scale: float = 1.0,
                       group: Optional[dist.ProcessGroup] = None, similarity_fn=None) -> torch.Tensor:
    """This rank's row block  scale * A_local @ T_all^T  of the global [W*B, W*B] matrix.

    `similarity_fn(a, t, scale)` defaults to the HIP fp32-MFMA kernel; the gloo tests inject a checker."""
    _, t_all = gather_embedding_banks(audio_emb, text_emb, group)
    if similarity_fn is None:
        from .model import similarity as similarity_fn      # HIP kernel; raises without the GPU library
    return similarity_fn(audio_emb.float(), t_all, scale)
