"""Retrieval scoring on top of the similarity matrix.

Reference: `audio_retrieval` (src/eval/eval_caco_torch.py:394-408) concatenates the embeddings, forms
`logits_ar = T @ A.T`, takes a full `argsort` per direction, copies the [N, N] index matrices to the host and calls
`compute_retrieval_metric` (src/eval/eval_utils.py:18-54), which only ever reads the first 10 columns.

Here the device does the selection (`topk`, C ABI `caco_topk`: one wave per row, no full sort, both directions on
the single stored matrix through strides) and only [N, 10] int32 indices travel to the host, where
`compute_retrieval_metric` restates the reference's string/dict bookkeeping, recall@{1,5,10} and mAP@10 with the same
argument meaning.  The jackknife confidence interval (astropy in the reference) is restated in NumPy.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, Mapping, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib

Tensor = torch.Tensor


def topk(sim: Tensor, k: int = 10, dim: int = -1) -> Tuple[Tensor, Tensor]:
    """First k columns of `argsort(-sim, dim)` (value descending, ties by ascending index) and the values.
    sim fp32 [R, C] on the GPU (any strides); dim = -1/1 ranks each row's columns, dim = 0 each column's rows
    (= the `logits_ar.T` direction, without materialising the transpose).  Returns (indices int32, values fp32)."""
    if sim.dim() != 2 or sim.dtype != torch.float32 or sim.device.type != "cuda":
        raise ValueError("topk: expected a 2-D fp32 CUDA tensor")
    if dim not in (-1, 1, 0):
        raise ValueError("topk: dim must be 0 or 1")
    lib = _lib.load()
    rs, cs = sim.stride(0), sim.stride(1)
    rows, cols = sim.shape
    if dim == 0:
        rows, cols, rs, cs = cols, rows, cs, rs
    k = min(int(k), cols)          # argsort(-sim)[:, :k] has min(k, cols) columns; never hand out -1 padding for cols < k
    if k < 1:
        raise ValueError("topk: k must be >= 1 and the ranked axis non-empty")
    idx = torch.empty(rows, k, dtype=torch.int32, device=sim.device)
    val = torch.empty(rows, k, dtype=torch.float32, device=sim.device)
    with torch.cuda.device(sim.device):
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(lib.caco_topk(C.c_void_p(sim.data_ptr()), rows, cols, rs, cs, int(k), C.c_void_p(idx.data_ptr()),
                                 C.c_void_p(val.data_ptr()), st), "topk")
    return idx, val


def jackknife_stats(data: np.ndarray, confidence_level: float = 0.95) -> Tuple[float, float, float, Tuple[float, float]]:
    """Jackknife estimate of the mean: (bias-corrected estimate, bias, standard error, confidence interval).
    Restates `astropy.stats.jackknife_stats(data, np.mean, 0.95)` as called at src/eval/eval_utils.py:57-67."""
    x = np.asarray(data, dtype=np.float64)
    n = x.size
    if n < 2:
        raise ValueError("jackknife_stats needs at least two samples")
    stat = x.mean()
    jack = (x.sum() - x) / (n - 1)                 # leave-one-out means
    mean_jack = jack.mean()
    bias = (n - 1) * (mean_jack - stat)
    std_err = math.sqrt((n - 1) * np.mean((jack - mean_jack) ** 2))
    estimate = stat - bias
    # two-sided normal quantile: sqrt(2) * erfinv(confidence_level)
    from scipy.special import erfinv
    z = math.sqrt(2.0) * float(erfinv(confidence_level))
    return float(estimate), float(bias), float(std_err), (float(estimate - z * std_err), float(estimate + z * std_err))


def compute_retrieval_metric(indices, all_querys: Sequence, all_keys: Sequence, gt_query_key: Mapping,
                             retrieval_type: str = "at", verbose: bool = False) -> Dict[str, object]:
    """src/eval/eval_utils.py:18-67 with the same arguments; `indices` is [num_queries, >= 10] (tensor or array,
    e.g. the output of `topk`).  Returns the per-query lists and the jackknife summaries instead of only printing."""
    if torch.is_tensor(indices):
        indices = indices.cpu().numpy()
    indices = np.asarray(indices)
    R1, R5, R10, mAP10 = [], [], [], []
    for i, query in enumerate(all_querys):
        # caco_topk pads a row with -1 when it has fewer than k selectable columns (cols < k, NaNs): the reference's
        # indices[i, :10] is simply shorter there - a -1 must never index all_keys (Python would wrap to the LAST key)
        pred_keys = [all_keys[int(idx)] for idx in indices[i, :10] if int(idx) >= 0]
        if retrieval_type == "at":
            preds, seen = [], []
            for pred in pred_keys:                      # a caption string counts once, and only if it belongs to this clip
                if pred not in seen and pred in gt_query_key[query]:
                    seen.append(pred)
                    preds.append(True)
                else:
                    preds.append(False)
            preds = np.asarray(preds)
        elif retrieval_type == "ta":
            preds = np.asarray([gt_query_key[query] == pred for pred in pred_keys])
        else:
            raise ValueError(f"retrieval_type must be 'at' or 'ta', got {retrieval_type!r}")
        R1.append(float(np.any(preds[:1])))
        R5.append(float(np.any(preds[:5])))
        R10.append(float(np.any(preds[:10])))
        positions = np.arange(1, 11, dtype=float)[: len(preds)][preds[:10] > 0]
        if len(positions) > 0:
            mAP10.append(float(np.mean(np.arange(1, len(positions) + 1, dtype=float) / positions)))
        else:
            mAP10.append(0.0)
    out: Dict[str, object] = {"R1": R1, "R5": R5, "R10": R10, "mAP10": mAP10}
    for name in ("R1", "R5", "R10", "mAP10"):
        est, _, _, ci = jackknife_stats(np.asarray(out[name]))
        out[name + "_estimate"] = est
        out[name + "_ci"] = ci
        if verbose:
            print(name, f"{est:.3f}", f"[{ci[0]:.3f}, {ci[1]:.3f}]")
    return out


def audio_retrieval_scores(audio_emb: Tensor, text_emb: Tensor, k: int = 10, sim: Optional[Tensor] = None):
    """The scoring tail of `audio_retrieval` (eval_caco_torch.py:394-408) on device: logits_ar = T @ A^T (unscaled),
    audio->text indices (rows = clips) and text->audio indices (rows = captions), k each."""
    from .model import similarity
    logits_ar = similarity(text_emb, audio_emb, 1.0, out=sim)
    at_idx, _ = topk(logits_ar, k, dim=0)       # argsort(-logits_ar.T)
    ta_idx, _ = topk(logits_ar, k, dim=1)       # argsort(-logits_ar)
    return logits_ar, at_idx, ta_idx


def zs_classification_scores(audio_emb: Tensor, class_text_emb: Tensor, target_idx, logit_scale: float = 0.0,
                             ks: Sequence[int] = (1,)) -> Dict[str, float]:
    """Zero-shot classification scoring, the arithmetic of `zs_classification` (src/eval/eval_caco_torch.py:326-340) for
    a whole set of clips at once: logits = exp(logit_scale) * A @ T_classes^T, a clip counts as correct at k when its
    target class is among the first k of argsort(-logits).  audio_emb [N, D], class_text_emb [C, D] (both as returned
    with normalize=True), target_idx int [N].  The similarity and the top-k run on the device (caco_similarity,
    caco_topk); only [N, max(ks)] indices come back.  Returns {"1": acc@1, ...} keyed like the reference's total_correct."""
    from .model import similarity
    kmax = int(max(ks))
    if kmax < 1 or kmax > 64:
        raise ValueError("zs_classification_scores: k must be in [1, 64]")
    logits = similarity(audio_emb, class_text_emb, float(np.exp(logit_scale)))
    idx, _ = topk(logits, kmax)
    idx = idx.cpu().numpy()
    tgt = np.asarray(target_idx.cpu() if torch.is_tensor(target_idx) else target_idx).astype(np.int64).reshape(-1)
    if tgt.shape[0] != idx.shape[0]:
        raise ValueError(f"zs_classification_scores: {tgt.shape[0]} targets for {idx.shape[0]} clips")
    return {str(int(k)): float((idx[:, :int(k)] == tgt[:, None]).any(axis=1).mean()) for k in ks}
