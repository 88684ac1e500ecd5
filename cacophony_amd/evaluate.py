"""The evaluation drivers that sit on the hot path in the reference, under the reference's names and argument meaning
(the starred `@no_grad` wrappers of src/eval/eval_caco_torch.py:231-261 and their callers
`compute_all_class_embeddings :265`, `zs_classification :289`, `audio_retrieval :343`, `audio_captioning :475`).

The reference walks a dataset one file at a time: load, log-mel on the host, one forward at batch 1, one `@` per
clip, and a full argsort of the final matrix on the host.  Here the same results come out of batches: clips of
different lengths go through the fused front end together (each masked at its own length, exactly what the
reference gets clip by clip), captions / class prompts are tokenised one by one as in the reference (so the ids are
the same) and encoded in batches, and the scoring - similarity, top-k, hits - runs on the device
(`cacophony_amd.retrieval`).  Only file names, label strings and [N, 10] indices are handled on the host.

Dataset processors and audio decoding are not part of this package: `dataprocessor` is the reference's object
(`get_filepaths_and_descriptions(current_split=...)`, `.config.sampling_rate`), and `load_audio` needs the
`soundfile` module like the reference's (src/eval/eval_utils.py:6-16) - pass `load_audio_fn` to use another decoder.
"""
from __future__ import annotations

import csv
import os
from typing import Callable, Dict, Iterator, List, Mapping, Optional, Sequence, Tuple

import numpy as np
import torch

from . import frontend, retrieval
from .config import DatasetConfig

Tensor = torch.Tensor


# ---------------------------------------------------------------------------------------------------------------------
# the @no_grad wrappers (eval_caco_torch.py:231-261)
# ---------------------------------------------------------------------------------------------------------------------
def compute_audio_embedding(model, audio_batch: Mapping[str, Tensor]) -> Tensor:
    """eval_caco_torch.py:231-246: normalised audio embedding [B, P] of a prepared batch."""
    return model.get_audio_embedding(audio_patches=audio_batch["audio_patches"], audio_time_inds=audio_batch["audio_time_inds"],
                                     audio_freq_inds=audio_batch["audio_freq_inds"], audio_mask=audio_batch["audio_mask"],
                                     deterministic=True, return_hidden_state=False, normalize=True)


def compute_text_embedding(model, text_batch: Mapping[str, Tensor]) -> Tensor:
    """eval_caco_torch.py:249-261: normalised text embedding [B, P] of a prepared batch."""
    return model.get_text_embedding(text_input_ids=text_batch["text_input_ids"], text_mask=text_batch["text_mask"],
                                    deterministic=True, return_hidden_state=False, normalize=True)


# ---------------------------------------------------------------------------------------------------------------------
# host bookkeeping (no device work): what the reference's loops build as they go
# ---------------------------------------------------------------------------------------------------------------------
def _chunks(n: int, size: int) -> Iterator[Tuple[int, int]]:
    size = max(1, int(size))
    for lo in range(0, n, size):
        yield lo, min(n, lo + size)


def audio_name_of(filepath: str) -> str:
    """`filepaths[i].split('/')[-1].split('.wav')[0]` (eval_caco_torch.py:318, 364)."""
    return filepath.split("/")[-1].split(".wav")[0]


def class_index_map(descriptions: Mapping[str, Mapping]) -> Tuple[List[str], Dict[str, int]]:
    """The unique class labels of a classification set (first description of every clip, eval_caco_torch.py:302-304).
    The reference takes `list(set(...))`, whose order changes from run to run; sorted here, which only decides ties."""
    labels = sorted({descriptions[a]["description"][0] for a in descriptions})
    return labels, {v: i for i, v in enumerate(labels)}


def retrieval_ground_truth(filepaths: Sequence[str], descriptions: Mapping[str, Mapping]):
    """The lists and dictionaries `audio_retrieval` builds (eval_caco_torch.py:356-391): clip names in file order, every
    caption in clip order, caption -> clip (a caption string shared by two clips keeps the LAST one, as there) and
    clip -> captions."""
    all_audio, all_text, gt_audio_text, gt_text_audio = [], [], {}, {}
    for fp in filepaths:
        name = audio_name_of(fp)
        gt_audio_text[name] = []
        for caption in descriptions[name]["description"]:
            gt_audio_text[name].append(caption)
            gt_text_audio[caption] = name
            all_text.append(caption)
        all_audio.append(name)
    return all_audio, all_text, gt_audio_text, gt_text_audio


def load_audio(audio_path: str, dataset_sampling_rate: int) -> np.ndarray:
    """src/eval/eval_utils.py:6-16: read, average the channels, resample to 16 kHz (scipy.signal.resample)."""
    try:
        import soundfile as sf
    except ImportError as e:       # the reference's decoder; not part of this package's requirements
        raise ImportError("load_audio needs the `soundfile` module (as the reference does); pass load_audio_fn=... instead") from e
    wav, _ = sf.read(audio_path)
    wav = wav.astype(np.float32)
    if wav.ndim > 1:
        wav = np.mean(wav, axis=-1)
    if dataset_sampling_rate != 16000:
        import scipy.signal
        wav = scipy.signal.resample(wav, round(wav.shape[-1] * 16000.0 / dataset_sampling_rate))
    return wav


def load_caco_torch(ckpt_path: str, device=None, tokenizer=None, use_decoder: bool = False, caco_config=None,
                    audio_config=None, weights_only: bool = True) -> Dict[str, object]:
    """eval_caco_torch.py:154-178: {'model', 'tokenizer', 'device'} from a checkpoint file.  The file may be the torch
    container (plain state dict or the `model_state_dict` / `state_dict` wrappers, :160-166) or the JAX side's Flax
    msgpack file (cacophony_amd.checkpoint).

    The two sides of the reference disagree on two hyper-parameters that do not change a single tensor shape (SURVEY Q5 /
    Q6): the torch model pools the audio tokens with 2 heads and LayerNorm eps 1e-5 (caco.py:20,294; mae.py:68,76,123), the
    JAX model the Flax file comes from with 8 heads (src/caco/load_model.py:46) and Flax's eps 1e-6 in the audio stack
    (src/caco/audio_models/mae.py:87,93,137).  A Flax file is therefore loaded into a model built with the JAX side's
    values unless `caco_config` / `audio_config` say otherwise; a torch file gets the torch defaults.
    The reference downloads `roberta-base`'s tokenizer; offline that only works from a local cache, so one can be passed."""
    from dataclasses import replace
    from .checkpoint import checkpoint_format, load_checkpoint
    from .config import default_audio_config, default_caco_config, default_text_config
    from .model import CACO
    if not torch.cuda.is_available():       # before the file is even opened: there is no CPU model to load it into
        raise RuntimeError("cacophony_amd needs a ROCm GPU (torch.cuda.is_available() is False); there is no CPU path")
    if checkpoint_format(ckpt_path) == "flax":
        if caco_config is None:
            caco_config = replace(default_caco_config(), num_attention_pool_heads=8)
        if audio_config is None:
            audio_config = replace(default_audio_config(), layer_norm_eps=1e-6)
    caco_config = caco_config if caco_config is not None else default_caco_config()
    audio_config = audio_config if audio_config is not None else default_audio_config()
    dec = replace(default_text_config(), num_hidden_layers=4) if use_decoder else None          # caco.py:297-309
    model = CACO(audio_config, default_text_config(), caco_config, decoder_config=dec, device=device)
    model.load_state_dict(load_checkpoint(ckpt_path, weights_only=weights_only))
    if tokenizer is None:
        from transformers import RobertaTokenizerFast
        try:
            tokenizer = RobertaTokenizerFast.from_pretrained("roberta-base", local_files_only=True)
        except Exception as e:
            raise RuntimeError("load_caco_torch: the `roberta-base` tokenizer is not in the local cache (no network here); "
                               "pass tokenizer=...") from e
    return {"model": model, "tokenizer": tokenizer, "device": model.device}


def task_dataset_config(task: str) -> DatasetConfig:
    """The DatasetConfig the reference's command line builds per task (eval_caco_torch.py:570-577, 605-613, 625-633):
    'zs' = 10 s clips (500 patches), 'ar' / 'caption' = 30 s clips (1500 patches); 16 x 16 patches, 100 text tokens."""
    if task == "zs":
        patches = 100 * 10 * 8 // 16
    elif task in ("ar", "caption"):
        patches = 16000 * 30 * 8 // 160 // 16
    else:
        raise ValueError(f"task must be 'zs', 'ar' or 'caption', got {task!r}")
    return DatasetConfig(batch_size=1, patches_seq_len=patches, time_patch_size=16, freq_patch_size=16, max_text_len=100,
                         synthetic_prob=0.8)


# ---------------------------------------------------------------------------------------------------------------------
# batched embedding of many items
# ---------------------------------------------------------------------------------------------------------------------
def embed_texts(model, tokenizer, texts: Sequence[str], max_text_len: int, device=None, batch_size: int = 256) -> Tensor:
    """Normalised embeddings [len(texts), P].  Each string is tokenised on its own with padding to `max_text_len`
    (prepare_text_batch, eval_caco_torch.py:209-227), so the ids equal the reference's; the rows are then encoded
    `batch_size` at a time."""
    if len(texts) == 0:
        raise ValueError("embed_texts: no texts (empty split or class list)")
    out = []
    for lo, hi in _chunks(len(texts), batch_size):
        rows = [frontend.prepare_text_batch(t, tokenizer, max_text_len, device) for t in texts[lo:hi]]
        batch = {"text_input_ids": torch.cat([r["text_input_ids"] for r in rows], 0),
                 "text_mask": torch.cat([r["text_mask"] for r in rows], 0)}
        out.append(compute_text_embedding(model, batch))
    return torch.cat(out, 0)


def embed_clips(model, clips: Sequence[np.ndarray], datasetconfig: DatasetConfig, device=None) -> Tensor:
    """Normalised embeddings [len(clips), P] of waveforms of different lengths in ONE front-end launch + ONE forward:
    every clip is masked at its own length (what the reference computes clip by clip, eval_caco_torch.py:181-206)."""
    return compute_audio_embedding(model, frontend.prepare_audio_batch(list(clips), datasetconfig, device))


def compute_all_class_embeddings(model, tokenizer, class_list: Sequence[str], max_text_len: int, device=None,
                                 prefix: str = "", batch_size: int = 256) -> Tensor:
    """eval_caco_torch.py:265-286: [len(class_list), P], one row per class prompt `prefix + class_text`."""
    return embed_texts(model, tokenizer, [prefix + c for c in class_list], max_text_len, device, batch_size)


# ---------------------------------------------------------------------------------------------------------------------
# the drivers (eval_caco_torch.py:289-408)
# ---------------------------------------------------------------------------------------------------------------------
def zs_classification(model, tokenizer, dataprocessor, datasetconfig: DatasetConfig, device=None, subdir_name: str = "",
                      text_prefix: str = "This is a sound of ", *, load_audio_fn: Optional[Callable] = None,
                      batch_size: int = 64, ks: Sequence[int] = (1,), verbose: bool = True) -> float:
    """Zero-shot classification (eval_caco_torch.py:289-340): top-1 accuracy of `exp(logit_scale) * A @ T_classes^T`
    over the split's clips.  Clips are decoded on the host and embedded `batch_size` at a time; similarity, top-k and
    the hit test run on the device.  Returns the accuracy at ks[0] like the reference (and prints every k)."""
    load = load_audio_fn or load_audio
    filepaths, descriptions, _ = dataprocessor.get_filepaths_and_descriptions(current_split=subdir_name)
    if len(filepaths) == 0:
        raise ValueError(f"zs_classification: split {subdir_name!r} has no files")
    class_labels, class_to_index = class_index_map(descriptions)
    class_emb = compute_all_class_embeddings(model, tokenizer, class_labels, datasetconfig.max_text_len, device, prefix=text_prefix)
    targets = [class_to_index[descriptions[audio_name_of(fp)]["description"][0]] for fp in filepaths]
    logit_scale = float(model.logit_scale)
    correct = {str(int(k)): 0.0 for k in ks}
    for lo, hi in _chunks(len(filepaths), batch_size):
        clips = [np.asarray(load(fp, dataprocessor.config.sampling_rate), dtype=np.float32) for fp in filepaths[lo:hi]]
        emb = embed_clips(model, clips, datasetconfig, device)
        acc = retrieval.zs_classification_scores(emb, class_emb, targets[lo:hi], logit_scale, ks)
        for k in correct:
            correct[k] += acc[k] * (hi - lo)
    n = max(1, len(filepaths))
    if verbose:
        for k in correct:
            print(f"top {k} accuracy: {correct[k] / n:.4f}")
    return correct[str(int(ks[0]))] / n


def audio_retrieval(model, tokenizer, dataprocessor, datasetconfig: DatasetConfig, device=None, eval_split: str = "test", *,
                    load_audio_fn: Optional[Callable] = None, batch_size: int = 64, verbose: bool = True) -> Dict[str, Dict[str, object]]:
    """Audio-text retrieval (eval_caco_torch.py:343-408): `logits_ar = T @ A^T` over every caption and clip of the split,
    recall@{1,5,10} and mAP@10 in both directions.  The reference prints the metrics; they are also returned here:
    {"audio_to_text": {...}, "text_to_audio": {...}} as `compute_retrieval_metric` gives them."""
    load = load_audio_fn or load_audio
    filepaths, descriptions, _ = dataprocessor.get_filepaths_and_descriptions(current_split=eval_split)
    if len(filepaths) == 0:
        raise ValueError(f"audio_retrieval: split {eval_split!r} has no files")
    all_audio, all_text, gt_audio_text, gt_text_audio = retrieval_ground_truth(filepaths, descriptions)
    text_emb = embed_texts(model, tokenizer, all_text, datasetconfig.max_text_len, device)
    audio_emb = []
    for lo, hi in _chunks(len(filepaths), batch_size):
        clips = [np.asarray(load(fp, dataprocessor.config.sampling_rate), dtype=np.float32) for fp in filepaths[lo:hi]]
        audio_emb.append(embed_clips(model, clips, datasetconfig, device))
    audio_emb = torch.cat(audio_emb, 0)
    _, at_idx, ta_idx = retrieval.audio_retrieval_scores(audio_emb, text_emb, k=10)
    if verbose:
        print("audio to text retrieval:")
    at = retrieval.compute_retrieval_metric(at_idx, all_audio, all_text, gt_audio_text, "at", verbose)
    if verbose:
        print("text to audio retrieval:")
    ta = retrieval.compute_retrieval_metric(ta_idx, all_text, all_audio, gt_text_audio, "ta", verbose)
    return {"audio_to_text": at, "text_to_audio": ta}


def audio_captioning(model, tokenizer, dataprocessor, datasetconfig: DatasetConfig, device=None, eval_split: str = "test",
                     output_dir: str = "./", *, load_audio_fn: Optional[Callable] = None, batch_size: int = 32,
                     max_decode_length: int = 100, temperature: float = 0.1, greedy: bool = False,
                     generator: Optional[torch.Generator] = None, verbose: bool = True) -> Dict[str, object]:
    """Audio captioning (eval_caco_torch.py:475-541): a caption per clip of the split, written with the references to
    `predictions.csv` / `gt.csv` in the reference's layout (commas removed from the references, five reference columns).
    The reference decodes one clip at a time through a full-prefix loop; here `batch_size` ragged clips decode together
    through the key / value-cached decoder (`captioning.decode_caption_ids`: per-row stop, temperature sampling as in
    `decode_caption` :459-461, or `greedy=True`).  Returns the paths and the lists that were written."""
    from . import captioning
    load = load_audio_fn or load_audio
    filepaths, descriptions, _ = dataprocessor.get_filepaths_and_descriptions(current_split=eval_split)
    names = [audio_name_of(fp) for fp in filepaths]
    predicted: List[str] = []
    for lo, hi in _chunks(len(filepaths), batch_size):
        clips = [np.asarray(load(fp, dataprocessor.config.sampling_rate), dtype=np.float32) for fp in filepaths[lo:hi]]
        batch = frontend.prepare_audio_batch(clips, datasetconfig, device)
        ids = captioning.decode_caption_ids(model, batch, max_decode_length, temperature, tokenizer.bos_token_id,
                                            tokenizer.eos_token_id, tokenizer.pad_token_id, greedy, generator)
        predicted.extend(t.strip() for t in tokenizer.batch_decode(ids, skip_special_tokens=True))
    references = [[d.replace(",", "") for d in descriptions[n]["description"]] for n in names]
    assert len(predicted) == len(references)
    pred_path, gt_path = os.path.join(output_dir, "predictions.csv"), os.path.join(output_dir, "gt.csv")
    with open(pred_path, "w", newline="") as fp:
        w = csv.writer(fp)
        w.writerow(["file_name", "caption_predicted"])
        w.writerows([n, c] for n, c in zip(names, predicted))
    with open(gt_path, "w", newline="") as fg:
        w = csv.writer(fg)
        w.writerow(["file_name"] + [f"caption_reference_{i:02d}" for i in range(1, 6)])
        for n, refs in zip(names, references):
            w.writerow([n] + refs + [""] * max(0, 5 - len(refs)))
    if verbose:
        print(f"Predictions saved to {pred_path}")
        print(f"Ground truth saved to {gt_path}")
    return {"predictions_path": pred_path, "gt_path": gt_path, "file_names": names, "predicted": predicted, "references": references}
