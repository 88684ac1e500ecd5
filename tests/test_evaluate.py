"""cacophony_amd.evaluate: the reference's evaluation drivers (src/eval/eval_caco_torch.py:231-408) in batched form.

CPU: the host bookkeeping against known answers, the wrappers' keyword contract, and the whole driver flow with the
device pieces (front end, towers, scoring) replaced by NumPy stand-ins that enforce the real functions' preconditions.
GPU: the same drivers on the real pieces against the reference's own procedure - one clip, one caption at a time,
host matmul and argsort."""
import math

import numpy as np
import pytest
import torch

from cacophony_amd import config as C
from cacophony_amd import evaluate as E
from cacophony_amd import frontend, retrieval, synth

TEXT_VOCAB = 1024


class StubTokenizer:
    """RobertaTokenizerFast as eval_caco_torch.py:216-221 calls it; ids in the tiny model's vocabulary."""

    def __call__(self, texts, padding=None, truncation=None, max_length=None, return_tensors=None):
        assert padding == "max_length" and truncation is True and return_tensors == "pt" and len(texts) == 1
        ids = torch.ones(len(texts), max_length, dtype=torch.int64)          # <pad> = 1
        mask = torch.zeros_like(ids)
        for i, t in enumerate(texts):
            toks = [0] + [4 + sum(ord(c) * (j + 1) for j, c in enumerate(w)) % (TEXT_VOCAB - 4) for w in t.split()][: max_length - 2] + [2]
            ids[i, : len(toks)] = torch.tensor(toks)
            mask[i, : len(toks)] = 1
        return {"input_ids": ids, "attention_mask": mask}


class ToyProcessor:
    """The slice of the reference's dataset processors the drivers touch."""

    class config:
        sampling_rate = 16000

    def __init__(self, clips, descriptions):
        self.clips, self.descriptions = clips, descriptions

    def get_filepaths_and_descriptions(self, current_split=""):
        return [f"/data/{current_split}/{name}.wav" for name in self.clips], self.descriptions, None

    def load(self, path, sr):
        assert sr == 16000
        return self.clips[E.audio_name_of(path)]


def _toy_dataset(n_clips=7, seed=5):
    rng = np.random.default_rng(seed)
    classes = ["dog bark", "rain", "siren"]
    clips, desc = {}, {}
    for i in range(n_clips):
        n = int(rng.integers(40000, 160000))
        clips[f"clip{i:02d}"] = synth.make_waveform(100 + i)[:n].astype(np.float32)
        label = classes[i % len(classes)]
        caps = [label] + [f"{label} take {i} version {j}" for j in range(1 + i % 3)]
        desc[f"clip{i:02d}"] = {"description": caps}
    return ToyProcessor(clips, desc), classes


# ----------------------------------------------------------------------------------------------------------- CPU
def test_host_bookkeeping_known_answers():
    assert E.audio_name_of("/a/b/c/1-100032-A-0.wav") == "1-100032-A-0"
    assert list(E._chunks(5, 2)) == [(0, 2), (2, 4), (4, 5)] and list(E._chunks(0, 4)) == []
    desc = {"x": {"description": ["rain", "heavy rain"]}, "y": {"description": ["dog"]}, "z": {"description": ["rain"]}}
    labels, idx = E.class_index_map(desc)
    assert labels == ["dog", "rain"] and idx == {"dog": 0, "rain": 1}
    all_audio, all_text, gt_at, gt_ta = E.retrieval_ground_truth(["/s/x.wav", "/s/y.wav", "/s/z.wav"], desc)
    assert all_audio == ["x", "y", "z"] and all_text == ["rain", "heavy rain", "dog", "rain"]
    assert gt_at == {"x": ["rain", "heavy rain"], "y": ["dog"], "z": ["rain"]}
    assert gt_ta == {"rain": "z", "heavy rain": "x", "dog": "y"}          # a shared caption keeps the last clip (:381)


def test_wrappers_pass_the_reference_keywords():
    seen = {}

    class Rec:
        def get_audio_embedding(self, **kw):
            seen["a"] = kw
            return torch.zeros(1, 4)

        def get_text_embedding(self, **kw):
            seen["t"] = kw
            return torch.zeros(1, 4)

    ab = {k: torch.zeros(1) for k in ("audio_patches", "audio_time_inds", "audio_freq_inds", "audio_mask")}
    E.compute_audio_embedding(Rec(), ab)
    E.compute_text_embedding(Rec(), {"text_input_ids": torch.zeros(1), "text_mask": torch.zeros(1)})
    assert {k: v for k, v in seen["a"].items() if k not in ab} == dict(deterministic=True, return_hidden_state=False, normalize=True)
    assert set(seen["a"]) - {"deterministic", "return_hidden_state", "normalize"} == set(ab)
    assert {k: v for k, v in seen["t"].items() if not k.startswith("text_")} == dict(deterministic=True, return_hidden_state=False, normalize=True)


def test_task_configs_and_loader_contract(tmp_path):
    assert E.task_dataset_config("zs").patches_seq_len == 500 and E.task_dataset_config("ar").patches_seq_len == 1500
    assert E.task_dataset_config("caption") == E.task_dataset_config("ar") and E.task_dataset_config("zs").max_text_len == 100
    with pytest.raises(ValueError):
        E.task_dataset_config("asr")
    empty = ToyProcessor({}, {})
    with pytest.raises(ValueError, match="no files"):
        E.zs_classification(object(), StubTokenizer(), empty, E.task_dataset_config("zs"), subdir_name="fold9", load_audio_fn=empty.load)
    with pytest.raises(ValueError, match="no files"):
        E.audio_retrieval(object(), StubTokenizer(), empty, E.task_dataset_config("ar"), load_audio_fn=empty.load)
    if not torch.cuda.is_available():       # no CPU model: the loader fails loudly instead of returning something slower
        with pytest.raises(RuntimeError):
            E.load_caco_torch(str(tmp_path / "missing.ckpt"), tokenizer=StubTokenizer())


def test_load_audio_without_decoder_says_so():
    with pytest.raises(ImportError, match="soundfile"):
        E.load_audio("/nonexistent.wav", 16000)


def _install_cpu_stand_ins(monkeypatch):
    """The device pieces replaced by host code with the same contracts (shapes, dtypes, argument meaning)."""
    dim = 16

    def prepare_text_batch(text, tokenizer, max_text_len, device=None):
        tok = tokenizer([text], padding="max_length", truncation=True, max_length=max_text_len, return_tensors="pt")
        return {"text_input_ids": tok["input_ids"], "text_mask": tok["attention_mask"]}

    def prepare_audio_batch(audio, datasetconfig, device=None, lengths=None):
        assert isinstance(audio, list) and all(np.ndim(c) == 1 and c.dtype == np.float32 for c in audio)
        feats = torch.tensor([[float(np.mean(c ** 2)), float(len(c))] for c in audio])
        return {"audio_patches": feats, "audio_time_inds": feats, "audio_freq_inds": feats, "audio_mask": feats}

    class Model:
        logit_scale = torch.tensor(math.log(10.0))

        def get_audio_embedding(self, audio_patches, audio_time_inds, audio_freq_inds, audio_mask, deterministic, return_hidden_state, normalize):
            assert deterministic and not return_hidden_state and normalize
            g = torch.Generator().manual_seed(0)
            proj = torch.randn(2, dim, generator=g)
            return torch.nn.functional.normalize(torch.tanh(audio_patches / torch.tensor([0.1, 1e5])) @ proj, dim=1)

        def get_text_embedding(self, text_input_ids, text_mask, deterministic, return_hidden_state, normalize):
            assert deterministic and not return_hidden_state and normalize and text_input_ids.dtype == torch.int64
            g = torch.Generator().manual_seed(1)
            table = torch.randn(TEXT_VOCAB, dim, generator=g)
            return torch.nn.functional.normalize((table[text_input_ids] * text_mask[..., None]).sum(1), dim=1)

    def zs_scores(audio_emb, class_text_emb, target_idx, logit_scale=0.0, ks=(1,)):
        logits = math.exp(logit_scale) * audio_emb @ class_text_emb.T
        order = torch.argsort(-logits, dim=-1, stable=True).numpy()
        tgt = np.asarray(target_idx).reshape(-1)
        assert tgt.shape[0] == order.shape[0]
        return {str(int(k)): float((order[:, :int(k)] == tgt[:, None]).any(1).mean()) for k in ks}

    def retrieval_scores(audio_emb, text_emb, k=10, sim=None):
        logits = text_emb @ audio_emb.T
        at = torch.argsort(-logits.T, dim=-1, stable=True)[:, :k].to(torch.int32)
        ta = torch.argsort(-logits, dim=-1, stable=True)[:, :k].to(torch.int32)
        return logits, at, ta

    monkeypatch.setattr(frontend, "prepare_text_batch", prepare_text_batch)
    monkeypatch.setattr(frontend, "prepare_audio_batch", prepare_audio_batch)
    monkeypatch.setattr(retrieval, "zs_classification_scores", zs_scores)
    monkeypatch.setattr(retrieval, "audio_retrieval_scores", retrieval_scores)
    return Model()


def _reference_procedure(model, tok, proc, cfg, classes_prefix, device=None):
    """The reference's loops as written (eval_caco_torch.py:289-408): one clip / one caption at a time, host matmul and
    full argsort.  Returns (top-1 accuracy, at metric dict, ta metric dict)."""
    filepaths, descriptions, _ = proc.get_filepaths_and_descriptions(current_split="test")
    labels, cmap = E.class_index_map(descriptions)
    class_emb = torch.cat([E.compute_text_embedding(model, frontend.prepare_text_batch(classes_prefix + c, tok, cfg.max_text_len, device))
                           for c in labels], 0).float().cpu()
    hits, a_embs, t_embs = 0, [], []
    all_audio, all_text, gt_at, gt_ta = E.retrieval_ground_truth(filepaths, descriptions)
    for fp in filepaths:
        name = E.audio_name_of(fp)
        emb = E.compute_audio_embedding(model, frontend.prepare_audio_batch([proc.load(fp, 16000)], cfg, device)).float().cpu()
        logits = float(torch.exp(model.logit_scale)) * emb @ class_emb.T
        hits += int(int(torch.argsort(-logits, dim=-1, stable=True)[0, 0]) == cmap[descriptions[name]["description"][0]])
        a_embs.append(emb)
        for cap in descriptions[name]["description"]:
            t_embs.append(E.compute_text_embedding(model, frontend.prepare_text_batch(cap, tok, cfg.max_text_len, device)).float().cpu())
    A, T = torch.cat(a_embs, 0), torch.cat(t_embs, 0)
    logits_ar = T @ A.T
    at = retrieval.compute_retrieval_metric(torch.argsort(-logits_ar.T, dim=-1, stable=True).numpy(), all_audio, all_text, gt_at)
    ta = retrieval.compute_retrieval_metric(torch.argsort(-logits_ar, dim=-1, stable=True).numpy(), all_text, all_audio, gt_ta, "ta")
    return hits / len(filepaths), at, ta


def test_drivers_flow_on_cpu_stand_ins(monkeypatch):
    """zs_classification / audio_retrieval end to end (batches of 3 over 7 clips: a ragged last batch) equal the
    reference's one-at-a-time procedure on the same stand-in model."""
    model = _install_cpu_stand_ins(monkeypatch)
    proc, _ = _toy_dataset()
    cfg = C.DatasetConfig(patches_seq_len=500, max_text_len=16)
    tok = StubTokenizer()
    acc = E.zs_classification(model, tok, proc, cfg, subdir_name="test", load_audio_fn=proc.load, batch_size=3, verbose=False)
    out = E.audio_retrieval(model, tok, proc, cfg, eval_split="test", load_audio_fn=proc.load, batch_size=3, verbose=False)
    ref_acc, ref_at, ref_ta = _reference_procedure(model, tok, proc, cfg, "This is a sound of ")
    assert acc == pytest.approx(ref_acc)
    for name in ("R1", "R5", "R10", "mAP10"):
        assert out["audio_to_text"][name] == ref_at[name] and out["text_to_audio"][name] == ref_ta[name]
    emb = E.compute_all_class_embeddings(model, tok, ["rain", "siren"], 16, prefix="This is a sound of ", batch_size=1)
    one = E.compute_text_embedding(model, frontend.prepare_text_batch("This is a sound of siren", tok, 16))
    assert emb.shape == (2, 16) and torch.allclose(emb[1:], one)


def test_audio_captioning_writes_the_reference_csv_layout(monkeypatch, tmp_path):
    """audio_captioning (eval_caco_torch.py:475-541) with the decoder replaced by a stand-in: batches of ragged clips go
    to the decoding loop with the tokenizer's special ids and the reference's defaults (100 tokens, temperature 0.1), and
    the two CSV files have the reference's header, comma-free references and five reference columns."""
    from cacophony_amd import captioning
    _install_cpu_stand_ins(monkeypatch)
    proc, _ = _toy_dataset(n_clips=4)
    proc.descriptions["clip01"]["description"] = ["rain, heavy", "rain on a roof, then thunder"]
    calls = []

    def decode_caption_ids(model, audio_batch, max_decode_length=100, temperature=0.1, bos_id=0, eos_id=2, pad_id=1, greedy=False,
                           generator=None, use_cache=True):
        n = audio_batch["audio_patches"].shape[0]
        calls.append((n, max_decode_length, temperature, bos_id, eos_id, pad_id, greedy))
        base = len(calls) * 10
        return torch.tensor([[bos_id, 50 + base + i, eos_id] for i in range(n)])

    class Tok(StubTokenizer):
        bos_token_id, eos_token_id, pad_token_id = 0, 2, 1

        def batch_decode(self, ids, skip_special_tokens=False):
            assert skip_special_tokens
            return [f"  caption {int(r[1])} " for r in ids]

    monkeypatch.setattr(captioning, "decode_caption_ids", decode_caption_ids)
    out = E.audio_captioning(object(), Tok(), proc, C.DatasetConfig(patches_seq_len=500), eval_split="test", output_dir=str(tmp_path),
                             load_audio_fn=proc.load, batch_size=3, verbose=False)
    assert calls == [(3, 100, 0.1, 0, 2, 1, False), (1, 100, 0.1, 0, 2, 1, False)]
    assert out["predicted"] == ["caption 60", "caption 61", "caption 62", "caption 70"]
    pred = (tmp_path / "predictions.csv").read_text().splitlines()
    gt = (tmp_path / "gt.csv").read_text().splitlines()
    assert pred == ["file_name,caption_predicted", "clip00,caption 60", "clip01,caption 61", "clip02,caption 62", "clip03,caption 70"]
    assert gt[0] == "file_name,caption_reference_01,caption_reference_02,caption_reference_03,caption_reference_04,caption_reference_05"
    assert gt[2] == "clip01,rain heavy,rain on a roof then thunder,,,"
    assert all(len(line.split(",")) == 6 for line in gt)


# ----------------------------------------------------------------------------------------------------------- GPU
def _check_drivers(model, expect_cuda):
    """Shared body of the device test and of its CPU dry run (same assertions on either backend)."""
    proc, _ = _toy_dataset()
    cfg = C.DatasetConfig(patches_seq_len=500, max_text_len=16)
    tok = StubTokenizer()
    filepaths, descriptions, _ = proc.get_filepaths_and_descriptions(current_split="test")
    clips = [proc.load(fp, 16000) for fp in filepaths]
    all_audio, all_text, gt_at, gt_ta = E.retrieval_ground_truth(filepaths, descriptions)
    labels, cmap = E.class_index_map(descriptions)

    A = torch.cat([E.embed_clips(model, clips[lo:hi], cfg) for lo, hi in E._chunks(len(clips), 3)], 0)
    T = E.embed_texts(model, tok, all_text, cfg.max_text_len)
    Cemb = E.compute_all_class_embeddings(model, tok, labels, cfg.max_text_len, prefix="This is a sound of ")
    assert A.shape == (len(clips), 768) and T.shape == (len(all_text), 768) and Cemb.shape == (len(labels), 768)
    assert A.is_cuda == expect_cuda
    for i in (0, 3, 6):                                                 # (1) batch of three ragged clips vs batch of one
        alone = E.compute_audio_embedding(model, frontend.prepare_audio_batch([clips[i]], cfg))
        assert torch.nn.functional.cosine_similarity(alone, A[i:i + 1]).item() > 0.9999

    A64, T64, C64 = (x.double().cpu().numpy() for x in (A, T, Cemb))
    logits = math.exp(float(model.logit_scale)) * A64 @ C64.T           # (2)
    top2 = -np.sort(-logits, axis=1)[:, :2]
    assert (top2[:, 0] - top2[:, 1]).min() > 1e-4
    target = np.array([cmap[descriptions[n]["description"][0]] for n in all_audio])
    acc = E.zs_classification(model, tok, proc, cfg, subdir_name="test", load_audio_fn=proc.load, batch_size=3, verbose=False)
    assert acc == pytest.approx(float((logits.argmax(1) == target).mean()))

    out = E.audio_retrieval(model, tok, proc, cfg, eval_split="test", load_audio_fn=proc.load, batch_size=3, verbose=False)   # (3)
    _, at_idx, ta_idx = retrieval.audio_retrieval_scores(A, T, k=10)
    ref_at = retrieval.compute_retrieval_metric(at_idx, all_audio, all_text, gt_at)
    ref_ta = retrieval.compute_retrieval_metric(ta_idx, all_text, all_audio, gt_ta, "ta")
    for name in ("R1", "R5", "R10", "mAP10"):
        assert out["audio_to_text"][name] == ref_at[name] and len(ref_at[name]) == len(all_audio), name
        assert out["text_to_audio"][name] == ref_ta[name] and len(ref_ta[name]) == len(all_text), name
    L64 = T64 @ A64.T
    for dev_idx, M in ((at_idx, L64.T), (ta_idx, L64)):
        dev_idx = dev_idx.cpu().numpy()
        k = dev_idx.shape[1]
        host_sorted = -np.sort(-M, axis=1)[:, :k]
        assert dev_idx.min() >= 0
        assert np.abs(np.take_along_axis(M, dev_idx.astype(np.int64), axis=1) - host_sorted).max() < 1e-5


def test_device_test_body_dry_run_on_the_oracle(monkeypatch, tiny_state):
    """The GPU case below, executed here with the oracle standing in for the library (front end, towers) and host code for
    the scoring kernels: the assertions, shapes and bookkeeping of that test are exercised before it ever meets a GPU."""
    from oracle import caco_oracle as O
    a, t, cc = C.tiny_configs(2)
    ref = O.CacoOracle(tiny_state, a, t, cc, backend="torch")

    class OracleModel:
        logit_scale = torch.tensor(float(cc.logit_scale_init_value))

        def get_audio_embedding(self, audio_patches, audio_time_inds, audio_freq_inds, audio_mask, deterministic=True,
                                return_hidden_state=True, normalize=False):
            assert deterministic
            with torch.no_grad():
                out = ref.get_audio_embedding(audio_patches.numpy(), audio_time_inds.numpy(), audio_freq_inds.numpy(),
                                              audio_mask.numpy(), return_hidden_state=return_hidden_state, normalize=normalize)
            return torch.as_tensor(np.asarray(out)) if not return_hidden_state else tuple(torch.as_tensor(np.asarray(o)) for o in out)

        def get_text_embedding(self, text_input_ids, text_mask, position_ids=None, deterministic=True, return_hidden_state=True,
                               normalize=False):
            assert deterministic and text_input_ids.dtype == torch.int64
            with torch.no_grad():
                out = ref.get_text_embedding(text_input_ids.numpy(), text_mask.numpy(), return_hidden_state=return_hidden_state,
                                             normalize=normalize)
            return torch.as_tensor(np.asarray(out)) if not return_hidden_state else tuple(torch.as_tensor(np.asarray(o)) for o in out)

    def prepare_audio_batch(audio, datasetconfig, device=None, lengths=None):          # clip by clip, as the reference does
        assert isinstance(audio, list)
        outs = [O.prepare_audio_batch(np.asarray(c, np.float32)[None], datasetconfig.patches_seq_len, backend="torch") for c in audio]
        return {k: torch.as_tensor(np.concatenate([o[k] for o in outs], 0)) for k in outs[0]}

    def prepare_text_batch(text, tokenizer, max_text_len, device=None):
        tok = tokenizer([text], padding="max_length", truncation=True, max_length=max_text_len, return_tensors="pt")
        return {"text_input_ids": tok["input_ids"], "text_mask": tok["attention_mask"]}

    def zs_scores(audio_emb, class_text_emb, target_idx, logit_scale=0.0, ks=(1,)):
        order = torch.argsort(-(math.exp(logit_scale) * audio_emb @ class_text_emb.T), dim=-1, stable=True).numpy()
        tgt = np.asarray(target_idx).reshape(-1)
        return {str(int(k)): float((order[:, :int(k)] == tgt[:, None]).any(1).mean()) for k in ks}

    def retrieval_scores(audio_emb, text_emb, k=10, sim=None):
        logits = text_emb @ audio_emb.T
        return (logits, torch.argsort(-logits.T, dim=-1, stable=True)[:, :k].to(torch.int32),
                torch.argsort(-logits, dim=-1, stable=True)[:, :k].to(torch.int32))

    monkeypatch.setattr(frontend, "prepare_audio_batch", prepare_audio_batch)
    monkeypatch.setattr(frontend, "prepare_text_batch", prepare_text_batch)
    monkeypatch.setattr(retrieval, "zs_classification_scores", zs_scores)
    monkeypatch.setattr(retrieval, "audio_retrieval_scores", retrieval_scores)
    torch.set_num_threads(min(8, torch.get_num_threads()))
    _check_drivers(OracleModel(), expect_cuda=False)


@pytest.mark.gpu
def test_drivers_on_the_device(tiny_state):
    """The batched drivers on the real front end, towers and device scoring.  A random-weight model maps every clip to
    nearly the same embedding (cosines up to 0.99997, rank gaps down to 1e-6 in the oracle), so the checks are made
    robust to near-ties: (1) a clip embedded in a ragged batch equals the clip embedded alone (the reference's way);
    (2) the zero-shot accuracy equals a float64 host scoring of the same embeddings (top-1 margins are checked to be
    far above fp32 noise); (3) the retrieval metrics are those of the device ranking of the same embeddings, lined up
    with the right names, and that ranking agrees with a float64 host ranking value for value at every rank."""
    from cacophony_amd.model import CACO
    a, t, cc = C.tiny_configs(2)
    assert t.vocab_size == TEXT_VOCAB
    model = CACO(a, t, cc, device="cuda:0").load_state_dict(tiny_state)
    _check_drivers(model, expect_cuda=True)


def test_device_test_body_on_the_simulator(monkeypatch, tiny_state):
    """The GPU case above with the REAL kernels - the mel front end with per-clip lengths, both towers, the exact-fp32
    similarity through the banks' strides and the top-k selection in both directions - executed by the wavesim build of the
    kernel sources (tools/wavesim) behind stand-ins that make the same C-ABI calls as frontend.py / model.py / retrieval.py
    do.  CPU only; what it adds to the oracle dry run is the device code of the scoring path under the drivers."""
    from tests import simlib
    if not simlib.available():
        pytest.skip("no host clang++ for the wavesim build")
    import ctypes
    sim = simlib.load()
    P = simlib.ptr
    a, t, cc = C.tiny_configs(1)                 # one layer of each tower (the first layer's weights of the 2-layer state): half the time
    m = simlib.SimModel(a, t, cc).load_state_dict(tiny_state)

    class SimBackedModel:
        logit_scale = torch.tensor(float(cc.logit_scale_init_value))

        def get_audio_embedding(self, audio_patches, audio_time_inds, audio_freq_inds, audio_mask, deterministic=True,
                                return_hidden_state=True, normalize=False):
            emb, hid = m.audio_forward(audio_patches, audio_time_inds, audio_freq_inds, audio_mask, normalize=normalize)
            return (emb, hid) if return_hidden_state else emb

        def get_text_embedding(self, text_input_ids, text_mask, position_ids=None, deterministic=True, return_hidden_state=True,
                               normalize=False):
            emb, hid = m.text_forward(text_input_ids, text_mask, normalize=normalize, position_ids=position_ids)
            return (emb, hid) if return_hidden_state else emb

    def prepare_audio_batch(audio, datasetconfig, device=None, lengths=None):     # frontend.prepare_audio_batch on a list of clips
        assert isinstance(audio, list)
        lens = torch.tensor([len(c) for c in audio], dtype=torch.int64)
        n = int(lens.max())
        wav = torch.zeros(len(audio), n)
        for i, c in enumerate(audio):
            wav[i, : len(c)] = torch.as_tensor(np.asarray(c, np.float32))
        S = datasetconfig.patches_seq_len
        out = {"audio_patches": torch.empty(len(audio), S, 256), "audio_time_inds": torch.empty(len(audio), S),
               "audio_freq_inds": torch.empty(len(audio), S), "audio_mask": torch.empty(len(audio), S)}
        simlib.check(sim.caco_mel_patches_lens(P(wav), P(lens), len(audio), n, S, 0.2, 0.9, P(out["audio_patches"]), 0,
                                               P(out["audio_time_inds"]), P(out["audio_freq_inds"]), P(out["audio_mask"]), None))
        return out

    def prepare_text_batch(text, tokenizer, max_text_len, device=None):
        tok = tokenizer([text], padding="max_length", truncation=True, max_length=max_text_len, return_tensors="pt")
        return {"text_input_ids": tok["input_ids"], "text_mask": tok["attention_mask"]}

    def similarity(x, y, scale=1.0):
        x, y = x.float().contiguous(), y.float().contiguous()
        out = torch.empty(x.shape[0], y.shape[0])
        simlib.check(sim.caco_similarity_ld(P(x), x.shape[0], x.stride(0), P(y), y.shape[0], y.stride(0), x.shape[1], float(scale),
                                            P(out), out.stride(0), None))
        return out

    def topk(s, k, dim=1):                                                          # retrieval.topk: dim 0 through the strides
        rows, cols, rs, cs = (s.shape[0], s.shape[1], s.stride(0), s.stride(1)) if dim == 1 else (s.shape[1], s.shape[0], s.stride(1), s.stride(0))
        k = min(int(k), cols)
        idx = torch.empty(rows, k, dtype=torch.int32)
        simlib.check(sim.caco_topk(P(s), rows, cols, rs, cs, k, P(idx), None, None))
        return idx

    def zs_scores(audio_emb, class_text_emb, target_idx, logit_scale=0.0, ks=(1,)):
        idx = topk(similarity(audio_emb, class_text_emb, float(np.exp(logit_scale))), int(max(ks))).numpy()
        tgt = np.asarray(target_idx).astype(np.int64).reshape(-1)
        return {str(int(k)): float((idx[:, :int(k)] == tgt[:, None]).any(axis=1).mean()) for k in ks}

    def retrieval_scores(audio_emb, text_emb, k=10, sim=None):
        logits = similarity(text_emb, audio_emb, 1.0)
        return logits, topk(logits, k, dim=0), topk(logits, k, dim=1)

    monkeypatch.setattr(frontend, "prepare_audio_batch", prepare_audio_batch)
    monkeypatch.setattr(frontend, "prepare_text_batch", prepare_text_batch)
    monkeypatch.setattr(retrieval, "zs_classification_scores", zs_scores)
    monkeypatch.setattr(retrieval, "audio_retrieval_scores", retrieval_scores)
    _check_drivers(SimBackedModel(), expect_cuda=False)
