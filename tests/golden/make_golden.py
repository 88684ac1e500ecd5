#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE itself.

Runs only in the build container (needs /root/reference, which never travels to the GPU box):
imports `src.caco_torch` and the pre-processing functions of `src/eval/eval_caco_torch.py`,
loads this build's seeded synthetic state dict into the reference modules
(`load_state_dict(strict=True)`), runs them on the seeded synthetic inputs of
`cacophony_amd.synth`, and stores inputs' checksums + expected outputs as small .npz files.
The fixtures are data only (no reference source, bytecode or text).

`src/eval/eval_caco_torch.py` imports torchaudio / soundfile / astropy, none installed here.
They are stubbed in `sys.modules`; the one function actually called on the path,
`torchaudio.functional.melscale_fbanks`, is served by the independent third-party
`transformers.audio_utils.mel_filter_bank` (HTK scale, norm=None), which is NOT the oracle's
restatement, so the mel goldens are a second opinion on that restatement.

Usage:  python tests/golden/make_golden.py           (from the repo root)
"""
from __future__ import annotations

import os
import sys
import types
import warnings
from dataclasses import replace

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)

from cacophony_amd import config as C  # noqa: E402
from cacophony_amd import synth  # noqa: E402

PROBE_ROWS = np.array([0, 1, 2, 3, 100, 247, 248, 495, 496, 499])


def _import_reference():
    sys.path.insert(0, REF)
    import importlib.machinery
    from transformers import RobertaTokenizerFast  # noqa: F401  (resolve the lazy import before stubbing)
    from transformers.audio_utils import mel_filter_bank  # noqa: F401
    ta = types.ModuleType("torchaudio")
    taf = types.ModuleType("torchaudio.functional")

    def melscale_fbanks(n_freqs, f_min, f_max, n_mels, sample_rate, norm=None, mel_scale="htk"):
        from transformers.audio_utils import mel_filter_bank
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            fb = mel_filter_bank(n_freqs, n_mels, f_min, f_max, sample_rate, norm=norm, mel_scale=mel_scale)
        return torch.from_numpy(np.asarray(fb, dtype=np.float32))

    taf.melscale_fbanks = melscale_fbanks
    ta.functional = taf
    ta.__spec__ = importlib.machinery.ModuleSpec("torchaudio", None)
    taf.__spec__ = importlib.machinery.ModuleSpec("torchaudio.functional", None)
    sys.modules["torchaudio"] = ta
    sys.modules["torchaudio.functional"] = taf
    for name in ("soundfile", "astropy", "astropy.stats"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["astropy.stats"].jackknife = None
    sys.modules["astropy"].stats = sys.modules["astropy.stats"]
    from src.caco_torch import caco as ref_caco
    from src.caco_torch.audio_models import mae as ref_mae
    from src.caco_torch.text_models import roberta as ref_roberta
    from src.eval import eval_caco_torch as ref_eval
    return ref_caco, ref_mae, ref_roberta, ref_eval


def _to_torch_state(sd):
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}


def _ref_audio_cfg(ref_mae, c: C.AudioTransformerConfig):
    return ref_mae.AudioTransformerConfig(
        hidden_size=c.hidden_size, num_layers=c.num_layers, num_heads=c.num_heads,
        intermediate_size=c.intermediate_size, patch_size=c.patch_size, max_time_ind=c.max_time_ind,
        num_freq_patches=c.num_freq_patches, dropout_rate=c.dropout_rate, drop_path_rate=c.drop_path_rate)


def _ref_text_cfg(ref_roberta, c: C.RobertaConfig):
    return ref_roberta.RobertaConfig(
        vocab_size=c.vocab_size, hidden_size=c.hidden_size, num_hidden_layers=c.num_hidden_layers,
        num_attention_heads=c.num_attention_heads, intermediate_size=c.intermediate_size,
        max_position_embeddings=c.max_position_embeddings, type_vocab_size=c.type_vocab_size,
        layer_norm_eps=c.layer_norm_eps, pad_token_id=c.pad_token_id)


def _build_ref_caco(ref_caco, ref_mae, ref_roberta, a, t, cc, seed=0):
    model = ref_caco.CACO(_ref_audio_cfg(ref_mae, a), _ref_text_cfg(ref_roberta, t),
                          ref_caco.CACOConfig(cc.projection_size, cc.num_attention_pool_heads,
                                              cc.logit_scale_init_value), decoder_config=None)
    sd = synth.make_caco_state(a, t, cc, seed=seed)
    model.load_state_dict(_to_torch_state(sd), strict=True)
    return model.eval(), sd


def _checksum(x: np.ndarray) -> np.ndarray:
    x = np.asarray(x, dtype=np.float64).reshape(-1)
    return np.array([x.sum(), np.abs(x).sum(), (x * np.cos(np.arange(x.size) * 0.37)).sum()])


def _hook_layers(layers, store, key):
    handles = []
    for n, layer in enumerate(layers):
        def hook(_m, _i, out, n=n):
            store[f"{key}{n}"] = out.detach().numpy().copy()
        handles.append(layer.register_forward_hook(hook))
    return handles


@torch.no_grad()
def golden_mel(ref_eval):
    out = {}
    wav = synth.make_waveforms(2)
    for i in range(2):
        mel = ref_eval.compute_mel_spectrogram(torch.from_numpy(wav[i]))
        out[f"mel{i}"] = mel.astype(np.float32 if i == 0 else np.float16)
        p = ref_eval.spectrogram_to_patches(mel, 16, 16, 500)
        out[f"patch_rows{i}"] = p["audio_patches"][PROBE_ROWS]
        out[f"time_inds{i}"] = p["audio_time_inds"]
        out[f"freq_inds{i}"] = p["audio_freq_inds"]
        out[f"mask{i}"] = p["audio_mask"]
    out["wav_checksum"] = np.stack([_checksum(wav[0]), _checksum(wav[1])])
    # ragged lengths: not a hop multiple / shorter than one patch row / 3 s clip / truncation branch
    for tag, n, max_p in (("short", 12345, 64), ("tiny", 700, 16), ("3s", 48000, 500), ("trunc", 48000, 100)):
        w = synth.make_waveform(7, n_samples=n)
        mel = ref_eval.compute_mel_spectrogram(torch.from_numpy(w))
        p = ref_eval.spectrogram_to_patches(mel, 16, 16, max_p)
        out[f"{tag}_mel"] = mel.astype(np.float32)
        for k, v in p.items():
            out[f"{tag}_{k}"] = v if k != "audio_patches" else v.astype(np.float16)
    np.savez_compressed(os.path.join(OUT, "mel.npz"), **out)
    print("mel.npz", {k: v.shape for k, v in out.items() if k.startswith("mel")})


def _inputs(ref_eval, batch, max_patches=500, n_samples=160000, start=0):
    wav = synth.make_waveforms(batch, n_samples, start=start)
    ps = [ref_eval.spectrogram_to_patches(ref_eval.compute_mel_spectrogram(torch.from_numpy(w)), 16, 16, max_patches)
          for w in wav]
    return {k: np.stack([p[k] for p in ps]) for k in ps[0]}


@torch.no_grad()
def golden_caco(refs, tag, a, t, cc, batch, full_text_hidden):
    ref_caco, ref_mae, ref_roberta, ref_eval = refs
    model, sd = _build_ref_caco(ref_caco, ref_mae, ref_roberta, a, t, cc)
    ab = _inputs(ref_eval, batch)
    ids, tmask = synth.make_captions(batch, 32, t.vocab_size)
    out = {"ids": ids, "tmask": tmask, "probe_rows": PROBE_ROWS}
    store = {}
    hs = _hook_layers(model.audio_module.layers, store, "audio_layer")
    hs += _hook_layers(model.text_module.encoder.layers, store, "text_layer")
    tt = {k: torch.from_numpy(v) for k, v in ab.items()}
    a_emb, a_hid = model.get_audio_embedding(tt["audio_patches"], tt["audio_time_inds"], tt["audio_freq_inds"],
                                             tt["audio_mask"])
    t_emb, t_hid = model.get_text_embedding(torch.from_numpy(ids), torch.from_numpy(tmask))
    a_n = model.get_audio_embedding(tt["audio_patches"], tt["audio_time_inds"], tt["audio_freq_inds"],
                                    tt["audio_mask"], return_hidden_state=False, normalize=True)
    t_n = model.get_text_embedding(torch.from_numpy(ids), torch.from_numpy(tmask), return_hidden_state=False,
                                   normalize=True)
    at, ta = model(tt["audio_patches"], tt["audio_time_inds"], tt["audio_freq_inds"], tt["audio_mask"],
                   torch.from_numpy(ids), torch.from_numpy(tmask))
    for h in hs:
        h.remove()
    out.update(audio_emb=a_emb.numpy(), audio_emb_norm=a_n.numpy(), text_emb=t_emb.numpy(), text_emb_norm=t_n.numpy(),
               at_logits=at.numpy(), ta_logits=ta.numpy(), audio_hidden_rows=a_hid.numpy()[:, PROBE_ROWS],
               text_hidden=t_hid.numpy())
    keep_layers = range(a.num_layers) if a.num_layers <= 2 else (0, a.num_layers // 2 - 1, a.num_layers - 1)
    for n in keep_layers:
        out[f"audio_layer{n}_rows"] = store[f"audio_layer{n}"][:, PROBE_ROWS]
    keep_t = range(t.num_hidden_layers) if full_text_hidden else (0, t.num_hidden_layers // 2 - 1)
    for n in keep_t:
        out[f"text_layer{n}"] = store[f"text_layer{n}"]
    # explicit position_ids path (roberta.py:292): offset positions, same weights
    pos = np.broadcast_to(np.arange(32) + 2, (batch, 32)).copy()
    t_pos = model.get_text_embedding(torch.from_numpy(ids), torch.from_numpy(tmask), position_ids=torch.from_numpy(pos),
                                     return_hidden_state=False)
    out["text_emb_pos2"] = t_pos.numpy()
    out["state_checksum"] = np.stack([_checksum(sd[k]) for k in sorted(sd)[:: max(1, len(sd) // 24)]])
    out["patch_checksum"] = _checksum(ab["audio_patches"])
    np.savez_compressed(os.path.join(OUT, f"caco_{tag}.npz"), **out)
    print(f"caco_{tag}.npz", at.numpy().round(3).tolist()[0])


@torch.no_grad()
def golden_varlen(refs):
    """3 s clips in a 500-patch window (arbitrary valid count) and the 30 s / S=1500 retrieval shape
    (src/eval/eval_caco_torch.py:607-617), 2-layer config to keep the fixture small."""
    ref_caco, ref_mae, ref_roberta, ref_eval = refs
    a, t, cc = C.tiny_configs(2)
    model, _ = _build_ref_caco(ref_caco, ref_mae, ref_roberta, a, t, cc)
    out = {"probe_rows": PROBE_ROWS}
    for tag, n, max_p in (("3s", 48000, 500), ("30s", 480000, 1500)):
        ab = _inputs(ref_eval, 2, max_p, n, start=20)
        tt = {k: torch.from_numpy(v) for k, v in ab.items()}
        emb, hid = model.get_audio_embedding(tt["audio_patches"], tt["audio_time_inds"], tt["audio_freq_inds"],
                                             tt["audio_mask"], normalize=True)
        rows = np.minimum(PROBE_ROWS * (3 if max_p == 1500 else 1), max_p - 1)
        out[f"{tag}_emb"] = emb.numpy()
        out[f"{tag}_rows"] = rows
        out[f"{tag}_hidden_rows"] = hid.numpy()[:, rows]
        out[f"{tag}_mask_sum"] = ab["audio_mask"].sum(1)
    np.savez_compressed(os.path.join(OUT, "caco_varlen.npz"), **out)
    print("caco_varlen.npz", out["3s_mask_sum"], out["30s_mask_sum"])


@torch.no_grad()
def golden_mae(refs, tag, layers):
    ref_caco, ref_mae, ref_roberta, ref_eval = refs
    enc = replace(C.default_audio_config(), num_layers=layers)
    dec = replace(C.default_audio_config(), num_layers=layers)
    model = ref_mae.AudioMAE(ref_mae.AudioMAEConfig(_ref_audio_cfg(ref_mae, enc), _ref_audio_cfg(ref_mae, dec))).eval()
    sd = synth.make_audiomae_state(enc, dec, seed=0)
    model.load_state_dict(_to_torch_state(sd), strict=True)
    batch = 2
    ab = _inputs(ref_eval, batch)
    sp = synth.make_mae_split(batch, 496, 100, 8)
    x = np.stack([ab["audio_patches"][i][sp["visible"][i]] for i in range(batch)])
    mask = np.ones((batch, 100), np.float32)
    rmask = np.ones((batch, 396), np.float32)
    y = model(torch.from_numpy(x), torch.from_numpy(mask), torch.from_numpy(sp["time_inds"]),
              torch.from_numpy(sp["freq_inds"]), torch.from_numpy(sp["restore_time_inds"]),
              torch.from_numpy(sp["restore_freq_inds"]), torch.from_numpy(rmask)).numpy()
    rows = np.array([0, 1, 50, 99, 100, 101, 250, 494, 495])
    np.savez_compressed(os.path.join(OUT, f"mae_{tag}.npz"), rows=rows, out_rows=y[:, rows],
                        out_checksum=_checksum(y), visible=sp["visible"], restore=sp["restore"])
    print(f"mae_{tag}.npz", y.shape, float(np.abs(y).mean()))


def golden_retrieval():
    """compute_retrieval_metric of the reference (src/eval/eval_utils.py:18-67) on the seeded scenario.  astropy is not
    installed: its jackknife_stats is replaced by a recorder, so the per-query R1/R5/R10/mAP10 lists ARE the
    reference's; the interval arithmetic is pinned separately against the closed form in the tests."""
    import contextlib
    import io
    from src.eval import eval_utils as ref_utils
    recorded = []

    class _Jack:
        @staticmethod
        def jackknife_stats(data, fn, conf):
            recorded.append(np.asarray(data, dtype=np.float64).copy())
            m = float(fn(data))
            return m, 0.0, 0.0, (m, m)

    ref_utils.jackknife = _Jack
    all_audio, all_text, gt_at, gt_ta, A, T = synth.make_retrieval_scenario()
    logits_ar = torch.from_numpy(T) @ torch.from_numpy(A).T                       # eval_caco_torch.py:398
    at_indices = torch.argsort(-logits_ar.T, dim=-1, stable=True).cpu().numpy()   # :403 (stable: ties by index)
    ta_indices = torch.argsort(-logits_ar, dim=-1, stable=True).cpu().numpy()     # :407
    with contextlib.redirect_stdout(io.StringIO()):
        ref_utils.compute_retrieval_metric(at_indices, all_audio, all_text, gt_at)
        ref_utils.compute_retrieval_metric(ta_indices, all_text, all_audio, gt_ta, "ta")
    names = ["R1", "R5", "R10", "mAP10"]
    out = {f"at_{n}": recorded[i] for i, n in enumerate(names)}
    out.update({f"ta_{n}": recorded[4 + i] for i, n in enumerate(names)})
    np.savez_compressed(os.path.join(OUT, "retrieval.npz"), logits_ar=logits_ar.numpy(), at_top10=at_indices[:, :10].astype(np.int32),
                        ta_top10=ta_indices[:, :10].astype(np.int32), **out)
    print("retrieval.npz", {k: float(v.mean()) for k, v in out.items()})


@torch.no_grad()
def golden_decoder(refs):
    """Caption decoder: CACO.get_decoder_logits (caco.py:212-240) on the 2-layer configuration with a
    2-layer RobertaDecoder, fed the reference's own audio hidden states: full 32-token captions (one padded) and a
    12-token prefix as the sampling loop would present it (eval_caco_torch.py:443-456 intends this call)."""
    ref_caco, ref_mae, ref_roberta, ref_eval = refs
    a, t, cc = C.tiny_configs(2)
    d = replace(t, num_hidden_layers=2)
    model = ref_caco.CACO(_ref_audio_cfg(ref_mae, a), _ref_text_cfg(ref_roberta, t),
                          ref_caco.CACOConfig(cc.projection_size, cc.num_attention_pool_heads, cc.logit_scale_init_value),
                          decoder_config=_ref_text_cfg(ref_roberta, d))
    sd = synth.make_caco_state(a, t, cc, seed=0, decoder_cfg=d)
    model.load_state_dict(_to_torch_state(sd), strict=True)
    model.eval()
    ab = _inputs(ref_eval, 2, start=40)
    tt = {k: torch.from_numpy(v) for k, v in ab.items()}
    _, a_hid = model.get_audio_embedding(tt["audio_patches"], tt["audio_time_inds"], tt["audio_freq_inds"], tt["audio_mask"])
    ids, tmask = synth.make_captions(2, 32, t.vocab_size, start=40)
    out = {"ids": ids, "tmask": tmask, "audio_hidden_checksum": _checksum(a_hid.numpy())}
    lg = model.get_decoder_logits(a_hid, tt["audio_mask"], torch.from_numpy(ids), torch.from_numpy(tmask))
    out["logits"] = lg.numpy()
    ids12, m12 = ids[:, :12].copy(), np.ones((2, 12), dtype=np.int64)
    lg12 = model.get_decoder_logits(a_hid, tt["audio_mask"], torch.from_numpy(ids12), torch.from_numpy(m12))
    out["logits_prefix12_last"] = lg12.numpy()[:, -1]
    # decoder alone on seeded random hidden states with a ragged audio mask (pure RobertaDecoder.forward)
    rng = np.random.RandomState(5)
    th = rng.randn(2, 20, t.hidden_size).astype(np.float32)
    ah = rng.randn(2, 70, t.hidden_size).astype(np.float32)
    tm = np.ones((2, 20), dtype=np.int64); tm[1, 13:] = 0
    am = np.ones((2, 70), dtype=np.float32); am[0, 50:] = 0
    lg_r = model.decoder_module(text_hidden_state=torch.from_numpy(th), attention_mask=torch.from_numpy(tm),
                                audio_hidden_state=torch.from_numpy(ah), audio_mask=torch.from_numpy(am))
    out["rand_logits"] = lg_r.numpy()
    np.savez_compressed(os.path.join(OUT, "decoder_tiny.npz"), **out)
    print("decoder_tiny.npz", lg.shape, lg.numpy()[0, :3].argmax(-1), float(np.abs(lg.numpy()).max()))


@torch.no_grad()
def golden_pool_heads(refs):
    """The audio pooler with 1, 4 and 8 heads on the SAME tensors (the JAX side pools with 8, src/caco/load_model.py:46; the
    torch default is 2, caco.py:20): the reference's own AudioAttentionPooler at each head count, 2-layer config, 3 s clips."""
    ref_caco, ref_mae, ref_roberta, ref_eval = refs
    a, t, cc = C.tiny_configs(2)
    ab = _inputs(ref_eval, 2, 150, 48000, start=30)
    tt = {k: torch.from_numpy(v) for k, v in ab.items()}
    out = {"mask_sum": ab["audio_mask"].sum(1), "patch_checksum": _checksum(ab["audio_patches"])}
    for heads in (1, 4, 8):
        model, _ = _build_ref_caco(ref_caco, ref_mae, ref_roberta, a, t, replace(cc, num_attention_pool_heads=heads))
        emb = model.get_audio_embedding(tt["audio_patches"], tt["audio_time_inds"], tt["audio_freq_inds"], tt["audio_mask"],
                                        return_hidden_state=False, normalize=True)
        out[f"emb_heads{heads}"] = emb.numpy()
    np.savez_compressed(os.path.join(OUT, "pool_heads.npz"), **out)
    print("pool_heads.npz", {k: v.shape for k, v in out.items()})


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    refs = _import_reference()
    golden_mel(refs[3])
    a, t, cc = C.tiny_configs(2)
    golden_caco(refs, "tiny", a, t, cc, batch=2, full_text_hidden=True)
    golden_caco(refs, "full", C.default_audio_config(), C.default_text_config(), C.default_caco_config(),
                batch=4, full_text_hidden=False)
    golden_varlen(refs)
    golden_mae(refs, "tiny", 2)
    golden_mae(refs, "full", 12)
    golden_retrieval()
    golden_decoder(refs)
    golden_pool_heads(refs)


if __name__ == "__main__":
    main()
