"""Pin the CPU oracle (oracle/caco_oracle.py) against outputs of the reference itself.

The fixtures in tests/golden/*.npz were produced by tests/golden/make_golden.py, which imports
/root/reference/src/caco_torch (+ the eval pre-processing functions) in the build container.
Tolerances: fp32 restatement vs fp32 reference -> 1e-5 relative on hidden states / embeddings,
1e-4 absolute on log-mel values.
"""
from dataclasses import replace

import os

import numpy as np
import pytest

from cacophony_amd import config as C
from cacophony_amd import synth
from oracle import caco_oracle as O
from tests.conftest import checksum, cosine_rows, load_golden, rel_l2


# ------------------------------------------------------------------ synth determinism
def test_synth_reproduces_golden_inputs(tiny_state):
    g = load_golden("mel.npz")
    wav = synth.make_waveforms(2)
    np.testing.assert_allclose(np.stack([checksum(wav[0]), checksum(wav[1])]), g["wav_checksum"], rtol=1e-12)
    t = load_golden("caco_tiny.npz")
    ids, mask = synth.make_captions(2, 32, 1024)
    np.testing.assert_array_equal(ids, t["ids"])
    np.testing.assert_array_equal(mask, t["tmask"])
    keys = sorted(tiny_state)[:: max(1, len(tiny_state) // 24)]
    np.testing.assert_allclose(np.stack([checksum(tiny_state[k]) for k in keys]), t["state_checksum"], rtol=1e-12)


def test_state_dict_contract(full_state):
    """465-tensor reference contract minus decoder_module.* (SURVEY.md section 8b)."""
    sd = full_state
    assert sd["audio_module.layers.0.attn.in_proj_weight"].shape == (2304, 768)
    assert sd["audio_module.input_proj.weight"].shape == (768, 256)
    assert sd["audio_attention_pool.kv_proj.weight"].shape == (1536, 768)
    assert sd["text_module.embeddings.word_embeddings.weight"].shape == (50265, 768)
    assert sd["text_module.pooler.attention_pool_query"].shape == (1, 768)
    assert sd["text_proj.weight"].shape == (768, 768)
    assert sd["logit_scale"].shape == ()
    n_audio = sum(v.size for k, v in sd.items() if k.startswith("audio_"))
    n_text = sum(v.size for k, v in sd.items() if k.startswith("text_"))
    assert abs(n_audio / 1e6 - (85.26 + 1.77)) < 0.05    # encoder 85.26 M + pooler
    assert abs(n_text / 1e6 - 125.23) < 0.7               # README.md:67-69 (+ text_proj)


# ------------------------------------------------------------------ front end
@pytest.mark.parametrize("backend", ["numpy", "torch"])
def test_mel_matches_reference(backend):
    g = load_golden("mel.npz")
    wav = synth.make_waveforms(2)
    mel0 = O.compute_mel_spectrogram(wav[0], backend=backend)
    assert mel0.shape == (1000, 128) and mel0.dtype == np.float32
    assert np.abs(mel0 - g["mel0"]).max() < 1e-4
    mel1 = O.compute_mel_spectrogram(wav[1], backend=backend)
    assert np.abs(mel1 - g["mel1"].astype(np.float32)).max() < 2e-3      # fp16-stored fixture
    # known answer: the all-zero HTK filter column is log(1e-5)*0.2+0.9
    assert np.allclose(mel0[:, 0], np.log(np.float32(1e-5)) * 0.2 + 0.9, atol=1e-6)
    p = O.spectrogram_to_patches(mel0, 16, 16, 500)
    np.testing.assert_allclose(p["audio_patches"][g["patch_rows0"].shape[0] * 0 + np.array([0, 1, 2, 3, 100, 247, 248, 495, 496, 499])],
                               g["patch_rows0"], atol=1e-4)
    for k in ("time_inds", "freq_inds", "mask"):
        np.testing.assert_array_equal(p["audio_" + k], g[k + "0"])
    assert p["audio_mask"].sum() == 496


@pytest.mark.parametrize("tag,n,max_p", [("short", 12345, 64), ("tiny", 700, 16), ("3s", 48000, 500), ("trunc", 48000, 100)])
def test_mel_ragged_lengths(tag, n, max_p):
    g = load_golden("mel.npz")
    w = synth.make_waveform(7, n_samples=n)
    mel = O.compute_mel_spectrogram(w)
    assert mel.shape == g[f"{tag}_mel"].shape
    assert np.abs(mel - g[f"{tag}_mel"]).max() < 1e-4
    p = O.spectrogram_to_patches(mel, 16, 16, max_p)
    for k in ("audio_time_inds", "audio_freq_inds", "audio_mask"):
        np.testing.assert_array_equal(p[k], g[f"{tag}_{k}"])
    assert np.abs(p["audio_patches"] - g[f"{tag}_audio_patches"].astype(np.float32)).max() < 2e-3


def test_mel_filterbank_known_answers():
    fb = O.melscale_fbanks_htk(257, 0.0, 8000.0, 128, 16000)
    assert fb.shape == (257, 128)
    assert (fb.sum(0) == 0).sum() == 1 and fb[:, 0].sum() == 0        # one empty filter
    assert ((fb != 0).sum(1) <= 2).all()                               # triangular overlap: <= 2 filters per bin
    assert fb.min() >= 0 and fb.max() <= 1.0


# ------------------------------------------------------------------ model
def _audio_inputs(batch, max_patches=500, n_samples=160000, start=0):
    return O.prepare_audio_batch(synth.make_waveforms(batch, n_samples, start=start), max_patches)


def _check_caco(g, state, a, t, cc, batch, backend, tol):
    m = O.CacoOracle(state, a, t, cc, backend=backend)
    ab = _audio_inputs(batch)
    np.testing.assert_allclose(checksum(ab["audio_patches"]), g["patch_checksum"], rtol=1e-4)
    probes = {}
    a_emb, a_hid = m.get_audio_embedding(ab["audio_patches"], ab["audio_time_inds"], ab["audio_freq_inds"],
                                         ab["audio_mask"], probes=probes)
    rows = g["probe_rows"]
    assert rel_l2(a_hid[:, rows], g["audio_hidden_rows"]) < tol
    for k in [k for k in g if k.startswith("audio_layer")]:
        n = int(k[len("audio_layer"):-len("_rows")])
        assert rel_l2(probes[f"audio_layer{n}"][:, rows], g[k]) < tol, k
    assert rel_l2(a_emb, g["audio_emb"]) < tol
    tprobes = {}
    t_emb, t_hid = m.get_text_embedding(g["ids"], g["tmask"], probes=tprobes)
    assert rel_l2(t_hid, g["text_hidden"]) < tol
    for k in [k for k in g if k.startswith("text_layer")]:
        assert rel_l2(tprobes[k], g[k]) < tol, k
    assert rel_l2(t_emb, g["text_emb"]) < tol
    a_n = m.get_audio_embedding(ab["audio_patches"], ab["audio_time_inds"], ab["audio_freq_inds"], ab["audio_mask"],
                                return_hidden_state=False, normalize=True)
    t_n = m.get_text_embedding(g["ids"], g["tmask"], return_hidden_state=False, normalize=True)
    assert cosine_rows(a_n, g["audio_emb_norm"]).min() > 1 - 1e-6
    assert cosine_rows(t_n, g["text_emb_norm"]).min() > 1 - 1e-6
    np.testing.assert_allclose(np.linalg.norm(a_n, axis=1), 1.0, atol=1e-5)
    at, ta = m.get_contrastive_logits(ab["audio_patches"], ab["audio_time_inds"], ab["audio_freq_inds"],
                                      ab["audio_mask"], g["ids"], g["tmask"])
    assert np.abs(at - g["at_logits"]).max() < 1e-3 and np.abs(ta - g["ta_logits"]).max() < 1e-3
    np.testing.assert_allclose(at, ta.T, atol=1e-5)
    pos = np.broadcast_to(np.arange(32) + 2, (batch, 32)).copy()
    t_pos = m.get_text_embedding(g["ids"], g["tmask"], position_ids=pos, return_hidden_state=False)
    assert rel_l2(t_pos, g["text_emb_pos2"]) < tol


@pytest.mark.parametrize("backend", ["numpy", "torch"])
def test_caco_tiny_matches_reference(tiny_state, backend):
    a, t, cc = C.tiny_configs(2)
    _check_caco(load_golden("caco_tiny.npz"), tiny_state, a, t, cc, 2, backend, 2e-5)


def test_caco_full_matches_reference(full_state):
    _check_caco(load_golden("caco_full.npz"), full_state, C.default_audio_config(), C.default_text_config(),
                C.default_caco_config(), 4, "torch", 5e-5)


def test_caco_varlen_matches_reference(tiny_state):
    """arbitrary valid-patch count (3 s clip in a 500 window) and the 30 s / S=1500 retrieval shape."""
    g = load_golden("caco_varlen.npz")
    a, t, cc = C.tiny_configs(2)
    m = O.CacoOracle(tiny_state, a, t, cc)
    for tag, n, max_p in (("3s", 48000, 500), ("30s", 480000, 1500)):
        ab = _audio_inputs(2, max_p, n, start=20)
        np.testing.assert_array_equal(ab["audio_mask"].sum(1), g[f"{tag}_mask_sum"])
        emb, hid = m.get_audio_embedding(ab["audio_patches"], ab["audio_time_inds"], ab["audio_freq_inds"],
                                         ab["audio_mask"], normalize=True)
        assert cosine_rows(emb, g[f"{tag}_emb"]).min() > 1 - 1e-6
        assert rel_l2(hid[:, g[f"{tag}_rows"]], g[f"{tag}_hidden_rows"]) < 2e-5


@pytest.mark.parametrize("tag,layers", [("tiny", 2), ("full", 12)])
def test_audiomae_matches_reference(tag, layers):
    g = load_golden(f"mae_{tag}.npz")
    enc = replace(C.default_audio_config(), num_layers=layers)
    sd = synth.make_audiomae_state(enc, enc)
    ab = _audio_inputs(2)
    sp = synth.make_mae_split(2, 496, 100, 8)
    np.testing.assert_array_equal(sp["visible"], g["visible"])
    x = np.stack([ab["audio_patches"][i][sp["visible"][i]] for i in range(2)])
    m = O.AudioMAEOracle(sd, enc, enc, backend="torch")
    y = m.forward(x, np.ones((2, 100), np.float32), sp["time_inds"], sp["freq_inds"], sp["restore_time_inds"],
                  sp["restore_freq_inds"], np.ones((2, 396), np.float32))
    assert y.shape == (2, 496, 256)
    assert rel_l2(y[:, g["rows"]], g["out_rows"]) < 5e-5
    np.testing.assert_allclose(checksum(y), g["out_checksum"], rtol=2e-3)


def test_normalize_zero_vector_is_finite():
    """x / ||x + 1e-10||: an all-zero embedding stays finite."""
    z = O.l2_normalize(O.get_ops("numpy"), np.zeros((1, 768), np.float32))
    assert np.isfinite(z).all() and (z == 0).all()


def test_audio_pooler_head_counts_match_reference(tiny_state):
    """1, 4 and 8 pooling heads on the same tensors (the JAX side's 8, src/caco/load_model.py:46): the oracle against the
    reference's own AudioAttentionPooler at each head count (tests/golden/pool_heads.npz)."""
    g = load_golden("pool_heads.npz")
    a, t, cc = C.tiny_configs(2)
    ab = O.prepare_audio_batch(synth.make_waveforms(2, 48000, start=30), 150)
    np.testing.assert_allclose(checksum(ab["audio_patches"]), g["patch_checksum"], rtol=1e-4)
    for heads in (1, 4, 8):
        o = O.CacoOracle(tiny_state, a, t, replace(cc, num_attention_pool_heads=heads), backend="torch")
        emb = o.get_audio_embedding(ab["audio_patches"], ab["audio_time_inds"], ab["audio_freq_inds"], ab["audio_mask"],
                                    return_hidden_state=False, normalize=True)
        assert rel_l2(emb, g[f"emb_heads{heads}"]) < 2e-5, heads
    assert rel_l2(g["emb_heads8"], g["emb_heads4"]) > 1e-2          # the head count matters: a silent fall-back would show


def test_rational_erf_of_the_device_gelu_is_erf_to_fp32_accuracy():
    """csrc/common.h erf_rational_f (the erf inside every GEMM kernel's erf-GELU epilogue) restated in NumPy fp32 with the same
    coefficients and operation order, against scipy.special.erf: <= 5e-7 absolute on [-6, 6], <= 3e-7 relative for |z| < 1; the
    GELU built on it is within 2e-6 of the oracle's exact-erf GELU - three orders below the bf16 rounding of the activation."""
    import re
    from scipy.special import erf
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cacophony_amd", "csrc", "common.h")).read()
    body = src[src.index("float erf_rational_f(float z)"):src.index("float gelu_erf_f(float x)")]
    coef = [np.float32(c) for c in re.findall(r"(-?\d\.\d+e-\d+)f", body)]
    assert len(coef) == 12, coef                      # 7 numerator + 5 denominator coefficients, read from the kernel source itself
    alpha, beta = coef[:7], coef[7:]

    def erf_dev(z):
        z = np.clip(z.astype(np.float32), np.float32(-4), np.float32(4))
        z2 = z * z
        p = np.full_like(z, alpha[0])
        for a in alpha[1:]:
            p = p * z2 + a
        q = np.full_like(z, beta[0])
        for b in beta[1:]:
            q = q * z2 + b
        return (z * p) / q

    z = np.linspace(-6, 6, 1200001).astype(np.float32)
    exact = erf(z.astype(np.float64))
    got = erf_dev(z).astype(np.float64)
    assert np.abs(got - exact).max() < 5e-7
    small = np.abs(z) < 1
    assert (np.abs(got - exact)[small] / np.maximum(np.abs(exact[small]), 1e-30)).max() < 3e-7
    x = z
    gelu_exact = 0.5 * x.astype(np.float64) * (1.0 + erf(x.astype(np.float64) / np.sqrt(2.0)))
    h = np.float32(0.5) * x
    gelu_dev = h * erf_dev(x * np.float32(0.70710678118654752440)) + h
    assert np.abs(gelu_dev.astype(np.float64) - gelu_exact).max() < 2e-6
