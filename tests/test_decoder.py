"""Caption decoder logits: CACO.get_decoder_logits (src/caco_torch/caco.py:212-240) ->
RobertaDecoder.forward (src/caco_torch/text_models/roberta.py:337-373).  Golden = the reference itself
(tests/golden/make_golden.py::golden_decoder)."""
from dataclasses import replace

import numpy as np
import pytest
import torch

from cacophony_amd import config as C
from cacophony_amd import synth
from oracle import caco_oracle as O
from tests.conftest import checksum, cosine_rows, load_golden, rel_l2

LOGIT_TOL = 1e-2          # rel-L2 of the logits: the same bar as hidden states; bf16 operands, fp32 accumulate
COS_TOL = 0.999           # per position


def _configs():
    a, t, cc = C.tiny_configs(2)
    return a, t, cc, replace(t, num_hidden_layers=2)


@pytest.fixture(scope="module")
def dec_state():
    a, t, cc, d = _configs()
    return synth.make_caco_state(a, t, cc, seed=0, decoder_cfg=d)


def test_oracle_decoder_matches_reference_golden(dec_state):
    g = load_golden("decoder_tiny.npz")
    a, t, cc, d = _configs()
    o = O.CacoOracle(dec_state, a, t, cc, backend="torch", decoder_cfg=d)
    batch = O.prepare_audio_batch(synth.make_waveforms(2, start=40), 500, backend="torch")
    _, ah = o.get_audio_embedding(batch["audio_patches"], batch["audio_time_inds"], batch["audio_freq_inds"], batch["audio_mask"])
    np.testing.assert_allclose(checksum(ah), g["audio_hidden_checksum"], rtol=2e-4)
    ids, tmask = synth.make_captions(2, 32, t.vocab_size, start=40)
    np.testing.assert_array_equal(ids, g["ids"])
    lg = o.get_decoder_logits(ah, batch["audio_mask"], ids, tmask)
    assert lg.shape == (2, 32, t.vocab_size)
    assert np.abs(lg - g["logits"]).max() < 2e-4
    lg12 = o.get_decoder_logits(ah, batch["audio_mask"], ids[:, :12], np.ones((2, 12), np.int64))
    assert np.abs(lg12[:, -1] - g["logits_prefix12_last"]).max() < 2e-4
    # the decoder alone on seeded random states with ragged masks on both sides
    rng = np.random.RandomState(5)
    th = rng.randn(2, 20, t.hidden_size).astype(np.float32)
    ah_r = rng.randn(2, 70, t.hidden_size).astype(np.float32)
    tm = np.ones((2, 20), dtype=np.int64); tm[1, 13:] = 0
    am = np.ones((2, 70), dtype=np.float32); am[0, 50:] = 0
    ops = o.ops
    lr = ops.to_numpy(O.roberta_decoder(ops, o.P, d, ops.f32(th), ops.i64(tm), ops.f32(ah_r), ops.f32(am)))
    assert np.abs(lr - g["rand_logits"]).max() < 2e-4
    with pytest.raises(ValueError, match="Decoder module not initialized"):
        O.CacoOracle(dec_state, a, t, cc, backend="torch").get_decoder_logits(ah, batch["audio_mask"], ids, tmask)


@pytest.fixture(scope="module")
def dec_model(dec_state):
    from cacophony_amd.model import CACO
    a, t, cc, d = _configs()
    return CACO(a, t, cc, decoder_config=d, device="cuda:0").load_state_dict(dec_state)


@pytest.mark.gpu
def test_decoder_logits_match_reference_golden(dec_model):
    from cacophony_amd import frontend
    g = load_golden("decoder_tiny.npz")
    wav = synth.make_waveforms(2, start=40)
    ab = frontend.mel_patches_device(torch.from_numpy(wav).cuda(), 500, torch.float32)
    _, ah = dec_model.get_audio_embedding(ab["audio_patches"], ab["audio_time_inds"], ab["audio_freq_inds"], ab["audio_mask"])
    lg = dec_model.get_decoder_logits(ah, ab["audio_mask"], g["ids"], g["tmask"]).cpu().numpy()
    assert lg.shape == g["logits"].shape
    keep = g["tmask"].astype(bool)                  # rows of padded caption positions are defined but never consumed
    assert rel_l2(lg[keep], g["logits"][keep]) < LOGIT_TOL
    assert cosine_rows(lg[keep], g["logits"][keep]).min() > COS_TOL
    assert (lg[keep].argmax(-1) == g["logits"][keep].argmax(-1)).mean() > 0.9
    # growing-prefix call of the sampling loop (eval_caco_torch.py:443-456): T = 12, no padding
    lg12 = dec_model.get_decoder_logits(ah, ab["audio_mask"], g["ids"][:, :12], np.ones((2, 12), np.int64)).cpu().numpy()
    assert rel_l2(lg12[:, -1], g["logits_prefix12_last"]) < LOGIT_TOL
    # causality: the logits of position p depend only on tokens <= p
    assert rel_l2(lg12, lg[:, :12]) < 2e-3


@pytest.mark.gpu
def test_decoder_alone_ragged_masks(dec_model):
    g = load_golden("decoder_tiny.npz")
    H = dec_model.text_config.hidden_size
    rng = np.random.RandomState(5)
    th = rng.randn(2, 20, H).astype(np.float32)
    ah = rng.randn(2, 70, H).astype(np.float32)
    tm = np.ones((2, 20), dtype=np.int64); tm[1, 13:] = 0
    am = np.ones((2, 70), dtype=np.float32); am[0, 50:] = 0
    lg = dec_model.decoder_module(text_hidden_state=th, attention_mask=tm, audio_hidden_state=ah, audio_mask=am).cpu().numpy()
    keep = tm.astype(bool)
    assert rel_l2(lg[keep], g["rand_logits"][keep]) < LOGIT_TOL
    assert cosine_rows(lg[keep], g["rand_logits"][keep]).min() > COS_TOL
    # masked audio tokens must not influence the logits: scramble them
    ah2 = ah.copy(); ah2[0, 50:] = 100.0
    lg2 = dec_model.decoder_module(th, tm, ah2, am).cpu().numpy()
    np.testing.assert_array_equal(lg2[0], lg[0])


@pytest.mark.gpu
def test_decoder_error_behaviour(dec_model, tiny_state):
    from cacophony_amd.model import CACO
    a, t, cc, d = _configs()
    plain = CACO(a, t, cc, device="cuda:0").load_state_dict(tiny_state)
    assert plain.decoder_module is None
    with pytest.raises(ValueError, match="Decoder module not initialized"):        # caco.py:223-224
        plain.get_decoder_logits(torch.zeros(1, 8, 768), torch.ones(1, 8), torch.zeros(1, 4, dtype=torch.long), torch.ones(1, 4))
    with pytest.raises(ValueError):
        CACO(a, t, cc, decoder_config=replace(d, hidden_size=512, num_attention_heads=8), device="cuda:0")
    with pytest.raises(ValueError):
        dec_model.decoder_module(torch.zeros(2, 4, 768), torch.ones(2, 4), torch.zeros(3, 8, 768), torch.ones(3, 8))
    # a model built WITH the decoder refuses a state dict that lacks decoder_module.*
    with pytest.raises((ValueError, RuntimeError), match="decoder_module"):
        CACO(a, t, cc, decoder_config=d, device="cuda:0").load_state_dict(tiny_state)


@pytest.mark.gpu
def test_greedy_caption_loop_vs_oracle(dec_model, dec_state):
    """decode_caption's loop (eval_caco_torch.py:412-461) with arg-max: token for token against the oracle's fp32 loop as
    long as the oracle's own top-1 / top-2 margin exceeds what bf16 operands can flip."""
    from cacophony_amd import captioning, frontend
    a, t, cc, d = _configs()
    o = O.CacoOracle(dec_state, a, t, cc, backend="torch", decoder_cfg=d)
    wav = synth.make_waveforms(2, start=40)
    ab = frontend.mel_patches_device(torch.from_numpy(wav).cuda(), 500, torch.float32)
    ids = captioning.decode_caption_ids(dec_model, ab, max_decode_length=8, greedy=True, bos_id=0, eos_id=2, pad_id=1).cpu().numpy()
    batch = O.prepare_audio_batch(wav, 500, backend="torch")
    _, ah = o.get_audio_embedding(batch["audio_patches"], batch["audio_time_inds"], batch["audio_freq_inds"], batch["audio_mask"])
    ref, margin = o.greedy_decode(ah, batch["audio_mask"], 8)
    assert ids.shape[0] == 2 and ids.shape[1] <= 9 and (ids[:, 0] == 0).all()
    compared = 0
    for b in range(2):
        for p in range(1, min(ids.shape[1], ref.shape[1])):
            if margin[b, p - 1] < 0.02:          # a near-tie: later tokens are conditioned on different prefixes
                break
            assert ids[b, p] == ref[b, p], f"clip {b} position {p}: {ids[b]} vs {ref[b]}"
            compared += 1
    assert compared >= 4
    # sampling path runs and respects the length bound and the per-row stop
    g = torch.Generator(device="cuda").manual_seed(0)
    sm = captioning.decode_caption_ids(dec_model, ab, max_decode_length=5, temperature=0.1, generator=g)
    assert sm.shape[0] == 2 and 2 <= sm.shape[1] <= 6 and int(sm.max()) < t.vocab_size

    class Tok:                                    # the three attributes + batch_decode decode_caption uses
        bos_token_id, eos_token_id, pad_token_id = 0, 2, 1
        def batch_decode(self, x, skip_special_tokens=True):
            return [" ".join(str(int(v)) for v in row if int(v) > 2) for row in x]
    text = captioning.decode_caption(dec_model, Tok(), ab, max_decode_length=4, greedy=True)
    assert text == " ".join(str(int(v)) for v in ids[0, 1:5] if int(v) > 2)
    from cacophony_amd.model import CACO
    with pytest.raises(ValueError, match="decoder module"):
        captioning.decode_caption_ids(CACO(a, t, cc, device="cuda:0").load_state_dict(synth.make_caco_state(a, t, cc)), ab)


@pytest.mark.gpu
def test_decoder_vocab_not_multiple_of_tile():
    """The real vocabulary (50265) is not a multiple of the GEMM tile: decoder_proj runs on rows padded to x256 and the
    logits are copied out with a row stride.  Exercised here with vocab 1000 against the oracle (itself pinned to the
    reference by decoder_tiny.npz at vocab 1024)."""
    from cacophony_amd.model import CACO
    a, t, cc = C.tiny_configs(1)
    t = replace(t, vocab_size=1000)
    d = replace(t, num_hidden_layers=1)
    state = synth.make_caco_state(a, t, cc, seed=3, decoder_cfg=d)
    model = CACO(a, t, cc, decoder_config=d, device="cuda:0").load_state_dict(state)
    o = O.CacoOracle(state, a, t, cc, backend="torch", decoder_cfg=d)
    rng = np.random.RandomState(11)
    th = rng.randn(3, 9, t.hidden_size).astype(np.float32)
    ah = rng.randn(3, 40, t.hidden_size).astype(np.float32)
    tm = np.ones((3, 9), dtype=np.int64); tm[2, 5:] = 0
    am = np.ones((3, 40), dtype=np.float32); am[1, 33:] = 0
    got = model.decoder_module(th, tm, ah, am).cpu().numpy()
    ops = o.ops
    ref = ops.to_numpy(O.roberta_decoder(ops, o.P, d, ops.f32(th), ops.i64(tm), ops.f32(ah), ops.f32(am)))
    assert got.shape == ref.shape == (3, 9, 1000)
    keep = tm.astype(bool)
    assert rel_l2(got[keep], ref[keep]) < LOGIT_TOL
    assert cosine_rows(got[keep], ref[keep]).min() > COS_TOL
    assert np.isfinite(got).all()


@pytest.mark.gpu
def test_decoder_full_vocabulary():
    """RoBERTa's real vocabulary (50265 -> 50432 padded rows, 197 column tiles): logits against the oracle on a 1+1-layer
    model, and the arg-max over the 50265 real columns never lands in the padding."""
    from cacophony_amd.model import CACO
    a, t, cc = C.tiny_configs(1)
    t = replace(t, vocab_size=50265)
    d = replace(t, num_hidden_layers=1)
    state = synth.make_caco_state(a, t, cc, seed=4, decoder_cfg=d)
    model = CACO(a, t, cc, decoder_config=d, device="cuda:0").load_state_dict(state)
    o = O.CacoOracle(state, a, t, cc, backend="torch", decoder_cfg=d)
    rng = np.random.RandomState(12)
    ids, tmask = synth.make_captions(2, 8, t.vocab_size, start=70)
    ah = rng.randn(2, 24, t.hidden_size).astype(np.float32)
    am = np.ones((2, 24), dtype=np.float32)
    got = model.get_decoder_logits(ah, am, ids, tmask).cpu().numpy()
    ref = o.get_decoder_logits(ah, am, ids, tmask)
    assert got.shape == ref.shape == (2, 8, 50265)
    keep = tmask.astype(bool)
    assert rel_l2(got[keep], ref[keep]) < LOGIT_TOL
    assert cosine_rows(got[keep], ref[keep]).min() > COS_TOL


@pytest.mark.gpu
def test_cached_decode_steps_equal_full_prefix(dec_model):
    """caco_decode_step (key / value caches) against the full-prefix form position by position, and against the
    reference's own logits for the 12-token prefix (decoder_tiny.npz)."""
    from cacophony_amd import captioning, frontend
    g = load_golden("decoder_tiny.npz")
    wav = synth.make_waveforms(2, start=40)
    ab = frontend.mel_patches_device(torch.from_numpy(wav).cuda(), 500, torch.float32)
    _, ah = dec_model.get_audio_embedding(ab["audio_patches"], ab["audio_time_inds"], ab["audio_freq_inds"], ab["audio_mask"])
    ids = torch.from_numpy(g["ids"][:, :12]).cuda()
    full = dec_model.get_decoder_logits(ah, ab["audio_mask"], ids, torch.ones_like(ids)).cpu().numpy()       # [2, 12, V]
    st = captioning.CaptionDecodeState(dec_model, ah, ab["audio_mask"], max_len=16)
    steps = np.stack([st.step(ids[:, p]).cpu().numpy() for p in range(12)], axis=1)
    assert steps.shape == full.shape
    for p in range(12):
        assert rel_l2(steps[:, p], full[:, p]) < 3e-3, p          # same arithmetic, different GEMM tilings (M = 2 vs 24)
    assert rel_l2(steps[:, -1], g["logits_prefix12_last"]) < LOGIT_TOL
    assert cosine_rows(steps[:, -1], g["logits_prefix12_last"]).min() > COS_TOL
    with pytest.raises(ValueError):
        st.step(ids[:1, 0])
    for p in range(4):
        st.step(ids[:, p])
    with pytest.raises((ValueError, RuntimeError), match="max_len"):
        st.step(ids[:, 0])
    st.close()
    # the decoding loop gives the same tokens with and without the caches (greedy)
    a_ids = captioning.decode_caption_ids(dec_model, ab, max_decode_length=6, greedy=True, use_cache=True).cpu().numpy()
    b_ids = captioning.decode_caption_ids(dec_model, ab, max_decode_length=6, greedy=True, use_cache=False).cpu().numpy()
    n = min(a_ids.shape[1], b_ids.shape[1])
    assert (a_ids[:, :min(n, 4)] == b_ids[:, :min(n, 4)]).all()
