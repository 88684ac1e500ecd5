"""HEAR embedding wrapper: scene = normalised pooled embedding, event = hidden states
averaged over the 8 frequency patches per time step + millisecond timestamps
(src/eval/heareval/embeddings/audio_embedding/caco_embeddings.py:97-131)."""
import numpy as np
import pytest
import torch

from cacophony_amd import config as C
from cacophony_amd import synth
from oracle import caco_oracle as O
from tests.conftest import cosine_rows, rel_l2


def test_oracle_event_pooling_known_answers():
    # token t of clip b carries the value 100 b + t in every channel: the mean over tokens 8k..8k+7 is 100 b + 8k + 3.5
    hid = (100.0 * np.arange(2)[:, None, None] + np.arange(499)[None, :, None] + np.zeros((1, 1, 8))).astype(np.float32)
    ev, ts = O.hear_event_embeddings(hid, audio_max_len=10)
    assert ev.shape == (2, 62, 8) and ts.shape == (62,)           # 499 // 8 = 62: 'VALID' drops the ragged tail
    np.testing.assert_allclose(ev[1, :, 0], 100 + 8 * np.arange(62) + 3.5)
    assert ts[0] == 0 and ts[-1] == 10000 and np.allclose(np.diff(ts), 10000 / 61)
    ev0, ts0 = O.hear_event_embeddings(np.zeros((1, 5, 4), np.float32))
    assert ev0.shape == (1, 0, 4) and ts0.shape == (0,)            # fewer tokens than one group: empty


@pytest.mark.gpu
@pytest.mark.parametrize("shape,group", [((3, 496, 768), 8), ((2, 500, 768), 8), ((1, 37, 64), 5), ((2, 7, 16), 8), ((1, 64, 4), 1)])
def test_token_group_mean_kernel_vs_oracle(shape, group):
    from cacophony_amd import hear
    rng = np.random.RandomState(sum(shape) + group)
    hid = rng.randn(*shape).astype(np.float32)
    got = hear.token_group_mean(torch.from_numpy(hid).cuda(), group).cpu().numpy()
    ref, _ = O.hear_event_embeddings(hid, group=group)
    assert got.shape == ref.shape
    np.testing.assert_allclose(got, ref, rtol=2e-6, atol=2e-6)    # fp32 sum of <= 8 terms vs float64 mean


@pytest.mark.gpu
def test_hear_wrapper_vs_oracle(tiny_state):
    from cacophony_amd import hear
    from cacophony_amd.model import CACO
    a, t, cc = C.tiny_configs(2)
    model = CACO(a, t, cc, device="cuda:0").load_state_dict(tiny_state)
    o = O.CacoOracle(tiny_state, a, t, cc, backend="torch")
    wav = synth.make_waveforms(2, start=90)
    emb = hear.Embedding(model, audio_max_len=10)
    assert emb.max_patches == 496
    scene = emb.get_scene_embeddings(torch.from_numpy(wav).cuda()).cpu().numpy()
    ev, ts = emb.get_timestamp_embeddings(torch.from_numpy(wav).cuda())
    # oracle: same clips through the fp32 CPU restatement at 496 patches, then the restated pooling
    batch = O.prepare_audio_batch(wav, 496)
    r_emb, r_hid = o.get_audio_embedding(batch["audio_patches"], batch["audio_time_inds"], batch["audio_freq_inds"],
                                         batch["audio_mask"], return_hidden_state=True, normalize=True)
    r_emb, r_hid = np.asarray(r_emb), np.asarray(r_hid)
    r_ev, r_ts = O.hear_event_embeddings(r_hid, 10)
    assert scene.shape == (2, cc.projection_size) and ev.shape == (2, 62, a.hidden_size)
    assert cosine_rows(scene, r_emb).min() > 0.999
    assert np.allclose(np.linalg.norm(scene, axis=1), 1.0, atol=1e-3)
    assert rel_l2(ev.cpu().numpy(), r_ev) < 1e-2
    np.testing.assert_allclose(ts.numpy(), r_ts)
    one, stamps = emb.get_embedding_as_numpy(torch.from_numpy(wav[:1]).cuda(), "event")
    assert one.shape == (1, 62, a.hidden_size) and len(stamps) == 1 and stamps[0].shape == (62,)
    assert emb.get_embedding_as_numpy(torch.from_numpy(wav[:1]).cuda()).shape == (cc.projection_size,)
