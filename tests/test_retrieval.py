"""Retrieval scoring: top-10 selection on the device, recall@k / mAP@10 bookkeeping on
the host, both against the reference's compute_retrieval_metric (golden produced by tests/golden/make_golden.py)."""
import numpy as np
import pytest
import torch

from cacophony_amd import retrieval, synth
from oracle import caco_oracle as O
from tests.conftest import load_golden

NAMES = ["R1", "R5", "R10", "mAP10"]


@pytest.fixture(scope="module")
def scenario():
    return synth.make_retrieval_scenario()


def test_oracle_ranking_and_hits_match_reference(scenario):
    g = load_golden("retrieval.npz")
    all_audio, all_text, gt_at, gt_ta, A, T = scenario
    np.testing.assert_allclose(T @ A.T, g["logits_ar"], atol=1e-6)
    at = O.argsort_desc(g["logits_ar"].T, 10)
    ta = O.argsort_desc(g["logits_ar"], 10)
    np.testing.assert_array_equal(at, g["at_top10"])
    np.testing.assert_array_equal(ta, g["ta_top10"])
    h_at = O.retrieval_hits(at, all_audio, all_text, gt_at, "at")
    h_ta = O.retrieval_hits(ta, all_text, all_audio, gt_ta, "ta")
    for n in NAMES:
        np.testing.assert_array_equal(np.asarray(h_at[n]), g[f"at_{n}"])
        np.testing.assert_array_equal(np.asarray(h_ta[n]), g[f"ta_{n}"])
    assert 0.2 < np.mean(g["ta_R1"]) < 0.95 and np.mean(g["at_R10"]) > np.mean(g["at_R1"])     # informative, not saturated


def test_host_metric_mirror_matches_reference(scenario):
    g = load_golden("retrieval.npz")
    all_audio, all_text, gt_at, gt_ta, _, _ = scenario
    m_at = retrieval.compute_retrieval_metric(g["at_top10"], all_audio, all_text, gt_at)
    m_ta = retrieval.compute_retrieval_metric(g["ta_top10"], all_text, all_audio, gt_ta, "ta")
    for n in NAMES:
        np.testing.assert_array_equal(np.asarray(m_at[n]), g[f"at_{n}"])
        np.testing.assert_array_equal(np.asarray(m_ta[n]), g[f"ta_{n}"])
    with pytest.raises(ValueError):
        retrieval.compute_retrieval_metric(g["at_top10"], all_audio, all_text, gt_at, "xx")


def test_jackknife_closed_form_equals_leave_one_out():
    rng = np.random.RandomState(0)
    for data in (rng.rand(37), (rng.rand(120) > 0.4).astype(float), np.array([0.0, 1.0, 1.0])):
        a = retrieval.jackknife_stats(data)
        b = O.jackknife_mean(data)
        np.testing.assert_allclose(a[0], b[0], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(a[2], b[2], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(a[3], b[3], rtol=1e-9, atol=1e-12)
        # for the mean the jackknife is unbiased and its standard error is the classical s / sqrt(n)
        assert abs(a[1]) < 1e-9
        np.testing.assert_allclose(a[2], np.std(data, ddof=1) / np.sqrt(len(data)), rtol=1e-9)


# ---------------------------------------------------------------------------------------------- device part
gpu = pytest.mark.gpu


@gpu
def test_topk_matches_oracle_both_directions():
    g = load_golden("retrieval.npz")
    sim = torch.from_numpy(g["logits_ar"]).cuda()
    ta, tav = retrieval.topk(sim, 10, dim=1)
    at, atv = retrieval.topk(sim, 10, dim=0)
    np.testing.assert_array_equal(ta.cpu().numpy(), g["ta_top10"])
    np.testing.assert_array_equal(at.cpu().numpy(), g["at_top10"])
    np.testing.assert_array_equal(tav.cpu().numpy(), np.take_along_axis(g["logits_ar"], g["ta_top10"].astype(np.int64), 1))
    np.testing.assert_array_equal(atv.cpu().numpy(), np.take_along_axis(g["logits_ar"].T, g["at_top10"].astype(np.int64), 1))


@gpu
@pytest.mark.parametrize("rows,cols,k", [(2048, 2048, 10), (5, 7, 10), (300, 1000, 16), (1, 64, 1), (33, 65, 64)])
def test_topk_ties_padding_strides(rows, cols, k):
    rng = np.random.RandomState(rows * 7 + cols)
    x = np.round(rng.randn(rows, cols) * 3).astype(np.float32) / 4      # heavy ties
    big = torch.full((rows, 2 * cols + 3), float("nan"), device="cuda")
    big[:, 1:2 * cols:2] = torch.from_numpy(x).cuda()
    view = big[:, 1:2 * cols:2]                                           # column stride 2, NaN in between
    idx, val = retrieval.topk(view, k, dim=1)
    ref = O.argsort_desc(x, k)
    kk = min(k, cols)
    np.testing.assert_array_equal(idx.cpu().numpy()[:, :kk], ref[:, :kk])
    if k > cols:
        assert (idx.cpu().numpy()[:, cols:] == -1).all()
    idx0, _ = retrieval.topk(view, min(k, rows), dim=0)
    np.testing.assert_array_equal(idx0.cpu().numpy(), O.argsort_desc(x.T, min(k, rows)))
    xn = x.copy()
    xn[0, :] = np.nan                                                    # a NaN row selects nothing
    idxn, _ = retrieval.topk(torch.from_numpy(xn).cuda(), 3, dim=1)
    assert (idxn[0].cpu().numpy() == -1).all() and (idxn[1:].cpu().numpy() == O.argsort_desc(x[1:], 3)).all()


@gpu
def test_audio_retrieval_scores_end_to_end(scenario):
    g = load_golden("retrieval.npz")
    all_audio, all_text, gt_at, gt_ta, A, T = scenario
    logits, at_idx, ta_idx = retrieval.audio_retrieval_scores(torch.from_numpy(A).cuda(), torch.from_numpy(T).cuda())
    assert np.abs(logits.cpu().numpy() - g["logits_ar"]).max() < 1e-6
    m_at = retrieval.compute_retrieval_metric(at_idx, all_audio, all_text, gt_at)
    m_ta = retrieval.compute_retrieval_metric(ta_idx, all_text, all_audio, gt_ta, "ta")
    # fp32 MFMA vs CPU matmul may swap exact near-ties; the metrics of this scenario are robust to that
    for n in NAMES:
        assert abs(np.mean(m_at[n]) - np.mean(g[f"at_{n}"])) < 0.03
        assert abs(np.mean(m_ta[n]) - np.mean(g[f"ta_{n}"])) < 0.03
    with pytest.raises(ValueError):
        retrieval.topk(logits.double(), 10)


def test_oracle_zero_shot_accuracy_known_answers():
    # 4 classes on the axes, clips near their class axis except clip 2, which sits nearer class 3 (second choice: class 2)
    T = np.eye(4, dtype=np.float32)
    A = np.array([[1, .1, 0, 0], [.1, 1, 0, 0], [0, 0, .6, .8], [0, 0, .1, 1]], dtype=np.float32)
    acc = O.zs_topk_accuracy(A, T, [0, 1, 2, 3], logit_scale=2.0, ks=(1, 2))
    assert acc == {"1": 0.75, "2": 1.0}


@pytest.mark.gpu
def test_zero_shot_scores_device_vs_oracle():
    rng = np.random.RandomState(3)
    T = rng.randn(37, 768).astype(np.float32); T /= np.linalg.norm(T, axis=1, keepdims=True)
    tgt = rng.randint(0, 37, size=300)
    A = (0.07 * T[tgt] + rng.randn(300, 768).astype(np.float32) / np.sqrt(768)).astype(np.float32)
    A /= np.linalg.norm(A, axis=1, keepdims=True)
    ref = O.zs_topk_accuracy(A, T, tgt, logit_scale=2.6592, ks=(1, 5))
    got = retrieval.zs_classification_scores(torch.from_numpy(A).cuda(), torch.from_numpy(T).cuda(), tgt, 2.6592, ks=(1, 5))
    assert 0.2 < ref["1"] < 0.98 and ref["5"] >= ref["1"]          # informative
    assert abs(got["1"] - ref["1"]) <= 1 / 300 + 1e-9 and abs(got["5"] - ref["5"]) <= 1 / 300 + 1e-9   # fp32 MFMA vs BLAS: at most one near-tie
    with pytest.raises(ValueError):
        retrieval.zs_classification_scores(torch.from_numpy(A).cuda(), torch.from_numpy(T).cuda(), tgt[:5])
