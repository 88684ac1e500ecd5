#!/usr/bin/env python3
"""Random-shape sweep of the kernels on the wavesim build against torch references: ragged and tiny sizes the hand-picked
cases do not reach (M = 1, S = 1, one key, every key masked, odd strides, every tile kind, guard rows behind every
output).  CPU only; seconds per hundred cases.

    python tools/wavesim/fuzz.py [--cases 300] [--seed 0] [--kinds gemm,attention,layernorm,mel,topk,model]

Exit status 0 and a final "FUZZ CLEAN n cases" line = no mismatch, no write past an output, nothing non-finite."""
import argparse
import math
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from tests import simlib  # noqa: E402

P = simlib.ptr
GUARD = 7.0


def guard_bf16(rows, cols, extra):
    return torch.full((rows + extra, cols), GUARD, dtype=torch.bfloat16)


def fuzz_gemm(sim, rng, log):
    tile = int(rng.choice([128, 2256, 8256, 4256, 4128, 256]))
    sim.caco_set_gemm_tile(tile)
    M = int(rng.choice([1, 2, 15, 16, 17, 63, 64, 65, 127, 128, 129, 255, 256, 257, 300, 511, 513, 700, 1025]))
    N = int(rng.choice([128, 256, 384, 512, 768, 1280]))
    K = int(rng.choice([64, 128, 192, 256, 320]))
    act = int(rng.integers(0, 3))
    a = torch.randn(M, K).bfloat16()
    w = (torch.randn(N, K) / math.sqrt(K)).bfloat16()
    bias = torch.randn(N) if rng.random() < 0.8 else None
    ref = a.float() @ w.float().T + (bias if bias is not None else 0)
    out = guard_bf16(M, N, 260)
    rc = sim.caco_op_gemm_bf16(P(a), P(w), P(bias), M, N, K, act, P(out), None)
    assert rc == 0, (sim.caco_last_error(), tile, M, N, K)
    r = torch.nn.functional.silu(ref) if act == 1 else torch.nn.functional.gelu(ref) if act == 2 else ref
    err = (out[:M].float() - r).abs()
    assert (err <= 2.0 ** -7 * r.abs() + 4e-3).all(), ("gemm bf16", tile, M, N, K, act, float(err.max()))
    assert (out[M:] == GUARD).all(), ("gemm bf16 wrote past row M", tile, M, N, K)
    x = torch.full((M + 260, N), GUARD)
    x0 = torch.randn(M, N)
    x[:M] = x0
    use_res = rng.random() < 0.7
    rc = sim.caco_op_gemm_bf16_f32out(P(a), P(w), P(bias), P(x) if use_res else None, M, N, K, P(x), None)
    assert rc == 0, sim.caco_last_error()
    r = ref + (x0 if use_res else 0)
    assert (x[:M] - r).abs().max().item() < 2e-4 * max(1.0, K / 64), ("gemm f32", tile, M, N, K, use_res)
    assert (x[M:] == GUARD).all(), ("gemm f32 wrote past row M", tile, M, N, K)
    sim.caco_set_gemm_tile(256)
    log.append(f"gemm tile {tile} M {M} N {N} K {K} act {act}")


def fuzz_skew(sim, rng, log):
    """Shapes that reach the skewed-row-block fp32 + residual kernel when the library is the `skew` variant build (CACO_SIM_LIB =
    tools/wavesim/libcaco_sim_skew.so; on the default build the same cases run gemm_bf16_w8): at least 16 tiles on the simulator's
    16 CUs, K >= 576, ragged and whole M, in place or with a separate residual, guard rows on both sides of the output."""
    N = int(rng.choice([256, 512, 768, 1024]))
    K = 64 * int(rng.choice([9, 10, 12, 13, 16, 17, 24]))
    panels = int(np.ceil(16 / (N // 256))) + int(rng.integers(0, 6))
    M = panels * 256 - int(rng.choice([0, 0, 1, 17, 100, 255]))
    a = torch.randn(M, K).bfloat16()
    w = (torch.randn(N, K) / math.sqrt(K)).bfloat16()
    bias, x0 = torch.randn(N), torch.randn(M, N)
    ref = a.double() @ w.double().T + bias.double() + x0.double()
    inplace = rng.random() < 0.6
    buf = torch.full((M + 8, N), GUARD)
    buf[4:4 + M] = x0 if inplace else 0.0
    out = buf[4:4 + M]
    sim.caco_set_gemm_tile(8256)
    rc = sim.caco_op_gemm_bf16_f32out(P(a), P(w), P(bias), P(out) if inplace else P(x0), M, N, K, P(out), None)
    sim.caco_set_gemm_tile(256)
    assert rc == 0, (sim.caco_last_error(), M, N, K)
    err = ((out.double() - ref).abs() / (ref.abs() + 1.0)).max().item()
    assert err < 4e-6 * max(1.0, K / 1024), ("skew gemm f32", M, N, K, inplace, err)
    assert (buf[:4] == GUARD).all() and (buf[4 + M:] == GUARD).all(), ("skew gemm f32 wrote outside its rows", M, N, K)
    log.append(f"skew M {M} N {N} K {K} inplace {inplace} err {err:.2e}")


def attn_ref(q, k, v, mask, heads, hd, causal):
    B, Sq, H = q.shape
    S = k.shape[1]
    qf = q.float().reshape(B, Sq, heads, hd).transpose(1, 2)
    kf = k.float().reshape(B, S, heads, hd).transpose(1, 2)
    vf = v.float().reshape(B, S, heads, hd).transpose(1, 2)
    s = qf @ kf.transpose(-1, -2) / math.sqrt(hd)
    allow = torch.ones(B, 1, Sq, S, dtype=torch.bool)
    if mask is not None:
        allow = allow & (mask != 0)[:, None, None, :]
    if causal:
        allow = allow & torch.tril(torch.ones(S, S, dtype=torch.bool))[None, None]
    s = s.masked_fill(~allow, float("-inf"))
    return (torch.softmax(s, -1) @ vf).transpose(1, 2).reshape(B, Sq, H)


def fuzz_attention(sim, rng, log):
    hd = int(rng.choice([64, 96]))
    heads = int(rng.integers(1, 4))
    B = int(rng.integers(1, 4))
    S = int(rng.choice([1, 2, 7, 31, 32, 33, 63, 64, 65, 100, 127, 128, 129, 191, 200, 257, 300]))
    causal = int(rng.random() < 0.4)
    Sq = S if (causal or rng.random() < 0.6) else int(rng.choice([1, 5, 32, 33, 64, 130]))
    os.environ["CACO_ATTN_SMALL"] = "1" if rng.random() < 0.5 else "0"
    simlib.sync_switches()
    H = heads * hd
    ld = 3 * H + (int(rng.choice([0, 8, 64])))              # padded row pitch
    qkv = torch.zeros(B, S, ld, dtype=torch.bfloat16)
    qkv[..., :3 * H] = (torch.randn(B, S, 3 * H) * 1.3).bfloat16()
    q = (torch.randn(B, Sq, H) * 1.3).bfloat16() if Sq != S else qkv[:, :, :H].contiguous()
    mask = torch.ones(B, S)
    for b in range(B):
        if rng.random() < 0.6:
            mask[b, int(rng.integers(0, S + 1)):] = 0        # may mask every key
    use_mask = rng.random() < 0.85
    out = torch.full((B * Sq + 40, H), GUARD, dtype=torch.bfloat16)
    rc = sim.caco_op_attention_qkv(P(q), H, Sq, P(qkv), ld, H, 2 * H, P(mask) if use_mask else None, B, S, heads, hd, causal, P(out), None)
    assert rc == 0, (sim.caco_last_error(), B, Sq, S, heads, hd, causal)
    ref = attn_ref(q, qkv[..., H:2 * H], qkv[..., 2 * H:3 * H], mask if use_mask else None, heads, hd, bool(causal))
    got = out[:B * Sq].float().reshape(B, Sq, H)
    live = torch.isfinite(ref).all(-1)
    assert torch.isfinite(got).all(), ("attention non-finite", B, Sq, S, heads, hd, causal)
    if live.any():
        assert (got[live] - ref[live]).abs().max().item() < 0.04, ("attention", B, Sq, S, heads, hd, causal, os.environ["CACO_ATTN_SMALL"])
    assert (got[~live] == 0).all(), ("attention fully masked rows", B, Sq, S, heads, hd)
    assert (out[B * Sq:] == GUARD).all(), ("attention wrote past its rows", B, Sq, S, heads, hd)
    log.append(f"attention B {B} Sq {Sq} S {S} heads {heads} hd {hd} causal {causal} small {os.environ['CACO_ATTN_SMALL']}")
    os.environ["CACO_ATTN_SMALL"] = "0"
    simlib.sync_switches()


def fuzz_layernorm(sim, rng, log):
    rows = int(rng.choice([1, 3, 4, 5, 63, 257, 1000]))
    dim = int(rng.choice([128, 256, 512, 768, 1024]))
    x = torch.randn(rows, dim) * float(rng.choice([0.1, 1.0, 30.0])) + float(rng.choice([0.0, 5.0]))
    g, b = torch.randn(dim), torch.randn(dim)
    of = torch.full((rows + 8, dim), GUARD)
    ob = torch.full((rows + 8, dim), GUARD, dtype=torch.bfloat16)
    assert sim.caco_op_layernorm(P(x), P(g), P(b), rows, dim, 1e-5, P(of), P(ob), None) == 0, sim.caco_last_error()
    ref = torch.nn.functional.layer_norm(x, (dim,), g, b, 1e-5)
    assert (of[:rows] - ref).abs().max().item() < 1e-4 and (of[rows:] == GUARD).all() and (ob[rows:] == GUARD).all(), ("layernorm", rows, dim)
    log.append(f"layernorm rows {rows} dim {dim}")


def fuzz_mel(sim, rng, log):
    from cacophony_amd import synth
    from oracle import caco_oracle as O
    B = int(rng.integers(1, 4))
    n = int(rng.choice([513, 700, 2560, 2561, 5119, 12345, 16000, 31999]))
    max_p = int(rng.choice([8, 16, 24, 100]))
    lens = [int(rng.integers(1, n + 1)) for _ in range(B)] if rng.random() < 0.5 else None
    wav = np.stack([synth.make_waveform(int(rng.integers(0, 1000)), n_samples=n) for _ in range(B)]).astype(np.float32)
    wt = torch.from_numpy(wav)
    patches = torch.full((B * max_p + 8, 256), GUARD)
    ti, fi, mk = torch.empty(B, max_p), torch.empty(B, max_p), torch.empty(B, max_p)
    ln = None if lens is None else torch.tensor(lens, dtype=torch.int64)
    rc = sim.caco_mel_patches_lens(P(wt), P(ln), B, n, max_p, 0.2, 0.9, P(patches), 0, P(ti), P(fi), P(mk), None)
    assert rc == 0, (sim.caco_last_error(), B, n, max_p, lens)
    assert (patches[B * max_p:] == GUARD).all(), ("mel wrote past its rows", B, n, max_p)
    got = patches[:B * max_p].reshape(B, max_p, 256).numpy()
    for b in range(B):
        L = n if lens is None else lens[b]
        ref = O.prepare_audio_batch(wav[b:b + 1, :L], max_p)
        np.testing.assert_array_equal(mk[b].numpy(), ref["audio_mask"][0])
        np.testing.assert_array_equal(ti[b].numpy(), ref["audio_time_inds"][0])
        np.testing.assert_array_equal(fi[b].numpy(), ref["audio_freq_inds"][0])
        nv = int(ref["audio_mask"].sum())
        if nv:
            assert np.abs(got[b, :nv] - ref["audio_patches"][0, :nv]).max() < 2e-3, ("mel", B, n, max_p, lens, b)
        assert (got[b, nv:] == 0).all(), ("mel padded rows", B, n, max_p, lens, b)
    log.append(f"mel B {B} n {n} max_p {max_p} lens {lens}")


def fuzz_topk(sim, rng, log):
    rows, cols = int(rng.integers(1, 40)), int(rng.choice([1, 5, 10, 63, 64, 65, 300]))
    k = int(rng.integers(1, 17))
    sim_m = torch.round(torch.randn(rows, cols) * 3) / 3          # ties on purpose
    idx = torch.full((rows, k), -7, dtype=torch.int32)
    val = torch.empty(rows, k)
    assert sim.caco_topk(P(sim_m), rows, cols, cols, 1, k, P(idx), P(val), None) == 0, sim.caco_last_error()
    rv, ri = torch.sort(sim_m, dim=1, descending=True, stable=True)
    kk = min(k, cols)
    np.testing.assert_array_equal(idx[:, :kk].numpy(), ri[:, :kk].numpy())
    assert (idx[:, kk:] == -1).all()
    log.append(f"topk rows {rows} cols {cols} k {k}")


_MODELS = {}


def _narrow_models():
    """hidden 256 (4 heads of 64), one and two layers: the whole towers at a ninth of the full-width cost."""
    if _MODELS:
        return _MODELS
    from dataclasses import replace
    from cacophony_amd import config as C, synth
    from oracle import caco_oracle as O
    for layers, pool in ((1, 2), (2, 4)):
        a = replace(C.default_audio_config(), num_layers=layers, hidden_size=256, num_heads=4, intermediate_size=512)
        t = replace(C.default_text_config(), num_hidden_layers=layers, hidden_size=256, num_attention_heads=4, intermediate_size=512,
                    vocab_size=300, max_position_embeddings=80)
        cc = replace(C.default_caco_config(), projection_size=256, num_attention_pool_heads=pool)
        state = synth.make_caco_state(a, t, cc, seed=layers)
        _MODELS[layers] = (simlib.SimModel(a, t, cc).load_state_dict(state), O.CacoOracle(state, a, t, cc, backend="torch"), t)
    return _MODELS


def fuzz_model(sim, rng, log):
    """Whole towers through the C ABI (mel -> audio tower -> pooler -> normalise; text tower; similarity) against the oracle,
    on random batch sizes, clip lengths (with and without per-clip lengths), caption lengths and kernel switches."""
    from cacophony_amd import synth
    from oracle import caco_oracle as O
    from tests.conftest import cosine_rows
    m, o, t = _narrow_models()[int(rng.choice([1, 2]))]
    os.environ["CACO_ATTN_SMALL"] = "1" if rng.random() < 0.5 else "0"
    os.environ["CACO_POS_FUSE"] = "1" if rng.random() < 0.5 else "0"
    os.environ["CACO_POOL_FUSE"] = "1" if rng.random() < 0.5 else "0"
    os.environ["CACO_PINGPONG"] = "1" if rng.random() < 0.5 else "0"
    simlib.sync_switches()
    tile = int(rng.choice([256, 8256, 128]))
    sim.caco_set_gemm_tile(tile)
    m.set_ln_fold(int(rng.random() < 0.3))
    B = int(rng.integers(1, 5))
    n = int(rng.choice([2560, 4000, 12345, 20480, 33000]))
    lens = [int(rng.integers(2560, n + 1)) for _ in range(B)] if rng.random() < 0.5 else None
    wav = np.stack([synth.make_waveform(int(rng.integers(0, 1000)), n_samples=n) for _ in range(B)]).astype(np.float32)
    if lens is not None:
        for b in range(B):
            wav[b, lens[b]:] = 0
    T = int(rng.choice([1, 2, 9, 31, 32, 33, 40, 64, 70]))
    ids, mask = synth.make_captions(B, T, t.vocab_size, start=int(rng.integers(0, 1000)))
    ea = m.encode_audio(wav, lengths=lens).numpy()
    et = m.encode_text(ids, mask).numpy()
    maxp = max(8, n * 8 // 160 // 16)
    if lens is None:
        ra = o.encode_audio(wav, maxp)
    else:
        ra = np.concatenate([o.encode_audio(wav[b:b + 1, :lens[b]], maxp) for b in range(B)], 0)
    rt = o.encode_text(ids, mask)
    assert np.isfinite(ea).all() and np.isfinite(et).all()
    ca, ct = cosine_rows(ea, ra).min(), cosine_rows(et, rt).min()
    desc = (f"model B {B} n {n} lens {lens} T {T} tile {tile} small {os.environ['CACO_ATTN_SMALL']} fuse {os.environ['CACO_POS_FUSE']} "
            f"pool {os.environ['CACO_POOL_FUSE']} pp {os.environ['CACO_PINGPONG']}")
    assert ca > 0.999 and ct > 0.999, (desc, ca, ct)
    sim.caco_set_gemm_tile(256)
    m.set_ln_fold(0)
    os.environ["CACO_ATTN_SMALL"] = os.environ["CACO_POS_FUSE"] = os.environ["CACO_POOL_FUSE"] = os.environ["CACO_PINGPONG"] = "0"
    simlib.sync_switches()
    log.append(desc + f" cos {ca:.5f} {ct:.5f}")


KINDS = {"model": fuzz_model, "gemm": fuzz_gemm, "skew": fuzz_skew, "attention": fuzz_attention, "layernorm": fuzz_layernorm, "mel": fuzz_mel, "topk": fuzz_topk}


def run(cases, seed, kinds, verbose=False):
    sim = simlib.load()
    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    log = []
    names = [k for k in kinds if k in KINDS]
    for i in range(cases):
        KINDS[names[i % len(names)]](sim, rng, log)
        if verbose:
            print(log[-1], flush=True)
    return log


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=300)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--kinds", default="gemm,attention,layernorm,mel,topk,model")
    ap.add_argument("-v", action="store_true")
    a = ap.parse_args()
    log = run(a.cases, a.seed, a.kinds.split(","), a.v)
    print(f"FUZZ CLEAN {len(log)} cases (seed {a.seed})")
