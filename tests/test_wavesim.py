"""The kernel SOURCES of cacophony_amd/csrc executed on the CPU by the wavesim functional model (tools/wavesim), through
the same C ABI and against the same checkers as tests/test_gpu_ops.py / test_gpu_model.py, at CPU-friendly sizes.

What this is: a check of the kernels' index arithmetic - MFMA fragment layouts, LDS swizzles, LDS-DMA addressing, counted
s_waitcnt vmcnt (LDS-DMA data lands only at the covering wait), transpose reads, buffer-descriptor range checks, masks,
tile-boundary handling - that needs no GPU.  What it is NOT: hardware evidence (the `-m gpu` suite is), a timing, or a
path the product can take (cacophony_amd never loads the simulator library).
The model is pinned the other way round as well: the kernels it executes here passed the GPU suite on MI355X in rounds
1 and 2, so a wrong instruction model would show up as a failure of a known-good kernel.
"""
import math
import os

import numpy as np
import pytest
import torch

from tests import simlib
from tests.conftest import cosine_rows, load_golden, rel_l2

pytestmark = pytest.mark.skipif(not simlib.available(), reason="no host clang++ for the wavesim build")

from cacophony_amd import config as C  # noqa: E402
from cacophony_amd import synth  # noqa: E402
from oracle import caco_oracle as O  # noqa: E402

P = simlib.ptr


@pytest.fixture(scope="module")
def sim():
    return simlib.load()


def _rand(shape, seed, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


# ------------------------------------------------------------------------------------------------ GEMM family
@pytest.mark.parametrize("tile", [128, 2256, 8256, 4256, 4128])
@pytest.mark.parametrize("M,N,K,act", [(300, 256, 128, 0), (520, 768, 256, 1), (257, 512, 192, 2), (64, 256, 64, 0)])
def test_gemm_bf16(sim, tile, M, N, K, act):
    assert sim.caco_set_gemm_tile(tile) == tile
    a = _rand((M, K), 1).bfloat16()
    w = _rand((N, K), 2, 1.0 / math.sqrt(K)).bfloat16()
    bias = _rand((N,), 3)
    out = torch.full((M, N), float("nan"), dtype=torch.bfloat16)
    simlib.check(sim.caco_op_gemm_bf16(P(a), P(w), P(bias), M, N, K, act, P(out), None))
    ref = a.float() @ w.float().T + bias
    ref = torch.nn.functional.silu(ref) if act == 1 else torch.nn.functional.gelu(ref) if act == 2 else ref
    err = (out.float() - ref).abs()
    assert torch.isfinite(out.float()).all()
    assert (err <= 2.0 ** -8 * ref.abs() + 2e-3).all(), f"max err {err.max().item():.4g}"
    sim.caco_set_gemm_tile(256)


@pytest.mark.parametrize("tile", [128, 2256, 8256, 4256, 4128])
def test_gemm_f32_residual_in_place_and_plain(sim, tile):
    sim.caco_set_gemm_tile(tile)
    M, N, K = 301, 768, 192
    a = _rand((M, K), 4).bfloat16()
    w = _rand((N, K), 5, 1.0 / math.sqrt(K)).bfloat16()
    bias = _rand((N,), 6)
    x = _rand((M, N), 7)
    ref = a.float() @ w.float().T + bias + x
    simlib.check(sim.caco_op_gemm_bf16_f32out(P(a), P(w), P(bias), P(x), M, N, K, P(x), None))
    assert (x - ref).abs().max().item() < 1e-4
    out = torch.empty(M, N)
    simlib.check(sim.caco_op_gemm_bf16_f32out(P(a), P(w), None, None, M, N, K, P(out), None))
    assert (out - a.float() @ w.float().T).abs().max().item() < 1e-4
    sim.caco_set_gemm_tile(256)


def _variant_sim(tag, defines):
    """A variant build of the simulator library (its own object files and .so, tools/wavesim/libcaco_sim_<tag>.so), bound like
    simlib.load() but NOT cached as the suite's library."""
    import ctypes as Ct
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "wavesim"))
    import build_sim
    from cacophony_amd import _lib
    # CACO_SIM_SKEW_LIB: a sanitizer build of the `skew` variant (tools/wavesim/tsan_check.py --skew --pytest ...)
    path = os.environ.get("CACO_SIM_SKEW_LIB") if tag == "skew" else None
    lib = Ct.CDLL(path or build_sim.build(defines=tuple(defines), tag=tag, verbose=False))
    for name, (res, args) in _lib._SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    return lib


@pytest.fixture(scope="module")
def skew_sim():
    return _variant_sim("skew", ["-DW8_F32_SKEW"])


@pytest.fixture(scope="module")
def noskew_sim(sim):
    """The other arm of every skew comparison: the build WITHOUT -DW8_F32_SKEW.  That is the suite's own library as long as the
    product build does not define it (cacophony_amd/build.py EXTRA_FLAGS); after a flip it is an explicit -U build, so that the
    cases below keep comparing the skewed kernel with gemm_bf16_w8 instead of with itself."""
    from cacophony_amd.build import EXTRA_FLAGS
    if "-DW8_F32_SKEW" not in EXTRA_FLAGS.get("gemm_w8.hip", []):
        return sim
    return _variant_sim("noskew", ["-UW8_F32_SKEW"])


def _skew_case(lib, M, N, K, seed, inplace=True, guard=3):
    a = _rand((M, K), seed).bfloat16()
    w = _rand((N, K), seed + 1, 1.0 / math.sqrt(K)).bfloat16()
    bias, x = _rand((N,), seed + 2), _rand((M, N), seed + 3)
    ref = a.float() @ w.float().T + bias + x                       # torch fp32, the bar of the GPU op tests
    ref64 = a.double() @ w.double().T + bias.double() + x.double()
    buf = torch.full((M + 2 * guard, N), 777.0)
    buf[guard:guard + M] = x if inplace else 0.0
    out = buf[guard:guard + M]
    lib.caco_set_gemm_tile(8256)
    try:
        rc = lib.caco_op_gemm_bf16_f32out(P(a), P(w), P(bias), P(out) if inplace else P(x), M, N, K, P(out), None)
        assert rc == 0, lib.caco_last_error()
    finally:
        lib.caco_set_gemm_tile(256)
    assert bool((buf[:guard] == 777.0).all()) and bool((buf[guard + M:] == 777.0).all()), "rows outside [0, M) were written"
    return out, ref, ref64


@pytest.mark.parametrize("M,N,K,inplace", [
    (2048, 768, 768, True),        # out-proj shape at 8 panels: D = 1, 5 teams of 3 workgroups on the simulator's 16 CUs
    (2000, 768, 768, True),        # ragged last panel: rows >= M read as zeros and are dropped by the descriptors
    (1800, 768, 640, False),       # nk = 10, separate residual; teams with 1 and 2 panels
    (4100, 768, 1024, True),       # nk = 16: D = 2; 17 panels over 5 teams (periods 3 and 4), ragged
    (2304, 512, 3072, True),       # fc2 K: nk = 48, D = 6; two n-tiles: 8 teams, 9 panels (one team walks two)
    (2304, 1024, 576, True),       # the smallest K the kernel takes (nk = 9); four n-tiles: 4 teams
])
def test_gemm_skewed_row_blocks(skew_sim, noskew_sim, M, N, K, inplace):
    """Variant build `skew` (-DW8_F32_SKEW, csrc/gemm_w8_skew.inc): the fp32 + bias + residual GEMM whose epilogue runs under its
    own K-loop.  Against torch fp32 at fp32 rounding (sum order differs from the other kernels: bias first, K-tiles rotated per
    row block), with guard rows around the output, in place and with a separate residual; and the DEFAULT build on the same
    case must differ from it in bits (otherwise the launch fell back to gemm_bf16_w8 and the case tests nothing)."""
    out, ref, ref64 = _skew_case(skew_sim, M, N, K, 40, inplace=inplace)
    err = ((out.double() - ref64).abs() / (ref64.abs() + 1.0)).max().item()
    err32 = ((ref.double() - ref64).abs() / (ref64.abs() + 1.0)).max().item()
    # measured against float64, as a fraction of 1 + |ref|: 1.8e-6 .. 2.4e-6 for K <= 1024 and 4.2e-6 at K = 3072; gemm_bf16_w8 on
    # the same cases 1.8e-6 resp. 4.8e-6 (fp32 accumulation of K products either way; the K-tiles are summed in a rotated order
    # here), torch's own fp32 product 1.1e-6
    assert err <= (3e-6 if K <= 1024 else 7e-6), f"max error {err:.3e} relative to 1 + |ref| (torch fp32 itself: {err32:.3e})"
    d32 = ((out - ref).abs() / (ref.abs() + 1.0)).max().item()      # two fp32 roundings apart: up to the sum of both errors
    print(f"[skew] M {M} N {N} K {K}: vs float64 {err:.2e} (torch fp32 {err32:.2e}), vs torch fp32 {d32:.2e}")
    assert d32 <= (3.5e-6 if K <= 1024 else 7e-6)
    base, _, _ = _skew_case(noskew_sim, M, N, K, 40, inplace=inplace)
    assert not torch.equal(base, out), "identical bits: the skewed kernel did not run"
    assert (base - out).abs().max().item() < 1e-4


def test_gemm_skewed_falls_back_where_it_does_not_apply(skew_sim, noskew_sim):
    """Launches that do not fill the chip, K < 576 and gathered / absent residuals stay on gemm_bf16_w8: bitwise the default's."""
    for M, N, K in ((1024, 768, 768), (2048, 768, 512), (2048, 256, 768)):
        a, _, _ = _skew_case(skew_sim, M, N, K, 50)
        b, _, _ = _skew_case(noskew_sim, M, N, K, 50)
        assert torch.equal(a, b), (M, N, K)


def test_gemm_skewed_linear_panel_list(noskew_sim):
    """The linear form of the same kernel (-DW8_SKEW_LINEAR, variant `skew_lin`: first-period blocks idle, a tail period at the
    end) - the other arm of the A/B against the circular panel list of `skew`."""
    lin = _variant_sim("skew_lin", ["-DW8_F32_SKEW", "-DW8_SKEW_LINEAR"])
    for M, N, K, inplace in ((2000, 768, 768, True), (2304, 512, 3072, True), (1800, 768, 640, False)):
        out, ref, ref64 = _skew_case(lin, M, N, K, 40, inplace=inplace)
        err = ((out.double() - ref64).abs() / (ref64.abs() + 1.0)).max().item()
        assert err <= (3e-6 if K <= 1024 else 7e-6), (M, N, K, err)
        base, _, _ = _skew_case(noskew_sim, M, N, K, 40, inplace=inplace)
        assert not torch.equal(base, out)


def test_gemm_skewed_other_team_geometries(skew_sim):
    """The team mapping at other workgroup counts (the CU count is read once per process: a subprocess with WAVESIM_CUS = 8 and 24):
    2 and 8 teams of 3 workgroups with idle workgroups left over, a single-column shape (teams of one), teams with unequal panel counts."""
    import subprocess
    import sys
    code = (
        "import sys, math, torch; sys.path.insert(0, %r)\n"
        "from tests.test_wavesim import _variant_sim, _skew_case\n"
        "from tests import simlib\n"
        "lib = _variant_sim('skew', ['-DW8_F32_SKEW'])\n"
        "for M, N, K in ((2048, 768, 768), (2100, 768, 640), (3000, 256, 1024)):\n"
        "    out, ref, ref64 = _skew_case(lib, M, N, K, 60)\n"
        "    err = ((out.double() - ref64).abs() / (ref64.abs() + 1.0)).max().item()\n"
        "    assert err <= 3e-6, (M, N, K, err)\n"
        "print('OK')\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    for cus in ("8", "24"):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, WAVESIM_CUS=cus), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0 and r.stdout.strip().endswith("OK"), (cus, r.stdout[-500:], r.stderr[-1500:])


@pytest.mark.parametrize("ln_fold", [0, 1])
def test_audio_layer_with_skewed_gemms_matches_the_default_build(skew_sim, noskew_sim, tiny_state, ln_fold):
    """One full-width audio layer at batch 8 with the persistent GEMM forced on (CACO_W8_MIN_TILES = 1): out-proj (K = 768, D = 1)
    and fc2 (K = 3072, D = 6) run on the skewed kernel inside the model's launch sequence - in-place residual stream, the real
    strides.  The embeddings must equal the default build's to fp32 rounding, and differ from them in bits (else it fell back).
    ln_fold = 1: the LayerNorm-folded stack - out-proj and fc2 are the fold PRODUCERS (bf16 copy of the new rows + per-row partial
    statistics), which the skewed kernel emits from its events (and its final burst) instead of gemm_bf16_w8's MODE 4 epilogue."""
    from dataclasses import replace
    a, t, cc = C.tiny_configs(2)
    a = replace(a, num_layers=1)
    state = {k: v for k, v in tiny_state.items() if ".layers.1." not in k and "layers_1" not in k}
    wav = torch.from_numpy(synth.make_waveforms(8, start=40))
    embs = []
    for lib in (noskew_sim, skew_sim):
        prev = lib.caco_get_switch(b"CACO_W8_MIN_TILES")
        assert lib.caco_set_switch(b"CACO_W8_MIN_TILES", 1) == 0
        try:
            m = simlib.SimModel(a, None, cc, lib=lib).load_state_dict({k: v for k, v in state.items() if k.startswith(("audio_", "logit_scale"))})
            assert m.set_ln_fold(ln_fold) == ln_fold
            embs.append(m.encode_audio(wav).numpy())
        finally:
            lib.caco_set_switch(b"CACO_W8_MIN_TILES", prev)
    base, skew = embs
    assert not np.array_equal(base, skew), "identical bits: the skewed kernel did not run inside the model"
    # folded form: the two builds round the statistics' partial sums in different orders
    tol = 1e-5 if ln_fold == 0 else 2e-5          # measured 6.5e-6 / 8.0e-6
    assert np.abs(base - skew).max() < tol * np.abs(base).max() + 1e-6, np.abs(base - skew).max() / np.abs(base).max()
    assert cosine_rows(base, skew).min() > (0.999999 if ln_fold == 0 else 0.99999)


def test_gemm_skewed_weak_wait_is_caught():
    """The skewed kernel waits for its residual registers itself (inline-assembly loads the compiler's wait-count pass does not
    see).  On the simulator such a load lands in its destination variable at the covering wait; a build whose wait leaves ONE
    more operation in flight (13 instead of 12) must produce stale sums - i.e. the model does check that count."""
    weak = _variant_sim("skew_weakwait", ["-DW8_F32_SKEW", "-DW8S_SIM_RD_WAIT=13"])
    out, ref, _ = _skew_case(weak, 2048, 768, 768, 40)
    assert (out - ref).abs().max().item() > 1e-2


def test_gemm_every_tile_kernel_gives_the_same_bits(sim):
    """What makes a clip's embedding independent of its batch mates (tests/test_gpu_model.py::test_odd_batch_sizes_...):
    the kernel families a batch size selects - 128 x 128, x, w8, w4q, w4h - accumulate K in the same order and round the
    same way in their epilogues (the w8 family writes SiLU as exp2(x * -log2 e) by hand, the others call __expf)."""
    M, N, K = 600, 768, 768
    a = _rand((M, K), 11).bfloat16()
    w = _rand((N, K), 12, 1.0 / math.sqrt(K)).bfloat16()
    bias, x = _rand((N,), 13), _rand((M, N), 14)
    tiles = (128, 2256, 8256, 4256, 4128)
    try:
        for act in (0, 1, 2):
            outs = []
            for tile in tiles:
                sim.caco_set_gemm_tile(tile)
                o = torch.zeros(M, N, dtype=torch.bfloat16)
                simlib.check(sim.caco_op_gemm_bf16(P(a), P(w), P(bias), M, N, K, act, P(o), None))
                outs.append(o.view(torch.int16))
            for tile, o in zip(tiles[1:], outs[1:]):
                assert torch.equal(o, outs[0]), f"act {act}: tile {tile} differs from tile 128 in {(o != outs[0]).sum().item()} elements"
        outs = []
        for tile in tiles:
            sim.caco_set_gemm_tile(tile)
            o = torch.zeros(M, N)
            simlib.check(sim.caco_op_gemm_bf16_f32out(P(a), P(w), P(bias), P(x), M, N, K, P(o), None))
            outs.append(o)
        for tile, o in zip(tiles[1:], outs[1:]):
            assert torch.equal(o, outs[0]), f"fp32 residual: tile {tile} differs from tile 128"
    finally:
        sim.caco_set_gemm_tile(256)


@pytest.mark.parametrize("tile", [8256, 4256, 4128], ids=["w8", "w4q", "w4h"])
@pytest.mark.parametrize("kind", ["f32r", "bf16", "silu"])
def test_gemm_w8_persistent_multi_tile_pipeline(sim, kind, tile):
    """More output tiles than workgroups (the simulator reports 16 CUs): operand loads prefetched across output-tile
    boundaries, the counted wait that leaves an epilogue's stores in flight, a ragged last M tile in mid-pipeline.
    w4q = the four-wave 128 x 128-per-wave experiment (gemm_w4q.hip, tile code 4256)."""
    assert sim.caco_set_gemm_tile(tile) == tile
    M, N, K = 1900, 768, 256          # 8 x 3 = 24 tiles on 16 workgroups
    a = _rand((M, K), 11).bfloat16()
    w = _rand((N, K), 12, 1.0 / math.sqrt(K)).bfloat16()
    bias = _rand((N,), 13)
    ref = a.float() @ w.float().T + bias
    if kind == "f32r":
        x0 = _rand((M, N), 14)
        x = x0.clone()
        simlib.check(sim.caco_op_gemm_bf16_f32out(P(a), P(w), P(bias), P(x), M, N, K, P(x), None))
        assert (x - (ref + x0)).abs().max().item() < 1e-4
    else:
        act = 1 if kind == "silu" else 0
        out = torch.full((M, N), float("nan"), dtype=torch.bfloat16)
        simlib.check(sim.caco_op_gemm_bf16(P(a), P(w), P(bias), M, N, K, act, P(out), None))
        r = torch.nn.functional.silu(ref) if act else ref
        assert ((out.float() - r).abs() / (r.abs() + 1.0)).max().item() < 1e-2
    sim.caco_set_gemm_tile(256)


def test_gemm_rejects_bad_shapes(sim):
    a = torch.zeros(64, 100, dtype=torch.bfloat16)
    with pytest.raises(ValueError):
        simlib.check(sim.caco_op_gemm_bf16(P(a), P(a), None, 64, 128, 100, 0, P(a), None))


def test_layernorm(sim):
    rows, dim = 203, 768
    x = _rand((rows, dim), 11, 3.0) + 0.7
    g, b = _rand((dim,), 12), _rand((dim,), 13)
    of = torch.empty_like(x)
    ob = torch.empty(rows, dim, dtype=torch.bfloat16)
    simlib.check(sim.caco_op_layernorm(P(x), P(g), P(b), rows, dim, 1e-5, P(of), P(ob), None))
    ref = torch.nn.functional.layer_norm(x, (dim,), g, b, 1e-5)
    assert (of - ref).abs().max().item() < 2e-5
    assert (ob.float() - ref).abs().max().item() < 0.04


# ------------------------------------------------------------------------------------------------ attention
def _attention_ref(q, k, v, key_mask, heads, hd, causal):
    B, Sq, H = q.shape
    S = k.shape[1]
    qf = q.float().reshape(B, Sq, heads, hd).transpose(1, 2)
    kf = k.float().reshape(B, S, heads, hd).transpose(1, 2)
    vf = v.float().reshape(B, S, heads, hd).transpose(1, 2)
    s = qf @ kf.transpose(-1, -2) / math.sqrt(hd)
    allow = torch.ones(B, 1, Sq, S, dtype=torch.bool)
    if key_mask is not None:
        allow = allow & (key_mask != 0)[:, None, None, :]
    if causal:
        allow = allow & torch.tril(torch.ones(S, S, dtype=torch.bool))[None, None]
    s = s.masked_fill(~allow, float("-inf"))
    return (torch.softmax(s, -1) @ vf).transpose(1, 2).reshape(B, Sq, H)


@pytest.mark.parametrize("B,S,heads,hd,causal,valid", [
    (2, 200, 2, 96, 0, [196, 77]),      # two query blocks per wave (S > 128), ragged last tile, key padding
    (3, 32, 3, 64, 1, [32, 12, 1]),     # the text tower's shape: causal AND padding
    (1, 496, 1, 96, 0, [496]),          # the audio tower's sequence length
    (2, 100, 2, 64, 1, [100, 37]), (1, 64, 2, 96, 0, [64]), (1, 130, 1, 96, 1, [129])])
def test_attention(sim, B, S, heads, hd, causal, valid):
    H = heads * hd
    qk = _rand((B, S, 2 * H), 20, 1.5).bfloat16()
    v = _rand((B, S, H), 21).bfloat16()
    mask = torch.zeros(B, S)
    for i, n in enumerate(valid):
        mask[i, :n] = 1
    qkv = torch.cat([qk, v], -1).contiguous()
    out = torch.full((B, S, H), float("nan"), dtype=torch.bfloat16)
    simlib.check(sim.caco_op_attention(P(qkv), 3 * H, H, 2 * H, P(mask), B, S, heads, hd, causal, P(out), None))
    ref = _attention_ref(qk[..., :H], qk[..., H:], v, mask, heads, hd, bool(causal))
    assert torch.isfinite(out.float()).all()
    assert (out.float() - ref).abs().max().item() < 0.03
    simlib.check(sim.caco_op_attention(P(qkv), 3 * H, H, 2 * H, None, B, S, heads, hd, causal, P(out), None))
    ref = _attention_ref(qk[..., :H], qk[..., H:], v, None, heads, hd, bool(causal))
    assert (out.float() - ref).abs().max().item() < 0.03


@pytest.mark.parametrize("B,Sq,S,heads,hd,valid", [(2, 32, 200, 2, 64, [196, 44]), (2, 1, 100, 2, 64, [100, 7]), (1, 200, 33, 1, 96, [33])])
def test_cross_attention(sim, B, Sq, S, heads, hd, valid):
    H = heads * hd
    q = _rand((B, Sq, H), 40, 1.5).bfloat16()
    kv = _rand((B, S, 2 * H), 41, 1.2).bfloat16()
    mask = torch.zeros(B, S)
    for i, n in enumerate(valid):
        mask[i, :n] = 1
    out = torch.full((B, Sq, H), float("nan"), dtype=torch.bfloat16)
    simlib.check(sim.caco_op_attention_qkv(P(q), H, Sq, P(kv), 2 * H, 0, H, P(mask), B, S, heads, hd, 0, P(out), None))
    ref = _attention_ref(q, kv[..., :H], kv[..., H:], mask, heads, hd, False)
    assert torch.isfinite(out.float()).all()
    assert (out.float() - ref).abs().max().item() < 0.03


@pytest.mark.parametrize("slope", [0.5, 3.0])
def test_attention_lazy_reference_ramp(sim, slope):
    """tests/test_gpu_ops.py::test_attention_lazy_reference_ramp on one head: scores that climb with the key index walk
    the lazy exponent reference through many tiles, in the two-block and in the one-block kernel."""
    B, S, heads, hd = 1, 500, 1, 96
    H = heads * hd
    qk = _rand((B, S, 2 * H), 40, 0.3)
    qk[:, :, 0] += 4.0
    ramp = torch.arange(S, dtype=torch.float32) / 64.0
    qk[:, :, H] += slope * ramp[None, :] * (math.sqrt(hd) * math.log(2.0) / 4.0)
    qk = qk.bfloat16()
    v = _rand((B, S, H), 41).bfloat16()
    qkv = torch.cat([qk, v], -1).contiguous()
    ref = _attention_ref(qk[..., :H], qk[..., H:], v, None, heads, hd, False)
    for Sq in (S, 100):
        out = torch.empty(B, Sq, H, dtype=torch.bfloat16)
        q = qkv[:, :Sq, :H].contiguous()
        simlib.check(sim.caco_op_attention_qkv(P(q), H, Sq, P(qkv), 3 * H, H, 2 * H, None, B, S, heads, hd, 0, P(out), None))
        assert torch.isfinite(out.float()).all()
        assert (out.float() - ref[:, :Sq]).abs().max().item() < 0.03, (slope, Sq)


# ------------------------------------------------------------------------------------------------ front end
def _mel_patches(sim, wav, max_p, dtype=torch.float32, lengths=None):
    wav = torch.as_tensor(wav).float().contiguous()
    B, n = wav.shape
    patches = torch.full((B, max_p, 256), float("nan"), dtype=dtype)
    tinds, finds, mask = torch.empty(B, max_p), torch.empty(B, max_p), torch.empty(B, max_p)
    dt = 1 if dtype == torch.bfloat16 else 0
    lens = None if lengths is None else torch.as_tensor(lengths).long().contiguous()
    simlib.check(sim.caco_mel_patches_lens(P(wav), P(lens), B, n, max_p, 0.2, 0.9, P(patches), dt, P(tinds), P(finds), P(mask), None))
    return {"audio_patches": patches, "audio_time_inds": tinds, "audio_freq_inds": finds, "audio_mask": mask}


def test_mel_spectrogram_matches_reference_golden(sim):
    g = load_golden("mel.npz")
    wav = torch.from_numpy(synth.make_waveforms(2))
    frames = int(sim.caco_mel_num_frames(wav.shape[1]))
    mel = torch.empty(2, frames, 128)
    simlib.check(sim.caco_mel_spectrogram(P(wav), 2, wav.shape[1], 0.2, 0.9, P(mel), None))
    mel = mel.numpy()
    assert mel.shape == (2, 1000, 128)
    assert np.abs(mel[0] - g["mel0"]).max() < 1e-3
    assert np.abs(mel[1] - g["mel1"].astype(np.float32)).max() < 3e-3
    assert np.allclose(mel[0][:, 0], math.log(1e-5) * 0.2 + 0.9, atol=1e-5)          # empty HTK filter, SURVEY Q13


@pytest.mark.parametrize("tag,n,max_p", [("short", 12345, 64), ("tiny", 700, 16), ("3s", 48000, 500), ("trunc", 48000, 100)])
def test_mel_ragged_lengths(sim, tag, n, max_p):
    g = load_golden("mel.npz")
    w = synth.make_waveform(7, n_samples=n)[None]
    for dt, tol in ((torch.float32, 3e-3), (torch.bfloat16, 2e-2)):
        p = _mel_patches(sim, w, max_p, dt)
        for k in ("audio_time_inds", "audio_freq_inds", "audio_mask"):
            np.testing.assert_array_equal(p[k][0].numpy(), g[f"{tag}_{k}"])
        assert np.abs(p["audio_patches"][0].float().numpy() - g[f"{tag}_audio_patches"].astype(np.float32)).max() < tol


def test_mel_per_clip_lengths_in_one_batch(sim):
    n = 48000
    lens = [48000, 20000, 7000]
    wav = np.zeros((3, n), np.float32)
    for i, L in enumerate(lens):
        wav[i, :L] = synth.make_waveform(30 + i, n_samples=L)
    p = _mel_patches(sim, wav, 150, lengths=lens)
    for i, L in enumerate(lens):
        ref = O.prepare_audio_batch(wav[i:i + 1, :L], 150)
        for k in ("audio_time_inds", "audio_freq_inds", "audio_mask"):
            np.testing.assert_array_equal(p[k][i].numpy(), ref[k][0])
        nv = int(ref["audio_mask"].sum())
        assert np.abs(p["audio_patches"][i, :nv].numpy() - ref["audio_patches"][0, :nv]).max() < 1e-3
        assert (p["audio_patches"][i, nv:] == 0).all()


# ------------------------------------------------------------------------------------------------ scoring
def test_similarity_normalize_topk(sim):
    a, t = _rand((37, 768), 40), _rand((300, 768), 41)
    an, tn = torch.empty_like(a), torch.empty_like(t)
    simlib.check(sim.caco_l2_normalize(P(a), 37, 768, P(an), None))
    simlib.check(sim.caco_l2_normalize(P(t), 300, 768, P(tn), None))
    assert np.abs(an.numpy() - O.l2_normalize(O.get_ops("numpy"), a.numpy())).max() < 1e-6
    out = torch.empty(37, 300)
    simlib.check(sim.caco_similarity(P(an), 37, P(tn), 300, 768, 14.28, P(out), 300, None))
    ref = 14.28 * (an.double() @ tn.double().T)
    assert (out.double() - ref).abs().max().item() < 1e-4
    idx = torch.empty(37, 10, dtype=torch.int32)
    val = torch.empty(37, 10)
    simlib.check(sim.caco_topk(P(out), 37, 300, 300, 1, 10, P(idx), P(val), None))
    rv, ri = torch.sort(out, dim=1, descending=True, stable=True)
    np.testing.assert_array_equal(idx.numpy(), ri[:, :10].numpy())
    np.testing.assert_array_equal(val.numpy(), rv[:, :10].numpy())


# ------------------------------------------------------------------------------------------------ whole towers
@pytest.mark.parametrize("ln_fold", [0, 1], ids=["ln_pass", "ln_folded"])
def test_tiny_config_matches_reference_golden(sim, tiny_state, ln_fold):
    """tests/test_gpu_model.py::test_tiny_config_matches_reference_golden on the simulator: 2-layer full-width towers,
    2 clips + 2 captions, against the outputs of the REFERENCE itself (tests/golden/caco_tiny.npz)."""
    g = load_golden("caco_tiny.npz")
    a, t, cc = C.tiny_configs(2)
    m = simlib.SimModel(a, t, cc).load_state_dict(tiny_state)
    assert m.set_ln_fold(ln_fold) == ln_fold
    wav = synth.make_waveforms(2)
    ab = _mel_patches(sim, wav, 500)
    rows = g["probe_rows"]
    a_emb, a_hid = m.audio_forward(ab["audio_patches"], ab["audio_time_inds"], ab["audio_freq_inds"], ab["audio_mask"])
    a_hid = a_hid.numpy()
    valid = rows[rows < 496]
    sel = np.isin(rows, valid)
    assert rel_l2(a_hid[:, valid], g["audio_hidden_rows"][:, sel]) < 1e-2
    assert cosine_rows(a_emb.numpy(), g["audio_emb"]).min() > 0.999
    assert rel_l2(a_emb.numpy(), g["audio_emb"]) < 1e-2
    ids, tmask = synth.make_captions(2, 32, 1024)
    t_emb, t_hid = m.text_forward(ids, tmask)
    keep = tmask.astype(bool)
    assert rel_l2(t_hid.numpy()[keep], g["text_hidden"][keep]) < 1e-2
    assert cosine_rows(t_emb.numpy(), g["text_emb"]).min() > 0.999
    a_n, _ = m.audio_forward(ab["audio_patches"], ab["audio_time_inds"], ab["audio_freq_inds"], ab["audio_mask"], normalize=True)
    t_n, _ = m.text_forward(ids, tmask, normalize=True)
    np.testing.assert_allclose(a_n.norm(dim=1).numpy(), 1.0, atol=1e-3)
    assert cosine_rows(a_n.numpy(), g["audio_emb_norm"]).min() > 0.999
    assert cosine_rows(t_n.numpy(), g["text_emb_norm"]).min() > 0.999
    sim_at = torch.empty(2, 2)
    scale = float(np.exp(2.6592))
    simlib.check(sim.caco_similarity(P(a_n), 2, P(t_n), 2, 768, scale, P(sim_at), 2, None))
    assert np.abs(sim_at.numpy() - g["at_logits"]).max() < 1e-3 * scale
    pos = np.broadcast_to(np.arange(32) + 2, (2, 32)).copy()
    t_pos, _ = m.text_forward(ids, tmask, position_ids=pos)
    assert cosine_rows(t_pos.numpy(), g["text_emb_pos2"]).min() > 0.999


def test_encode_audio_and_text_vs_oracle(sim, tiny_state):
    """wav -> embedding (mel kernel, bf16 patches, towers, pooling, normalise) against the oracle's fp32 path."""
    a, t, cc = C.tiny_configs(2)
    m = simlib.SimModel(a, t, cc).load_state_dict(tiny_state)
    o = O.CacoOracle(tiny_state, a, t, cc, backend="torch")
    wav = synth.make_waveforms(2, start=60)
    ids, tmask = synth.make_captions(2, 32, 1024, start=60)
    ea, et = m.encode_audio(wav), m.encode_text(ids, tmask)
    ra, rt = o.encode_audio(wav), o.encode_text(ids, tmask)
    assert cosine_rows(ea.numpy(), ra).min() > 0.999
    assert cosine_rows(et.numpy(), rt).min() > 0.999


def test_jax_side_hyperparameters_8_pool_heads_eps_1e6(sim, tiny_state):
    """The JAX model pools with 8 heads and LayerNorm eps 1e-6 on the SAME tensor shapes (SURVEY Q5 / Q6,
    src/caco/load_model.py:46): what evaluate.load_caco_torch builds for a Flax checkpoint.  Against the oracle, and
    different from the 2-head model (a silent fall-back to 2 heads would pass a shape check)."""
    from dataclasses import replace
    a, t, cc = C.tiny_configs(2)
    a8, cc8 = replace(a, layer_norm_eps=1e-6), replace(cc, num_attention_pool_heads=8)
    m = simlib.SimModel(a8, None, cc8).load_state_dict({k: v for k, v in tiny_state.items() if not k.startswith(("text_", "logit"))})
    o = O.CacoOracle(tiny_state, a8, t, cc8, backend="torch")
    wav = synth.make_waveforms(2, n_samples=32000, start=5)
    ab = _mel_patches(sim, wav, 100)
    emb, hid = m.audio_forward(ab["audio_patches"], ab["audio_time_inds"], ab["audio_freq_inds"], ab["audio_mask"], normalize=True)
    host = {k: v.numpy() for k, v in ab.items()}
    r_emb, r_hid = o.get_audio_embedding(host["audio_patches"], host["audio_time_inds"], host["audio_freq_inds"], host["audio_mask"], normalize=True)
    nv = int(host["audio_mask"][0].sum())
    assert rel_l2(hid.numpy()[:, :nv], r_hid[:, :nv]) < 1e-2
    assert cosine_rows(emb.numpy(), r_emb).min() > 0.999
    o2 = O.CacoOracle(tiny_state, a8, t, cc, backend="torch")
    r2, _ = o2.get_audio_embedding(host["audio_patches"], host["audio_time_inds"], host["audio_freq_inds"], host["audio_mask"], normalize=True)
    assert rel_l2(emb.numpy(), r_emb) < 0.5 * rel_l2(emb.numpy(), r2)      # the two head counts differ by several per cent


def test_mel_lengths_bound_the_sample_fetch(sim):
    """A row that holds something else past lengths[b] (a clip cut out of a longer buffer): the tail frames must see the
    STFT's zero padding, exactly as when the reference pre-processes that clip alone (mel.hip)."""
    n = 24000
    lens = [24000, 10100, 5130]          # 64 and 33 frames: the last used frames reach past the clip
    wav = synth.make_waveform(50, n_samples=n)[None].repeat(3, 0).copy()       # every row full of signal
    p = _mel_patches(sim, wav, 80, lengths=lens)
    for i, L in enumerate(lens):
        ref = O.prepare_audio_batch(wav[i:i + 1, :L], 80)
        np.testing.assert_array_equal(p["audio_mask"][i].numpy(), ref["audio_mask"][0])
        nv = int(ref["audio_mask"].sum())
        assert np.abs(p["audio_patches"][i, :nv].numpy() - ref["audio_patches"][0, :nv]).max() < 1e-3


@pytest.mark.parametrize("ngroup", ["0", "2", "4", None])
def test_gemm_w8_n_tile_groups(sim, ngroup, caco_switch):
    """The persistent kernel's tile order in groups of n-tiles (w4_decode; default since round 3: groups of 3-4 at
    K <= 1024): equal groups, a ragged last group (5 n-tiles in groups of 2 / 4), one group; both cursors (operand
    prefetch and epilogue) must decode the same order."""
    if ngroup is None:
        caco_switch(sim, "CACO_W_NGROUP", None)
    else:
        caco_switch(sim, "CACO_W_NGROUP", ngroup)
    sim.caco_set_gemm_tile(8256)
    for (M, N, K) in ((700, 1280, 128), (520, 2304, 64)):      # 3 x 5 and 3 x 9 tiles
        a = _rand((M, K), 1).bfloat16()
        w = _rand((N, K), 2, 1.0 / math.sqrt(K)).bfloat16()
        bias = _rand((N,), 3)
        out = torch.full((M, N), float("nan"), dtype=torch.bfloat16)
        simlib.check(sim.caco_op_gemm_bf16(P(a), P(w), P(bias), M, N, K, 0, P(out), None))
        ref = a.float() @ w.float().T + bias
        assert ((out.float() - ref).abs() <= 2.0 ** -8 * ref.abs() + 2e-3).all(), (ngroup, M, N, K)
    sim.caco_set_gemm_tile(256)


@pytest.mark.parametrize("B,Sq,S,heads,causal,valid", [
    (3, 32, 32, 12, 1, [32, 12, 1]),       # the text tower: 36 units, causal AND padding
    (5, 32, 32, 3, 0, [32, 7, 20, 32, 1]),  # 15 units: the last workgroup has an idle wave
    (2, 64, 64, 2, 1, [64, 33]),           # two key tiles, two query blocks, causal skips the upper-right tile
    (2, 37, 37, 2, 1, [37, 5]),            # ragged second tile
    (2, 20, 50, 2, 0, [50, 33]),           # cross-attention lengths
    (2, 1, 64, 2, 0, [64, 2]),             # a single query row (a cached decode step)
    (1, 33, 33, 1, 0, [0])])               # every key masked: rows of zeros, no NaN
def test_attention_small_kernel(sim, caco_switch, B, Sq, S, heads, causal, valid):
    """attention_small.hip (one wave per (clip, head, 32-query block), opt-in) against the same checker as the big kernel,
    and against the big kernel itself."""
    hd = 64
    H = heads * hd
    q = _rand((B, Sq, H), 50, 1.5).bfloat16()
    kv = _rand((B, S, 2 * H), 51, 1.2).bfloat16()
    mask = torch.zeros(B, S)
    for i, n in enumerate(valid):
        mask[i, :n] = 1
    outs = {}
    for flag in ("1", "0"):
        caco_switch(sim, "CACO_ATTN_SMALL", flag)
        out = torch.full((B, Sq, H), float("nan"), dtype=torch.bfloat16)
        simlib.check(sim.caco_op_attention_qkv(P(q), H, Sq, P(kv), 2 * H, 0, H, P(mask), B, S, heads, hd, causal, P(out), None))
        outs[flag] = out.float()
    assert torch.isfinite(outs["1"]).all()
    if valid == [0]:
        assert (outs["1"] == 0).all()
        return
    ref = _attention_ref(q, kv[..., :H], kv[..., H:], mask, heads, hd, bool(causal))
    live = torch.isfinite(ref).all(-1)                      # rows with at least one visible key
    assert (outs["1"][live] - ref[live]).abs().max().item() < 0.03
    assert (outs["1"][~live] == 0).all()
    assert (outs["1"] - outs["0"]).abs().max().item() < 0.02
    # no mask pointer
    caco_switch(sim, "CACO_ATTN_SMALL", "1")
    out = torch.full((B, Sq, H), float("nan"), dtype=torch.bfloat16)
    simlib.check(sim.caco_op_attention_qkv(P(q), H, Sq, P(kv), 2 * H, 0, H, None, B, S, heads, hd, causal, P(out), None))
    ref = _attention_ref(q, kv[..., :H], kv[..., H:], None, heads, hd, bool(causal))
    assert (out.float() - ref).abs().max().item() < 0.03


def test_pos_embed_in_the_patch_embed_epilogue(sim, tiny_state, caco_switch):
    """CACO_POS_FUSE=1: the positional embedding as a gathered residual of the patch-embed GEMM (w8 MODE 5) instead of a
    separate pass.  Same hidden states as the separate kernel (fp32 re-association only), also for positions that are not
    small integers (those rows are finished by the exact per-row kernel) and for a ragged last M tile."""
    from dataclasses import replace
    a, t, cc = C.tiny_configs(2)
    a1 = replace(a, num_layers=0)                      # patch embed + positional embedding + final LayerNorm only
    m = simlib.SimModel(a1, None, cc).load_state_dict({k: v for k, v in tiny_state.items() if k.startswith("audio_") and ".layers." not in k})
    wav = synth.make_waveforms(3, n_samples=41000, start=3)
    ab = _mel_patches(sim, wav, 130)                   # M = 390: one full and one ragged 256-row tile
    tin = ab["audio_time_inds"].clone()
    tin[1, 5] = 2.5                                     # not an integer
    tin[2, 7] = 4000.0                                  # integer, far outside the table
    tin[0, 9] = -1.0
    sim.caco_set_gemm_tile(8256)
    try:
        outs = {}
        for flag in ("0", "1"):
            caco_switch(sim, "CACO_POS_FUSE", flag)
            _, hid = m.audio_forward(ab["audio_patches"], tin, ab["audio_freq_inds"], ab["audio_mask"])
            outs[flag] = hid.numpy()
    finally:
        sim.caco_set_gemm_tile(256)
    assert np.isfinite(outs["1"]).all()
    assert np.abs(outs["1"] - outs["0"]).max() < 2e-5
    assert (outs["1"] != outs["0"]).any()               # (x + te) + fe vs x + (te + fe): the other path did run
    for (b, p) in ((1, 5), (2, 7), (0, 9)):             # the exact path's rows are bit-identical
        np.testing.assert_array_equal(outs["1"][b, p], outs["0"][b, p])
    o = O.CacoOracle(tiny_state, a1, t, cc, backend="torch")
    _, ref = o.get_audio_embedding(ab["audio_patches"].numpy(), tin.numpy(), ab["audio_freq_inds"].numpy(), ab["audio_mask"].numpy())
    assert rel_l2(outs["1"], ref) < 5e-3


def test_tiny_config_golden_with_round3_switches(sim, tiny_state, caco_switch):
    """The 2-layer towers against the reference golden with every round-3 opt-in on at once (fused positional embedding,
    short-sequence attention kernel), on the persistent GEMM."""
    caco_switch(sim, "CACO_POS_FUSE", "1")
    caco_switch(sim, "CACO_ATTN_SMALL", "1")
    sim.caco_set_gemm_tile(8256)
    try:
        test_tiny_config_matches_reference_golden(sim, tiny_state, 0)
    finally:
        sim.caco_set_gemm_tile(256)


@pytest.mark.skipif(os.environ.get("CACO_SIM_ASAN", "0") in ("", "0"), reason="set CACO_SIM_ASAN=1: builds the ASan variant of the simulator (~2 min)")
def test_kernels_and_c_abi_are_asan_clean():
    """tools/wavesim/asan_check.py: every kernel and the C-ABI layer under AddressSanitizer with exactly-sized buffers and
    ragged shapes (the GPU pool has no device sanitizer; result recorded in profiles/r3_cpu/asan.txt)."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(simlib.REPO, "tools", "wavesim", "asan_check.py")], capture_output=True, text=True)
    assert r.returncode == 0 and "ASAN CLEAN" in r.stdout, r.stdout[-4000:]


# ------------------------------------------------------------------------------------------------ the other entry points
def test_audiomae_matches_reference_golden(sim):
    """AudioMAE.forward (mae.py:217-247): 2-layer encoder on 100 visible patches + 2-layer decoder on all 496, against the
    reference's own output rows (tests/golden/mae_tiny.npz)."""
    from dataclasses import replace
    g = load_golden("mae_tiny.npz")
    enc = replace(C.default_audio_config(), num_layers=2)
    m = simlib.SimModel(enc, None, C.default_caco_config(), mae_decoder_layers=2).load_state_dict(synth.make_audiomae_state(enc, enc))
    ab = _mel_patches(sim, synth.make_waveforms(2), 500)
    sp = synth.make_mae_split(2, 496, 100, 8)
    np.testing.assert_array_equal(sp["visible"], g["visible"])
    vis = torch.from_numpy(sp["visible"])
    x = torch.stack([ab["audio_patches"][i][vis[i]] for i in range(2)]).contiguous()
    f = lambda a: torch.as_tensor(a).float().contiguous()
    out = torch.empty(2, 496, 256)
    args = [f(torch.ones(2, 100)), f(sp["time_inds"]), f(sp["freq_inds"]), f(sp["restore_time_inds"]), f(sp["restore_freq_inds"]),
            f(torch.ones(2, 396))]                      # kept alive across the call: P() only takes the address
    simlib.check(sim.caco_mae_forward(m.h, P(x), 0, *[P(t) for t in args], 2, 100, 396, P(out), None), "mae_forward")
    assert rel_l2(out.numpy()[:, g["rows"]], g["out_rows"]) < 1e-2


def _decoder_model(sim):
    from dataclasses import replace
    a, t, cc = C.tiny_configs(2)
    d = replace(t, num_hidden_layers=2)
    state = synth.make_caco_state(a, t, cc, seed=0, decoder_cfg=d)
    return simlib.SimModel(a, t, cc, caption_decoder_layers=2).load_state_dict(state), t


def test_caption_decoder_ragged_masks_and_cached_steps(sim):
    """RobertaDecoder.forward (roberta.py:337-373) on seeded states with ragged masks on both sides against the reference's
    logits (decoder_tiny.npz `rand_logits`), the masked-audio-token invariance, and caco_decode_step (key / value caches)
    against the full-prefix form position by position."""
    import ctypes
    g = load_golden("decoder_tiny.npz")
    m, t = _decoder_model(sim)
    H, V = t.hidden_size, t.vocab_size
    rng = np.random.RandomState(5)
    th = torch.from_numpy(rng.randn(2, 20, H).astype(np.float32))
    ah = torch.from_numpy(rng.randn(2, 70, H).astype(np.float32))
    tm = torch.ones(2, 20, dtype=torch.int64); tm[1, 13:] = 0
    am = torch.ones(2, 70); am[0, 50:] = 0

    def dec(th_, tm_, ah_, am_):
        lg = torch.empty(th_.shape[0], th_.shape[1], V)
        simlib.check(sim.caco_decoder_forward(m.h, P(th_), P(tm_), P(ah_), P(am_), th_.shape[0], th_.shape[1], ah_.shape[1], P(lg), None), "decoder")
        return lg.numpy()
    lg = dec(th, tm, ah, am)
    keep = tm.numpy().astype(bool)
    assert rel_l2(lg[keep], g["rand_logits"][keep]) < 1e-2
    assert cosine_rows(lg[keep], g["rand_logits"][keep]).min() > 0.999
    ah2 = ah.clone(); ah2[0, 50:] = 100.0
    np.testing.assert_array_equal(dec(th, tm, ah2, am)[0], lg[0])
    # cached steps == text tower + decoder on the growing prefix
    ids = torch.from_numpy(g["ids"][:, :6]).contiguous()
    _, thid = m.text_forward(ids, torch.ones_like(ids))
    full = dec(thid.contiguous(), torch.ones_like(ids), ah, am)
    st = ctypes.c_void_p()
    simlib.check(sim.caco_decode_begin(m.h, P(ah), P(am), 2, 70, 8, ctypes.byref(st), None), "decode_begin")
    try:
        for p in range(6):
            tok = ids[:, p].contiguous()
            lgp = torch.empty(2, V)
            simlib.check(sim.caco_decode_step(st, P(tok), P(lgp), None), "decode_step")
            assert rel_l2(lgp.numpy(), full[:, p]) < 3e-3, p
    finally:
        sim.caco_decode_end(st)


def test_token_group_mean_and_strided_similarity(sim):
    x = _rand((2, 37, 768), 60)
    out = torch.empty(2, 4, 768)
    simlib.check(sim.caco_token_group_mean(P(x), 2, 37, 768, 8, P(out), None))
    np.testing.assert_allclose(out.numpy(), x[:, :32].reshape(2, 4, 8, 768).mean(2).numpy(), atol=1e-6)
    bank = torch.nn.functional.normalize(_rand((9, 2, 768), 61), dim=-1).contiguous()      # packed [B, 2, P] exchange buffer
    sim_m = torch.full((9, 12), float("nan"))
    simlib.check(sim.caco_similarity_ld(P(bank[:, 0]), 9, 2 * 768, P(bank[:, 1]), 9, 2 * 768, 768, 2.0, P(sim_m), 12, None))
    ref = 2.0 * bank[:, 0].double() @ bank[:, 1].double().T
    assert (sim_m[:, :9].double() - ref).abs().max().item() < 1e-5 and torch.isnan(sim_m[:, 9:]).all()


@pytest.mark.parametrize("tile", [128, 2256, 8256, 4256, 4128])
def test_gemm_ragged_m_writes_nothing_past_row_m(sim, tile):
    """tests/test_gpu_ops.py::test_gemm_ragged_m_writes_nothing_past_row_m on the simulator, whose buffer descriptors
    range-check the per-lane offset only (as the hardware does): guard rows behind the output keep their sentinel."""
    sim.caco_set_gemm_tile(tile)
    M, N, K, G = 300, 768, 128, 256
    a = _rand((M, K), 1).bfloat16()
    w = _rand((N, K), 2, 1.0 / math.sqrt(K)).bfloat16()
    bias = _rand((N,), 3)
    out = torch.full((M + G, N), 7.0, dtype=torch.bfloat16)
    simlib.check(sim.caco_op_gemm_bf16(P(a), P(w), P(bias), M, N, K, 1, P(out), None))
    x = torch.full((M + G, N), 7.0)
    x[:M] = _rand((M, N), 4)
    simlib.check(sim.caco_op_gemm_bf16_f32out(P(a), P(w), P(bias), P(x), M, N, K, P(x), None))
    assert (out[M:] == 7.0).all() and (x[M:] == 7.0).all()
    sim.caco_set_gemm_tile(256)


def test_results_do_not_depend_on_the_schedule(sim, monkeypatch):
    """The same launches with waves and lanes run first-to-last and last-to-first between rendezvous points, and with LDS-DMA
    landing late (at the covering wait) and at once: bitwise the same outputs.  A kernel that only works because wave 0
    happens to run first (a missing barrier, a read of another wave's data before its covering wait) would differ."""
    M, N, K = 700, 768, 192
    a = _rand((M, K), 1).bfloat16()
    w = _rand((N, K), 2, 1.0 / math.sqrt(K)).bfloat16()
    bias = _rand((N,), 3)
    B, S, heads, hd = 2, 200, 2, 96
    qkv = _rand((B, S, 3 * heads * hd), 4, 1.2).bfloat16()
    mask = torch.ones(B, S)
    mask[1, 150:] = 0
    wav = torch.from_numpy(synth.make_waveforms(1, n_samples=20000))
    outs = {}
    for order in ("forward", "reverse", "random:5"):
        monkeypatch.setenv("WAVESIM_ORDER", order)
        got = []
        for tile in (8256, 4256, 4128, 2256):
            sim.caco_set_gemm_tile(tile)
            o = torch.empty(M, N, dtype=torch.bfloat16)
            simlib.check(sim.caco_op_gemm_bf16(P(a), P(w), P(bias), M, N, K, 1, P(o), None))
            x = _rand((M, N), 5)
            simlib.check(sim.caco_op_gemm_bf16_f32out(P(a), P(w), P(bias), P(x), M, N, K, P(x), None))
            got += [o, x]
        sim.caco_set_gemm_tile(256)
        o = torch.empty(B, S, heads * hd, dtype=torch.bfloat16)
        simlib.check(sim.caco_op_attention(P(qkv), 3 * heads * hd, heads * hd, 2 * heads * hd, P(mask), B, S, heads, hd, 0, P(o), None))
        got.append(o)
        got.append(_mel_patches(sim, wav, 80)["audio_patches"])
        outs[order] = got
    for other in ("reverse", "random:5"):
        for x, y in zip(outs["forward"], outs[other]):
            assert torch.equal(x.float(), y.float()), other


def test_bad_arguments_are_refused_not_executed(sim, tiny_state):
    """Every entry point with arguments it cannot serve returns a status and a message - on the simulator a missing check
    would dereference a null / run a kernel off its buffers and take the process down (tests/test_abi.py covers the calls
    that need no device; these need one)."""
    import ctypes
    z = torch.zeros(64, 768)
    zb = torch.zeros(64, 768, dtype=torch.bfloat16)
    bad = []

    def refused(rc, what):
        if rc == 0:
            bad.append(what)
        else:
            assert sim.caco_last_error(), what

    refused(sim.caco_op_gemm_bf16(P(zb), P(zb), None, 64, 100, 768, 0, P(zb), None), "gemm N % 128")
    refused(sim.caco_op_gemm_bf16(P(zb), P(zb), None, 64, 128, 100, 0, P(zb), None), "gemm K % 64")
    refused(sim.caco_op_gemm_bf16(P(zb), P(zb), None, 0, 128, 64, 0, P(zb), None), "gemm M = 0")
    refused(sim.caco_op_gemm_bf16(None, P(zb), None, 64, 128, 64, 0, P(zb), None), "gemm null A")
    refused(sim.caco_op_gemm_bf16(P(zb), P(zb), None, 64, 128, 64, 7, P(zb), None), "gemm unknown activation")
    refused(sim.caco_op_gemm_bf16_strided(P(zb), 60, P(zb), 64, None, 64, 128, 64, 0, P(zb), 128, None), "gemm lda < K")
    refused(sim.caco_op_attention(P(zb), 768, 256, 512, None, 1, 64, 4, 80, 0, P(zb), None), "attention head_dim 80")
    refused(sim.caco_op_attention(P(zb), 770, 256, 512, None, 1, 64, 4, 64, 0, P(zb), None), "attention ld % 8")
    refused(sim.caco_op_attention(None, 768, 256, 512, None, 1, 64, 4, 64, 0, P(zb), None), "attention null")
    refused(sim.caco_op_attention(P(zb), 768, 256, 512, None, 0, 64, 4, 64, 0, P(zb), None), "attention batch 0")
    refused(sim.caco_op_attention_qkv(P(zb), 256, 20, P(zb), 768, 256, 512, None, 1, 64, 4, 64, 1, P(zb), None), "causal with Sq != S")
    refused(sim.caco_topk(P(z), 4, 16, 16, 1, 0, P(z), None, None), "topk k = 0")
    refused(sim.caco_topk(P(z), 4, 16, 16, 1, 65, P(z), None, None), "topk k > 64")
    refused(sim.caco_similarity(None, 4, P(z), 4, 768, 1.0, P(z), 4, None), "similarity null")
    refused(sim.caco_similarity(P(z), 4, P(z), 4, 768, 1.0, P(z), 2, None), "similarity ld_out < nt")
    refused(sim.caco_mel_patches(P(z), 1, 0, 16, 0.2, 0.9, P(z), 0, P(z), P(z), P(z), None), "mel n_samples 0")
    refused(sim.caco_mel_patches(P(z), 1, 700, 16, 0.2, 0.9, P(z), 5, P(z), P(z), P(z), None), "mel dtype")
    refused(sim.caco_token_group_mean(P(z), 1, 64, 766, 8, P(z), None), "group mean dim % 4")
    # model-level: calls before the weights are final, on the wrong tower, with nulls
    a, t, cc = C.tiny_configs(1)
    m = simlib.SimModel(a, None, cc)
    e = torch.zeros(2, 768)
    refused(sim.caco_audio_forward(m.h, P(z), 0, P(z), P(z), P(z), 2, 8, 0, P(e), None, None), "audio forward before finalize")
    m.load_state_dict({k: v for k, v in synth.make_caco_state(a, t, cc).items() if k.startswith("audio_")})
    refused(sim.caco_audio_forward(m.h, None, 0, P(z), P(z), P(z), 2, 8, 0, P(e), None, None), "audio forward null patches")
    refused(sim.caco_audio_forward(m.h, P(z), 0, P(z), P(z), P(z), 0, 8, 0, P(e), None, None), "audio forward batch 0")
    ids = torch.zeros(2, 8, dtype=torch.int64)
    refused(sim.caco_text_forward(m.h, P(ids), P(ids), None, 2, 8, 0, P(e), None, None), "text forward on an audio-only model")
    refused(sim.caco_decoder_forward(m.h, P(z), P(ids), P(z), P(z), 2, 8, 8, P(z), None), "decoder not initialised")
    st = ctypes.c_void_p()
    refused(sim.caco_decode_begin(m.h, P(z), P(z), 2, 8, 4, ctypes.byref(st), None), "decode_begin without a decoder")
    refused(sim.caco_mae_forward(m.h, P(z), 0, P(z), P(z), P(z), P(z), P(z), P(z), 2, 4, 4, P(z), None), "mae forward without a decoder")
    refused(sim.caco_encode_audio_ex(m.h, P(z), None, 2, 0, 8, P(e), 0, None), "encode_audio n_samples 0")
    refused(sim.caco_encode_audio_ex(m.h, P(z), None, 2, 700, 8, P(e), 100, None), "encode_audio ld_emb < projection")
    cfg = _bad_config()
    h = ctypes.c_void_p()
    refused(sim.caco_create(ctypes.byref(cfg), ctypes.byref(h)), "create with 3 pool heads")
    assert not bad, bad


def _bad_config():
    import ctypes
    from cacophony_amd import _lib
    cfg = _lib.CacoConfigC()
    simlib.load().caco_default_config(ctypes.byref(cfg))
    cfg.pool_heads = 3
    return cfg


def test_random_shape_sweep(sim):
    """A seeded slice of tools/wavesim/fuzz.py (random ragged / tiny shapes for every kernel family and for the whole narrow
    towers, guard rows behind every output).  Longer runs: `python tools/wavesim/fuzz.py --cases 400 --seed N`
    (profiles/r3_cpu/fuzz.txt)."""
    import sys
    sys.path.insert(0, os.path.join(simlib.REPO, "tools", "wavesim"))
    import fuzz
    log = fuzz.run(45, 123, ["gemm", "attention", "layernorm", "mel", "topk"])
    log += fuzz.run(6, 124, ["model"])
    assert len(log) == 51


@pytest.mark.parametrize("pool_heads", [2, 8])
def test_final_layernorm_inside_the_pooler(sim, tiny_state, caco_switch, pool_heads):
    """CACO_POOL_FUSE=1: encode_audio's final LayerNorm applied inside the pooling kernel (no normalised rows written).
    Same embeddings as the two-launch form up to the bf16 rounding of the rows it no longer takes, and within the parity
    bars of the oracle; ragged clip lengths (masked tokens, a clip shorter than the window)."""
    from dataclasses import replace
    a, t, cc = C.tiny_configs(1)
    cc = replace(cc, num_attention_pool_heads=pool_heads)
    m = simlib.SimModel(a, None, cc).load_state_dict({k: v for k, v in tiny_state.items() if k.startswith("audio_")})
    o = O.CacoOracle(tiny_state, a, t, cc, backend="torch")
    n = 30000
    lens = [30000, 11000, 4000]
    wav = np.stack([synth.make_waveform(70 + i, n_samples=n) for i in range(3)]).astype(np.float32)
    for i, L in enumerate(lens):
        wav[i, L:] = 0
    embs = {}
    for flag in ("0", "1"):
        caco_switch(sim, "CACO_POOL_FUSE", flag)
        embs[flag] = m.encode_audio(wav, lengths=lens).numpy()
    ref = np.concatenate([o.encode_audio(wav[i:i + 1, :L], max(8, n * 8 // 160 // 16)) for i, L in enumerate(lens)], 0)
    assert np.isfinite(embs["1"]).all()
    assert (embs["1"] != embs["0"]).any()                          # the other path did run
    assert cosine_rows(embs["1"], embs["0"]).min() > 0.99999
    assert cosine_rows(embs["1"], ref).min() > 0.999 and rel_l2(embs["1"], ref) <= rel_l2(embs["0"], ref) * 1.05


@pytest.mark.parametrize("heads", [1, 4, 8])
def test_audio_pooler_head_counts_match_reference(sim, tiny_state, heads):
    """The pooling kernel at 1, 4 and 8 heads against the reference's own outputs (tests/golden/pool_heads.npz)."""
    from dataclasses import replace
    g = load_golden("pool_heads.npz")
    a, t, cc = C.tiny_configs(2)
    m = simlib.SimModel(a, None, replace(cc, num_attention_pool_heads=heads)).load_state_dict(
        {k: v for k, v in tiny_state.items() if k.startswith("audio_")})
    ab = _mel_patches(sim, synth.make_waveforms(2, 48000, start=30), 150)
    emb, _ = m.audio_forward(ab["audio_patches"], ab["audio_time_inds"], ab["audio_freq_inds"], ab["audio_mask"], normalize=True)
    ref = g[f"emb_heads{heads}"]
    other = g["emb_heads4" if heads != 4 else "emb_heads8"]
    assert cosine_rows(emb.numpy(), ref).min() > 0.999
    assert rel_l2(emb.numpy(), ref) < 0.5 * rel_l2(emb.numpy(), other)


@pytest.mark.skipif(os.environ.get("CACO_SIM_FULL", "0") in ("", "0"), reason="set CACO_SIM_FULL=1: the 12 + 12-layer model on the simulator (~5 min)")
@pytest.mark.parametrize("variant", ["default", "w8_and_round3_switches"])
def test_full_config_matches_reference_golden(sim, full_state, caco_switch, variant):
    """tests/test_gpu_model.py::test_full_config_matches_reference_golden on the simulator: the full 12 + 12-layer model,
    4 clips + 4 captions, against the reference's own outputs (tests/golden/caco_full.npz), incl. the centred cosine -
    with the kernels a batch of 4 gets by default, and with the persistent GEMM forced plus every round-3 switch on.
    Recorded in profiles/r3_cpu/wavesim_runs.txt."""
    if variant != "default":
        for k in ("CACO_ATTN_SMALL", "CACO_POS_FUSE", "CACO_POOL_FUSE"):
            caco_switch(sim, k, 1)
        sim.caco_set_gemm_tile(8256)
    try:
        _full_config_golden(sim, full_state)
    finally:
        sim.caco_set_gemm_tile(256)


def _full_config_golden(sim, full_state):
    g = load_golden("caco_full.npz")
    m = simlib.SimModel(C.default_audio_config(), C.default_text_config(), C.default_caco_config()).load_state_dict(full_state)
    ab = _mel_patches(sim, synth.make_waveforms(4), 500)
    rows = g["probe_rows"]
    a_emb, a_hid = m.audio_forward(ab["audio_patches"], ab["audio_time_inds"], ab["audio_freq_inds"], ab["audio_mask"])
    valid = rows[rows < 496]
    sel = np.isin(rows, valid)
    assert rel_l2(a_hid.numpy()[:, valid], g["audio_hidden_rows"][:, sel]) < 1e-2
    assert cosine_rows(a_emb.numpy(), g["audio_emb"]).min() > 0.999
    ids, tmask = synth.make_captions(4, 32, 50265)
    t_emb, t_hid = m.text_forward(ids, tmask)
    keep = tmask.astype(bool)
    assert rel_l2(t_hid.numpy()[keep], g["text_hidden"][keep]) < 1e-2
    assert cosine_rows(t_emb.numpy(), g["text_emb"]).min() > 0.999
    a_n = m.encode_audio(synth.make_waveforms(4), 500)                   # the fused front end + (optionally) fused pooler path
    t_n, _ = m.text_forward(ids, tmask, normalize=True)
    for got, key in ((a_n.numpy(), "audio_emb_norm"), (t_n.numpy(), "text_emb_norm")):
        mu = g[key].mean(0, keepdims=True)
        assert cosine_rows(got, g[key]).min() > 0.999
        assert cosine_rows(got - mu, g[key] - mu).min() > 0.99
    scale = float(np.exp(2.6592))
    at = torch.empty(4, 4)
    simlib.check(sim.caco_similarity(P(a_n), 4, P(t_n), 4, 768, scale, P(at), 4, None))
    assert np.abs(at.numpy() - g["at_logits"]).max() < 1e-3 * scale


@pytest.mark.parametrize("B,n", [(9, 20480), (16, 5120), (3, 41000)])
def test_pingpong_traversal_changes_nothing_but_the_order(sim, tiny_state, caco_switch, B, n):
    """CACO_PINGPONG=1: consecutive kernels of a layer walk the rows in opposite directions inside the 8 ranges the XCDs own
    (reversed tile lists, range-ordered LayerNorm, contiguous clips per XCD in attention).  Pure re-ordering of independent
    work: hidden states and embeddings are BITWISE those of the default order - for batches that are and are not multiples
    of 8, with row counts that leave ragged ranges."""
    a, t, cc = C.tiny_configs(2)
    m = simlib.SimModel(a, None, cc).load_state_dict({k: v for k, v in tiny_state.items() if k.startswith("audio_")})
    wav = np.stack([synth.make_waveform(90 + i, n_samples=n) for i in range(B)]).astype(np.float32)
    ab = _mel_patches(sim, wav, max(8, n * 8 // 160 // 16))
    sim.caco_set_gemm_tile(8256)
    try:
        outs = {}
        for flag in ("0", "1"):
            caco_switch(sim, "CACO_PINGPONG", flag)
            emb, hid = m.audio_forward(ab["audio_patches"], ab["audio_time_inds"], ab["audio_freq_inds"], ab["audio_mask"], normalize=True)
            outs[flag] = (emb.numpy().copy(), hid.numpy().copy())
    finally:
        sim.caco_set_gemm_tile(256)
    np.testing.assert_array_equal(outs["1"][1], outs["0"][1])
    np.testing.assert_array_equal(outs["1"][0], outs["0"][0])
    assert np.isfinite(outs["1"][0]).all()


def test_argument_validation_returns_status_codes(sim):
    """Null pointers, zero / negative sizes, unsupported head sizes, null model handles: every entry point answers with a status
    code and a message (include/caco_hip.h: no exceptions, no crashes across the ABI).  The validation is host code, identical in
    the product and the simulator build; here a miss would be a segfault of the test process."""
    import ctypes as C
    Pn = lambda x: C.c_void_p(x.ctypes.data)
    a = np.zeros((64, 64), np.uint16); w = np.zeros((128, 64), np.uint16); b = np.zeros(128, np.float32); o = np.zeros((64, 128), np.uint16)
    xf = np.zeros((4, 128), np.float32); g = np.ones(128, np.float32); of = np.zeros((4, 768), np.float32)
    qkv = np.zeros((8, 288), np.uint16); ao = np.zeros((8, 96), np.uint16)
    sm = np.zeros((2, 8), np.float32); idx = np.zeros((2, 3), np.int32); val = np.zeros((2, 3), np.float32)
    wav = np.zeros((1, 16000), np.float32); pat = np.zeros((1, 16, 256), np.float32)
    calls = {
        "gemm null A": lambda: sim.caco_op_gemm_bf16(None, Pn(w), Pn(b), 64, 128, 64, 0, Pn(o), None),
        "gemm null out": lambda: sim.caco_op_gemm_bf16(Pn(a), Pn(w), Pn(b), 64, 128, 64, 0, None, None),
        "gemm fp32 null out": lambda: sim.caco_op_gemm_bf16_f32out(Pn(a), Pn(w), Pn(b), None, 64, 128, 64, None, None),
        "gemm M < 0": lambda: sim.caco_op_gemm_bf16(Pn(a), Pn(w), Pn(b), -5, 128, 64, 0, Pn(o), None),
        "layernorm null x": lambda: sim.caco_op_layernorm(None, Pn(g), Pn(g), 4, 128, 1e-5, Pn(of), None, None),
        "layernorm no output": lambda: sim.caco_op_layernorm(Pn(xf), Pn(g), Pn(g), 4, 128, 1e-5, None, None, None),
        "layernorm 0 rows": lambda: sim.caco_op_layernorm(Pn(xf), Pn(g), Pn(g), 0, 128, 1e-5, Pn(of), None, None),
        "attention null qkv": lambda: sim.caco_op_attention(None, 288, 96, 192, None, 1, 8, 1, 96, 0, Pn(ao), None),
        "attention null out": lambda: sim.caco_op_attention(Pn(qkv), 288, 96, 192, None, 1, 8, 1, 96, 0, None, None),
        "attention seq 0": lambda: sim.caco_op_attention(Pn(qkv), 288, 96, 192, None, 1, 0, 1, 96, 0, Pn(ao), None),
        "attention head_dim 80": lambda: sim.caco_op_attention(Pn(qkv), 288, 96, 192, None, 1, 8, 1, 80, 0, Pn(ao), None),
        "similarity null": lambda: sim.caco_similarity(None, 2, None, 2, 768, 1.0, None, 2, None),
        "l2_normalize null": lambda: sim.caco_l2_normalize(None, 2, 768, None, None),
        "topk null": lambda: sim.caco_topk(None, 2, 8, 8, 1, 3, None, None, None),
        "topk k = 0": lambda: sim.caco_topk(Pn(sm), 2, 8, 8, 1, 0, Pn(idx), Pn(val), None),
        "mel null": lambda: sim.caco_mel_patches(None, 1, 16000, 16, 0.2, 0.9, None, 1, None, None, None, None),
        "mel batch 0": lambda: sim.caco_mel_patches(Pn(wav), 0, 16000, 16, 0.2, 0.9, Pn(pat), 0, None, None, None, None),
        "mel samples < 0": lambda: sim.caco_mel_patches(Pn(wav), 1, -3, 16, 0.2, 0.9, Pn(pat), 0, None, None, None, None),
        "token_group_mean null": lambda: sim.caco_token_group_mean(None, 1, 16, 128, 8, None, None),
        "audio forward, null model": lambda: sim.caco_audio_forward(None, Pn(pat), 0, None, None, None, 1, 16, 1, Pn(of), None, None),
        "text forward, null model": lambda: sim.caco_text_forward(None, None, None, None, 1, 4, 1, None, None, None),
        "encode_audio, null model": lambda: sim.caco_encode_audio(None, Pn(wav), 1, 16000, 16, Pn(of), None),
        "load_tensor, null model": lambda: sim.caco_load_tensor(None, b"x", Pn(xf), None, 0),
        "decode_step null": lambda: sim.caco_decode_step(None, None, None, None),
    }
    for name, call in calls.items():
        assert call() != 0, f"{name}: accepted"
        assert len(sim.caco_last_error()) > 0, name
    sim.caco_destroy(None)             # no-ops on null handles
    sim.caco_decode_end(None)
    assert sim.caco_profile_report(None, 0) >= 0
