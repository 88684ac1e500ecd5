#!/usr/bin/env python3
"""Headline benchmark: audio-text pairs/sec embedded + scored on MI355X (BASELINE.json metric).

One "step" = one pass of the whole hot path over one synthetic batch that is already resident in HBM:
  wav fp32 [256, 160000] -> fused log-mel patches (bf16) -> AudioMAE-ViT encoder + pooler -> L2 norm  \\  both towers write
  ids/mask int64 [256, 32] -> causal RoBERTa encoder + pooler + projection -> L2 norm                  /  ONE [256, 2, 768] buffer
  [N > 1: ONE RCCL all-gather of that buffer]  -> similarity row block [256, N*256] (banks read through their row stride)
Per-GPU work is fixed as N grows (weak scaling); value = N * 256 * steps / wall, wall = max over ranks
between two barrier + synchronize brackets.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line; it also carries
  roofline      : the dominant kernel (the 768 -> 3072 SiLU bf16 MFMA GEMM of the audio MLP), achieved algorithmic TFLOP/s
                  from its average launch duration measured with HIP events recorded inside the library on the launch
                  stream, vs the 2.5 PFLOP/s dense bf16 MFMA peak; `traffic` = HBM / fabric bytes of one launch of that kernel from
                  the TCC counters, collected by this run itself AFTER the timed region (N = 1, rank 0: tools/pmc_hbm.sh in
                  child processes - separate `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `WRITE_SIZE` passes over the same
                  launch, calibrated on a stream of known size as MI355X_MICROARCH.md prescribes; --no-live-traffic or
                  CACO_BENCH_LIVE_TRAFFIC=0 turns it off), else from the committed measurement named in `traffic_source`;
  stages        : the same per-launch-group timing for every stage (mel: HBM GB/s vs 8 TB/s);
  extra_configs : BASELINE configs[1] (audio tower only, batch 256) and configs[4] (AudioMAE stage-1 forward, batch 256),
                  measured in this run outside the timed region;
  cpu_baseline  : the CPU oracle (torch CPU ops, "port" of the reference's src/caco_torch path) timed on this box's host
                  cores on a bounded sample (batch 4; thread sweep, best + 1-thread figures, per-stage times), rank 0 at
                  N = 1 only.

CACO_BENCH_DRYRUN=1: no GPU, no library - the same control flow (warm-up, fences, timed loop, MAX all-reduce, teardown
order) AND the same step closure (make_step: encode_pairs -> dist.gather_packed -> similarity into the row block) on the
gloo backend with CPU stand-ins for the towers and the similarity kernel: tests/test_bench_dryrun.py runs it at world size 2.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

B_PER_GPU = 256
N_SAMPLES = 160000
SEQ = 500          # 496 valid patches padded to patches_seq_len = 500 (src/eval/eval_caco_torch.py:573)
SEQ_RUN = 496      # what the audio tower runs on: encode_audio drops the 4 all-padding positions (api.hip caco_encode_audio_ex)
TEXT_LEN = 32
H, I, P = 768, 3072, 256
PEAK_BF16_TFLOPS = 2500.0      # dense, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0
MAE_GFLOP_PER_CLIP = 111.0     # 12-layer encoder on 100 visible + 12-layer decoder on 496 patches
DRYRUN = os.environ.get("CACO_BENCH_DRYRUN", "0") not in ("", "0")


# algorithmic FLOPs (2 * MACs) per launch group at batch 256; SURVEY.md section 8d
def _flops(batch):
    M, Mt = batch * SEQ_RUN, batch * TEXT_LEN
    S_attn = 2 * 2 * SEQ_RUN * SEQ_RUN * H    # QK^T + PV per clip, all heads
    T_attn = 2 * 2 * TEXT_LEN * TEXT_LEN * H
    return {
        "audio.patch_embed": 2 * M * P * H, "audio.gemm_qkv": 2 * M * H * 3 * H,
        "audio.attention": batch * S_attn, "audio.gemm_out": 2 * M * H * H, "audio.gemm_fc1": 2 * M * H * I,
        "audio.gemm_fc2": 2 * M * I * H,
        "text.gemm_qkv": 2 * Mt * H * 3 * H, "text.attention": batch * T_attn,
        "text.gemm_out": 2 * Mt * H * H, "text.gemm_fc1": 2 * Mt * H * I, "text.gemm_fc2": 2 * Mt * I * H,
    }


def _audio_tower_flops(batch):
    f = _flops(batch)
    per_layer = f["audio.gemm_qkv"] + f["audio.attention"] + f["audio.gemm_out"] + f["audio.gemm_fc1"] + f["audio.gemm_fc2"]
    return f["audio.patch_embed"] + 12 * per_layer


def _bytes(batch):
    M = batch * SEQ_RUN
    return {
        "mel.patches": batch * (N_SAMPLES * 4 + 496 * 256 * 2),      # fp32 samples in + bf16 patches out
        "audio.ln": M * H * (4 + 2),                                  # fp32 residual in, bf16 operand out
        "audio.pos_embed": M * H * 8,
    }


def _make_inputs(batch, rank, device):
    from cacophony_amd import synth
    base = synth.make_waveforms(16, N_SAMPLES, start=100 + 16 * rank)       # 16 distinct structured clips per rank
    reps = (batch + 15) // 16
    gains = (0.35 + 0.65 * (np.arange(reps) + 1) / reps).astype(np.float32)
    wav = np.concatenate([base * g for g in gains], 0)[:batch]
    ids, mask = synth.make_captions(batch, TEXT_LEN, 50265, start=batch * rank)
    return (torch.from_numpy(wav).to(device), torch.from_numpy(ids).to(device), torch.from_numpy(mask).to(device))


def _measure_traffic_live(timeout_s: float = 300.0, tag: str = "bench_live_hbm"):
    """HBM / fabric bytes of ONE fc1 launch from the TCC counters, measured by child processes of this run: tools/pmc_hbm.sh =
    four `rocprofv3 --kernel-trace --pmc <one counter>` passes (FETCH_SIZE and WRITE_SIZE cannot share a pass; kernel-trace only, no
    other tracing domain) - two over a streaming probe of known size (the calibration MI355X_MICROARCH.md asks for: on gfx950
    FETCH_SIZE reports half of a wide coalesced read, WRITE_SIZE is uncalibrated) and two over tools/gemm_bench.py --only fc1, the
    same kernel, shape, tile order and library as the step's fc1 launch.  rocprofv3 serialises the dispatches it counts, so the
    figure is per launch.  Returns (record | None, reason): never raises, bounded by `timeout_s`, kills the whole process group
    on a timeout.  Not inside the timed region; this process idles on a synchronised device while the children run."""
    import shutil
    import signal
    import subprocess
    if os.environ.get("ROCP_TOOL_LIBRARIES"):
        return None, "this process itself runs under rocprofv3 (nested counter collection is not attempted)"
    if shutil.which("rocprofv3") is None and not os.path.exists("/opt/rocm/bin/rocprofv3"):
        return None, "rocprofv3 not found"
    out_dir = os.path.join(REPO, "gpurun_out", tag)
    rec_path = os.path.join(out_dir, "hbm_traffic.json")
    try:
        if os.path.exists(rec_path):
            os.remove(rec_path)
        env = dict(os.environ, PATH=os.pathsep.join([os.environ.get("PATH", ""), os.path.dirname(sys.executable), "/opt/rocm/bin"]))
        env.pop("CACO_BENCH_DRYRUN", None)
        t0 = time.perf_counter()
        proc = subprocess.Popen(["bash", os.path.join(REPO, "tools", "pmc_hbm.sh"), tag, "fc1", "256"], cwd=REPO, env=env,
                                stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, start_new_session=True)
        try:
            log, _ = proc.communicate(timeout=timeout_s)
        except subprocess.TimeoutExpired:
            os.killpg(proc.pid, signal.SIGKILL)
            proc.communicate()
            return None, f"tools/pmc_hbm.sh did not finish within {timeout_s:.0f} s (killed)"
        if not os.path.exists(rec_path):
            return None, f"tools/pmc_hbm.sh left no record (exit {proc.returncode}): {(log or '').strip().splitlines()[-1:]}"
        rec = json.load(open(rec_path))
        if not (rec.get("hbm_bytes", 0) > 0 and rec.get("dispatches", 0) >= 1):
            return None, f"counter record unusable: {rec}"
        rec["collect_s"] = round(time.perf_counter() - t0, 1)
        return rec, "measured by this run"
    except Exception as e:          # a side measurement: the bench line must survive anything here
        return None, f"live counter pass failed: {e!r}"


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _cpu_baseline(state, budget_s: float = 45.0):
    """Oracle (torch CPU ops) on a bounded sample of the same workload: batch 4, full-size model.  A batch of 4 cannot feed
    every core of a large host (round 1 timed it at torch's default of all 128 threads: 4x slower than at 8), so the thread
    count is swept and the best figure reported next to the 1-thread one; per-stage times at the best setting."""
    from cacophony_amd import config as Cfg
    from cacophony_amd import synth
    from oracle import caco_oracle as O
    b = 4
    ncores = os.cpu_count() or 1
    ref = O.CacoOracle(state, Cfg.default_audio_config(), Cfg.default_text_config(), Cfg.default_caco_config(), backend="torch")
    wav = synth.make_waveforms(b, N_SAMPLES, start=100)
    ids, mask = synth.make_captions(b, TEXT_LEN)

    def run(stages=None):
        with torch.no_grad():
            t0 = time.perf_counter()
            pb = O.prepare_audio_batch(wav, SEQ, backend="torch") if stages is not None else None
            t1 = time.perf_counter()
            if stages is not None:
                ea = ref.get_audio_embedding(pb["audio_patches"], pb["audio_time_inds"], pb["audio_freq_inds"], pb["audio_mask"],
                                             return_hidden_state=False, normalize=True)
            else:
                ea = ref.encode_audio(wav, SEQ)
            t2 = time.perf_counter()
            et = ref.encode_text(ids, mask)
            t3 = time.perf_counter()
            sim = O.similarity(ea, et)
            t4 = time.perf_counter()
            if stages is not None:
                stages.update(mel_ms=(t1 - t0) * 1e3, audio_ms=(t2 - t1) * 1e3, text_ms=(t3 - t2) * 1e3, sim_ms=(t4 - t3) * 1e3)
            return sim

    default_threads = torch.get_num_threads()
    # (all cores of a 256-thread host on a batch of 4 is pathological - measured 0.025 pairs/s - and would eat the time budget)
    sweep = sorted({t for t in (8, 16, 32, 64) if t <= ncores} | {min(8, ncores)})
    t_start = time.time()
    results = {}
    try:
        for th in sweep + [1]:
            if results and th != 1 and time.time() - t_start > budget_s * 0.6:
                continue
            torch.set_num_threads(th)
            if th != 1:
                run()                       # warm-up (the single-thread point is one cold run: it is ~10 s of CPU work)
            ts = []
            for _ in range(3 if th != 1 else 1):
                t0 = time.perf_counter()
                run()
                ts.append(time.perf_counter() - t0)
                if time.time() - t_start > budget_s:
                    break
            results[th] = float(np.median(ts))
        best = min((t for t in results if t != 1), key=lambda t: results[t]) if len(results) > (1 in results) else 1
        torch.set_num_threads(best)
        stages = {}
        run(stages)
        # the optional larger point of BASELINE.md 2: batch 32 feeds more threads than batch 4 does (one warm-up + one timed
        # run per thread count, ~2-4 s each)
        b32 = {}
        b_big = 32
        wav4, ids4, mask4 = wav, ids, mask
        wav = synth.make_waveforms(b_big, N_SAMPLES, start=200)
        ids, mask = synth.make_captions(b_big, TEXT_LEN)
        for th in sorted({t for t in (best, 2 * best, 64) if t <= ncores}):
            if time.time() - t_start > budget_s * 1.6:
                break
            torch.set_num_threads(th)
            run()
            t0 = time.perf_counter()
            run()
            b32[th] = time.perf_counter() - t0
        wav, ids, mask = wav4, ids4, mask4
    finally:
        torch.set_num_threads(default_threads)
    value, cores, sample_b, per_batch = b / results[best], best, b, results[best]
    if b32:
        th32 = min(b32, key=lambda t: b32[t])
        if b_big / b32[th32] > value:
            value, cores, sample_b, per_batch = b_big / b32[th32], th32, b_big, b32[th32]
    out = {"value": round(value, 3), "unit": "pairs/s", "cores": int(cores), "kind": "port",
           "sample": f"batch {sample_b} x 10 s clips + {TEXT_LEN}-token captions, full 12+12-layer model, oracle torch-CPU backend "
                     f"({per_batch * 1e3:.0f} ms per batch at {cores} threads); best of a thread sweep at batch 4 (median of 3 runs "
                     f"per thread count) and at batch 32 (one run per thread count)",
           "cpu_model": _cpu_model(), "host_cores": int(ncores),
           "thread_sweep_pairs_per_s": {str(t): round(b / v, 3) for t, v in sorted(results.items())},
           "batch32_thread_sweep_pairs_per_s": {str(t): round(b_big / v, 3) for t, v in sorted(b32.items())},
           "stages_ms_at_best": {k: round(v, 1) for k, v in stages.items()}}
    if 1 in results:
        out["one_thread_pairs_per_s"] = round(b / results[1], 3)
    return out


# ---- the control flow the driver's contract prescribes (also what the gloo dry run executes) ----------------------------
def timed_loop(step, steps, warmup, world, sync, tensor_device):
    """W untimed steps, then exactly K steps between two (synchronize, barrier, synchronize) fences; returns the MAX of the
    ranks' wall times."""
    import torch.distributed as dist

    def fence():
        sync()
        if world > 1:
            dist.barrier()
        sync()

    for _ in range(warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=tensor_device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    return elapsed


def make_step(model, wav, ids, mask, sim_out, similarity_fn, gather_fn, audio_streams=1, check_sizes=False):
    """THE step of the benchmark (also what the gloo dry run executes, with CPU stand-ins for `model` and `similarity_fn`):
    both towers write one packed [B, 2, P] buffer = the all-gather payload (text tower on a side stream next to the audio
    tower, one workspace per tower and stream inside the library), ONE all-gather, this rank's row block
    sim_out[B, W*B] = A_local . T_all^T read through the banks' row strides.  No other device work in between.
    `check_sizes` adds dist.gather_packed's shard-size check (one tiny all-reduce): off in the timed path, where every rank
    holds B_PER_GPU rows by construction."""
    def step():
        bank = model.encode_pairs(wav, ids, mask, SEQ, audio_streams=audio_streams, packed=True)
        allb = gather_fn(bank, check_sizes=check_sizes)
        return similarity_fn(bank[:, 0], allb[:, 1], 1.0, out=sim_out)
    return step


class _DryrunModel:
    """CPU stand-in with encode_pairs' contract (packed fp32 [B, 2, P] bank, rows L2-normalised, a function of the inputs
    only), so that the dry run drives the real step closure and the real exchange; `delay` makes ranks unequally fast."""

    def __init__(self, dim, delay):
        self.dim, self.delay = dim, delay

    def encode_pairs(self, wav, ids, mask, max_patches=None, audio_streams=1, packed=False):
        time.sleep(self.delay)
        B = wav.shape[0]
        k = torch.arange(self.dim, dtype=torch.float32)[None, :]
        ea = torch.cos(wav[:, :1] * (k + 1.0)) + wav[:, 1:2]
        et = torch.sin((ids[:, :1].float() + mask.sum(1, keepdim=True).float()) * 0.01 * (k + 1.0))
        bank = torch.stack([torch.nn.functional.normalize(ea, dim=1), torch.nn.functional.normalize(et, dim=1)], 1).contiguous()
        assert packed and bank.shape == (B, 2, self.dim)
        return bank


def _dryrun_similarity(a, t, scale=1.0, out=None):
    r = scale * a.double() @ t.double().T
    out[:, :r.shape[1]].copy_(r)
    return out


def _timeit(fn, n, sync):
    fn()
    sync()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    sync()
    return (time.perf_counter() - t0) / n * 1e3


SWITCH_NAMES = ("CACO_PINGPONG", "CACO_POS_FUSE", "CACO_POOL_FUSE", "CACO_ATTN_SMALL", "CACO_ATTN_ROWS", "CACO_W_NGROUP",
                "CACO_W8_MIN_TILES", "CACO_W4H_MAX_TILES")


def _lib_path():
    from cacophony_amd import _lib
    return _lib.LIB_PATH


def _lib_is_product():
    from cacophony_amd import _lib
    return os.path.realpath(_lib.LIB_PATH) == os.path.realpath(_lib.PRODUCT_LIB_PATH)


def _switches_in_force(lib):
    """What the library itself reports (caco_get_switch), plus the LayerNorm-fold default: an A/B line says what it measured."""
    if lib is None:
        return None
    d = {n: int(lib.caco_get_switch(n.encode())) for n in SWITCH_NAMES}
    d["CACO_LN_FOLD"] = int(lib.caco_set_ln_fold(-99))
    return d


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--profile-steps", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-configs", action="store_true")
    ap.add_argument("--no-live-traffic", action="store_true", help="do not collect the fc1 launch's TCC counters in child rocprofv3 passes after the timed region")
    ap.add_argument("--audio-streams", type=int, default=1, help="streams the clip batch is split over (1 = one audio stream; the text tower always runs on its own)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
        args.gpus = world
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")

    lib = model = None
    state = None
    check_sizes = os.environ.get("CACO_BENCH_CHECK_SIZES", "0") not in ("", "0")
    if DRYRUN:
        # the REAL step closure and the REAL exchange (dist.gather_packed on the gloo backend); only the towers and the
        # similarity kernel are CPU stand-ins.  CACO_BENCH_DRYRUN_UNEQUAL=1: rank 0 holds one row more than the others -
        # with CACO_BENCH_CHECK_SIZES=1 the step must refuse that on every rank instead of hanging in the collective.
        device = torch.device("cpu")
        if world > 1:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        sync = lambda: None
        from cacophony_amd.dist import gather_packed
        b_here = B_PER_GPU + (1 if (rank == 0 and os.environ.get("CACO_BENCH_DRYRUN_UNEQUAL", "0") not in ("", "0")) else 0)
        g = torch.Generator().manual_seed(1000 + rank)
        wav = torch.rand(b_here, 4, generator=g)
        ids = torch.randint(3, 50265, (b_here, TEXT_LEN), generator=g)
        mask = (torch.arange(TEXT_LEN)[None, :] < torch.randint(8, TEXT_LEN + 1, (b_here, 1), generator=g)).long()
        sim_out = torch.zeros(b_here, world * B_PER_GPU, dtype=torch.float32)
        model = _DryrunModel(8, 0.002 * (1 + rank))                # ranks of unequal speed: the MAX must win
        step = make_step(model, wav, ids, mask, sim_out, _dryrun_similarity, gather_packed, args.audio_streams, check_sizes)
        finite = True
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a ROCm GPU: the HIP path has no CPU fallback")
        device = torch.device(f"cuda:{local_rank}")
        torch.cuda.set_device(device)
        if world > 1:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        from cacophony_amd import _lib, config as Cfg, synth
        from cacophony_amd.dist import gather_packed
        from cacophony_amd.model import create_caco_model, similarity

        lib = _lib.load()          # refuses any library but cacophony_amd/libcaco_hip.so unless CACO_ALLOW_VARIANT_LIB=1
        state = synth.make_caco_state(Cfg.default_audio_config(), Cfg.default_text_config(), Cfg.default_caco_config())
        model = create_caco_model(device=device).load_state_dict(state)
        wav, ids, mask = _make_inputs(B_PER_GPU, rank, device)
        sim_out = torch.empty(B_PER_GPU, world * B_PER_GPU, dtype=torch.float32, device=device)
        sync = torch.cuda.synchronize
        step = make_step(model, wav, ids, mask, sim_out, similarity, gather_packed, args.audio_streams, check_sizes)

    elapsed = timed_loop(step, args.steps, args.warmup, world, sync, device)
    ms_per_step = elapsed / args.steps * 1e3
    value = world * B_PER_GPU * args.steps / elapsed
    dry_err = None
    if not DRYRUN:
        finite = bool(torch.isfinite(sim_out).all().item())
    else:
        # what the closure left in sim_out must be this rank's row block of the global matrix, in rank order
        bank = model.encode_pairs(wav, ids, mask, SEQ, packed=True)
        parts = [torch.empty_like(bank) for _ in range(world)] if world > 1 else [bank]
        if world > 1:
            dist.all_gather(parts, bank)
        ref = bank[:, 0].double() @ torch.cat([p[:, 1] for p in parts], 0).double().T
        dry_err = float((sim_out.double() - ref).abs().max())

    # ---- per-launch-group timing with HIP events on the launch stream (separate, un-timed pass) -----------------
    stages, roofline, extra = {}, None, None
    if rank == 0 and not DRYRUN:
        try:
            def serial_step():      # per-launch timings need one stream: overlapped launches stretch each other's event pairs
                ea = model.encode_audio(wav, SEQ)
                et = model.encode_text(ids, mask, check_ids=False)
                return similarity(ea, et, 1.0, out=sim_out[:, :B_PER_GPU])

            lib.caco_profile_enable(1)
            for _ in range(max(1, args.profile_steps)):
                serial_step()
            torch.cuda.synchronize()
            buf = C.create_string_buffer(1 << 16)
            lib.caco_profile_report(buf, len(buf))
            lib.caco_profile_enable(0)
            prof = json.loads(buf.value.decode())
            fl, by = _flops(B_PER_GPU), _bytes(B_PER_GPU)
            psteps = max(1, args.profile_steps)
            for name, rec in sorted(prof.items()):
                avg_ms = rec["ms"] / rec["n"]
                ent = {"ms_per_step": round(rec["ms"] / psteps, 4), "launches_per_step": rec["n"] // psteps, "avg_launch_ms": round(avg_ms, 5)}
                if name in fl:
                    tf = fl[name] / (avg_ms * 1e-3) / 1e12
                    ent.update(bound="mfma", achieved_tflops=round(tf, 1), frac=round(tf / PEAK_BF16_TFLOPS, 4))
                elif name in by:
                    gbs = by[name] / (avg_ms * 1e-3) / 1e9
                    ent.update(bound="hbm", achieved_gbs=round(gbs, 1), frac=round(gbs / PEAK_HBM_GBS, 4))
                stages[name] = ent
            dom = stages.get("audio.gemm_fc1")
            # HBM bytes of one fc1 launch: a committed TCC-counter measurement (tools/pmc_hbm.sh), quoted ONLY when it was taken
            # with the tile order this run uses (its "w_ngroup" field; a file without the field is of unknown order and never quoted).  Anything else would pair this run's time with another schedule's bytes.
            traffic, traffic_source, traffic_rec = None, None, None
            live_reason = "off (--no-live-traffic / CACO_BENCH_LIVE_TRAFFIC=0)"
            if world == 1 and not args.no_live_traffic and os.environ.get("CACO_BENCH_LIVE_TRAFFIC", "1") not in ("", "0"):
                torch.cuda.synchronize()
                traffic_rec, live_reason = _measure_traffic_live()
            try:
                import glob
                ngroup_now = int(lib.caco_get_switch(b"CACO_W_NGROUP"))
                if traffic_rec is not None and int(traffic_rec.get("w_ngroup", -99)) == ngroup_now:
                    traffic = int(traffic_rec["hbm_bytes"])
                    traffic_source = (f"measured by this run after the timed region ({traffic_rec['collect_s']} s): tools/pmc_hbm.sh, separate rocprofv3 "
                                      f"--kernel-trace --pmc FETCH_SIZE / WRITE_SIZE passes over {traffic_rec['dispatches']} fc1 launches of this library with "
                                      f"the tile order in force (CACO_W_NGROUP = {ngroup_now}), calibrated on a stream of known size "
                                      f"({traffic_rec['bytes_per_fetch_unit']:.0f} B per FETCH_SIZE unit, {traffic_rec['bytes_per_write_unit']:.0f} B per WRITE_SIZE "
                                      f"unit); read {traffic_rec['hbm_read_bytes'] / 1e9:.3f} GB + write {traffic_rec['hbm_write_bytes'] / 1e9:.3f} GB per launch")
                for cand in ([] if traffic is not None else sorted(glob.glob(os.path.join(REPO, "profiles", "*", "hbm_traffic.json")), reverse=True)):
                    rec = json.load(open(cand))
                    if "w_ngroup" in rec and int(rec["w_ngroup"]) == ngroup_now:      # no field = tile order unknown: never quoted
                        traffic = int(rec["hbm_bytes"])
                        traffic_source = (os.path.relpath(cand, REPO) + ": committed TCC FETCH_SIZE + WRITE_SIZE measurement of one fc1 "
                                          f"launch (tools/pmc_hbm.sh, calibrated on a 1 GiB stream); NOT measured in this run (live pass: {live_reason})")
                        break
                else:
                    if traffic is None:
                        traffic_source = (f"live counter pass: {live_reason}; no committed hbm_traffic.json was measured with the tile order in "
                                          f"force (CACO_W_NGROUP = {ngroup_now})")
            except Exception:
                traffic = None
            # algorithmic bytes of one fc1 launch: A [M,768] bf16 in + W [3072,768] bf16 in + out [M,3072] bf16
            alg_bytes = B_PER_GPU * SEQ_RUN * H * 2 + I * H * 2 + B_PER_GPU * SEQ_RUN * I * 2
            if dom:
                roofline = {"kernel": "gemm_bf16_w8_kernel<EPI_BF16, SiLU> (audio MLP fc1: [126976,768] x [3072,768]^T)",
                            "bound": "mfma", "achieved": dom["achieved_tflops"], "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                            "frac": dom["frac"], "traffic": traffic, "traffic_source": traffic_source,
                            "algorithmic_flops_per_launch": fl["audio.gemm_fc1"], "avg_launch_ms": dom["avg_launch_ms"],
                            "algorithmic_bytes_per_launch": alg_bytes,
                            "traffic_over_algorithmic": round(traffic / alg_bytes, 3) if traffic else None,
                            "note": "peak = the nominal dense bf16 MFMA figure of MI355X_MICROARCH.md; what bounds this kernel in "
                                    "practice (package power cap, vendor-library calibration) is DESIGN.md section 4 with the "
                                    "measurements under profiles/"}

            if not args.no_extra_configs:
                extra = {}
                ms = _timeit(lambda: model.encode_audio(wav, SEQ), 5, torch.cuda.synchronize)
                tf = _audio_tower_flops(B_PER_GPU) / (ms * 1e-3) / 1e12
                extra["configs[1] audio encoder only (mel + ViT + pooler), batch 256, bf16"] = {
                    "ms_per_batch": round(ms, 3), "clips_per_s": round(B_PER_GPU / ms * 1e3, 1), "achieved_tflops": round(tf, 1),
                    "frac_of_mfma_peak": round(tf / PEAK_BF16_TFLOPS, 4)}
                try:
                    from cacophony_amd import config as Cfg, synth
                    from cacophony_amd.model import AudioMAE
                    enc = Cfg.default_audio_config()
                    mae = AudioMAE(Cfg.AudioMAEConfig(enc, enc), device=device).load_state_dict(synth.make_audiomae_state(enc, enc))
                    V, R = 100, 396
                    g = torch.Generator().manual_seed(0)
                    x = torch.randn(B_PER_GPU, V, 256, generator=g).to(device)
                    perm = torch.stack([torch.randperm(496, generator=g) for _ in range(B_PER_GPU)])
                    vis, res = perm[:, :V].sort(1).values, perm[:, V:].sort(1).values
                    f = lambda t: t.float().to(device)
                    margs = (x, f(torch.ones(B_PER_GPU, V)), f(vis // 8), f(vis % 8), f(res // 8), f(res % 8), f(torch.ones(B_PER_GPU, R)))
                    ms = _timeit(lambda: mae.forward(*margs), 3, torch.cuda.synchronize)
                    tf = MAE_GFLOP_PER_CLIP * B_PER_GPU / ms
                    extra["configs[4] AudioMAE stage-1 forward (100 visible + 396 restored patches, 12 + 12 layers), batch 256, bf16"] = {
                        "ms_per_batch": round(ms, 3), "clips_per_s": round(B_PER_GPU / ms * 1e3, 1), "achieved_tflops": round(tf, 1),
                        "frac_of_mfma_peak": round(tf / PEAK_BF16_TFLOPS, 4)}
                    del mae
                except Exception as e:          # the headline number must not depend on the side measurement
                    extra["configs[4] AudioMAE stage-1 forward"] = {"error": repr(e)}

        except Exception as e:      # the timed headline above must not be lost to the un-timed profile pass
            roofline = {"error": repr(e)}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not DRYRUN:
        try:
            cpu = _cpu_baseline(state)
        except Exception as e:          # the headline number must not be lost to the side measurement (host memory, thread limits ...)
            cpu = {"value": None, "unit": "pairs/s", "cores": 0, "kind": "port", "sample": "failed", "error": repr(e)}

    if rank == 0:
        out = {
            "metric": "audio-text pairs/sec embedded+scored, 10s@16kHz, batch 256, 1/2/4/8 GPU",
            "value": round(value, 2), "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic" if not DRYRUN else "none (CACO_BENCH_DRYRUN: control flow only, not a measurement)",
            "config": {"workload": "BASELINE configs[2] per GPU (batch=256 audio+text encoders + similarity GEMM, bf16 MFMA, "
                                   "fp32 accumulate/residual/softmax/LayerNorm), sharded as configs[3] when n_gpus > 1 "
                                   "(one RCCL all-gather of the packed [256,2,768] fp32 banks, local [256, N*256] row block)",
                       "global_batch": world * B_PER_GPU, "clip": "10 s @ 16 kHz (160000 samples, 500 patches, 496 valid)",
                       "caption_tokens": TEXT_LEN, "parallelism": f"dp{world}", "weights": "seeded random init (no checkpoint offline)",
                       # which library was timed: a headline line and an A/B line of a variant build are told apart after the fact
                       "lib_path": os.path.relpath(os.path.realpath(_lib_path()), REPO) if lib is not None else None,
                       "lib_is_product": _lib_is_product() if lib is not None else None,
                       "caco_version": lib.caco_version().decode() if lib is not None else None,
                       "gemm_tile": int(lib.caco_set_gemm_tile(0)) if lib is not None else None,
                       # run-time switches in force: an A/B line says what it measured
                       "switches": _switches_in_force(lib)},
            "roofline": roofline, "cpu_baseline": cpu, "extra_configs": extra, "stages": stages, "outputs_finite": finite,
        }
        if DRYRUN:
            out["dryrun_row_block_max_err"] = dry_err
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()          # rank 0's un-timed passes are over: every rank tears the group down together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
