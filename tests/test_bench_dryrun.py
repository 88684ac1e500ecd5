"""bench.py's multi-process control flow, executed before any 8-GPU driver run: warm-up, the (synchronize, barrier,
synchronize) fences, the timed loop, the MAX all-reduce of the ranks' wall times, rank 0's single JSON line and the
teardown order - on the gloo backend at world size 2 and at the target's world size 8 (CACO_BENCH_DRYRUN=1: no GPU, no library).
The per-rank step is bench.make_step itself - encode_pairs(packed) -> dist.gather_packed -> similarity into this rank's
row block - with CPU stand-ins for the towers and the similarity kernel only: the exchange, the strided views and the row
block placement are the real ones, and the row block is checked against an independent all-gather."""
import json
import os
import socket
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(cmd, expect_failure=False, **extra_env):
    env = dict(os.environ, CACO_BENCH_DRYRUN="1", OMP_NUM_THREADS="1", **extra_env)
    r = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=300)
    if expect_failure:
        return r
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout          # ONE JSON line, from rank 0 only
    return json.loads(lines[0])


def test_bench_control_flow_two_ranks_gloo():
    out = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--steps", "4", "--warmup", "2"])
    assert out["n_gpus"] == 2 and out["steps"] == 4 and out["warmup"] == 2 and out["scaling"] == "weak"
    assert out["config"]["global_batch"] == 512 and out["config"]["parallelism"] == "dp2"
    # rank 1's step sleeps 4 ms, rank 0's 2 ms: the reported time must be the slower rank's (MAX over ranks)
    assert out["ms_per_step"] >= 3.9
    assert abs(out["value"] - 2 * 256 / (out["ms_per_step"] * 1e-3)) / out["value"] < 1e-3
    assert out["data"].startswith("none")      # a dry run can never be mistaken for a measurement
    assert out["dryrun_row_block_max_err"] < 1e-6      # the step closure left rank 0's [256, 512] row block in place


def test_bench_control_flow_eight_ranks_gloo():
    """The driver's own N = 8 command line (SCALE_rNN.json) with gloo in RCCL's place: eight ranks of unequal speed, one
    packed all-gather per step, rank 0's [256, 2048] row block in rank order, ONE JSON line, clean teardown."""
    out = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                "--master-port", str(_free_port()), "bench.py", "--gpus", "8", "--steps", "3", "--warmup", "1"],
               CACO_BENCH_CHECK_SIZES="1")
    assert out["n_gpus"] == 8 and out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["config"]["global_batch"] == 2048 and out["config"]["parallelism"] == "dp8"
    assert out["ms_per_step"] >= 15.9          # the slowest rank (rank 7 sleeps 16 ms per step) sets the time
    assert abs(out["value"] - 8 * 256 / (out["ms_per_step"] * 1e-3)) / out["value"] < 1e-3      # whole-job pairs/s
    assert out["data"].startswith("none") and out["cpu_baseline"] is None
    assert out["dryrun_row_block_max_err"] < 1e-6


def test_bench_step_closure_refuses_unequal_shards_on_every_rank():
    """Rank 0 holds 257 rows, rank 1 holds 256: with the shard-size check on, bench.py's step (make_step ->
    dist.gather_packed(check_sizes=True)) must raise on BOTH ranks - no hang inside all_gather_into_tensor, no corrupted
    row block - and torchrun must report the failure."""
    r = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
              "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1"],
             expect_failure=True, CACO_BENCH_DRYRUN_UNEQUAL="1", CACO_BENCH_CHECK_SIZES="1")
    assert r.returncode != 0
    assert r.stderr.count("pad the shards to one size") >= 2, r.stderr[-3000:]      # one ValueError per rank
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]           # and no result line


def test_bench_step_closure_with_size_check_on_equal_shards():
    out = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1"],
               CACO_BENCH_CHECK_SIZES="1")
    assert out["n_gpus"] == 2 and out["dryrun_row_block_max_err"] < 1e-6


def test_bench_control_flow_single_process():
    out = _run([sys.executable, "bench.py", "--steps", "3", "--warmup", "1"])
    assert out["n_gpus"] == 1 and out["ms_per_step"] >= 1.9 and out["cpu_baseline"] is None


def test_bench_real_step_and_profile_pass_on_the_simulator():
    """Not the dry run: bench.py's real branch - library load, create_caco_model, make_step over the kernels, the per-launch
    profile pass, the roofline object - on tools/wavesim (tests/bench_on_sim.py: 2 pairs, one layer per tower).  Checks the
    contract of the printed line; every number in it is simulator wall-clock."""
    import pytest
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import simlib
    if not simlib.available():
        pytest.skip("no host clang++ for tools/wavesim")
    env = {k: v for k, v in os.environ.items() if k != "CACO_BENCH_DRYRUN"}
    r = subprocess.run([sys.executable, os.path.join(REPO, "tests", "bench_on_sim.py")], cwd=REPO, env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in out, key
    assert out["steps"] == 2 and out["warmup"] == 1 and out["n_gpus"] == 1 and out["outputs_finite"] is True
    assert out["unit"] == "pairs/s" and out["dtype"] == "bf16" and out["vs_baseline"] is None and "workload" in out["config"]
    rf = out["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "avg_launch_ms", "note"):
        assert key in rf, key
    assert rf["bound"] == "mfma" and rf["peak"] == 2500.0 and rf["avg_launch_ms"] > 0
    st = out["stages"]
    for name in ("mel.patches", "audio.patch_embed", "audio.gemm_qkv", "audio.attention", "audio.gemm_out", "audio.gemm_fc1",
                 "audio.gemm_fc2", "audio.ln", "text.attention", "text.gemm_fc1", "similarity"):
        assert name in st and st[name]["launches_per_step"] >= 1, name
