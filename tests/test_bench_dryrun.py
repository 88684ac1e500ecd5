"""bench.py's multi-process control flow, executed before any 8-GPU driver run: warm-up, the (synchronize, barrier,
synchronize) fences, the timed loop, the MAX all-reduce of the ranks' wall times, rank 0's single JSON line and the
teardown order - on the gloo backend at world size 2 with a stand-in step (CACO_BENCH_DRYRUN=1: no GPU, no library).
The per-rank step there is `gather_packed` on a [256, 2, 8] bank, i.e. the real exchange of the data-parallel path."""
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


def _run(cmd):
    env = dict(os.environ, CACO_BENCH_DRYRUN="1", OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=300)
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


def test_bench_control_flow_single_process():
    out = _run([sys.executable, "bench.py", "--steps", "3", "--warmup", "1"])
    assert out["n_gpus"] == 1 and out["ms_per_step"] >= 1.9 and out["cpu_baseline"] is None
