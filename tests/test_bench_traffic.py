"""bench.py's live counter pass (roofline.traffic): the plumbing around tools/pmc_hbm.sh without a GPU.  `rocprofv3` and `python` are
shadowed by the logging fakes of tools/session_fakes.sh (every counter reads 1000), so what is checked is the control flow: the four
separate --pmc passes, the record's fields, the timeout with a process-group kill, the refusal to nest under rocprofv3 - never a number."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _call(tmp_path, body, path_first, **env_extra):
    code = (f"import sys, json; sys.path.insert(0, {REPO!r})\nimport bench\n" + body)
    env = dict(os.environ, PATH=f"{path_first}{os.pathsep}{os.environ['PATH']}", **env_extra)
    env.pop("ROCP_TOOL_LIBRARIES", None) if "ROCP_TOOL_LIBRARIES" not in env_extra else None
    r = subprocess.run([sys.executable, "-c", code], cwd=REPO, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_live_traffic_pass_runs_four_separate_counter_passes(tmp_path):
    fakes, log = tmp_path / "fakes", tmp_path / "commands.log"
    fakes.mkdir()
    subprocess.run(["bash", "-c", f"source tools/session_fakes.sh {fakes} {log}"], cwd=REPO, check=True)
    tag = f"test_live_hbm_{os.getpid()}"
    try:
        out = _call(tmp_path, f"rec, why = bench._measure_traffic_live(60.0, {tag!r}); print(json.dumps([rec, why]))", fakes)
        rec, why = out
        assert why == "measured by this run" and rec is not None
        for key in ("hbm_bytes", "hbm_read_bytes", "hbm_write_bytes", "bytes_per_fetch_unit", "bytes_per_write_unit", "dispatches", "w_ngroup", "collect_s"):
            assert key in rec, key
        assert rec["w_ngroup"] == -1 and rec["shape"] == "fc1" and rec["dispatches"] >= 1
        # every fake counter reads 1000 and the calibration stream averages 432 MiB per launch: the arithmetic of pmc_hbm.sh, not a measurement
        assert rec["hbm_bytes"] == 2 * 432 * (1 << 20)
        cmds = [ln for ln in open(log) if ln.startswith("rocprofv3 ")]
        assert len(cmds) == 4
        assert sum("--pmc FETCH_SIZE" in c for c in cmds) == 2 and sum("--pmc WRITE_SIZE" in c for c in cmds) == 2
        assert all("--kernel-trace" in c for c in cmds)
        assert not any(d in c for c in cmds for d in ("--sys-trace", "--hip-trace", "--hsa-trace", "--runtime-trace", "--memory-copy-trace",
                                                      "--marker-trace", "--scratch-memory-trace", " -s ", " -r "))
        assert sum("tools/gemm_bench.py --iters 3 --only fc1 --tile 256" in c for c in cmds) == 2
    finally:
        subprocess.run(["rm", "-rf", os.path.join(REPO, "gpurun_out", tag)])


def test_live_traffic_pass_is_bounded_and_never_raises(tmp_path):
    slow = tmp_path / "slow"
    slow.mkdir()
    (slow / "rocprofv3").write_text("#!/bin/bash\nsleep 60\n")
    (slow / "rocprofv3").chmod(0o755)
    tag = f"test_live_hbm_slow_{os.getpid()}"
    try:
        rec, why = _call(tmp_path, f"import time; t0 = time.time(); rec, why = bench._measure_traffic_live(2.0, {tag!r}); "
                                   "assert time.time() - t0 < 30; print(json.dumps([rec, why]))", slow)
        assert rec is None and "did not finish" in why
        left = subprocess.run(["pgrep", "-f", f"pmc_hbm.sh {tag}"], capture_output=True, text=True).stdout.split()
        assert not left, f"the timed-out counter pass left processes behind: {left}"
        # under rocprofv3 itself (tools/profile_bench.sh, tools/pmc_run.sh run bench.py that way) no nested collection is attempted
        rec, why = _call(tmp_path, f"rec, why = bench._measure_traffic_live(2.0, {tag!r}); print(json.dumps([rec, why]))", slow,
                         ROCP_TOOL_LIBRARIES="/opt/rocm/lib/rocprofiler-sdk/librocprofiler-sdk-tool.so")
        assert rec is None and "under rocprofv3" in why
    finally:
        subprocess.run(["rm", "-rf", os.path.join(REPO, "gpurun_out", tag)])
