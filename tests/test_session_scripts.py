"""The GPU session scripts (tools/gpu_session.sh and what it calls) are the only way measurements enter this repo, and they run
where a typo costs GPU minutes.  tools/session_dryrun.sh executes every part of them in a throw-away copy of the tree with `python`,
`rocprofv3` and `timeout` shadowed by logging fakes, then checks every logged python command line: the script exists and takes the
flags it is given.  No GPU, nothing measured."""
import os
import subprocess

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_part_of_the_gpu_session_dry_runs_clean():
    r = subprocess.run(["bash", os.path.join(REPO, "tools", "session_dryrun.sh")], cwd=REPO, capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0 and "PASS" in r.stdout.splitlines()[-1], tail
    assert "all scripts and flags exist" in r.stdout, tail
    for part in ("truth", "ab", "variants", "pmc", "bisect"):
        assert f"== dry run: gpu_session.sh {part}" in r.stdout
    # the records the later steps depend on were produced by the scripts' own plumbing (fabricated counters, real csv readers)
    for name in ("pytest_gpu.txt", "smoke.txt", "bench.json", "kernel_stats_default.csv", "hbm_traffic.json", "pmc_fc1_tcc1.csv",
                 "ab_switches.json", "ab_variants.json", "predictions_vs_measured.txt", "pytest_arm_r2.txt"):
        assert name in r.stdout, name


def test_shell_scripts_parse():
    tools = os.path.join(REPO, "tools")
    for f in sorted(os.listdir(tools)):
        if f.endswith(".sh"):
            r = subprocess.run(["bash", "-n", os.path.join(tools, f)], capture_output=True, text=True)
            assert r.returncode == 0, (f, r.stderr)
