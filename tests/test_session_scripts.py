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
    for part in ("quick", "truth", "ab", "variants", "pmc", "bisect"):
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


def test_hardware_history_lists_in_conftest_name_existing_gpu_cases():
    """tests/conftest.py orders the `-m gpu` cases by hand-maintained name lists (seen on hardware in round 1 / round 2, golden
    model cases).  A renamed test would silently drop to 'never met an MI355X': every listed name must be a collected gpu case."""
    import re
    import sys
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import conftest
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(REPO, "tests"), "-m", "gpu", "--collect-only", "-q"], cwd=REPO,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    funcs = {re.split(r"[\[ ]", ln.split("::")[-1])[0] for ln in r.stdout.splitlines() if "::" in ln}
    assert len(funcs) > 50, len(funcs)          # 61 distinct gpu test functions (183 parametrised cases) at round 6
    for name, group in (("_ON_HARDWARE_R1", conftest._ON_HARDWARE_R1), ("_ON_HARDWARE_R2", conftest._ON_HARDWARE_R2),
                        ("_GOLDEN_MODEL_CASES", conftest._GOLDEN_MODEL_CASES)):
        missing = sorted(set(group) - funcs)
        assert not missing, f"tests/conftest.py {name} lists cases that no longer exist: {missing}"
    for f in conftest._F_ROW_FILES:
        assert os.path.exists(os.path.join(REPO, "tests", f)), f
