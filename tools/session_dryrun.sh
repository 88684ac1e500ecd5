#!/bin/bash
# Dry run of tools/gpu_session.sh WITHOUT a GPU (round 6): catches script errors before they cost GPU minutes.
#   bash tools/session_dryrun.sh [parts...]        default: quick truth ab variants pmc bisect
# How: `python`, `rocprofv3` and `timeout` are shadowed by logging fakes (a temp directory first on PATH); the fake rocprofv3
# fabricates the csv files the scripts look for and then runs its command through the fake python.  Afterwards every logged python
# command line is checked: the script exists, and every --flag it was given is one the script's own --help lists.
# Nothing here measures anything; the only outputs are the log and a PASS / FAIL line (exit code 1 on FAIL).
set -u
cd "$(dirname "$0")/.."
PARTS=${*:-quick truth ab variants pmc bisect}
T=$(mktemp -d); LOG=$T/commands.log; : > "$LOG"
# the scripts write under gpurun_out/ of the tree they run in: run them in a throw-away copy, so that no fabricated record
# (an hbm_traffic.json made of the fake counters, say) can ever sit next to real ones
mkdir "$T/repo"
tar c --exclude=.git --exclude=gpurun_out --exclude=tools/wavesim/_gen --exclude='*.o' --exclude=__pycache__ . | tar x -C "$T/repo"
cd "$T/repo"
source tools/session_fakes.sh "$T" "$LOG"
export OUT=$T/out          # gpu_session.sh's own record directory
FAIL=0
for part in $PARTS; do
  echo "== dry run: gpu_session.sh $part"
  (cd "$PWD" && PATH="$T:$PATH" bash tools/gpu_session.sh "$part") > "$T/$part.stdout" 2> "$T/$part.stderr" || { echo "FAIL: gpu_session.sh $part exited $?"; FAIL=1; }
  grep -n "No such file\|command not found\|syntax error\|unbound variable\|Traceback\|not found" "$T/$part.stdout" "$T/$part.stderr" && FAIL=1
  grep -q "session done" "$T/$part.stdout" || { echo "FAIL: $part did not reach 'session done'"; FAIL=1; }
done
grep -n "REFUSED-BY-GPURUN" "$LOG" && FAIL=1
echo "== files the session left under \$OUT"; (cd "$OUT" && ls | tr '\n' ' '); echo
echo "== checking $(grep -c '^python ' "$LOG") python command lines"
"$REAL_PY" - "$LOG" <<'PY' || FAIL=1
import os, re, shlex, subprocess, sys
bad, seen = 0, {}
for line in open(sys.argv[1]):
    if not line.startswith("python "):
        continue
    argv = shlex.split(line)[1:]
    if not argv or argv[0] in ("-", "-c"):
        continue
    if argv[0] == "-m":
        if argv[1] == "pytest":
            for a in argv[2:]:
                if a.startswith("tests") and not os.path.exists(a.split("::")[0]):
                    print("FAIL: pytest path does not exist:", a); bad += 1
        continue
    script = argv[0]
    if not os.path.exists(script):
        print("FAIL: no such script:", script, "<-", line.strip()); bad += 1
        continue
    flags = [a.split("=")[0] for a in argv[1:] if a.startswith("--")]
    if not flags:
        continue
    if script not in seen:
        if script == "__graft_entry__.py":
            seen[script] = None
        else:
            r = subprocess.run([sys.executable, script, "--help"], capture_output=True, text=True, timeout=300,
                               env=dict(os.environ, CACO_BENCH_DRYRUN="1"))
            seen[script] = r.stdout + r.stderr if r.returncode == 0 else None
            if r.returncode != 0:
                print(f"note: {script} --help exits {r.returncode} (flags not checked): {(r.stderr or r.stdout).strip().splitlines()[-1:]}" )
    if seen[script] is None:
        continue
    for f in flags:
        if not re.search(r"(?<![\w-])" + re.escape(f) + r"(?![\w-])", seen[script]):
            print(f"FAIL: {script} has no flag {f}  <- {line.strip()}"); bad += 1
print("python command lines:", "all scripts and flags exist" if not bad else f"{bad} problem(s)")
sys.exit(1 if bad else 0)
PY
[ $FAIL = 0 ] && { echo "PASS ($(grep -c . "$LOG") commands logged)"; cd /; rm -rf "$T"; } || { echo "FAIL (log: $LOG, outputs: $T)"; exit 1; }
