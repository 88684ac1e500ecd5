#!/bin/bash
# Dry run of tools/gpu_session.sh WITHOUT a GPU (round 6): catches script errors before they cost GPU minutes.
#   bash tools/session_dryrun.sh [parts...]        default: truth ab variants pmc bisect
# How: `python`, `rocprofv3` and `timeout` are shadowed by logging fakes (a temp directory first on PATH); the fake rocprofv3
# fabricates the csv files the scripts look for and then runs its command through the fake python.  Afterwards every logged python
# command line is checked: the script exists, and every --flag it was given is one the script's own --help lists.
# Nothing here measures anything; the only outputs are the log and a PASS / FAIL line (exit code 1 on FAIL).
set -u
cd "$(dirname "$0")/.."
PARTS=${*:-truth ab variants pmc bisect}
T=$(mktemp -d); LOG=$T/commands.log; : > "$LOG"
# the scripts write under gpurun_out/ of the tree they run in: run them in a throw-away copy, so that no fabricated record
# (an hbm_traffic.json made of the fake counters, say) can ever sit next to real ones
mkdir "$T/repo"
tar c --exclude=.git --exclude=gpurun_out --exclude=tools/wavesim/_gen --exclude='*.o' --exclude=__pycache__ . | tar x -C "$T/repo"
cd "$T/repo"
REAL_PY=$(command -v python)
cat > "$T/python" <<FAKE
#!/bin/bash
echo "python \$*" >> "$LOG"
case "\$1" in
  -)  exec "$REAL_PY" "\$@" ;;                                  # inline scripts (pmc_hbm.sh's summary) run for real on the fabricated csv files
  tools/summarize_*.py|tools/check_predictions.py) exec "$REAL_PY" "\$@" ;;      # pure record readers: run for real
esac
# a bench line for the scripts that read one; a minimal record at --out for the A/B tools' readers
A=("\$@"); for ((i = 0; i < \${#A[@]}; i++)); do [ "\${A[i]}" = --out ] && echo '{"rows": [{"variant": "default", "stages_ms": {}, "delta_mean": 0.0, "verdict": "-"}]}' > "\${A[i+1]}"; done
case "\$*" in *bench.py*) echo '{"metric": "dry run", "value": 0, "ms_per_step": 0, "roofline": {}, "config": {}}' ;; esac
exit 0
FAKE
cat > "$T/timeout" <<FAKE
#!/bin/bash
shift; exec "\$@"
FAKE
cat > "$T/rocprofv3" <<FAKE
#!/bin/bash
echo "rocprofv3 \$*" >> "$LOG"
D=""; PMC=""; A=("\$@")
for ((i = 0; i < \${#A[@]}; i++)); do
  [ "\${A[i]}" = -d ] && D=\${A[i+1]}
  if [ "\${A[i]}" = --pmc ]; then j=\$((i + 1)); while [ \$j -lt \${#A[@]} ] && [[ "\${A[j]}" != -* ]]; do PMC="\$PMC \${A[j]}"; j=\$((j + 1)); done; fi
  if [ "\${A[i]}" = -- ]; then CMD=("\${A[@]:i+1}"); fi
done
case "\$*" in *--sys-trace*|*--runtime-trace*|*--hip-trace*|*--hsa-trace*|*--memory-copy-trace*|*--marker-trace*|*--scratch-memory-trace*)
  [ -n "\$PMC" ] && echo "REFUSED-BY-GPURUN: --pmc combined with a tracing domain: \$*" >> "$LOG" ;; esac
mkdir -p "\$D/host/1"
printf '"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","StdDev"\n"void caco::gemm_bf16_w8_kernel<0,1,1>(caco::W8Args)",12,1000,83.3,100.0,80,90,1.0\n' > "\$D/host/1/p_kernel_stats.csv"
cp "\$D/host/1/p_kernel_stats.csv" "\$D/host/1/bench_kernel_stats.csv"
printf '"Kind","Agent_Id","Queue_Id","Stream_Id","Thread_Id","Dispatch_Id","Kernel_Id","Kernel_Name","Correlation_Id","Start_Timestamp","End_Timestamp","Private_Segment_Size","Group_Segment_Size","Workgroup_Size_X","Workgroup_Size_Y","Workgroup_Size_Z","Grid_Size_X","Grid_Size_Y","Grid_Size_Z"\n"KERNEL_DISPATCH",1,1,1,1,1,1,"void caco::gemm_bf16_w8_kernel<0,1,1>(caco::W8Args)",1,1000,2000,0,0,512,1,1,131072,1,1\n' > "\$D/host/1/p_kernel_trace.csv"
{ printf '"Correlation_Id","Dispatch_Id","Agent_Id","Queue_Id","Process_Id","Thread_Id","Grid_Size","Kernel_Id","Kernel_Name","Workgroup_Size","LDS_Block_Size","Scratch_Size","VGPR_Count","Accum_VGPR_Count","SGPR_Count","Counter_Name","Counter_Value","Start_Timestamp","End_Timestamp"\n'
  for c in \$PMC; do for k in "void caco::gemm_bf16_w8_kernel<0,1,1>(caco::W8Args)" "rd(float const*, float*, unsigned long)" "wr(float*, unsigned long)" "void caco::attention_kernel<96,false,4,2>(caco::AttnArgs)" "void caco::mel_kernel<0>(caco::MelArgs)"; do
    printf '1,1,1,1,1,1,131072,1,"%s",512,0,0,128,0,96,"%s",1000.0,1000,2000\n' "\$k" "\$c"; done; done; } > "\$D/host/1/p_counter_collection.csv"
"\${CMD[@]}"
FAKE
cat > "$T/hipcc" <<FAKE
#!/bin/bash
echo "hipcc \$*" >> "$LOG"; exit 0
FAKE
chmod +x "$T"/python "$T"/timeout "$T"/rocprofv3 "$T"/hipcc
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
