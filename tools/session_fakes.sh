#!/bin/bash
# Logging fakes of `python`, `rocprofv3`, `timeout` and `hipcc` for dry runs WITHOUT a GPU (tools/session_dryrun.sh, tests/test_bench_traffic.py):
#   source tools/session_fakes.sh <dir> <log>      -> executables in <dir> (put it first on PATH), every invocation appended to <log>
# The fake rocprofv3 fabricates the csv files the scripts look for (kernel stats, kernel trace, counter collection with the value 1000
# for every counter asked for) and then runs its command; the fake python logs, runs pure record readers and inline scripts for real,
# prints a bench line for bench.py and writes a minimal record at --out.  Nothing they produce is a measurement.
T=$1; LOG=$2
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
