#!/bin/bash
# Retry a gpurun call until the pool answers with something other than "refused" / "no box" (a refused call costs nothing).
#   bash tools/gpu_probe.sh <log> <timeout_s> <command...>
LOG=$1; shift; TMO=$1; shift
while true; do
  /usr/local/graft/bin/gpurun --timeout "$TMO" -- "$@" > "$LOG" 2>&1; rc=$?
  if ! grep -q 'status=refused' "$LOG" && [ $rc -ne 3 ]; then echo "gpurun answered rc=$rc"; tail -30 "$LOG"; exit $rc; fi
  date -u +%T >> "$LOG.attempts"
  sleep 300
done
