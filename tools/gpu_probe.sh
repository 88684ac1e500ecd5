#!/bin/bash
# Retry a gpurun call until the pool answers with something other than "refused" / "no box" (a refused call costs nothing).
#   bash tools/gpu_probe.sh <log> <timeout_s> <command...>
# Gives up after MAX_ATTEMPTS tries (default 17 = ~85 min at one try per SLEEP_S = 300 s), exit code 75 (EX_TEMPFAIL):
# the caller decides whether to start another round of tries.  Exit code otherwise = gpurun's.
LOG=$1; shift; TMO=$1; shift
MAX_ATTEMPTS=${MAX_ATTEMPTS:-17}; SLEEP_S=${SLEEP_S:-300}
n=0
while [ $n -lt "$MAX_ATTEMPTS" ]; do
  n=$((n + 1))
  /usr/local/graft/bin/gpurun --timeout "$TMO" -- "$@" > "$LOG" 2>&1; rc=$?
  if ! grep -q 'status=refused' "$LOG" && [ $rc -ne 3 ]; then echo "gpurun answered rc=$rc (attempt $n)"; tail -30 "$LOG"; exit $rc; fi
  echo "$(date -u +%FT%TZ) attempt $n: rc=$rc $(grep -o 'status=[a-z_]*' "$LOG" | head -1)" >> "$LOG.attempts"
  [ $n -lt "$MAX_ATTEMPTS" ] && sleep "$SLEEP_S"
done
echo "gpurun still closed after $n attempts"; tail -3 "$LOG"
exit 75
