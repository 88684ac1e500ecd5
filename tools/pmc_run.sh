#!/bin/bash
# PMC counters for any command (separate rocprofv3 passes; kernel-trace only, no other tracing domains).
#   bash tools/pmc_run.sh <tag> <kernel-name-substring> -- <command...>
# PMC_PASSES="tcc1 tcc2" limits the passes (default: all five); every pass re-runs the command.
TAG=$1; FILT=$2; shift 3
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
run() { # name counters... (command in "$CMD")
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/raw_$name -o p -- "${CMD[@]}" > $OUT/$name.log 2>&1
  local f=$(find $OUT/raw_$name -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python tools/summarize_pmc.py "$f" | grep -i "kernel,\|$FILT" > $OUT/$name.csv; cat $OUT/$name.csv; else echo "no counters for $name"; tail -3 $OUT/$name.log; fi
  local kt=$(find $OUT/raw_$name -name "*kernel_trace.csv" | head -1)   # wall time of the same dispatches (clock = cycles / time)
  if [ -n "$kt" ]; then python tools/summarize_trace.py "$kt" "$FILT" | tee $OUT/$name.dur; fi
  rm -rf $OUT/raw_$name
}
CMD=("$@")
PASSES=" ${PMC_PASSES:-sq1 sq2 sq3 tcc1 tcc2} "
want_pass() { [[ "$PASSES" == *" $1 "* ]]; }
want_pass sq1 && run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
want_pass sq2 && run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS
want_pass sq3 && run sq3 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_VALU_TRANS SQ_THREAD_CYCLES_VALU
want_pass tcc1 && run tcc1 FETCH_SIZE GRBM_GUI_ACTIVE
want_pass tcc2 && run tcc2 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
true
