#!/bin/bash
# PMC counters for the GEMM micro-benchmark (separate rocprofv3 passes, no tracing domains besides kernel-trace).
#   bash tools/pmc_gemm.sh <tag> <only> [tile]
TAG=${1:-pmc}; ONLY=${2:-fc1}; TILE=${3:-256}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
run() { # name counters...
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/raw_$name -o p -- python tools/gemm_bench.py --iters 3 --only $ONLY --tile $TILE > $OUT/$name.log 2>&1
  local f=$(find $OUT/raw_$name -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python tools/summarize_pmc.py "$f" > $OUT/$name.csv; cat $OUT/$name.csv; else echo "no counters for $name"; tail -3 $OUT/$name.log; fi
  rm -rf $OUT/raw_$name
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS
run tcc1 FETCH_SIZE GRBM_GUI_ACTIVE
run tcc2 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
