#!/bin/bash
# The GPU sessions that rounds 3-5 prepared without a GPU (tools/README.md has the runbook):
#   gpurun --timeout 3000 -- 'bash tools/gpu_session.sh truth'                       then, one call each:
#   gpurun --timeout 1800 -- 'bash tools/gpu_session.sh ab'       bash tools/gpurun_variants.sh --timeout 2400 -- 'bash tools/gpu_session.sh variants'
#   gpurun --timeout 1800 -- 'bash tools/gpu_session.sh pmc'      bash tools/gpurun_variants.sh --timeout 1500 -- 'bash tools/gpu_session.sh bisect'
# 1. truth: hardware record of the DEFAULT build: pytest -m gpu (default path), smoke, bench, rocprofv3 kernel stats, then the
#    experimental cases  -> gpurun_out/r6_v0/   (copy to profiles/r6_v0/)
# 2. ab: every run-time switch against the default, interleaved in one process (tools/ab_switches.py: flip / delete verdicts),
#    rocprofv3 kernel stats with the fusions on, GEMM kernels per shape (w8 / w4q / w4h / x / 128), fc1 inside chains
# 3. variants: compile-time A/B libraries (tools/build_variants.sh; they travel only with tools/gpurun_variants.sh), the skew and
#    attn_lean libraries under the product's own tests, tools/check_predictions.py      4. pmc: SQ / TCC counters, HBM traffic of fc1
# 5. bisect (never part of `all`): the arms r2 / classic / libm_erf / r2addr under the GEMM cases, after a red truth run
# Every part is wrapped in its own timeout so that a hang cannot eat the call.
set -u
PART=${1:-all}
OUT=${OUT:-gpurun_out/r6_v0}
mkdir -p "$OUT"
export TMPDIR=/tmp
want() { [ "$PART" = all ] || [ "$PART" = "$1" ]; }
# 0. quick (never part of `all`): the shortest hardware record of the default build - pytest -m gpu (default path), smoke, bench with its
#    own counter pass; no rocprofv3 kernel stats, no experimental cases.  For a pool that opens with little session time left.
if [ "$PART" = quick ]; then
echo "HEAD $(cat .git/HEAD 2>/dev/null) $(date -u +%FT%TZ) (quick)" > "$OUT/session.txt"
(timeout 900 python -m pytest tests -m gpu -q -rfE --tb=short 2>&1 | tail -120) > "$OUT/pytest_gpu.txt"
tail -3 "$OUT/pytest_gpu.txt"
(timeout 300 python __graft_entry__.py smoke 2>&1 | tail -5) > "$OUT/smoke.txt"
cat "$OUT/smoke.txt"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.err"
head -c 1500 "$OUT/bench.json"; echo
fi
if want truth; then
echo "HEAD $(cat .git/HEAD 2>/dev/null) $(date -u +%FT%TZ)" > "$OUT/session.txt"
(timeout 1200 python -m pytest tests -m gpu -q -rfE --tb=short 2>&1 | tail -120) > "$OUT/pytest_gpu.txt"                  # the product: default path only
tail -3 "$OUT/pytest_gpu.txt"
(timeout 300 python __graft_entry__.py smoke 2>&1 | tail -5) > "$OUT/smoke.txt"
cat "$OUT/smoke.txt"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.err"
head -c 600 "$OUT/bench.json"; echo
bash tools/profile_bench.sh r6_default --steps 10 --warmup 3 --no-extra-configs > "$OUT/prof_default.txt" 2>&1
cp gpurun_out/prof_r6_default/kernel_stats_summary.csv "$OUT/kernel_stats_default.csv" 2>/dev/null
# the experimental cases last: the product's record (pytest, smoke, bench, kernel stats) must exist before anything optional runs
(CACO_RUN_EXPERIMENTAL=1 timeout 900 python -m pytest tests -m "gpu and experimental" -q 2>&1 | tail -40) > "$OUT/pytest_gpu_experimental.txt"   # never-default kernels, opt-in switches (no -x)
tail -3 "$OUT/pytest_gpu_experimental.txt"
fi
if want ab; then
# every run-time switch against the default, interleaved inside ONE process (caco_set_switch), with the flip / delete verdict
(timeout 900 python tools/ab_switches.py --reps 5 --steps 10 --out "$OUT/ab_switches.json" 2>&1 | tail -40) | tee "$OUT/ab_switches.txt"
CACO_ATTN_SMALL=1 CACO_POS_FUSE=1 CACO_POOL_FUSE=1 bash tools/profile_bench.sh r6_switches --steps 10 --warmup 3 --no-extra-configs > "$OUT/prof_switches.txt" 2>&1
cp gpurun_out/prof_r6_switches/kernel_stats_summary.csv "$OUT/kernel_stats_switches.csv" 2>/dev/null
{ for t in 8256 4256 8256 4256; do echo "tile $t"; timeout 120 python tools/gemm_bench.py --tile $t --only qkv,out,fc1,fc2,t_fc1,t_fc2 --iters 20; done;
  for t in 128 2256 8256 4256 4128; do echo "text shapes, tile $t"; timeout 120 python tools/gemm_bench.py --tile $t --only t_qkv,t_out,t_fc1,t_fc2 --iters 50; done; } > "$OUT/gemm_w8_vs_w4q.txt" 2>&1; cat "$OUT/gemm_w8_vs_w4q.txt"
(timeout 300 python tools/gemm_chain_bench.py 2>&1 | tail -8) > "$OUT/gemm_chain.txt"; cat "$OUT/gemm_chain.txt"
fi
# bisection arms (round 5): only worth running when `truth` shows a red GEMM / model case, or to settle whether
# rounds 1-2's epilogue addressing wrote past row M on hardware (r2 and r2addr under test_gemm_ragged_m_writes_nothing_past_row_m).
#   r2 = the whole library of commit cccbeef (last binary an MI355X ran), classic = r4's peeled K-tile + packed bias epilogue reverted,
#   libm_erf = r4's rational erf reverted, r2addr = r3's per-lane row offsets reverted.  (tools/build_variants.sh builds all four.)
if [ "$PART" = bisect ]; then
  for v in r2 classic libm_erf r2addr; do
    L=$PWD/cacophony_amd/_variants/libcaco_hip_$v.so
    [ -f "$L" ] || { echo "$v: not built"; continue; }
    (CACO_ALLOW_VARIANT_LIB=1 CACO_LIB_PATH=$L timeout 600 python -m pytest tests/test_gpu_ops.py -q -m "gpu and not experimental" -k "gemm or ragged" 2>&1 | tail -15) > "$OUT/pytest_arm_$v.txt"
    echo "== arm $v"; tail -3 "$OUT/pytest_arm_$v.txt"
  done
  (timeout 900 python tools/ab_variants.py --reps 3 --steps 10 --out "$OUT/ab_arms.json" default r2 classic libm_erf r2addr 2>&1 | tail -30) | tee "$OUT/ab_arms.txt"
fi
# compile-time variants prepared by tools/build_variants.sh (cacophony_amd/_variants/; they travel only with tools/gpurun_variants.sh)
if want variants; then
if ls cacophony_amd/_variants/libcaco_hip_skew.so >/dev/null 2>&1; then
  # parity first: the skewed-row-block fp32 epilogue under the product's own GEMM cases (the chip-filling fp32 + residual shapes of
  # test_gemm_persistent_multi_tile_pipeline reach it; everything else falls back to gemm_bf16_w8) and the model goldens
  # (CACO_ALLOW_VARIANT_LIB: the suite otherwise refuses any library but the product's)
  for v in skew skew_lin; do
    (CACO_ALLOW_VARIANT_LIB=1 CACO_LIB_PATH=$PWD/cacophony_amd/_variants/libcaco_hip_$v.so timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -q -m "gpu and not experimental" -k "gemm or golden or chip_filling or full_batch" 2>&1 | tail -3) > "$OUT/pytest_$v.txt"
    echo "== $v"; cat "$OUT/pytest_$v.txt"
  done
  (CACO_ALLOW_VARIANT_LIB=1 CACO_LIB_PATH=$PWD/cacophony_amd/_variants/libcaco_hip_attn_lean.so timeout 600 python -m pytest tests/test_gpu_ops.py -q -m "gpu and not experimental" -k "attention" 2>&1 | tail -3) > "$OUT/pytest_attn_lean.txt"
  cat "$OUT/pytest_attn_lean.txt"
  # isolated launches of the two fp32 + residual shapes: default, skew, skew_lin (twice, alternating: the box's own spread)
  { for rep in 1 2; do for v in default skew skew_lin; do
      if [ $v = default ]; then unset CACO_LIB_PATH; else export CACO_LIB_PATH=$PWD/cacophony_amd/_variants/libcaco_hip_$v.so; fi
      echo "== $v (rep $rep)"; CACO_ALLOW_VARIANT_LIB=1 timeout 120 python tools/gemm_bench.py --only out,fc2 --iters 20
    done; done; unset CACO_LIB_PATH; } > "$OUT/gemm_f32r_isolated.txt" 2>&1; cat "$OUT/gemm_f32r_isolated.txt"
  # every variant library against the default, interleaved inside ONE process (tools/ab_variants.py: one model per library, rotated
  # order); every arm has a written prediction (tools/build_variants.sh, tools/check_predictions.py): 11 arms x 5 reps
  (timeout 1200 python tools/ab_variants.py --reps 5 --steps 10 --out "$OUT/ab_variants.json" default default+fold skew skew+fold skew_lin classic attn_lean attn_lean_k2 wb hints 2>&1 | tail -50) | tee "$OUT/ab_variants.txt"
  [ -s "$OUT/bench.json" ] && B="$OUT/bench.json" || B=""
  [ -s "$OUT/ab_variants.json" ] && python tools/check_predictions.py "$OUT/ab_variants.json" $B 2>&1 | tee "$OUT/predictions_vs_measured.txt"      # stated before, checked after
else
  echo "variants skipped: libraries not present (build with tools/build_variants.sh, push with tools/gpurun_variants.sh)"
fi
fi
# PMC counters of the shipped kernels, one fresh session, per-dispatch min / max next to the means (round-2 verdict item 6)
if want pmc; then
  # first the one record bench.py quotes (roofline.traffic): fabric bytes of a fc1 launch with the DEFAULT tile order (w_ngroup -1) -
  # copy it to profiles/r6_v0/hbm_traffic.json and commit it before the driver's end-of-round bench
  (unset CACO_W_NGROUP; bash tools/pmc_hbm.sh r6_hbm fc1 256) > "$OUT/pmc_hbm.log" 2>&1; cp gpurun_out/r6_hbm/hbm_traffic.json "$OUT/" 2>/dev/null && cat "$OUT/hbm_traffic.json"
  bash tools/pmc_run.sh r6_pmc_fc1 gemm_bf16_w8 -- python tools/gemm_bench.py --iters 3 --only fc1 > /dev/null 2>&1
  bash tools/pmc_run.sh r6_pmc_attn attention_kernel -- python tools/attn_bench.py > /dev/null 2>&1
  bash tools/pmc_run.sh r6_pmc_mel mel_kernel -- python tools/mel_bench.py > /dev/null 2>&1
  for k in fc1 attn mel; do for f in sq1 sq2 sq3 tcc1 tcc2; do cat gpurun_out/r6_pmc_$k/$f.csv gpurun_out/r6_pmc_$k/$f.dur > "$OUT/pmc_${k}_$f.csv" 2>/dev/null; done; done
  # fabric reads of every kernel of the step (TCC passes only), default order vs one n-tile group (round 2's order, what profiles/r2_v3 measured)
  PMC_PASSES="tcc1 tcc2" bash tools/pmc_run.sh r6_pmc_step kernel -- python bench.py --steps 2 --warmup 1 --profile-steps 1 --no-cpu-baseline --no-extra-configs > /dev/null 2>&1
  PMC_PASSES="tcc1 tcc2" bash tools/pmc_run.sh r6_pmc_step_ng0 kernel -- env CACO_W_NGROUP=0 python bench.py --steps 2 --warmup 1 --profile-steps 1 --no-cpu-baseline --no-extra-configs > /dev/null 2>&1
  for k in step step_ng0; do for f in tcc1 tcc2; do cp gpurun_out/r6_pmc_$k/$f.csv "$OUT/pmc_${k}_$f.csv" 2>/dev/null; done; done
  ls "$OUT"
fi
echo "session done"
