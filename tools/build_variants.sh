#!/bin/bash
# Rebuild the variant libraries of the GPU sessions from the CURRENT sources (in-tree, git-ignored; they travel only with
# tools/gpurun_variants.sh).  Run after any kernel change.
#   bash tools/build_variants.sh            the A/B set of `gpu_session.sh variants`: 7 libraries, every one with a written prediction
#   bash tools/build_variants.sh bisect     the bisection arms of `gpu_session.sh bisect` (only after a red `truth` run): r2 classic libm_erf r2addr
#   bash tools/build_variants.sh all        both
# Round 6 (no GPU): arms without a prediction of their own were dropped or merged - skew_d2 (depth sweep of skew: only
# matters once skew has won), kpipe2 (contained in attn_lean_k2), f32direct (the fp32 epilogue straight from the accumulator layout with the
# matrix pipe idle: skew's events are that form UNDER the K-loop), bf16_wb / f32_wb / f32_wb_ld0 / st_plain (four cuts of one question -> `wb`),
# a_nt / w_nt / attn_nt / ln_nt (four cuts of one question -> `hints`).  If a merged arm moves the step, split it then.
set -e
cd "$(dirname "$0")/.."
WHAT=${1:-ab}
rm -f cacophony_amd/_variants/*.so cacophony_amd/_variants/*.o
if [ "$WHAT" = ab ] || [ "$WHAT" = all ]; then
# round 5: the fp32 + residual epilogue UNDER the K-loop (csrc/gemm_w8_skew.inc: skewed row blocks, one 16-row block completing every
# D K-tiles; simulator-checked, 246 VGPRs, 0 scratch).  Predictions, stated before any measurement: skew (circular panel list: no
# launch tail) out-proj 0.23 -> <= 0.20 ms, fc2 0.54 -> <= 0.41 ms, step -1.8 .. -2.2 ms of 28; skew_lin (linear list: 7D + 1 extra
# K-tiles per launch) <= 0.21 / <= 0.46 ms, step -1.0 .. -1.3.
bash tools/build_variant.sh skew gemm_w8.hip -DW8_F32_SKEW
bash tools/build_variant.sh skew_lin gemm_w8.hip -DW8_F32_SKEW -DW8_SKEW_LINEAR
# round 5: the two-block attention kernel without its spilled register (geometry of the DMA pieces packed 9 -> 3 registers, K row
# offset re-derived per tile: 0 bytes of scratch in every attention kernel) and with the output epilogue's normalisation + bf16
# conversion on register pairs (-156 instructions per wave); bitwise the default's results on the simulator.  Prediction: attention
# 0.264 -> 0.255-0.262 ms; attn_lean_k2 adds round 4's K fragment reads pinned 2 steps ahead of their MFMAs: a further 0 .. -0.01 ms.
bash tools/build_variant.sh attn_lean attention.hip -DATTN_LEAN
bash tools/build_variant.sh attn_lean_k2 attention.hip -DATTN_LEAN -DATTN_KPIPE=2
# round 4 made two forms the default WITHOUT a timing (profiles/r4_cpu/epilogue_budget.txt) - the peeled first K-tile with C = 0 and
# the bias-only bf16 epilogue on register pairs; `classic` is the previous form.  Prediction (round 4): default faster by 0.2 .. 1.1 ms.
bash tools/build_variant.sh classic gemm_w8.hip -DW8_CLASSIC
# every epilogue store / residual load of the persistent GEMM with the DEFAULT cache policy instead of nt / sc1: on gfx950's in-order
# vmcnt queue the next tile's operand loads cannot be confirmed before the epilogue's stores are acknowledged, and a store acknowledged
# at the L2 shortens that wait (profiles/r4_cpu/epilogue_budget.txt); round 1 chose nt inside round 1's kernels.  Prediction: within
# +-0.3 ms of the default (the box's own spread) - a larger move in either direction is the finding.
bash tools/build_variant.sh wb gemm_w8.hip -DW8_ST_AUX=0 -DW8_ST_AUX_F32=0 -DW8_LD_AUX=0
# streaming hints on everything that is read or written once and is not hinted yet: GEMM A operand (nt), W operand (nt), attention
# output stores (nt), LayerNorm output stores (nt).  Prediction: within +-0.3 ms; W with nt may LOSE (weights should stay in L2).
bash tools/build_variant.sh hints gemm_w8.hip,attention.hip,norm.hip -DW8_A_AUX=2 -DW8_W_AUX=2 -DATTN_ST_AUX=2 -DLN_ST_NT
fi
if [ "$WHAT" = bisect ] || [ "$WHAT" = all ]; then
# single-change bisection arms: round 4's peeled K-tile + packed bias epilogue reverted (classic), the erf-GELU on
# the device library's erff as in rounds 1-2 (every GEMM translation unit), rounds 1-2's epilogue addressing (row block in the scalar
# offset), and the arm that changes everything at once: the whole library of commit cccbeef, the last binary an MI355X has run
[ -f cacophony_amd/_variants/libcaco_hip_classic.so ] || bash tools/build_variant.sh classic gemm_w8.hip -DW8_CLASSIC
bash tools/build_variant.sh libm_erf gemm.hip,gemm_x.hip,gemm_w8.hip,gemm_w4q.hip,gemm_w4h.hip -DCACO_LIBM_ERF
bash tools/build_variant.sh r2addr gemm_w8.hip -DW8_R2_ADDR
bash tools/build_r2_arm.sh || echo "r2 arm skipped (commit cccbeef not in this clone?)"
fi
python -m cacophony_amd.build >/dev/null
# every variant must resolve all its symbols (a kernel-side signature change breaks a variant silently otherwise)
python - <<'PY'
import ctypes, glob, sys
bad = 0
for f in sorted(glob.glob("cacophony_amd/_variants/*.so")):
    try:
        ctypes.CDLL(f)
        print("ok  ", f)
    except OSError as e:
        bad += 1
        print("FAIL", f, str(e)[:160])
sys.exit(1 if bad else 0)
PY
