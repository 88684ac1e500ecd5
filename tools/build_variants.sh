#!/bin/bash
# Rebuild the A/B variant libraries of the GPU session (tools/gpu_session.sh variants) from the CURRENT sources (they travel with gpurun: in-tree,
# git-ignored).  Run after any kernel change, before `gpurun -- bash tools/gpu_session.sh variants`.
set -e
cd "$(dirname "$0")/.."
rm -f cacophony_amd/_variants/*.so cacophony_amd/_variants/*.o
# (round 5: the parked max-free attention pass `fastpass` - 13 spilled registers in its two-block kernel - is no longer built; the source
# stays under tools/experimental/; SRC_OVERRIDE in tools/build_variant.sh builds it without touching csrc/)
# round 4: K fragment reads of the score phase pinned 1 / 2 steps ahead of their MFMAs (the default build's ISA waits lgkmcnt(0) after
# every read there: one fragment buffer at 256 VGPRs); same registers, same results (simulator)
# (round 5: attn_sc1, ln_2rows, a_sc1 - no written hypothesis -, kpipe1, f32direct4 / f32direct16 / classic_f32direct - depth sweeps that only
# matter if their base form wins - are no longer built: VERDICT r4 item 5)
# round 5 (CPU part): the two-block attention kernel without its spilled register (geometry of the DMA pieces packed
# 9 -> 3 registers, K row offset re-derived per tile: 0 bytes of scratch in every attention kernel) and with the output epilogue's
# normalisation + bf16 conversion on register pairs (-156 instructions per wave, no v_perm / v_alignbit); bitwise the default's
# results on the simulator.  Prediction: attention 0.264 -> 0.255-0.262 ms (3 % fewer instructions in an issue-bound kernel, ~2 % more
# in its tile loop); attn_lean_k2 adds kpipe2's pinned K reads on top.
bash tools/build_variant.sh attn_lean attention.hip -DATTN_LEAN
bash tools/build_variant.sh attn_lean_k2 attention.hip -DATTN_LEAN -DATTN_KPIPE=2
bash tools/build_variant.sh kpipe2 attention.hip -DATTN_KPIPE=2
bash tools/build_variant.sh attn_nt attention.hip -DATTN_ST_AUX=2
bash tools/build_variant.sh ln_nt norm.hip -DLN_ST_NT
bash tools/build_variant.sh a_nt gemm_w8.hip -DW8_A_AUX=2
bash tools/build_variant.sh w_nt gemm_w8.hip -DW8_W_AUX=2
# epilogue stores with the default cache policy (instead of nt / sc1): for the ping-pong experiment, in case the streaming
# hints keep a producer's output out of the Infinity Cache
bash tools/build_variant.sh st_plain gemm_w8.hip -DW8_ST_AUX=0 -DW8_ST_AUX_F32=0 -DW8_LD_AUX=0
# round 4: the plain fp32 epilogues (out-proj, fc2, patch-embed) straight from the accumulator layout - no LDS transposition,
# 12 resp. 16 residual blocks in flight per wave (gemm_w8_epilogue.h W8_F32_DIRECT); checked on the simulator build
bash tools/build_variant.sh f32direct gemm_w8.hip -DW8_F32_DIRECT=12
# round 4: the static instruction budget (profiles/r4_cpu/epilogue_budget.txt) made two forms the default WITHOUT a timing - the peeled
# first K-tile with C = 0 and the bias-only bf16 epilogue on register pairs.  `classic` is the previous form: the other arm of the A/B.
# fp32 epilogue stores with the default (write-back) policy: on gfx950's in-order vmcnt queue the next tile's operand loads cannot be
# confirmed before the epilogue's stores are acknowledged; a store acknowledged at the L2 instead of at memory shortens that wait
# (profiles/r4_cpu/epilogue_budget.txt).  Round 2 compared nt and sc1 for this epilogue, not the default policy.
bash tools/build_variant.sh bf16_wb gemm_w8.hip -DW8_ST_AUX=0          # the bf16 epilogue's stores with the default policy (round 1 chose nt over it inside round 1's kernels)
bash tools/build_variant.sh f32_wb gemm_w8.hip -DW8_ST_AUX_F32=0
bash tools/build_variant.sh f32_wb_ld0 gemm_w8.hip -DW8_ST_AUX_F32=0 -DW8_LD_AUX=0
bash tools/build_variant.sh classic gemm_w8.hip -DW8_CLASSIC
# round 5: single-change bisection arms for the items `classic` does not isolate - the erf-GELU on the device
# library's erff as in rounds 1-2 (every GEMM translation unit), and rounds 1-2's epilogue addressing (row block in the scalar offset)
bash tools/build_variant.sh libm_erf gemm.hip,gemm_x.hip,gemm_w8.hip,gemm_w4q.hip,gemm_w4h.hip -DCACO_LIBM_ERF
bash tools/build_variant.sh r2addr gemm_w8.hip -DW8_R2_ADDR
# round 5: the fp32 + residual epilogue UNDER the K-loop (csrc/gemm_w8_skew.inc: skewed row blocks, one 16-row block completing every
# D K-tiles; simulator-checked, 246 VGPRs, 0 scratch).  Predictions, stated before any measurement: skew (circular panel list: no
# launch tail) out-proj 0.23 -> <= 0.20 ms, fc2 0.54 -> <= 0.41 ms, step -1.8 .. -2.2 ms of 28; skew_lin (linear list: 7D + 1 extra
# K-tiles per launch) <= 0.21 / <= 0.46 ms, step -1.0 .. -1.3; skew_d2 (at most 2 K-tiles between events) within 0.02 ms of skew.
bash tools/build_variant.sh skew gemm_w8.hip -DW8_F32_SKEW
bash tools/build_variant.sh skew_d2 gemm_w8.hip -DW8_F32_SKEW -DW8_SKEW_D=2
bash tools/build_variant.sh skew_lin gemm_w8.hip -DW8_F32_SKEW -DW8_SKEW_LINEAR
# ... and the arm that changes everything at once: the whole library of commit cccbeef, the last binary an MI355X has run
bash tools/build_r2_arm.sh
python -m cacophony_amd.build --force >/dev/null
# every variant must resolve all its symbols (a kernel-side signature change breaks the parked attention variant silently otherwise)
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
