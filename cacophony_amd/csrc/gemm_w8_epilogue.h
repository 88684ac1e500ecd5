// Epilogue of the 128 x 64 per-wave output tile of gemm_bf16_w8 (8 x 4 blocks of v_mfma_f32_16x16x32_bf16, operands swapped
// so that a lane owns 4 consecutive n of one output row m).  (The 32x32x16 form's epilogue, also used by the parked d4 / s8
// kernels: tools/experimental/gemm_w8_mf32_epilogue.h.)
#pragma once
#include "common.h"
#include "kernels.h"

namespace caco {
namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int ACT>
__device__ __forceinline__ float w4_epi_act(float x) {
  if constexpr (ACT == ACT_SILU) return silu_f(x);
  if constexpr (ACT == ACT_GELU) return gelu_erf_f(x);
  return x;
}

// Epilogue of one 256x256 tile, this wave's 128 x 64 part.  MODE selects what is known at compile time, so that the hot
// forms carry no per-element branches or phi copies (the generic form measured 400 v_mov + 70 branches per tile):
//   0 generic: bias / residual / LayerNorm-fold consumer / producer outputs all tested at run time
//   1 bias only        2 bias + residual (EPI_F32)
//   3 bias + LayerNorm-fold consumer (EPI_BF16)        4 bias + residual + fold producer (EPI_F32: bf16 copy + row sums)
//   5 bias + GATHERED residual (EPI_F32): row m adds resid[resid_idx[m], :] of a small table (row stride ldc); an index
//     < 0 wraps past the descriptor's range and reads as 0.  The patch-embed GEMM's positional embedding (api.hip).
// Global I/O goes through raw buffer descriptors anchored at the wave's tile corner: 32-bit offsets, and rows past M
// fall outside num_records, so stores need no exec mask and always count NST in vmcnt.  The ROW part of every offset is in
// the per-lane (VGPR) offset: under the LLVM-documented model of the raw.buffer intrinsics that is the only part the range
// check covers (the scalar offset operand is "excluded from bounds checking"); the gfx9-family ISA manuals read the other
// way for raw buffers (out of range when offset >= num_records - sgpr_offset), and no GPU has been available since round 2
// to settle it on this chip.  The per-lane form is safe under BOTH models.  Rounds 1-2 stepped through the row blocks with
// the scalar offset; under the LLVM model that leaves the rows of a ragged last M tile unprotected (the wavesim build, whose
// descriptor follows that model, corrupts the heap at M % 256 != 0 with the old form) - whether the hardware of rounds 1-2
// really wrote past row M is NOT established (tests/test_gpu_ops.py::test_gemm_ragged_m_writes_nothing_past_row_m is the
// hardware check).  -DW8_R2_ADDR (bisection arm `r2addr`, tools/build_variants.sh) puts the row-block part back into the
// scalar offset - rounds 1-2's addressing and nothing else - so that one GPU run of that test on that arm settles the
// question; tools/build_r2_arm.sh builds the whole round-2 library (commit cccbeef) as the arm that changes everything at
// once.  Only the column half j * 128, which is always inside a valid row's 256-byte span, still rides in the scalar offset.
// W8_VS(v, rows, col) expands to the (per-lane offset, scalar offset) argument pair of a buffer access.
// (Measured dead end: storing the accumulator layout directly - 8-byte pieces, no LDS transposition, no barrier after
// the epilogue - is 30-40 % SLOWER on the QKV / fc1 shapes: partial-line writes from 32 rows per instruction.)
// Cache-policy bits of the epilogue's stores / residual loads: 2 = nt (streaming).  The outputs are far larger than the
// L2 and are not re-read by this kernel; marking them streaming keeps the weight / activation tiles of the K-loop
// resident instead: QKV -5 %, fc1 -4.6 %, out-proj -5.8 % (nt residual loads), fc2 +-0; sc0 / sc1 variants equal.
#ifdef W8_R2_ADDR
#define W8_VS(v, rows, col) (v), (rows) + (col)
#else
#define W8_VS(v, rows, col) (v) + (rows), (col)
#endif
#ifndef W8_ST_AUX
#define W8_ST_AUX 2
#endif
// fp32 (residual) epilogue: sc1 = write-through.  Found while building the parked GEMM + LayerNorm launch
// (tools/experimental/ln_tail): out-proj 253 -> 231 us isolated, fc2 541 -> 530; in the step -0.15 ms (A/B on one box,
// 29.03 -> 28.88).  The bf16 epilogue keeps nt (sc1 there: +0.7 ms per step).
#ifndef W8_ST_AUX_F32
#define W8_ST_AUX_F32 16
#endif
#ifndef W8_LD_AUX
#define W8_LD_AUX 2
#endif
#ifndef W8_RES_AHEAD
#define W8_RES_AHEAD 1      // residual slabs in flight ahead of the one being processed (2 and 3 measured: no gain)
#endif
// 16x16x32 accumulator layout: acc[ib][jb] is a 16 (m) x 16 (n) block, lane l holds m = ib*16 + (l & 15) and the four
// consecutive n = jb*16 + 4*(l >> 4) + r.  Slabs are 32 rows x 128 bytes; a write instruction covers 16 rows per 16-lane
// group, so the chunk swizzle is (row >> 1) & 7 (the operand tiles' one).
// bf16: per 32-row block row a 32 x 64 bf16 slab.  Bias (and the LayerNorm-fold column sums) are re-read from L1 per
// 4-column group instead of being held in 32-64 registers across the whole epilogue: the accumulators already fill half
// the register file.  fp32: eight 32 x 32 fp32 slabs, the residual of slab s+1 fetched while slab s is processed.
// The per-lane offsets voff + block * rowb are loop-invariant across the output tiles of a persistent workgroup; left alone
// the compiler hoists all 16 of them out of the tile loop and carries 14 more VGPRs through the K-loop (238 -> 252 of the
// 256 a two-waves-per-SIMD kernel has).  Making the row pitch opaque once per epilogue keeps them epilogue-local: one
// v_add with a scalar operand per access instead.
// W8_EPI_SLAB_FENCE(): a scheduling fence between the slabs of an epilogue.  Empty for gemm_w8 (its instruction streams are
// the hardware-verified ones).  gemm_w4q defines it as sched_barrier: its 256 accumulators live in AGPRs and, left free, the
// scheduler hoists the v_accvgpr_reads of later slabs over the current one until the 256 VGPRs overflow (18-40 spilled).
#ifndef W8_EPI_SLAB_FENCE
#define W8_EPI_SLAB_FENCE() ((void)0)
#endif
#ifdef WAVESIM
#define W8_EPI_LOCAL(x) ((void)0)
#else
#define W8_EPI_LOCAL(x) asm volatile("" : "+s"(x))
#endif
template <int EPI, int ACT, int MODE>
__device__ __forceinline__ void w16_epilogue(const f32x4 (&acc)[8][4], const GemmArgs& p, int64_t m0, int n0, int wm, int wn, int lane, char* slab) {
  const int l16 = lane & 15, lq = lane >> 4;
  const int rrow = lane >> 3, c8 = lane & 7;            // read-back: 8 rows x 128 B per instruction
  const int64_t mw = m0 + wm * 128;
  const int nw = n0 + wn * 64;
  const int rows = (int)min((int64_t)128, p.M - mw);    // valid rows of this wave's part (may be <= 0)
  const bool has_bias = MODE ? true : p.bias != nullptr;
  if constexpr (EPI == EPI_BF16) {
    int rowb = p.ldc * 2;
    W8_EPI_LOCAL(rowb);
    const __amdgpu_buffer_rsrc_t out_r = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<bf16_t*>(p.out) + mw * p.ldc + nw, 0, rows > 0 ? (rows - 1) * rowb + 128 : 0, 0x00020000);
    const int voff = rrow * rowb + c8 * 16;
    const bool fold = MODE ? MODE == 3 : p.fold_mr != nullptr;
    const float* bias_l = has_bias ? p.bias + nw + lq * 4 : nullptr;
    const float* c1_l = fold ? p.fold_c1 + nw + lq * 4 : nullptr;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      W8_EPI_SLAB_FENCE();
#pragma unroll
      for (int ib = 0; ib < 2; ++ib) {
        float nmr = 0.f, rstd = 1.f;                 // out = rstd * acc + (-mean * rstd) * c1 + bias
        if (fold) {
          const int64_t m = min(mw + i * 32 + ib * 16 + l16, p.M - 1);
          const float2 mr = *reinterpret_cast<const float2*>(p.fold_mr + 2 * m);
          rstd = mr.y;
          nmr = -mr.x * mr.y;
        }
        const int row = ib * 16 + l16;
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) {
          f32x4 bb = has_bias ? *reinterpret_cast<const f32x4*>(bias_l + jb * 16) : f32x4{0.f, 0.f, 0.f, 0.f};
          if (fold) {
            const f32x4 cc = *reinterpret_cast<const f32x4*>(c1_l + jb * 16);
#pragma unroll
            for (int r = 0; r < 4; ++r) bb[r] = __builtin_fmaf(nmr, cc[r], bb[r]);
          }
          const f32x4 a = acc[i * 2 + ib][jb];
          bf16x4 o;
          if constexpr (ACT == ACT_SILU) {   // packed pairs, see w8_epilogue
            typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int r = 0; r < 4; r += 2) {
              f32x2 x;
              x[0] = MODE == 1 ? a[r] + bb[r] : __builtin_fmaf(a[r], rstd, bb[r]);
              x[1] = MODE == 1 ? a[r + 1] + bb[r + 1] : __builtin_fmaf(a[r + 1], rstd, bb[r + 1]);
              const f32x2 t = x * f32x2{-1.4426950408889634f, -1.4426950408889634f};
              const f32x2 d = f32x2{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])} + f32x2{1.f, 1.f};
              const f32x2 y = x * f32x2{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
              o[r] = (bf16_t)y[0];
              o[r + 1] = (bf16_t)y[1];
            }
          } else {
#ifndef W8_CLASSIC
            // Written on register pairs like the SiLU branch.  Left to the scalar loop below, hipcc 7.2 turns the bias-only
            // epilogue (QKV) into 64 v_add + 64 v_pk_add + 96 v_cvt_pk + 96 permute / align / pk_mov = 320 VALU per wave and
            // tile where 64 v_pk_add + 64 v_cvt_pk do the work (static count, profiles/r4_cpu/epilogue_budget.txt); the epilogue
            // is issue-bound.  Bit-identical results.  Default since round 4, not yet timed: -DW8_CLASSIC
            // (variant build `classic`) is the scalar form.
            typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int r = 0; r < 4; r += 2) {
              f32x2 x;
              if constexpr (MODE == 1) {
                x = f32x2{a[r], a[r + 1]} + f32x2{bb[r], bb[r + 1]};
              } else {
                x[0] = __builtin_fmaf(a[r], rstd, bb[r]);
                x[1] = __builtin_fmaf(a[r + 1], rstd, bb[r + 1]);
              }
              o[r] = (bf16_t)w4_epi_act<ACT>(x[0]);
              o[r + 1] = (bf16_t)w4_epi_act<ACT>(x[1]);
            }
#else
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float x = MODE == 1 ? a[r] + bb[r] : __builtin_fmaf(a[r], rstd, bb[r]);
              o[r] = (bf16_t)w4_epi_act<ACT>(x);
            }
#endif
          }
          *reinterpret_cast<bf16x4*>(slab + row * 128 + (((jb * 2 + (lq >> 1)) ^ ((row >> 1) & 7)) << 4) + (lq & 1) * 8) = o;
        }
      }
      CACO_WAVE_LDS_SYNC();
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        const int row = tt * 8 + rrow;
        const u32x4 v = *reinterpret_cast<const u32x4*>(slab + row * 128 + ((c8 ^ ((row >> 1) & 7)) << 4));
        __builtin_amdgcn_raw_buffer_store_b128(v, out_r, W8_VS(voff, (i * 32 + tt * 8) * rowb, 0), W8_ST_AUX);
      }
      CACO_WAVE_LDS_SYNC();
    }
  } else {   // EPI_F32: eight 32 x 32 fp32 slabs; the residual of slab s+1 is fetched while slab s is processed
    int rowb = p.ldc * 4;
    W8_EPI_LOCAL(rowb);
#ifdef W8_F32_DIRECT
#include "gemm_w8_epilogue_direct.inc"
#endif
    const int bytes = rows > 0 ? (rows - 1) * rowb + 256 : 0;
    const bool has_resid = MODE ? (MODE == 2 || MODE == 4 || MODE == 5) : p.resid != nullptr;
    const bool produce_xb = MODE ? MODE == 4 : p.xb_out != nullptr;
    const bool produce_st = MODE ? MODE == 4 : p.stats_part != nullptr;
    const __amdgpu_buffer_rsrc_t out_r =
        __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(p.out) + mw * p.ldc + nw, 0, bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t res_r =
        MODE == 5 ? __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.resid) + nw, 0, 0x7fffffff, 0x00020000)
                  : __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(has_resid ? p.resid : reinterpret_cast<const float*>(p.out)) + mw * p.ldc + nw,
                                                      0, has_resid ? bytes : 0, 0x00020000);
    int gidx[4][4];                     // MODE 5: table row of output row i*32 + tt*8 + rrow (rows past M: the last row's; never stored)
    if constexpr (MODE == 5) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) gidx[i][tt] = p.resid_idx[min(mw + i * 32 + tt * 8 + rrow, p.M - 1)];
    }
    const int voff = rrow * rowb + c8 * 16;
    const __amdgpu_buffer_rsrc_t xb_r = __builtin_amdgcn_make_buffer_rsrc(
        produce_xb ? p.xb_out + mw * p.ldc + nw : reinterpret_cast<bf16_t*>(p.out), 0, produce_xb ? bytes >> 1 : 0, 0x00020000);
    constexpr int RA = W8_RES_AHEAD, RN = RA + 1;
    u32x4 res[RN][4];
    float st1[4][4], st2[4][4];
    auto fetch = [&](int s, u32x4 (&dst)[4]) {
      const int i = s >> 1, j = s & 1;
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        if constexpr (MODE == 5) dst[tt] = __builtin_amdgcn_raw_buffer_load_b128(res_r, gidx[i][tt] * rowb + c8 * 16, j * 128, 0);   // table rows are re-used: default cache policy
        else dst[tt] = __builtin_amdgcn_raw_buffer_load_b128(res_r, W8_VS(voff, (i * 32 + tt * 8) * rowb, j * 128), W8_LD_AUX);
      }
    };
    if (has_resid) {
#pragma unroll
      for (int s0 = 0; s0 < RA; ++s0) fetch(s0, res[s0 % RN]);
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int i = s >> 1, j = s & 1;
      W8_EPI_SLAB_FENCE();
      if (has_resid && s + RA < 8) fetch(s + RA, res[(s + RA) % RN]);
#pragma unroll
      for (int ib = 0; ib < 2; ++ib)
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
          f32x4 v = acc[i * 2 + ib][j * 2 + jb];
          if (has_bias) v += *reinterpret_cast<const f32x4*>(p.bias + nw + j * 32 + jb * 16 + lq * 4);
          const int row = ib * 16 + l16;
          *reinterpret_cast<f32x4*>(slab + row * 128 + (((jb * 4 + lq) ^ ((row >> 1) & 7)) << 4)) = v;
        }
      CACO_WAVE_LDS_SYNC();
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        const int row = tt * 8 + rrow;
        f32x4 v = *reinterpret_cast<const f32x4*>(slab + row * 128 + ((c8 ^ ((row >> 1) & 7)) << 4));
        if (has_resid) v += __builtin_bit_cast(f32x4, res[s % RN][tt]);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), out_r, W8_VS(voff, (i * 32 + tt * 8) * rowb, j * 128), W8_ST_AUX_F32);
        if (produce_xb) {
          bf16x4 b;
#pragma unroll
          for (int r = 0; r < 4; ++r) b[r] = (bf16_t)v[r];
          typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
          #ifdef W8_R2_ADDR
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, b), xb_r, voff >> 1, ((i * 32 + tt * 8) * rowb + j * 128) >> 1, 0);
#else
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, b), xb_r, (voff + (i * 32 + tt * 8) * rowb) >> 1, (j * 128) >> 1, 0);
#endif
        }
        if (produce_st) {
          const float a1 = (v[0] + v[1]) + (v[2] + v[3]);
          const float a2 = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
          if (j == 0) { st1[i][tt] = a1; st2[i][tt] = a2; } else { st1[i][tt] += a1; st2[i][tt] += a2; }
        }
      }
      CACO_WAVE_LDS_SYNC();
    }
    if (produce_st) {
      const int nslot = p.N >> 6, slot = nw >> 6;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
          float a1 = st1[i][tt], a2 = st2[i][tt];
#pragma unroll
          for (int o = 1; o < 8; o <<= 1) {
            a1 += __shfl_xor(a1, o, 64);
            a2 += __shfl_xor(a2, o, 64);
          }
          const int64_t m = mw + i * 32 + tt * 8 + rrow;
          if (c8 == 0 && m < p.M) *reinterpret_cast<float2*>(p.stats_part + (m * nslot + slot) * 2) = make_float2(a1, a2);
        }
    }
  }
}

}  // namespace
}  // namespace caco
