// Epilogue of one wave's 128 x 64 output tile (8 x 4 MFMA 16x16 blocks, fp32 accumulators), shared by the
// 256x128 and the 256x256 GEMM kernels.  Bias / activation / residual are applied in registers; each 16-row slab is
// then transposed through a per-wave LDS scratch slab (4 KiB, XOR-swizzled instead of padded) so that EVERY global
// store instruction writes whole 128 B (bf16) or 256 B (fp32) row segments instead of the MFMA layout's 32 B pieces.
#pragma once
#include "common.h"
#include "kernels.h"

namespace caco {

constexpr int EPI_SCRATCH_BYTES = 4096;   // per wave

template <int ACT>
__device__ __forceinline__ float epi_act(float x) {
  if constexpr (ACT == ACT_SILU) return silu_f(x);
  if constexpr (ACT == ACT_GELU) return gelu_erf_f(x);
  return x;
}

// acc[i][j]: block (rows mw + 16 i, cols nw + 16 j), swapped MFMA form: lane = (m = lane & 15, 4 consecutive n at
// (lane >> 4) * 4).
template <int EPI, int ACT>
__device__ __forceinline__ void wave_epilogue_128x64(const f32x4 (&acc)[8][4], const GemmArgs& p, int64_t mw, int nw,
                                                     int lane, char* sc) {
  const int lm = lane & 15, lg = lane >> 4;
  if constexpr (EPI == EPI_BF16) {
    constexpr int PITCH = 144;                    // 64 bf16 + 16 B pad
    f32x4 b4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      b4[j] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + nw + j * 16 + lg * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 v = acc[i][j] + b4[j];
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (bf16_t)epi_act<ACT>(v[r]);
        *reinterpret_cast<bf16x4*>(sc + lm * PITCH + (j * 16 + lg * 4) * 2) = o;
      }
      CACO_WAVE_LDS_SYNC();
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {            // 8 rows x 128 B per store instruction
        const int row = tt * 8 + (lane >> 3), c16 = lane & 7;
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(sc + row * PITCH + c16 * 16);
        const int64_t m = mw + i * 16 + row;
        if (m < p.M) *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16_t*>(p.out) + m * p.ldc + nw + c16 * 8) = v;
      }
      CACO_WAVE_LDS_SYNC();
    }
  } else if constexpr (EPI == EPI_F32) {
    constexpr int PITCH = 256;                    // 64 fp32; 16-byte chunk c of row r lives at chunk c ^ (r & 15)
    f32x4 b4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      b4[j] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + nw + j * 16 + lg * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        *reinterpret_cast<f32x4*>(sc + lm * PITCH + (((j * 4 + lg) ^ lm) << 4)) = acc[i][j] + b4[j];
      CACO_WAVE_LDS_SYNC();
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {            // 4 rows x 256 B per store instruction
        const int row = tt * 4 + (lane >> 4), c4 = lane & 15;
        f32x4 v = *reinterpret_cast<const f32x4*>(sc + row * PITCH + ((c4 ^ row) << 4));
        const int64_t m = mw + i * 16 + row;
        if (m < p.M) {
          const int64_t o = m * p.ldc + nw + c4 * 4;
          if (p.resid) v += *reinterpret_cast<const f32x4*>(p.resid + o);
          *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + o) = v;
        }
      }
      CACO_WAVE_LDS_SYNC();
    }
  }
}

}  // namespace caco
