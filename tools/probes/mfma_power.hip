// Probe: energy of the matrix pipe by instruction shape and wave tile.  Register-resident operands (no LDS, no memory in
// the loop), two operand sets alternating like the K-loop's double buffer; random or zero operand bits come from `seed`.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC mfma_power.hip -o libmfma_power.so ; driven by tools/mfma_power.py
// Variants: 0 = 32x32x16, wave tile 128x64, 8 waves/CU (the w8 kernel's arithmetic)   1 = 16x16x32, 128x64, 8 waves/CU
//           2 = 16x16x32, wave tile 128x128, 4 waves/CU (the vendor library's choice)  3 = 32x32x16, 128x128, 4 waves/CU
#include <hip/hip_runtime.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NX, int NW>
__device__ __forceinline__ void load_ops(const bf16x8* seed, bf16x8 (&x)[2][NX], bf16x8 (&w)[2][NW]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int k = (blockIdx.x * 8 + wave) * 64 * (2 * (NX + NW)) + lane;
  for (int s = 0; s < 2; ++s) {
    for (int i = 0; i < NX; ++i, k += 64) x[s][i] = seed[k & 0xfffff];
    for (int j = 0; j < NW; ++j, k += 64) {
      bf16x8 v = seed[k & 0xfffff];
      for (int e = 0; e < 8; ++e) v[e] = (__bf16)((float)v[e] * 0.036f);
      w[s][j] = v;
    }
  }
}

template <int NX, int NW, int THREADS>
__global__ __launch_bounds__(THREADS) void k32(const bf16x8* seed, float* out, int iters) {
  bf16x8 x[2][NX], w[2][NW];
  load_ops<NX, NW>(seed, x, w);
  f32x16 acc[NX][NW];
  for (int i = 0; i < NX; ++i) for (int j = 0; j < NW; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < NX; ++i)
#pragma unroll
        for (int j = 0; j < NW; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[s][j], x[s][i], acc[i][j], 0, 0, 0);
  }
  float t = 0.f;
  for (int i = 0; i < NX; ++i) for (int j = 0; j < NW; ++j) for (int e = 0; e < 16; ++e) t += acc[i][j][e];
  if (t == 1234.5f) out[threadIdx.x] = t;
}
template <int NX, int NW, int THREADS>
__global__ __launch_bounds__(THREADS) void k16(const bf16x8* seed, float* out, int iters) {
  bf16x8 x[2][NX], w[2][NW];
  load_ops<NX, NW>(seed, x, w);
  f32x4 acc[NX][NW];
  for (int i = 0; i < NX; ++i) for (int j = 0; j < NW; ++j) for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < NX; ++i)
#pragma unroll
        for (int j = 0; j < NW; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[s][j], x[s][i], acc[i][j], 0, 0, 0);
  }
  float t = 0.f;
  for (int i = 0; i < NX; ++i) for (int j = 0; j < NW; ++j) for (int e = 0; e < 4; ++e) t += acc[i][j][e];
  if (t == 1234.5f) out[threadIdx.x] = t;
}

// returns flops per launch
extern "C" double mfma_power_run(int variant, const void* seed, void* out, int iters, int grid, hipStream_t st) {
  const bf16x8* s = (const bf16x8*)seed;
  float* o = (float*)out;
  switch (variant) {
    case 0: k32<4, 2, 512><<<grid, 512, 0, st>>>(s, o, iters); return 2.0 * 32 * 32 * 16 * 16 * 8.0 * grid * iters;
    case 1: k16<8, 4, 512><<<grid, 512, 0, st>>>(s, o, iters); return 2.0 * 16 * 16 * 32 * 64 * 8.0 * grid * iters;
    case 2: k16<8, 8, 256><<<grid, 256, 0, st>>>(s, o, iters); return 2.0 * 16 * 16 * 32 * 128 * 4.0 * grid * iters;
    case 3: k32<4, 4, 256><<<grid, 256, 0, st>>>(s, o, iters); return 2.0 * 32 * 32 * 16 * 32 * 4.0 * grid * iters;
  }
  return 0;
}
