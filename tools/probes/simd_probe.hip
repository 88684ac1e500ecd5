// Probe: where do the 4 waves of a 256-thread / 160 KiB-LDS / 512-register workgroup land (SIMD ids), and how fast
// does a bare v_mfma_f32_32x32x16_bf16 stream run in that configuration?   hipcc --offload-arch=gfx950 -O3 simd_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void probe(unsigned* ids, float* out, int iters) {
  extern __shared__ char smem[];
  f32x16 acc[16];
  for (int i = 0; i < 16; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(threadIdx.x * 0.001f + e); b[e] = (__bf16)(e * 0.5f - threadIdx.x * 0.002f); }
  unsigned hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  if ((threadIdx.x & 63) == 0) ids[blockIdx.x * 4 + (threadIdx.x >> 6)] = hw;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
  if (s == 12345.f) out[threadIdx.x] = s + smem[threadIdx.x];
}

// Same stream with operands that toggle the way a real GEMM's do: 4 A x 4 B fragments of hashed (mode 1) or zero (mode 0)
// bf16 values, acc[i*4+j] += A[i]*B[j].  The gap between the two modes is the DVFS/power cost of operand toggling alone.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void probe_data(float* out, int iters, int mode) {
  f32x16 acc[16];
  for (int i = 0; i < 16; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  bf16x8 a[4], b[4];
  unsigned h = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) {
    h = h * 1664525u + 1013904223u; float va = ((h >> 8) & 0xffff) * (1.f / 32768.f) - 1.f;
    h = h * 1664525u + 1013904223u; float vb = ((h >> 8) & 0xffff) * (1.f / 32768.f) - 1.f;
    a[i][e] = (__bf16)(mode ? va : 0.f); b[i][e] = (__bf16)(mode ? vb * 0.03f : 0.f);
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i * 4 + j], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
  if (s == 12345.f) out[threadIdx.x] = s;
}

int main() {
  unsigned* ids; float* out;
  hipMalloc(&ids, 256 * 4 * 4); hipMalloc(&out, 1024);
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int lds : {163840, 65536, 0}) {
    probe<<<256, 256, lds>>>(ids, out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<<<256, 256, lds>>>(ids, out, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = 256.0 * 4 * iters * 16 * 2.0 * 32 * 32 * 16;
    unsigned h[1024]; hipMemcpy(h, ids, sizeof(h), hipMemcpyDeviceToHost);
    printf("lds=%d: %.3f ms  %.1f TFLOP/s  wg0 simd ids:", lds, ms, flops / ms / 1e9);
    for (int w = 0; w < 4; ++w) printf(" [simd %u cu %u se %u]", (h[w] >> 4) & 3, (h[w] >> 8) & 15, (h[w] >> 13) & 7);
    int hist[4] = {0, 0, 0, 0}; int bad = 0;
    for (int g = 0; g < 256; ++g) { int m = 0; for (int w = 0; w < 4; ++w) m |= 1 << ((h[g * 4 + w] >> 4) & 3); if (m != 15) ++bad; }
    printf("  workgroups not on 4 distinct SIMDs: %d / 256\n", bad);
  }
  for (int rep = 0; rep < 2; ++rep)
    for (int mode : {1, 0}) {
      probe_data<<<256, 256>>>(out, 100, mode);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      probe_data<<<256, 256>>>(out, iters, mode);
      hipEventRecord(e1);
      hipDeviceSynchronize();
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double flops = 256.0 * 4 * iters * 16 * 2.0 * 32 * 32 * 16;
      printf("bare MFMA stream, %s operands: %.3f ms  %.1f TFLOP/s\n", mode ? "hashed" : "zero", ms, flops / ms / 1e9);
    }
  return 0;
}
