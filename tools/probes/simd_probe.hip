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
  return 0;
}
