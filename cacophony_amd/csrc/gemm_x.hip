// gemm_bf16_x: 256x128 bf16 MFMA GEMM, two workgroups per CU.  out = epilogue(A[M,K] x W[N,K]^T), fp32 accumulate.
//
// Design point (measured, see DESIGN.md "GEMM"): at K = 768 the encoder GEMMs have only 12-24 K-steps per output
// tile, so a one-workgroup-per-CU kernel loses a lot around its K-loop: the epilogue's stores, the next tile's first
// loads and every barrier stall leave the matrix pipe idle.  This kernel runs TWO independent 4-wave workgroups per
// CU (2 waves per SIMD, one from each), so that one workgroup's epilogue / prologue / barrier wait is covered by
// the other's MFMAs.  The price is operand reuse: 85 FLOP per byte pulled from L2 against 128 for the 256x256 tile.
//
//   tile      256 (M) x 128 (N) x 32 (K-step), 4 waves as 2 x 2, wave tile 128 x 64 = 8 x 4 MFMA 16x16x32 blocks
//             (128 fp32 accumulators per lane)
//   LDS       3-stage ring of K-steps, 24 KiB each (A 256 rows + B 128 rows, 64 B per row) = 72 KiB -> 2 per CU
//   loads     global_load_lds_dwordx4 straight into the ring, issued TWO K-steps ahead, retired by a counted
//             vmcnt(6) (never 0 in the loop) + ONE raw s_barrier per K-step
//   LDS image lane-linear (a DMA constraint); bank conflicts are removed by permuting the 16-byte chunks of a row
//             on the SOURCE address and on the ds_read address with the same involution
//   epilogue  gemm_epilogue.h: bias / activation / residual in registers, 16-row slabs transposed through the (now
//             idle) ring so that every global store instruction writes whole 128 B / 256 B row segments
//
// Reference ops replaced: nn.Linear + activation + residual add (audio_models/mae.py:55-61,92-97;
// text_models/roberta.py:77-83,120,156-157,174).
#include "common.h"
#include "gemm_epilogue.h"
#include "kernels.h"

namespace caco {
namespace {

constexpr int XBM = 256, XBN = 128, XBK = 32;
constexpr int XROW = XBK * 2;                       // 64 bytes per row per K-step
constexpr int XA_BYTES = XBM * XROW;                // 16 KiB
constexpr int XB_BYTES = XBN * XROW;                // 8 KiB
constexpr int XSTAGE = XA_BYTES + XB_BYTES;         // 24 KiB
constexpr int XSTAGES = 3;
constexpr int XSMEM = XSTAGES * XSTAGE;             // 72 KiB

typedef __attribute__((address_space(3))) void* lds_vptr;
typedef const __attribute__((address_space(1))) void* gbl_vptr;

// 16-byte chunk permutation of a 64-byte row: with f = (-(row >> 2)) & 3 every ds_read_b128 lane group
// ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ...) touches 16 distinct 16-byte slots of the 256-byte bank row.
__device__ __forceinline__ int xswz(int row) { return (0 - (row >> 2)) & 3; }

// rows x 32 bf16 from row-major global memory -> lane-linear LDS image (one wave instruction = 16 rows = 1 KiB)
template <int ROWS>
__device__ __forceinline__ void xstage(const bf16_t* __restrict__ g, int64_t row0, int64_t last_row, int ld, int k0,
                                       char* lds, int wave, int lane) {
#pragma unroll
  for (int it = 0; it < ROWS / 64; ++it) {
    const int grp = it * 4 + wave;
    const int r = grp * 16 + (lane >> 2);
    const int c = (lane & 3) ^ xswz(r);
    int64_t grow = row0 + r;
    grow = grow < last_row ? grow : last_row;      // clamped rows are computed and never stored
    const bf16_t* src = g + grow * (int64_t)ld + k0 + c * 8;
    __builtin_amdgcn_global_load_lds((gbl_vptr)src, (lds_vptr)(lds + grp * 1024), 16, 0, 0);
  }
}

__device__ __forceinline__ bf16x8 xfrag(const char* tile, int row, int chunk) {
  return *reinterpret_cast<const bf16x8*>(tile + row * XROW + ((chunk ^ xswz(row)) << 4));
}

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}

template <int EPI, int ACT>
__global__ __launch_bounds__(256, 2) void gemm_bf16_x_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  const int tiles_n = p.N / XBN;
  const int tiles_m = (int)((p.M + XBM - 1) / XBM);
  const int t = xcd_remap(blockIdx.x, tiles_m * tiles_n);   // n fastest: the tiles_n workgroups of one A tile are neighbours
  const int64_t m0 = (int64_t)(t / tiles_n) * XBM;
  const int n0 = (t % tiles_n) * XBN;
  const int nks = p.K / XBK;
  const int lda = p.lda ? p.lda : p.K, ldw = p.ldw ? p.ldw : p.K;

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto issue = [&](int ks) {   // 4 + 2 direct-to-LDS loads per wave
    char* st = smem + (ks % XSTAGES) * XSTAGE;
    xstage<XBM>(p.A, m0, p.M - 1, lda, ks * XBK, st, wave, lane);
    xstage<XBN>(p.W, n0, p.N - 1, ldw, ks * XBK, st + XA_BYTES, wave, lane);
  };
  issue(0);
  if (nks > 1) issue(1);

  const int frow = lane & 15, fchunk = lane >> 4;
  for (int ks = 0; ks < nks; ++ks) {
    // K-step ks has landed once at most the 6 loads of K-step ks+1 are outstanding
    if (ks + 1 < nks) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                 // everyone's share landed; everyone is done reading K-step ks-1
    if (ks + 2 < nks) issue(ks + 2);              // overwrites the slot K-step ks-1 lived in
    const char* st = smem + (ks % XSTAGES) * XSTAGE;
    const char* a_t = st + wm * 128 * XROW;
    const char* b_t = st + XA_BYTES + wn * 64 * XROW;
    bf16x8 bf[4], af[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) bf[j] = xfrag(b_t, j * 16 + frow, fchunk);
#pragma unroll
    for (int i = 0; i < 8; ++i) af[i] = xfrag(a_t, i * 16 + frow, fchunk);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)    // operands swapped: D[i = n][j = m], a lane owns 4 consecutive n of one row m
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j], af[i], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  }
  __builtin_amdgcn_s_barrier();                   // the ring is idle: reuse it as per-wave transpose slabs
  wave_epilogue_128x64<EPI, ACT>(acc, p, m0 + wm * 128, n0 + wn * 64, lane, smem + wave * EPI_SCRATCH_BYTES);
}

template <int EPI, int ACT>
int launch_x(const GemmArgs& p, hipStream_t st) {
  auto kern = gemm_bf16_x_kernel<EPI, ACT>;
  CACO_TRY_RC(prepare_launch(reinterpret_cast<const void*>(kern), XSMEM, nullptr));
  const int tiles = (int)((p.M + XBM - 1) / XBM) * (p.N / XBN);
  hipLaunchKernelGGL(kern, dim3(tiles), dim3(256), XSMEM, st, p);
  return check_hip(hipGetLastError(), "gemm_bf16_x launch");
}

}  // namespace

int gemm_bf16_x(const GemmArgs& p, int epi, int act, hipStream_t st) {
  if (epi == EPI_BF16) {
    if (act == ACT_NONE) return launch_x<EPI_BF16, ACT_NONE>(p, st);
    if (act == ACT_SILU) return launch_x<EPI_BF16, ACT_SILU>(p, st);
    if (act == ACT_GELU) return launch_x<EPI_BF16, ACT_GELU>(p, st);
  } else if (epi == EPI_F32 && act == ACT_NONE) {
    return launch_x<EPI_F32, ACT_NONE>(p, st);
  }
  set_error("gemm_bf16_x: unsupported epilogue %d / activation %d", epi, act);
  return CACO_ERR_INVALID;
}

}  // namespace caco
