// bf16 MFMA GEMMs for the encoder stacks (dispatch + two of the three kernels), and an exact-fp32 MFMA GEMM for
// the small scoring ops.
//
//   out[M,N] = epilogue( A[M,K] (bf16, row-major) x W[N,K]^T (bf16, torch Linear layout) )
//
// Both operands are K-contiguous, which is the natural MFMA feed on CDNA4: every lane's 8-element fragment is one
// 16-byte LDS read.  Tiles are staged HBM/L2 -> LDS with 16-byte direct loads (global_load_lds_dwordx4); the LDS
// image is lane-linear, so the bank-conflict swizzle is applied to the per-lane SOURCE address and undone on the
// ds_read side (same involution on both sides).  Accumulation is fp32; bias / residual / SiLU / erf-GELU are fused
// into the epilogue so the activations make one HBM round trip per GEMM.
//
// Kernels (measurements and the reasoning behind the choice per shape: DESIGN.md section "GEMM"):
//   gemm_bf16_p8_kernel  (this file) 256x256x64 tile, 8 waves (2 x 4), ONE persistent workgroup per CU, 128 KiB LDS =
//                        two K-tiles, each split in four 16 KiB half-tiles (B0 B1 A0 A1).  The K-loop runs in
//                        PHASES of 16 MFMAs (one 64x32 quadrant of the wave's 128x64 output over the whole K-tile);
//                        every phase reads only the fragments its quadrant still lacks and issues ONE half-tile of
//                        loads six half-tiles ahead of its use; loads stay in flight across the raw barriers and
//                        are retired by one counted vmcnt per K-tile.  The two wave rows run one barrier slot apart,
//                        so on every SIMD one wave issues MFMAs while the other reads LDS.  Highest operand reuse:
//                        128 FLOP per byte pulled from L2, which is what bounds these GEMMs on MI355X.
//   gemm_bf16_x_kernel   (gemm_x.hip) 256x128x32 tile, 4 waves, TWO workgroups per CU covering each other's stalls.
//   gemm_bf16_kernel     (this file) 128x128x64 tile, 4 waves, plain double buffering: small-M shapes.
//
// Reference ops replaced: every nn.Linear on the path (audio_models/mae.py:51-52,116,133;
// nn.MultiheadAttention in/out projections mae.py:69-74; text_models/roberta.py:62-64,110,153,164;
// caco.py:35-37,113) and their following F.silu / F.gelu / residual adds.
#include "common.h"
#include "gemm_epilogue.h"
#include "kernels.h"

namespace caco {

namespace {

constexpr int BK = 64;             // K-tile (bf16 elements) = 128 bytes per row = 8 chunks of 16 B
constexpr int ROW_BYTES = BK * 2;

typedef __attribute__((address_space(3))) void* lds_vptr;
typedef const __attribute__((address_space(1))) void* gbl_vptr;

// chunk swizzle: 16 consecutive rows x one k-chunk land on 16 distinct 16-byte slots of the
// 256-byte LDS bank row (rows are 128 B, so two rows share a bank row).
__device__ __forceinline__ int swz(int row) { return (row >> 1) & 7; }

// Stage ROWS x 64 bf16 (ROWS*128 B) from row-major global memory into a lane-linear LDS image.
template <int ROWS, int NWAVES>
__device__ __forceinline__ void stage_tile(const bf16_t* __restrict__ g, int64_t row0, int64_t last_row, int ld, int k0,
                                           char* lds_tile, int wave, int lane) {
  constexpr int ITER = ROWS / 8 / NWAVES;
#pragma unroll
  for (int it = 0; it < ITER; ++it) {
    const int grp = it * NWAVES + wave;          // 8-row group, one wave instruction (1 KiB)
    const int r = grp * 8 + (lane >> 3);
    const int c = (lane & 7) ^ swz(r);           // source chunk that must land at LDS position (lane&7)
    int64_t grow = row0 + r;
    grow = grow < last_row ? grow : last_row;    // clamp: out-of-range rows are computed, never stored
    const bf16_t* src = g + grow * (int64_t)ld + k0 + c * 8;
    char* dst = lds_tile + grp * 1024;           // wave-uniform; hardware adds lane * 16
    __builtin_amdgcn_global_load_lds((gbl_vptr)src, (lds_vptr)dst, 16, 0, 0);
  }
}

__device__ __forceinline__ bf16x8 lds_frag(const char* tile, int row, int chunk) {
  return *reinterpret_cast<const bf16x8*>(tile + row * ROW_BYTES + ((chunk ^ swz(row)) << 4));
}

// XCD-aware, bijective block remap: consecutive logical tiles share one XCD's L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}

// One MFMA step, operands swapped: the weight fragment is the A operand, so that D[i = n][j = m] and every lane
// owns 4 CONSECUTIVE n of one output row m.
__device__ __forceinline__ f32x4 mma(const bf16x8& xa, const bf16x8& wb, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb, xa, c, 0, 0, 0);
}

// Direct (register -> global) epilogue for the small kernel.
template <int TM, int TN, int EPI, int ACT>
__device__ __forceinline__ void epilogue_direct(const f32x4 (&acc)[TM][TN], const GemmArgs& p, int64_t mw, int nw, int lane) {
  {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = nw + j * 16 + (lane >> 4) * 4;
      f32x4 b4 = f32x4{0.f, 0.f, 0.f, 0.f};
      if (p.bias) b4 = *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int64_t m = mw + i * 16 + (lane & 15);
        if (m >= p.M) continue;
        f32x4 v = acc[i][j] + b4;
        if constexpr (EPI == EPI_BF16) {
          bf16x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = (bf16_t)epi_act<ACT>(v[r]);
          *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(p.out) + m * p.ldc + n) = o;
        } else {  // EPI_F32: optional residual (may alias out), fp32 store
          float* op = reinterpret_cast<float*>(p.out) + m * p.ldc + n;
          if (p.resid) v += *reinterpret_cast<const f32x4*>(p.resid + m * p.ldc + n);
          *reinterpret_cast<f32x4*>(op) = v;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// 128x128 tile, plain double buffering
// ------------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, int EPI, int ACT>
__global__ __launch_bounds__(WM* WN * 64) void gemm_bf16_kernel(GemmArgs p) {
  constexpr int NW = WM * WN;
  constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
  constexpr int A_BYTES = BM * ROW_BYTES, B_BYTES = BN * ROW_BYTES, BUF = A_BYTES + B_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int lda = p.lda ? p.lda : p.K, ldw = p.ldw ? p.ldw : p.K;

  const int tiles_n = p.N / BN;
  const int tiles_m = (int)((p.M + BM - 1) / BM);
  const int t = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int tile_m = t / tiles_n, tile_n = t % tiles_n;
  const int64_t m0 = (int64_t)tile_m * BM;
  const int n0 = tile_n * BN;

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / BK;
  stage_tile<BM, NW>(p.A, m0, p.M - 1, lda, 0, smem, wave, lane);
  stage_tile<BN, NW>(p.W, n0, p.N - 1, ldw, 0, smem + A_BYTES, wave, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const int frow = lane & 15, fchunk = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    const char* cur = smem + (kt & 1) * BUF;
    if (kt + 1 < nk) {
      char* nxt = smem + ((kt + 1) & 1) * BUF;
      stage_tile<BM, NW>(p.A, m0, p.M - 1, lda, (kt + 1) * BK, nxt, wave, lane);
      stage_tile<BN, NW>(p.W, n0, p.N - 1, ldw, (kt + 1) * BK, nxt + A_BYTES, wave, lane);
    }
    const char* a_t = cur + (wm * (BM / WM)) * ROW_BYTES;
    const char* b_t = cur + A_BYTES + (wn * (BN / WN)) * ROW_BYTES;
#pragma unroll
    for (int kk = 0; kk < BK / 32; ++kk) {
      bf16x8 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = lds_frag(a_t, i * 16 + frow, kk * 4 + fchunk);
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = lds_frag(b_t, j * 16 + frow, kk * 4 + fchunk);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = mma(af[i], bf[j], acc[i][j]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  epilogue_direct<TM, TN, EPI, ACT>(acc, p, m0 + wm * (BM / WM), n0 + wn * (BN / WN), lane);
}

template <int BM, int BN, int WM, int WN, int EPI, int ACT>
int launch_cfg(const GemmArgs& p, hipStream_t st) {
  constexpr int smem = 2 * (BM + BN) * ROW_BYTES;
  auto kern = gemm_bf16_kernel<BM, BN, WM, WN, EPI, ACT>;
  CACO_TRY_RC(prepare_launch(reinterpret_cast<const void*>(kern), smem, nullptr));
  const int tiles = (int)((p.M + BM - 1) / BM) * (p.N / BN);
  hipLaunchKernelGGL(kern, dim3(tiles), dim3(WM * WN * 64), smem, st, p);
  return check_hip(hipGetLastError(), "gemm_bf16 launch");
}

template <int EPI, int ACT>
int launch_epi(const GemmArgs& p, hipStream_t st, int epi, int act) {
  const int cfg = gemm_tile_config();
  const int64_t tiles_x = ((p.M + 255) / 256) * (int64_t)(p.N / 128);
  if (p.fold_mr || p.xb_out || p.stats_part || p.resid_idx) {      // LayerNorm folding / gathered residual: 8-wave kernel only
    CACO_REQUIRE(gemm_bf16_w8_ok(p, epi), "gemm_bf16: LayerNorm folding / a gathered residual need N %% 256 == 0 and K %% 64 == 0");
    return gemm_bf16_w8(p, epi, act, st);
  }
  // the experimental 4-wave kernels carry the plain epilogues only (bias; bias + residual in fp32): anything else falls
  // through to the default choice, as caco_hip.h promises for a forced kernel
  const bool w4_ok = gemm_bf16_w8_ok(p, epi) && p.bias && ((epi == EPI_BF16 && !p.resid) || (epi == EPI_F32 && act == ACT_NONE));
  // forced kernels (tests, A/B runs): 8256 = persistent 256x256 (w8), 2256 = 256x128 two workgroups per CU
  if (cfg == 8256 && gemm_bf16_w8_ok(p, epi)) return gemm_bf16_w8(p, epi, act, st);
  if (cfg == 4256 && w4_ok) return gemm_bf16_w4q(p, epi, act, st);     // experiment: 4 waves of 128 x 128
  if (cfg == 4128 && w4_ok) return gemm_bf16_w4h(p, epi, act, st);     // experiment: 128 x 256 tiles (mid M)
  if (cfg == 2256) return gemm_bf16_x(p, epi, act, st);
  if (cfg != 128) {      // 256, or a forced kernel that does not carry this shape / epilogue: the default choice
    // chip-filling shapes: the 256x256 persistent 8-wave pipelined kernel (gemm_w8.hip: most reuse per L2 byte) when every
    // CU gets >= 2 tiles, else the two-workgroups-per-CU 256x128 kernel when its grid covers the chip, else 128x128.
    // Threshold in 256x128-tile units.  128 also sends the text tower's M = 8192 GEMMs to w8: in isolation the 128x128
    // kernel is faster for its N = 768 shapes, but the text tower runs NEXT TO the audio tower and a few persistent
    // 160 KiB workgroups interleave with the audio GEMMs better than many small ones (step 32.2 -> 31.7 ms measured)
    const int w8_min = sw(SW_W8_MIN_TILES);
    // CACO_W4H_MAX_TILES=<n> (experiment, round 3): shapes with fewer than n 256 x 256 tiles - the text tower's N = 768 GEMMs
    // have 96 - take 128 x 256 tiles instead (gemm_w4h.hip)
    const int w4h_max = sw(SW_W4H_MAX_TILES);
    if (w4h_max > 0 && w4_ok && tiles_x >= w8_min && ((p.M + 255) / 256) * (int64_t)(p.N / 256) < w4h_max)
      return gemm_bf16_w4h(p, epi, act, st);
    if (gemm_bf16_w8_ok(p, epi) && tiles_x >= w8_min) return gemm_bf16_w8(p, epi, act, st);
    if (tiles_x >= 256) return gemm_bf16_x(p, epi, act, st);
  }
  return launch_cfg<128, 128, 2, 2, EPI, ACT>(p, st);
}

}  // namespace

// the dispatch rule of launch_epi (same order of tests) for callers that only take a w8-specific form when w8 would run the
// plain form anyway.  A forced 4-wave kernel (4256 / 4128) that does not carry the shape / epilogue falls through to the
// default choice there, and so does this function.
bool gemm_bf16_picks_w8(const GemmArgs& p, int epi) {
  const int cfg = gemm_tile_config();
  if (!gemm_bf16_w8_ok(p, epi) || p.K % BK != 0 || p.N % 128 != 0) return false;
  if (cfg == 8256) return true;
  if (cfg == 128 || cfg == 2256) return false;
  const bool w4_ok = p.bias && ((epi == EPI_BF16 && !p.resid) || epi == EPI_F32);
  if ((cfg == 4256 || cfg == 4128) && w4_ok) return false;          // the forced 4-wave kernel carries it
  const int64_t tiles_x = ((p.M + 255) / 256) * (int64_t)(p.N / 128);
  const int w8_min = sw(SW_W8_MIN_TILES), w4h_max = sw(SW_W4H_MAX_TILES);
  if (w4h_max > 0 && w4_ok && tiles_x >= w8_min && ((p.M + 255) / 256) * (int64_t)(p.N / 256) < w4h_max) return false;
  return tiles_x >= w8_min;
}

static int g_tile_cfg = -1;
int gemm_tile_config() {
  if (g_tile_cfg < 0) g_tile_cfg = 256;
  return g_tile_cfg;
}
int set_gemm_tile_config(int tile) {
  if (tile == 128 || tile == 256 || tile == 2256 || tile == 8256 || tile == 4256 || tile == 4128) g_tile_cfg = tile;   // > 256: force a kernel
  return gemm_tile_config();
}

int gemm_bf16(const GemmArgs& p, int epi, int act, hipStream_t st) {
  CACO_REQUIRE(p.K % BK == 0 && p.K >= BK, "gemm_bf16: K=%d must be a positive multiple of %d", p.K, BK);
  CACO_REQUIRE(p.N % 128 == 0, "gemm_bf16: N=%d must be a multiple of 128", p.N);
  CACO_REQUIRE(p.M > 0, "gemm_bf16: M=%lld must be positive", (long long)p.M);
  CACO_REQUIRE(p.A && p.W && p.out, "gemm_bf16: null operand");
  CACO_REQUIRE((p.lda == 0 || (p.lda >= p.K && p.lda % 8 == 0)) && (p.ldw == 0 || (p.ldw >= p.K && p.ldw % 8 == 0)),
               "gemm_bf16: operand row strides must be >= K and multiples of 8 elements");
  if (epi == EPI_BF16) {
    if (act == ACT_NONE) return launch_epi<EPI_BF16, ACT_NONE>(p, st, epi, act);
    if (act == ACT_SILU) return launch_epi<EPI_BF16, ACT_SILU>(p, st, epi, act);
    if (act == ACT_GELU) return launch_epi<EPI_BF16, ACT_GELU>(p, st, epi, act);
  } else if (epi == EPI_F32) {
    if (act == ACT_NONE) return launch_epi<EPI_F32, ACT_NONE>(p, st, epi, act);
  }
  set_error("gemm_bf16: unsupported epilogue %d / activation %d", epi, act);
  return CACO_ERR_INVALID;
}

// ------------------------------------------------------------------------------------------------
// exact fp32 GEMM on v_mfma_f32_32x32x2_f32 (bitwise an fmaf chain): out = scale * A x B^T + bias.
// Used for the [B,768] projections after pooling and for the audio x text similarity matrix, where
// rounding operands to bf16 would eat into the 1e-3 parity budget and the FLOPs are negligible.
// One wave per 32x32 output tile; each lane streams float4 along K straight from global memory.
// ------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                       const float* __restrict__ bias, float* __restrict__ C, int M,
                                                       int N, int K, int ldc, float scale, int lda, int ldb) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int tiles_n = (N + 31) / 32;
  const int tile = blockIdx.x * 4 + wave;
  const int tm = tile / tiles_n, tn = tile % tiles_n;
  if (tm * 32 >= M) return;
  const int ra = min(tm * 32 + (lane & 31), M - 1);
  const int rb = min(tn * 32 + (lane & 31), N - 1);
  const int half = lane >> 5;
  const float* ap = A + (int64_t)ra * lda + half * 4;
  const float* bp = B + (int64_t)rb * ldb + half * 4;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int k = 0; k + 8 <= K; k += 8) {
    const f32x4 a4 = *reinterpret_cast<const f32x4*>(ap + k);
    const f32x4 b4 = *reinterpret_cast<const f32x4*>(bp + k);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j], b4[j], acc, 0, 0, 0);
  }
  const int n = tn * 32 + (lane & 31);
  if (n >= N) return;
  const float bs = bias ? bias[n] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
    if (m < M) C[(int64_t)m * ldc + n] = acc[r] * scale + bs;
  }
}
}  // namespace

int gemm_f32(const float* A, const float* B, const float* bias, float* C, int M, int N, int K, int ldc, float scale,
             hipStream_t st, int lda, int ldb) {
  CACO_REQUIRE(M > 0 && N > 0 && K > 0 && (K % 8 == 0), "gemm_f32: need M,N > 0 and K %% 8 == 0 (got %d,%d,%d)", M, N, K);
  CACO_REQUIRE(ldc >= N, "gemm_f32: ldc %d < N %d", ldc, N);
  const int tiles = ((M + 31) / 32) * ((N + 31) / 32);
  CACO_REQUIRE(lda == 0 || (lda >= K && lda % 4 == 0), "gemm_f32: lda %d must be >= K and a multiple of 4", lda);
  CACO_REQUIRE(ldb == 0 || (ldb >= K && ldb % 4 == 0), "gemm_f32: ldb %d must be >= K and a multiple of 4", ldb);
  hipLaunchKernelGGL(gemm_f32_kernel, dim3((tiles + 3) / 4), dim3(256), 0, st, A, B, bias, C, M, N, K, ldc, scale, lda ? lda : K, ldb ? ldb : K);
  return check_hip(hipGetLastError(), "gemm_f32 launch");
}

}  // namespace caco
