// bf16 MFMA GEMM for the encoder stacks, and an exact-fp32 MFMA GEMM for the small scoring ops.
//
//   out[M,N] = epilogue( A[M,K] (bf16, row-major) x W[N,K]^T (bf16, torch Linear layout) )
//
// Both operands are K-contiguous, which is the natural MFMA feed on CDNA4: every lane's 8-element
// fragment is one 16-byte LDS read.  Tiles are staged HBM -> LDS with 16-byte direct loads
// (global_load_lds_dwordx4); the LDS image is lane-linear, so the bank-conflict swizzle is applied
// to the per-lane SOURCE address and undone on the ds_read side (same involution on both sides).
// Accumulation is fp32; bias / residual / SiLU / erf-GELU are fused into the epilogue so the
// activations make one HBM round trip per GEMM.
//
// Reference ops replaced: every nn.Linear on the path (audio_models/mae.py:51-52,116,133;
// nn.MultiheadAttention in/out projections mae.py:69-74; text_models/roberta.py:62-64,110,153,164;
// caco.py:35-37,113) and their following F.silu / F.gelu / residual adds.
#include "common.h"
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

template <int ACT>
__device__ __forceinline__ float apply_act(float x) {
  if constexpr (ACT == ACT_SILU) return silu_f(x);
  if constexpr (ACT == ACT_GELU) return gelu_erf_f(x);
  return x;
}

template <int BM, int BN, int WM, int WN, int EPI, int ACT>
__global__ __launch_bounds__(WM* WN * 64) void gemm_bf16_kernel(GemmArgs p) {
  constexpr int NW = WM * WN;
  constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
  constexpr bool SWAP = (EPI != EPI_VT);       // SWAP: lane owns 4 consecutive n of one row m
  constexpr int A_BYTES = BM * ROW_BYTES, B_BYTES = BN * ROW_BYTES, BUF = A_BYTES + B_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

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
  stage_tile<BM, NW>(p.A, m0, p.M - 1, p.K, 0, smem, wave, lane);
  stage_tile<BN, NW>(p.W, n0, p.N - 1, p.K, 0, smem + A_BYTES, wave, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const int frow = lane & 15, fchunk = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    const char* cur = smem + (kt & 1) * BUF;
    if (kt + 1 < nk) {
      char* nxt = smem + ((kt + 1) & 1) * BUF;
      stage_tile<BM, NW>(p.A, m0, p.M - 1, p.K, (kt + 1) * BK, nxt, wave, lane);
      stage_tile<BN, NW>(p.W, n0, p.N - 1, p.K, (kt + 1) * BK, nxt + A_BYTES, wave, lane);
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
        for (int j = 0; j < TN; ++j) {
          if constexpr (SWAP)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j], af[i], acc[i][j], 0, 0, 0);
          else
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ------------------------------------------------------------------ epilogue
  const int64_t mw = m0 + wm * (BM / WM);
  const int nw = n0 + wn * (BN / WN);
  if constexpr (SWAP) {
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
          for (int r = 0; r < 4; ++r) o[r] = (bf16_t)apply_act<ACT>(v[r]);
          *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(p.out) + m * p.ldc + n) = o;
        } else {  // EPI_F32: optional residual (may alias out), fp32 store
          float* op = reinterpret_cast<float*>(p.out) + m * p.ldc + n;
          if (p.resid) v += *reinterpret_cast<const f32x4*>(p.resid + m * p.ldc + n);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = apply_act<ACT>(v[r]);
          *reinterpret_cast<f32x4*>(op) = v;
        }
      }
    }
  } else {
    // transposed per-clip store: vt[b][n][s] <- (m = b*S + s, n); lane owns 4 consecutive m of one n
    bf16_t* vt = reinterpret_cast<bf16_t*>(p.out);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = nw + j * 16 + (lane & 15);
      const float bs = p.bias ? p.bias[n] : 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int64_t m4 = mw + i * 16 + (lane >> 4) * 4;
        if (m4 >= p.M) continue;
        const int b = (int)(m4 / p.seq), s = (int)(m4 % p.seq);
        f32x4 v = acc[i][j];
        if (s + 3 < p.seq && (p.seq & 3) == 0) {
          bf16x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = (bf16_t)(v[r] + bs);
          *reinterpret_cast<bf16x4*>(vt + ((int64_t)b * p.N + n) * p.seq_pad + s) = o;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int64_t m = m4 + r;
            if (m < p.M) {
              const int bb = (int)(m / p.seq), ss = (int)(m % p.seq);
              vt[((int64_t)bb * p.N + n) * p.seq_pad + ss] = (bf16_t)(v[r] + bs);
            }
          }
        }
      }
    }
  }
}

template <int BM, int BN, int WM, int WN, int EPI, int ACT>
int launch_cfg(const GemmArgs& p, hipStream_t st) {
  constexpr int smem = 2 * (BM + BN) * ROW_BYTES;
  auto kern = gemm_bf16_kernel<BM, BN, WM, WN, EPI, ACT>;
  static bool attr_done = false;   // per instantiation; benign race (idempotent)
  if (!attr_done) {
    CACO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    attr_done = true;
  }
  const int tiles = (int)((p.M + BM - 1) / BM) * (p.N / BN);
  hipLaunchKernelGGL(kern, dim3(tiles), dim3(WM * WN * 64), smem, st, p);
  return check_hip(hipGetLastError(), "gemm_bf16 launch");
}

template <int EPI, int ACT>
int launch_epi(const GemmArgs& p, hipStream_t st) {
  const int cfg = gemm_tile_config();
  if (cfg == 256 && p.N % 256 == 0 && p.M >= 2048) return launch_cfg<256, 256, 2, 4, EPI, ACT>(p, st);
  return launch_cfg<128, 128, 2, 2, EPI, ACT>(p, st);
}

}  // namespace

static int g_tile_cfg = -1;
int gemm_tile_config() {
  if (g_tile_cfg < 0) {
    const char* e = getenv("CACO_GEMM_TILE");
    g_tile_cfg = (e && atoi(e) == 128) ? 128 : 256;
  }
  return g_tile_cfg;
}
int set_gemm_tile_config(int tile) {
  if (tile == 128 || tile == 256) g_tile_cfg = tile;
  return gemm_tile_config();
}

int gemm_bf16(const GemmArgs& p, int epi, int act, hipStream_t st) {
  CACO_REQUIRE(p.K % BK == 0 && p.K >= BK, "gemm_bf16: K=%d must be a positive multiple of %d", p.K, BK);
  CACO_REQUIRE(p.N % 128 == 0, "gemm_bf16: N=%d must be a multiple of 128", p.N);
  CACO_REQUIRE(p.M > 0, "gemm_bf16: M=%lld must be positive", (long long)p.M);
  CACO_REQUIRE(p.A && p.W && p.out, "gemm_bf16: null operand");
  if (epi == EPI_BF16) {
    if (act == ACT_NONE) return launch_epi<EPI_BF16, ACT_NONE>(p, st);
    if (act == ACT_SILU) return launch_epi<EPI_BF16, ACT_SILU>(p, st);
    if (act == ACT_GELU) return launch_epi<EPI_BF16, ACT_GELU>(p, st);
  } else if (epi == EPI_F32) {
    if (act == ACT_NONE) return launch_epi<EPI_F32, ACT_NONE>(p, st);
  } else if (epi == EPI_VT) {
    CACO_REQUIRE(p.seq > 0 && p.seq_pad >= p.seq, "gemm_bf16: bad seq / seq_pad for the transposed store");
    if (act == ACT_NONE) return launch_epi<EPI_VT, ACT_NONE>(p, st);
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
                                                       int N, int K, int ldc, float scale) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int tiles_n = (N + 31) / 32;
  const int tile = blockIdx.x * 4 + wave;
  const int tm = tile / tiles_n, tn = tile % tiles_n;
  if (tm * 32 >= M) return;
  const int ra = min(tm * 32 + (lane & 31), M - 1);
  const int rb = min(tn * 32 + (lane & 31), N - 1);
  const int half = lane >> 5;
  const float* ap = A + (int64_t)ra * K + half * 4;
  const float* bp = B + (int64_t)rb * K + half * 4;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  int k = 0;
  for (; k + 8 <= K; k += 8) {
    const f32x4 a4 = *reinterpret_cast<const f32x4*>(ap + k);
    const f32x4 b4 = *reinterpret_cast<const f32x4*>(bp + k);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j], b4[j], acc, 0, 0, 0);
  }
  for (; k < K; k += 2) {  // K tail (K even is required by the host wrapper)
    const float a1 = A[(int64_t)ra * K + k + half], b1 = B[(int64_t)rb * K + k + half];
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc, 0, 0, 0);
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
             hipStream_t st) {
  CACO_REQUIRE(M > 0 && N > 0 && K > 0 && (K % 8 == 0), "gemm_f32: need M,N > 0 and K %% 8 == 0 (got %d,%d,%d)", M, N, K);
  CACO_REQUIRE(ldc >= N, "gemm_f32: ldc %d < N %d", ldc, N);
  const int tiles = ((M + 31) / 32) * ((N + 31) / 32);
  hipLaunchKernelGGL(gemm_f32_kernel, dim3((tiles + 3) / 4), dim3(256), 0, st, A, B, bias, C, M, N, K, ldc, scale);
  return check_hip(hipGetLastError(), "gemm_f32 launch");
}

}  // namespace caco
