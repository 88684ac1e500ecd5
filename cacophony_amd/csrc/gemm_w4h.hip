// gemm_bf16_w4h: the persistent bf16 GEMM of gemm_w8.hip on 128 (M) x 256 (N) x 64 tiles with FOUR waves (1 x 4), each
// owning the same 128 x 64 part as a wave of w8 - same fragments, same MFMA phases, same epilogue.  A "mid-M" path
// (round-2 verdict item 5): at M = 8192 (the text tower at batch 256) the N = 768 GEMMs are 96 tiles of 256 x 256 - 37 % of
// the CUs get one tile, the rest none - but 192 tiles of 128 x 256.  The price: one 112 KiB workgroup per CU = ONE wave per
// SIMD (nobody fills this wave's stalls), and every weight K-tile is loaded for 128 rows instead of 256 (85 FLOP per byte
// of L2 -> LDS traffic instead of 128).
// Written in round 3 WITHOUT a GPU: tile code 4128 (caco_set_gemm_tile) or CACO_W4H_MAX_TILES=<n> (used instead of w8 when the
// shape has fewer than n 256 x 256 tiles); never picked by default; verified on the wavesim build - not yet timed.
//
//   LDS      A ring 3 x 16 KiB (128 rows x 128 B), W ring 2 x 32 KiB = 112 KiB; the free A slot is the four waves' 4 KiB slabs
//   K-loop   gemm_w8.hip's: P0 P1 P2 | vmcnt(4) lgkmcnt(0) s_barrier | P3, four A pieces per wave and K-tile (as in w8) and
//            eight W pieces (w8: four)
//   epilogue gemm_w8_epilogue.h, plain forms (bias; bias + residual)
// a kernel that has not run on hardware yet: the intra-wave LDS hand-offs are also fenced for the compiler (common.h)
#define CACO_WAVE_SYNC_FENCE 1
#include "common.h"
#include "kernels.h"
#include "gemm_w8_common.h"
#include "gemm_w8_epilogue.h"

namespace caco {
namespace {

constexpr int H_ASLOT = 128 * WROWB;          // 16 KiB
constexpr int H_AOFF = 0;
constexpr int H_WOFF = 3 * H_ASLOT;           // 48 KiB
constexpr int H_SMEM = 3 * H_ASLOT + 2 * W_SLOT;   // 112 KiB

struct HCurA {
  __amdgpu_buffer_rsrc_t r;
  int voff[4];        // four 32-row groups of the 128-row tile
  int li, kt;
};

__device__ __forceinline__ void h_setup_a(HCurA& C, const GemmArgs& p, int t, int tiles_n, int tiles_m, int lda, int wave, int lane) {
  int tm, tn;
  w4_decode(t, tiles_n, tiles_m, p.ngroup, tm, tn);
  const int64_t m0 = (int64_t)tm * 128;
  C.r = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + m0 * lda), 0, 0x7fffffff, 0x00020000);
  const int r8 = wave * 8 + (lane >> 3);
  const int chunk = (lane & 7) ^ ((wave * 4 + (lane >> 4)) & 7);
  const int last = (int)min((int64_t)128, p.M - m0) - 1;
#pragma unroll
  for (int it = 0; it < 4; ++it) C.voff[it] = min(it * 32 + r8, last) * lda * 2 + chunk * 16;
}
__device__ __forceinline__ void h_setup_w(W4CurW& C, const GemmArgs& p, int t, int tiles_n, int tiles_m, int ldw, int wave, int lane) {
  int tm, tn;
  w4_decode(t, tiles_n, tiles_m, p.ngroup, tm, tn);
  C.r = __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + (int64_t)tn * 256 * ldw), 0, 0x7fffffff, 0x00020000);
  const int r8 = wave * 8 + (lane >> 3);
  const int chunk = (lane & 7) ^ ((wave * 4 + (lane >> 4)) & 7);
  C.voff = r8 * ldw * 2 + chunk * 16;
}
__device__ __forceinline__ void h_piece_a(const HCurA& C, int it, char* slot, int wave) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(C.r, (lds_vptr)(slot + (it * 4 + wave) * 1024), 16, C.voff[it], W4_KOFF(C.kt), 0, W8_A_AUX);
}

#define H16_MFMAS(XC, WC, IB)                                                                               \
  _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_)                                                          \
  _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                                          \
    acc[(IB) * 4 + q_][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WC[j_], XC[q_], acc[(IB) * 4 + q_][j_], 0, 0, 0);
#define H16_MFMAS_Z(XC, WC, IB)                                                                               \
  _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_)                                                          \
  _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                                          \
    acc[(IB) * 4 + q_][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WC[j_], XC[q_], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#define H16_SCHED(NRD, VM_MASK)                                                                             \
  _Pragma("unroll") for (int n_ = 0; n_ < 16; ++n_) {                                                       \
    __builtin_amdgcn_sched_group_barrier(W4_SGB_MFMA, 1, 0);                                                \
    if (n_ < (NRD)) __builtin_amdgcn_sched_group_barrier(W4_SGB_DSRD, 1, 0);                                \
    if (((VM_MASK) >> n_) & 1) __builtin_amdgcn_sched_group_barrier(W4_SGB_VMEM, 1, 0);                     \
  }                                                                                                         \
  __builtin_amdgcn_sched_barrier(0);

template <int EPI, int ACT, int MODE>
__device__ __forceinline__ void h16_body(const GemmArgs& p, char* smem) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave;                                   // 1 x 4 waves: every wave owns all 128 rows, 64 of the 256 columns
  const int lda = p.lda ? p.lda : p.K, ldw = p.ldw ? p.ldw : p.K;

  const int tiles_n = p.N / 256;
  const int tiles_m = (int)((p.M + 127) / 128);
  const int nwg = tiles_m * tiles_n;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
  const int q = nwg >> 3, r = nwg & 7;
  const int cnt = q + (xcd < r ? 1 : 0);
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  if (slot >= cnt) return;
  const int nk = p.K / WBK;

  const int l16 = lane & 15, lq = lane >> 4;
  const int w_off = wn * 64 * WROWB;

  HCurA CA;
  W4CurW CW;
  CA.li = CW.li = slot;
  CA.kt = CW.kt = 0;
  h_setup_a(CA, p, base + slot, tiles_n, tiles_m, lda, wave, lane);
  h_setup_w(CW, p, base + slot, tiles_n, tiles_m, ldw, wave, lane);
  auto advance_a = [&]() {
    if (++CA.kt == nk) {
      CA.kt = 0;
      if (CA.li + slots < cnt) { CA.li += slots; h_setup_a(CA, p, base + CA.li, tiles_n, tiles_m, lda, wave, lane); }
    }
  };
  auto advance_w = [&]() {
    if (++CW.kt == nk) {
      CW.kt = 0;
      if (CW.li + slots < cnt) { CW.li += slots; h_setup_w(CW, p, base + CW.li, tiles_n, tiles_m, ldw, wave, lane); }
    }
  };

  int a_c = H_AOFF, a_1 = H_AOFF + H_ASLOT, a_2 = H_AOFF + 2 * H_ASLOT;
  int w_c = H_WOFF, w_1 = H_WOFF + W_SLOT;

  // prologue: A(0) W(0) | A(1) W(1): 4 + 8 pieces per wave and K-tile
#pragma unroll
  for (int it = 0; it < 4; ++it) h_piece_a(CA, it, smem + a_c, wave);
  advance_a();
#pragma unroll
  for (int it = 0; it < 8; ++it) w4_piece_w<4>(CW, it, ldw, smem + w_c, wave);
  advance_w();
#pragma unroll
  for (int it = 0; it < 4; ++it) h_piece_a(CA, it, smem + a_1, wave);
  advance_a();
#pragma unroll
  for (int it = 0; it < 8; ++it) w4_piece_w<4>(CW, it, ldw, smem + w_1, wave);
  advance_w();
  asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#define H16_F(OPER, BLK, S) w4_frag(OPER, (BLK) * 16 + l16, (S) * 4 + lq)
  bf16x8 xa[4], xb[4], wc[4], wn_[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) xa[i] = H16_F(smem + a_c, i, 0);
#pragma unroll
  for (int j = 0; j < 4; ++j) wc[j] = H16_F(smem + w_c + w_off, j, 0);

  constexpr int NST = (EPI == EPI_BF16) ? 16 : 32;
  bool stores_pending = false;
  int c_li = slot;
  while (true) {
    f32x4 acc[8][4];          // first touched by the peeled K-tile's MFMAs with C = 0

    {
#define H16_MF_FIRST H16_MFMAS_Z
#include "gemm_w4h_ktile.inc"
#undef H16_MF_FIRST
    }
    for (int kt = 1; kt < nk; ++kt) {
#define H16_MF_FIRST H16_MFMAS
#include "gemm_w4h_ktile.inc"
#undef H16_MF_FIRST
    }
    const int t = base + c_li;
    int tm_, tn_;
    w4_decode(t, tiles_n, tiles_m, p.ngroup, tm_, tn_);
    w16_epilogue<EPI, ACT, MODE>(acc, p, (int64_t)tm_ * 128, tn_ * 256, 0, wn, lane, smem + a_2 + wave * 4096);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < 4; ++i) xa[i] = H16_F(smem + a_c, i, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) wc[j] = H16_F(smem + w_c + w_off, j, 0);
    stores_pending = (EPI == EPI_F32);
    c_li += slots;
    if (c_li >= cnt) break;
  }
}

template <int EPI, int ACT, int MODE>
// (the register budget is held to 256 as in w8: with 512 allowed, hipcc puts part of the accumulators into AGPRs and copies
// them in and out around the MFMAs - 127 VALU per K-tile instead of 23; occupancy is one workgroup per CU either way: LDS)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_bf16_w4h_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  h16_body<EPI, ACT, MODE>(p, smem);
}

template <int EPI, int ACT, int MODE>
int launch_w4h(const GemmArgs& p, hipStream_t st) {
  void (*kern)(GemmArgs) = gemm_bf16_w4h_kernel<EPI, ACT, MODE>;
  int num_cu = 0;
  CACO_TRY_RC(prepare_launch(reinterpret_cast<const void*>(kern), H_SMEM, &num_cu));
  const int tiles = (int)((p.M + 127) / 128) * (p.N / 256);
  const int grid = tiles < num_cu ? (tiles + 7) / 8 * 8 : num_cu / 8 * 8;
  GemmArgs q = p;
  q.ngroup = p.N / 256;                     // one group: the shapes this kernel is for have at most a few n-tiles per M panel
  if (p.reverse) q.ngroup = -q.ngroup;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), H_SMEM, st, q);
  return check_hip(hipGetLastError(), "gemm_bf16_w4h launch");
}

}  // namespace

int gemm_bf16_w4h(const GemmArgs& p, int epi, int act, hipStream_t st) {
  CACO_REQUIRE(gemm_bf16_w8_ok(p, epi), "gemm_bf16_w4h: shape not supported");
  CACO_REQUIRE(p.bias && !p.fold_mr && !p.xb_out && !p.stats_part && !p.resid_idx, "gemm_bf16_w4h: plain epilogues only");
  if (epi == EPI_BF16 && !p.resid) {
    if (act == ACT_NONE) return launch_w4h<EPI_BF16, ACT_NONE, 1>(p, st);
    if (act == ACT_SILU) return launch_w4h<EPI_BF16, ACT_SILU, 1>(p, st);
    if (act == ACT_GELU) return launch_w4h<EPI_BF16, ACT_GELU, 1>(p, st);
  }
  if (epi == EPI_F32 && act == ACT_NONE) {
    if (p.resid) return launch_w4h<EPI_F32, ACT_NONE, 2>(p, st);
    return launch_w4h<EPI_F32, ACT_NONE, 1>(p, st);
  }
  set_error("gemm_bf16_w4h: unsupported epilogue %d / activation %d", epi, act);
  return CACO_ERR_INVALID;
}

}  // namespace caco
