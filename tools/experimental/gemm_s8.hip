// NOT BUILT since the end of round 2 (the product ships what wins).  To try it again: copy this file back
// to cacophony_amd/csrc/, add it to SOURCES in cacophony_amd/build.py, declare gemm_bf16_s8_ok / gemm_bf16_s8 in kernels.h
// and give launch_epi (gemm.hip) a forced mode for it; tools/experimental/s8_timing.py reads its -DS8_TIMING stamps.
// Its test (every element, three repetitions, 1-6 panels per team) was tests/test_gpu_ops.py::test_gemm_skewed_row_groups
// at commit 354541b.
//
// gemm_bf16_s8: 256x256x64 bf16 MFMA GEMM whose epilogue runs UNDER ITS OWN K-LOOP ("skewed row groups").
//   out[M,N] = epilogue( A[M,K] (bf16) x W[N,K]^T (bf16) ), fp32 accumulate on v_mfma_f32_32x32x16_bf16.
//
// Why (tools/w8_timing.py): in the 256x256 persistent kernel of gemm_w4.hip (w8) a tile's epilogue takes
// 25-30 % of the tile period on the bf16 shapes and ~50 % on the fp32 + residual shapes.  It is bound by the CU's own
// store path (~16 B/clk/CU: 128 wave stores of 1 KiB per bf16 tile), not by HBM (the same with 64 CUs active), and runs
// with the matrix pipe idle because all eight waves hold one accumulator set that the next tile needs at once.  Two
// workgroups per CU (gemm_d4.hip) or half-size tiles with two accumulator sets (gemm_v8.hip) buy the overlap with 1.5x the
// operand traffic and lose more than they gain.  This kernel keeps the 256x256 tile and ONE accumulator set:
//
//   * a workgroup keeps ONE n-tile for its whole run and walks a list of 256-row panels, so the W operand stream is
//     periodic: K-tile k of every panel period is the same W[n-tile, k];
//   * each wave's 128 x 64 part is four 32-row groups; row group i accumulates K-tiles 3i .. 11 of panel period j and
//     K-tiles 0 .. 3i-1 of period j+1 (the A pieces of a row group are simply fetched from the panel it is working on, W[k]
//     is what every group needs at that moment anyway).  Row group i therefore completes a tile at the START of K-tile 3i
//     of each period - one group every three K-tiles, never all four together;
//   * the 32 finished accumulators are copied to 32 spare registers and converted / transposed / stored as filler
//     instructions in the slots of the following three K-tiles, while the group's own registers restart from the bias block
//     (bias enters through one extra 32x32x8 MFMA per block: bf16 hi + lo parts times a ones fragment).
//
// MEASURED (tools/s8_timing.py, gemm_bench.py --tile 6256, round 2): correct (tests/test_gpu_ops.py::test_gemm_skewed_row_groups)
// and a steady-state panel period of 31.5 k (QKV) / 33.8 k (fc1 + SiLU) cycles against 38.1 k / 39.6 k for w8 - but a run pays
// one extra period for the idle row groups of its first and last period (1/19, 1/25), and the whole launch lands at
// 486 us vs 498 us (QKV) and 610 vs 604 us (fc1): a draw, so launch_epi does not select it (caco_set_gemm_tile(6256) does).
// What the experiment established: in this K-loop a wave's in-order ISSUE is as saturated as the matrix pipe (MFMA + fragment
// read + DMA piece + scalar bookkeeping fill the ~64 cycles a wave has per MFMA of its own with two waves per SIMD), and the
// two waves of a SIMD run the same slot at the same time, so every filler instruction adds its issue time to the K-tile
// (ablations: each of parking copies / activation / read-back / stores added 1:1); the epilogue can be moved under the
// K-loop but not made free.  Kept as the measurement behind DESIGN.md 4.1; at the 1400 W power cap the saved cycles come back as a lower clock (power_probe: 587 vs 592 us at 1.68 vs 1.75 GHz).
//
//   waves     8 = 2 (M) x 4 (N), wave tile 128 x 64, two per SIMD
//   LDS       A ring 2 x 32 KiB, W ring 2 x 32 KiB, 8 slabs x 4 KiB = 160 KiB; lane-linear images, bank swizzle
//             (chunk ^ ((row >> 1) & 7)) on the DMA source and on the ds_read_b128 side
//   K-loop    rotated: ks0 ks1 ks2 | s_waitcnt ; s_barrier | ks3; every 16-deep step is eight fenced slots (one MFMA each)
//   DMA       per wave and K-tile g: ks0: W(g+1) x4, ks3: A(g+2) x4 (the slot of A(g) is free after the barrier)
//   drain     steps 1..7 after an event: activation + packing + one ds_write_b64 per 8-column group; steps 8..11: one
//             128-byte-row read-back and one store per step
//   schedule  logical workgroup w (XCD-major) -> team w / tiles_n, column w % tiles_n; team t owns panels t, t + T, ...:
//             the tiles_n workgroups of a team share each A panel through their XCD's L2 at the same time
//
// Shapes: M % 256 == 0, N % 256 == 0, K = 768 (twelve K-tiles per period, all unrolled), bf16 output with bias (+ SiLU);
// everything else goes to the w8 kernel.
//
// Reference ops replaced: nn.Linear + activation (audio_models/mae.py:55-61,69-74,92-97; text_models/roberta.py:62-64,153).
#include "common.h"
#include "kernels.h"

namespace caco {
namespace {

constexpr int SBK = 64;
constexpr int SROWB = SBK * 2;                 // 128 bytes per row per K-tile
constexpr int S_SLOT = 256 * SROWB;            // 32 KiB: one operand's K-tile
constexpr int S_AOFF = 0;
constexpr int S_WOFF = 2 * S_SLOT;
constexpr int S_SLAB_OFF = 4 * S_SLOT;
constexpr int S_SLAB = 4096;
constexpr int S_SMEM = S_SLAB_OFF + 8 * S_SLAB;   // 163840 = 160 KiB

typedef __attribute__((address_space(3))) void* lds_vptr;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));


template <int ACT>
__device__ __forceinline__ float s8_act(float x) {
  if constexpr (ACT == ACT_SILU) return silu_f(x);
  if constexpr (ACT == ACT_GELU) return gelu_erf_f(x);
  return x;
}

// Park one finished accumulator: a REAL register copy (early-clobber output).  With a plain assignment the compiler renames
// instead of copying, every row group's accumulators end a period in other registers than they started it in, and the
// loop back-edge then rotates 160 registers through scratch memory.
__device__ __forceinline__ void s8_park1(f32x16& d, const f32x16& a, int e, bool on) {
#ifdef S8_NOPARK
  on = false;
#endif
  if (on) {
    float t;
    asm volatile("v_mov_b32 %0, %1" : "=&v"(t) : "v"(a[e]));
    d[e] = t;
  }
}

__device__ __forceinline__ bf16x8 s8_frag(const char* oper, int row, int chunk) {
  return *reinterpret_cast<const bf16x8*>(oper + row * SROWB + ((chunk ^ ((row >> 1) & 7)) << 4));
}

// One 16-deep step = EIGHT FENCED SLOTS.  A slot is one MFMA, at most one fragment read for the next step, at most two DMA
// pieces and one small piece of a finished row group's epilogue (F(n)); __builtin_amdgcn_sched_barrier(0) after every slot
// pins that interleave.  (sched_group_barrier pipelines were not honoured here: with VALU filler present the MFMAs of a
// step ended up bunched behind it, and since the two waves of a SIMD run the same step at the same time - one barrier per
// K-tile - a bunched wave leaves the matrix pipe idle instead of leaving it to its partner.)
// The W fragments are double-buffered (WC current, WN next); the four A fragments are SINGLE-buffered: x[i] is re-read for
// the next step right behind the two MFMAs that consume it (row-major MFMA order), six MFMA slots before its next use.
// ZI: row group whose accumulators restart in this step (-1: none): its two blocks start from the bias block.
#define S8_MF(I, J) acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WC_[J], x[I], acc[I][J], 0, 0, 0);
#define S8_FENCE __builtin_amdgcn_sched_barrier(0);
// plain step: row groups in the order G0 G1 G2 G3 (a permutation of 0..3)
#define S8_STEP_G(WN, XA, WW, KS, WC, G0, G1, G2, G3, DMA0, DMA1, DMA2, DMA3, F)                  \
  {                                                                                              \
    const bf16x8(&WC_)[2] = WC;                                                                  \
    S8_MF(G0, 0) WN[0] = s8_frag(WW, 0 * 32 + frow, (KS) * 2 + fhalf); DMA0; F(0); S8_FENCE      \
    S8_MF(G0, 1) x[G0] = s8_frag(XA, (G0) * 32 + frow, (KS) * 2 + fhalf); DMA1; F(1); S8_FENCE   \
    S8_MF(G1, 0) WN[1] = s8_frag(WW, 1 * 32 + frow, (KS) * 2 + fhalf); DMA2; F(2); S8_FENCE      \
    S8_MF(G1, 1) x[G1] = s8_frag(XA, (G1) * 32 + frow, (KS) * 2 + fhalf); DMA3; F(3); S8_FENCE   \
    S8_MF(G2, 0) F(4); S8_FENCE                                                                  \
    S8_MF(G2, 1) x[G2] = s8_frag(XA, (G2) * 32 + frow, (KS) * 2 + fhalf); F(5); S8_FENCE         \
    S8_MF(G3, 0) F(6); S8_FENCE                                                                  \
    S8_MF(G3, 1) x[G3] = s8_frag(XA, (G3) * 32 + frow, (KS) * 2 + fhalf); F(7); S8_FENCE         \
  }
#define S8_STEP(WN, XA, WW, KS, WC, DMA0, DMA1, DMA2, DMA3, F) S8_STEP_G(WN, XA, WW, KS, WC, 0, 1, 2, 3, DMA0, DMA1, DMA2, DMA3, F)
// restart step of row group EV (ks0 of its event K-tile): the other three groups first, while EV's 32 finished accumulators
// are copied to the parking registers (5-6 copies per slot), then EV's two blocks restart from the bias block
#define S8_PARK(EV, E0, E1) \
  _Pragma("unroll") for (int e_ = (E0); e_ < (E1); ++e_) { s8_park1(d0, acc[EV][0], e_ & 15, e_ < 16); s8_park1(d1, acc[EV][1], e_ & 15, e_ >= 16); }
#define S8_STEP_RESTART(WN, XA, WW, KS, WC, EV, DMA0, DMA1, DMA2, DMA3, F)                        \
  {                                                                                              \
    const bf16x8(&WC_)[2] = WC;                                                                  \
    constexpr int G0_ = ((EV) + 1) & 3, G1_ = ((EV) + 2) & 3, G2_ = ((EV) + 3) & 3;              \
    S8_MF(G0_, 0) WN[0] = s8_frag(WW, 0 * 32 + frow, (KS) * 2 + fhalf); DMA0; S8_PARK(EV, 0, 6) F(0); S8_FENCE      \
    S8_MF(G0_, 1) x[G0_] = s8_frag(XA, G0_ * 32 + frow, (KS) * 2 + fhalf); DMA1; S8_PARK(EV, 6, 12) F(1); S8_FENCE  \
    S8_MF(G1_, 0) WN[1] = s8_frag(WW, 1 * 32 + frow, (KS) * 2 + fhalf); DMA2; S8_PARK(EV, 12, 17) F(2); S8_FENCE    \
    S8_MF(G1_, 1) x[G1_] = s8_frag(XA, G1_ * 32 + frow, (KS) * 2 + fhalf); DMA3; S8_PARK(EV, 17, 22) F(3); S8_FENCE \
    S8_MF(G2_, 0) S8_PARK(EV, 22, 27) F(4); S8_FENCE                                             \
    S8_MF(G2_, 1) x[G2_] = s8_frag(XA, G2_ * 32 + frow, (KS) * 2 + fhalf); S8_PARK(EV, 27, 32) F(5); S8_FENCE       \
    acc[EV][0] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(bias_f[0], ones_f, zero16, 0, 0, 0);   \
    acc[EV][1] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(bias_f[1], ones_f, zero16, 0, 0, 0);   \
    S8_FENCE                                                                                     \
    S8_MF(EV, 0) F(6); S8_FENCE                                                                  \
    S8_MF(EV, 1) x[EV] = s8_frag(XA, (EV) * 32 + frow, (KS) * 2 + fhalf); F(7); S8_FENCE         \
  }

template <int ACT>
__device__ __forceinline__ void s8_body(const GemmArgs& p, char* smem) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int whi = wave >> 2;                      // DMA pieces: row group of piece `it` = (it & 1) * 2 + whi
  const int lda = p.lda ? p.lda : p.K, ldw = p.ldw ? p.ldw : p.K;

  const int tiles_n = p.N / 256;
  const int tiles_m = (int)(p.M / 256);
  const int slots = gridDim.x >> 3;
  const int logical = (blockIdx.x & 7) * slots + (blockIdx.x >> 3);     // XCD-major: a team sits on one XCD (or two)
  const int teams = (int)gridDim.x / tiles_n;
  const int team = logical / tiles_n, col = logical - team * tiles_n;
  if (team >= teams || team >= tiles_m) return;
  const int R = (tiles_m - team + teams - 1) / teams;                   // panels team, team + teams, ...
  const int nk = p.K / SBK;

  const int frow = lane & 31, fhalf = lane >> 5;
  const int x_off = wm * 128 * SROWB, w_off = wn * 64 * SROWB;
  char* slab = smem + S_SLAB_OFF + wave * S_SLAB;
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  // ---- operand streams -------------------------------------------------------------------------------------------------
  const __amdgpu_buffer_rsrc_t a_r = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t w_r =
      __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + (int64_t)col * 256 * ldw), 0, 0x7fffffff, 0x00020000);
  const int chunk = (lane & 7) ^ ((wave * 4 + (lane >> 4)) & 7);        // (row >> 1) & 7 is the same for every 64-row span
  const int a_voff = (wave * 8 + (lane >> 3)) * lda * 2 + chunk * 16;      // piece `it` adds it * 64 rows through the scalar offset
  const int w_voff = (wave * 8 + (lane >> 3)) * ldw * 2 + chunk * 16;
  const int panel_bytes = 256 * lda * 2;
  // byte offsets of the panels of period j-1, j, j+1 (clamped at the ends of the run: valid memory, results unused)
  int s_prev = team * panel_bytes, s_cur = s_prev, s_next = (R > 1 ? team + teams : team) * panel_bytes;
  int p_prev = team, p_cur = team, p_next = R > 1 ? team + teams : team;      // the same as panel indices

  // piece IT of the A K-tile at byte offset KB along K.  Its row group is g = (IT & 1) * 2 + whi; groups 0 .. SEL take the
  // target period's own panel, the others the panel before it.  NXT: the target K-tile belongs to the NEXT period.
#define S8_PIECE_A(NXT, SEL, IT, SLOT, KB)                                                                               \
  {                                                                                                                      \
    const bool own_ = (((IT) & 1) * 2 + whi) <= (SEL);                                                                   \
    const int so_ = (NXT) ? (own_ ? s_next : s_cur) : (own_ ? s_cur : s_prev);                                           \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(a_r, (lds_vptr)(smem + (SLOT) + ((IT) * 8 + wave) * 1024), 16, a_voff,     \
                                             so_ + (KB) + (IT) * 64 * lda * 2, 0, 0);                                    \
  }
#define S8_PIECE_W(IT, SLOT, KB)                                                                                         \
  __builtin_amdgcn_raw_ptr_buffer_load_lds(w_r, (lds_vptr)(smem + (SLOT) + ((IT) * 8 + wave) * 1024), 16, w_voff,        \
                                           (KB) + (IT) * 64 * ldw * 2, 0, 0)

  int a_c = S_AOFF, a_1 = S_AOFF + S_SLOT;
  int w_c = S_WOFF, w_1 = S_WOFF + S_SLOT;

  // ---- drain state -----------------------------------------------------------------------------------------------------
  // Bias enters through the matrix pipe: a restarting block's first C operand is bias_f[j] x ones (one extra MFMA per block
  // and tile).  bias_f[j]: lane (n = lane & 31, k half = lane >> 5) holds (hi, lo, 0, ...) at k = 0, 1 with hi + lo = bias to
  // 2^-17 relative; ones_f holds 1 at k = 0, 1.  (A per-lane bias table in the accumulator layout would be 32 registers or
  // 2 KiB of LDS; neither is left.)
  const int ncol0 = col * 256 + wn * 64;
  // (32x32x8 form: its operands are two registers each, so the three fragments cost six registers instead of twelve)
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  s16x4 bias_f[2], ones_f;
#pragma unroll
  for (int e = 0; e < 4; ++e) { bias_f[0][e] = 0; bias_f[1][e] = 0; ones_f[e] = 0; }
  if (fhalf == 0) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float b = p.bias ? p.bias[ncol0 + j * 32 + frow] : 0.f;
      const bf16_t hi = (bf16_t)b;
      const bf16_t lo = (bf16_t)(b - (float)hi);
      bias_f[j][0] = __builtin_bit_cast(short, hi);
      bias_f[j][1] = __builtin_bit_cast(short, lo);
    }
    ones_f[0] = 0x3f80;
    ones_f[1] = 0x3f80;
  }
  const int rowb = p.ldc * 2;
  __amdgpu_buffer_rsrc_t out_r = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, 0, 0x00020000);   // zero-size: stores are no-ops
  u32x4 rows_v = u32x4{0u, 0u, 0u, 0u};
  f32x16 d0 = zero16, d1 = zero16;
  int period = 0;                                 // panel period j (0 .. R; R = tail)

  auto set_drain = [&](int i) {                   // row group i of the tile finished in period j-1
    const int64_t row0 = (int64_t)p_prev * 256 + wm * 128 + i * 32;
    bf16_t* ob = reinterpret_cast<bf16_t*>(p.out) + row0 * p.ldc + ncol0;
    out_r = __builtin_amdgcn_make_buffer_rsrc(ob, 0, period > 0 ? 31 * rowb + 128 : 0, 0x00020000);
  };
  // The drain's per-lane addresses are re-derived from the lane id at every use (a handful of VALU instructions, opaque to
  // loop-invariant hoisting): held across the K-loop they are exactly the registers that would spill, and a scratch reload
  // inside the loop costs a vmcnt(0) behind freshly issued DMA pieces.
  // The epilogue of a parked row group (d0 = block (i,0), d1 = block (i,1): 32 rows x 64 columns) rides in the slots of the
  // THREE K-tiles that follow its completion, a few instructions per slot: the two waves of a SIMD run the same slot at the
  // same time, so whatever a slot carries beyond ~6 vector instructions or one memory instruction idles the matrix pipe
  // (measured: every filler instruction of a denser schedule added 1:1 to the K-tile).  Slot s = step * 8 + n, step 0 = ks0
  // of the event K-tile (which carries the parking copies instead):
  //   steps 1..7   activation + bf16 packing of the eight 8-column groups q (step 1: q = 0 and 1, one accumulator per slot;
  //                step t >= 2: q = t, one accumulator per two slots), each ending in ONE ds_write_b64 into the wave's slab
  //                (32 rows x 64 bf16, chunk c of row r at c ^ (r & 7))
  //   steps 8..11  (the third K-tile) slot 1: read 8 rows x 128 B of the slab back, slot 5: store them - behind the step's
  //                DMA pieces (slots 0..3), so that the K-tile's barrier can leave the stores in flight (vmcnt(3))
  // Three per-lane addresses stay resident: in this loop every instruction a wave issues costs issue time that nothing hides
  // (tools/s8_timing.py: recomputing them per use made the drain ~150 instructions per event instead of ~40 and cost 20 %).
  const int slab_w = (lane & 31) * 128 + ((lane & 7) << 4) + (lane >> 5) * 8;        // slab write, chunk 0 (chunk constant XORed in)
  const int slab_r = (lane >> 3) * 128 + (((lane & 7) ^ ((lane >> 3) & 7)) << 4);    // slab read-back, row slab 0: (tt*8 + rrow) & 7 == rrow & 7
  const int st_voff = (lane >> 3) * rowb + (lane & 7) * 16;
  float tq[4];
  auto slab_store = [&](int q) {
    bf16x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = (bf16_t)tq[r];
    *reinterpret_cast<bf16x4*>(slab + (slab_w ^ (q << 4))) = o;            // q = j * 4 + g
  };
  auto drain_slot = [&](int sl) {
#ifdef S8_NODRAIN
    return;
#endif
    const int step = sl >> 3, n = sl & 7;
    if (step == 1) {                              // q = 0 (slots 0..3), q = 1 (slots 4..7): both in block 0
      const int q = n >> 2, r = n & 3;
      tq[r] = s8_act<ACT>(d0[q * 4 + r]);
      if (r == 3) slab_store(q);
    } else if (step >= 2 && step <= 7) {
      const int q = step, j = q >> 2, g = q & 3;
      if (!(n & 1)) tq[n >> 1] = s8_act<ACT>((j ? d1 : d0)[g * 4 + (n >> 1)]);
      if (n == 7) slab_store(q);
    } else if (step >= 8) {
      const int tt = step - 8;
      if (n == 1) rows_v = *reinterpret_cast<const u32x4*>(slab + slab_r + tt * 1024);
      if (n == 5) {
#ifdef S8_NOSTORE
        if (rows_v[0] == 0x12345u)
#endif
        __builtin_amdgcn_raw_buffer_store_b128(rows_v, out_r, st_voff, tt * 8 * rowb, 2);
      }
    }
  };

  // ---- prologue: A(0) W(0) A(1) W(1)[0,1] ---------------------------------------------------------------------------------
#pragma unroll
  for (int it = 0; it < 4; ++it) S8_PIECE_A(false, 3, it, a_c, 0)
#pragma unroll
  for (int it = 0; it < 4; ++it) S8_PIECE_W(it, w_c, 0);
#pragma unroll
  for (int it = 0; it < 4; ++it) S8_PIECE_A(false, 3, it, a_1, 128)
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  bf16x8 x[4], w0[2], w1[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) x[i] = s8_frag(smem + a_c + x_off, i * 32 + frow, fhalf);
#pragma unroll
  for (int j = 0; j < 2; ++j) w0[j] = s8_frag(smem + w_c + w_off, j * 32 + frow, fhalf);

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = zero16;

#ifdef S8_TIMING
#define S8_STAMP()                                                                                                       \
  if (wave == 0 && lane == 0 && stamp_i < 64)                                                                            \
    reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(p.out) + (size_t)p.M * p.ldc * 2)[(size_t)blockIdx.x * 64 + stamp_i] = \
        __builtin_amdgcn_s_memtime();                                                                                    \
  ++stamp_i;
  int stamp_i = 0;
#else
#define S8_STAMP()
#endif
  // One K-tile of the rotated loop (K = 768: twelve per panel period, all unrolled: every one of them has its own role).
  //   K      K-tile index in the period; row group K / 3 completes at the start of K-tiles 0, 3, 6, 9 (stage 0), stages 1, 2
  //          = the two K-tiles after it
  //   DMA    ks0: W(K+1) x4 -> the slot of W(K-1); ks3: A(K+2) x4 -> the slot of A(K) (free after this K-tile's barrier);
  //          row group i of A K-tile k' comes from the period's own panel if k' >= 3 i, else from the previous one
#define S8_SEL_OF(KT) (((KT) % 12) / 3)           /* row groups 0 .. SEL use the target period's own panel */
#define S8_KTILE(K)                                                                                                        \
  {                                                                                                                      \
    S8_STAMP()                                                                                                           \
    constexpr int EV_ = (K) / 3, ST_ = (K) % 3;                                                                          \
    constexpr int KW_ = (((K) + 1) % 12) * 128, KA_ = (((K) + 2) % 12) * 128;                                            \
    constexpr bool NXT_ = (K) + 2 >= 12;          /* the A pieces issued here belong to the next period */               \
    constexpr int SEL_ = S8_SEL_OF((K) + 2);                                                                             \
    const char* xa = smem + a_c + x_off;                                                                                 \
    const char* ww = smem + w_c + w_off;                                                                                 \
    auto f0_ = [&](int n) { if (ST_ == 0) { if (n == 0) set_drain(EV_); } else drain_slot(ST_ * 32 + n); };              \
    auto f1_ = [&](int n) { drain_slot(ST_ * 32 + 8 + n); };                                                             \
    auto f2_ = [&](int n) { drain_slot(ST_ * 32 + 16 + n); };                                                            \
    auto f3_ = [&](int n) { drain_slot(ST_ * 32 + 24 + n); };                                                            \
    if constexpr (ST_ == 0) {                                                                                            \
      S8_STEP_RESTART(w1, xa, ww, 1, w0, EV_, S8_PIECE_W(0, w_1, KW_), S8_PIECE_W(1, w_1, KW_), S8_PIECE_W(2, w_1, KW_),   \
                      S8_PIECE_W(3, w_1, KW_), f0_)                                                                      \
    } else {                                                                                                             \
      S8_STEP(w1, xa, ww, 1, w0, S8_PIECE_W(0, w_1, KW_), S8_PIECE_W(1, w_1, KW_), S8_PIECE_W(2, w_1, KW_),               \
              S8_PIECE_W(3, w_1, KW_), f0_)                                                                              \
    }                                                                                                                    \
    S8_STEP(w0, xa, ww, 2, w1, (void)0, (void)0, (void)0, (void)0, f1_)                                                   \
    S8_STEP(w1, xa, ww, 3, w0, (void)0, (void)0, (void)0, (void)0, f2_)                                                   \
    /* A(K+1), W(K+1) landed; in a stage-2 K-tile its three stores (issued behind the W pieces) may stay in flight */     \
    if (ST_ == 2) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");                                             \
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                      \
    __builtin_amdgcn_s_barrier();                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                                   \
    S8_STEP(w0, smem + a_1 + x_off, smem + w_1 + w_off, 0, w1, S8_PIECE_A(NXT_, SEL_, 0, a_c, KA_),                        \
            S8_PIECE_A(NXT_, SEL_, 1, a_c, KA_), S8_PIECE_A(NXT_, SEL_, 2, a_c, KA_), S8_PIECE_A(NXT_, SEL_, 3, a_c, KA_), f3_) \
    { const int t_ = a_c; a_c = a_1; a_1 = t_; }                                                                         \
    { const int t_ = w_c; w_c = w_1; w_1 = t_; }                                                                         \
  }

  while (true) {
    S8_KTILE(0) S8_KTILE(1) S8_KTILE(2) S8_KTILE(3) S8_KTILE(4) S8_KTILE(5)
    S8_KTILE(6) S8_KTILE(7) S8_KTILE(8) S8_KTILE(9) S8_KTILE(10) S8_KTILE(11)
    if (period == R) break;                       // the tail period drained the last tile's four row groups
    ++period;
    s_prev = s_cur; p_prev = p_cur;
    s_cur = s_next; p_cur = p_next;
    if (period + 1 < R) { p_next = team + (period + 1) * teams; s_next = p_next * panel_bytes; }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may outlive the workgroup's LDS allocation
}

template <int ACT>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_bf16_s8_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  s8_body<ACT>(p, smem);
}

template <int ACT>
int launch_s8(const GemmArgs& p, hipStream_t st) {
  void (*kern)(GemmArgs) = gemm_bf16_s8_kernel<ACT>;
  int num_cu = 0;
  CACO_TRY_RC(prepare_launch(reinterpret_cast<const void*>(kern), S_SMEM, &num_cu));
  const int grid = num_cu / 8 * 8;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), S_SMEM, st, p);
  return check_hip(hipGetLastError(), "gemm_bf16_s8 launch");
}

}  // namespace

// bf16 output + bias (+ activation), whole 256-row panels, enough panels to fill every team's pipeline
bool gemm_bf16_s8_ok(const GemmArgs& p, int epi) {
  const int lda = p.lda ? p.lda : p.K;
  return epi == EPI_BF16 && !p.resid && !p.fold_mr && !p.xb_out && !p.stats_part && p.M % 256 == 0 && p.N % 256 == 0 &&
         p.N / 256 <= 32 && p.K == 12 * SBK && p.M * lda * 2 < 0x7fffffffLL &&
         (int64_t)256 * (p.ldw ? p.ldw : p.K) * 2 < 0x7fffffff && p.M / 256 >= 2 * (256 / (p.N / 256));
}

int gemm_bf16_s8(const GemmArgs& p, int epi, int act, hipStream_t st) {
  CACO_REQUIRE(gemm_bf16_s8_ok(p, epi), "gemm_bf16_s8: shape / epilogue not supported");
  if (act == ACT_NONE) return launch_s8<ACT_NONE>(p, st);
  if (act == ACT_SILU) return launch_s8<ACT_SILU>(p, st);
  set_error("gemm_bf16_s8: unsupported activation %d", act);
  return CACO_ERR_INVALID;
}

}  // namespace caco
