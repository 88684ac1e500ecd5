// gemm_bf16_v8: 128x256x64 bf16 MFMA GEMM whose EPILOGUE RUNS UNDER THE NEXT TILE'S K-LOOP.
//   out[M,N] = epilogue( A[M,K] (bf16) x W[N,K]^T (bf16) ), fp32 accumulate on v_mfma_f32_32x32x16_bf16.
//
// Why (DESIGN.md "GEMM", measured on the 256x256 kernels of gemm_w4.hip): at K = 768 a 256x256 tile's epilogue costs
// 18-27 % of the GEMM because it runs with the matrix pipe idle - its VALU work (bias, SiLU = two quarter-rate
// transcendentals per element, bf16 packing) and the wait for its stores cannot overlap anything when all eight waves
// hold one accumulator set that the next tile needs at once.  Here a wave owns a 64x64 output sub-tile = 64 fp32
// accumulators, and keeps TWO sets: while set A accumulates tile t+1, set B (tile t) is converted, transposed through
// the wave's LDS slab and stored, its instructions riding as fillers between tile t+1's MFMAs (first four K-tiles of
// the tile: one 32x32 block each; stores leave a few per K-tile instead of as one burst).  The price is operand
// traffic: 128x256 tiles pull 1.5x the L2 bytes per FLOP of 256x256 tiles and re-stream the weights per 128 rows.
//
//   waves     8 = 2 (M) x 4 (N), two per SIMD, unphased (whichever stalls leaves the matrix pipe to the other)
//   LDS       A ring 3 x 16 KiB (128 rows x 128 B, two K-tiles ahead), W ring 2 x 32 KiB (one ahead), 8 slabs x 4 KiB
//             = 144 KiB; images lane-linear, bank swizzle (chunk ^ ((row >> 1) & 7)) on the DMA source + the read
//   K-loop    rotated: ks0 ks1 ks2 | s_waitcnt vmcnt(2) lgkmcnt(0) ; s_barrier | ks3, fragments double-buffered one
//             16-deep step ahead across K-tiles and output tiles; 4 MFMAs + 4 ds_read_b128 per step
//   DMA       per wave and K-tile: W(g+1) pieces in ks3 / ks0, A(g+2) pieces in ks1 / ks2; vector-memory operations
//             retire in order, the epilogue's stores are issued at the head of ks3, i.e. BEFORE the loads the next
//             barrier waits for, so the counted wait stays exact
//   stores    buffer_store with the descriptor's num_records as the M bound (rows past M are dropped by hardware; a
//             zero-size descriptor turns the drain of a not-yet-existing tile into a no-op): no branch in the loop
//
// Reference ops replaced: nn.Linear + activation + residual add (audio_models/mae.py:55-61,69-74,92-97,133;
// text_models/roberta.py:62-64,110,153,164).
#include "common.h"
#include "kernels.h"

namespace caco {
namespace {

constexpr int VBK = 64;
constexpr int VROWB = VBK * 2;                  // 128 bytes per row per K-tile
constexpr int V_ASLOT = 128 * VROWB;            // 16 KiB
constexpr int V_WSLOT = 256 * VROWB;            // 32 KiB
constexpr int V_AOFF = 0;
constexpr int V_WOFF = 3 * V_ASLOT;
constexpr int V_SLAB_OFF = V_WOFF + 2 * V_WSLOT;
constexpr int V_SLAB = 4096;
constexpr int V_BIAS_OFF = V_SLAB_OFF + 8 * V_SLAB;   // per wave: the 64 bias values of the tile being drained
constexpr int V_SMEM = V_BIAS_OFF + 8 * 256;          // 149504 = 146 KiB

typedef __attribute__((address_space(3))) void* lds_vptr;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define V8_SGB_VALU 0x002
#define V8_SGB_MFMA 0x008
#define V8_SGB_VMEM 0x010
#define V8_SGB_DSRD 0x100
#define V8_SGB_DSWR 0x200

template <int ACT>
__device__ __forceinline__ float v8_act(float x) {
  if constexpr (ACT == ACT_SILU) return silu_f(x);
  if constexpr (ACT == ACT_GELU) return gelu_erf_f(x);
  return x;
}

struct V8CurA {
  __amdgpu_buffer_rsrc_t r;
  int voff[2];
  int li, kt;
};
struct V8CurW {
  __amdgpu_buffer_rsrc_t r;
  int voff;
  int li, kt;
};

__device__ __forceinline__ void v8_setup_a(V8CurA& C, const GemmArgs& p, int t, int tiles_n, int lda, int wave, int lane) {
  const int64_t m0 = (int64_t)(t / tiles_n) * 128;
  C.r = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + m0 * lda), 0, 0x7fffffff, 0x00020000);
  const int r8 = wave * 8 + (lane >> 3);
  const int chunk = (lane & 7) ^ ((wave * 4 + (lane >> 4)) & 7);
  const int last = (int)min((int64_t)128, p.M - m0) - 1;
#pragma unroll
  for (int it = 0; it < 2; ++it) C.voff[it] = min(it * 64 + r8, last) * lda * 2 + chunk * 16;
}
__device__ __forceinline__ void v8_setup_w(V8CurW& C, const GemmArgs& p, int t, int tiles_n, int ldw, int wave, int lane) {
  const int n0 = (t % tiles_n) * 256;
  C.r = __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + (int64_t)n0 * ldw), 0, 0x7fffffff, 0x00020000);
  const int r8 = wave * 8 + (lane >> 3);
  const int chunk = (lane & 7) ^ ((wave * 4 + (lane >> 4)) & 7);
  C.voff = r8 * ldw * 2 + chunk * 16;
}
__device__ __forceinline__ void v8_piece_a(const V8CurA& C, int it, char* slot, int wave) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(C.r, (lds_vptr)(slot + (it * 8 + wave) * 1024), 16, C.voff[it], C.kt * (VBK * 2), 0, 0);
}
__device__ __forceinline__ void v8_piece_w(const V8CurW& C, int it, int ldw, char* slot, int wave) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(C.r, (lds_vptr)(slot + (it * 8 + wave) * 1024), 16, C.voff,
                                           C.kt * (VBK * 2) + it * 64 * ldw * 2, 0, 0);
}
__device__ __forceinline__ bf16x8 v8_frag(const char* oper, int row, int chunk) {
  return *reinterpret_cast<const bf16x8*>(oper + row * VROWB + ((chunk ^ ((row >> 1) & 7)) << 4));
}

// ---- drain state of the PREVIOUS tile (its accumulators are being emptied under the current K-loop) ----------------
struct V8Drain {
  __amdgpu_buffer_rsrc_t out_r, res_r;     // bounded descriptors of the previous tile's rows (zero-size: no-op)
  int col0;                                // previous tile's first column
  int row_off;                             // per-lane byte offset of (row rrow, column group c8) inside the tile
};

// block (i, j) of the wave's 64x64 sub-tile -> wave slab.  MFMA layout (operands swapped): lane = (m = lane & 31,
// n = j*32 + g*8 + (lane >> 5)*4 + r) for accumulator register g*4 + r.
// The bias comes from the wave's LDS row (staged once per tile): a vector-memory load consumed inside the K-loop would
// make the compiler wait for every older DMA piece as well.
template <int EPI, int ACT>
__device__ __forceinline__ void v8_block_to_slab(const f32x16& a, const float* bias_lds, int j, int lane, char* slab) {
  const int lm = lane & 31, lh = lane >> 5;
  if constexpr (EPI == EPI_BF16) {      // slab: 32 rows x 64 bf16 (128-byte pitch), chunk c of row r at c ^ (r & 7)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 bb = *reinterpret_cast<const f32x4*>(bias_lds + j * 32 + g * 8 + lh * 4);
      bf16x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = (bf16_t)v8_act<ACT>(a[g * 4 + r] + bb[r]);
      *reinterpret_cast<bf16x4*>(slab + lm * 128 + (((j * 4 + g) ^ (lm & 7)) << 4) + lh * 8) = o;
    }
  } else {                              // slab: 32 rows x 32 fp32 (128-byte pitch), one block per fill
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 v;
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = a[g * 4 + r];
      v += *reinterpret_cast<const f32x4*>(bias_lds + j * 32 + g * 8 + lh * 4);
      *reinterpret_cast<f32x4*>(slab + lm * 128 + (((g * 2 + lh) ^ (lm & 7)) << 4)) = v;
    }
  }
}
// read the slab back as whole 128-byte row segments: 8 rows per instruction
__device__ __forceinline__ void v8_slab_read(const char* slab, int lane, u32x4 (&v)[4]) {
  const int rrow = lane >> 3, c8 = lane & 7;
#pragma unroll
  for (int tt = 0; tt < 4; ++tt) {
    const int row = tt * 8 + rrow;
    v[tt] = *reinterpret_cast<const u32x4*>(slab + row * 128 + ((c8 ^ (row & 7)) << 4));
  }
}

// one 16-deep step: 4 MFMAs on the current fragments, 4 fragment reads for the next step, up to 2 DMA pieces, fillers
#define V8_STEP(XN, WN, XA, WW, KS, XC, WC, ACC, ZC, DMA0, DMA1, FILL)                             \
  XN[0] = v8_frag(XA, frow, (KS) * 2 + fhalf);                                                   \
  WN[0] = v8_frag(WW, frow, (KS) * 2 + fhalf);                                                   \
  DMA0;                                                                                          \
  XN[1] = v8_frag(XA, 32 + frow, (KS) * 2 + fhalf);                                              \
  WN[1] = v8_frag(WW, 32 + frow, (KS) * 2 + fhalf);                                              \
  DMA1;                                                                                          \
  FILL;                                                                                          \
  ACC[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WC[0], XC[0], (ZC) ? zero16 : ACC[0][0], 0, 0, 0); \
  ACC[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WC[0], XC[1], (ZC) ? zero16 : ACC[1][0], 0, 0, 0); \
  ACC[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WC[1], XC[0], (ZC) ? zero16 : ACC[0][1], 0, 0, 0); \
  ACC[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WC[1], XC[1], (ZC) ? zero16 : ACC[1][1], 0, 0, 0); \
  _Pragma("unroll") for (int n_ = 0; n_ < 4; ++n_) {                                             \
    __builtin_amdgcn_sched_group_barrier(V8_SGB_MFMA, 1, 0);                                     \
    __builtin_amdgcn_sched_group_barrier(V8_SGB_DSRD, 1, 0);                                     \
    if (n_ == 0 || n_ == 2) __builtin_amdgcn_sched_group_barrier(V8_SGB_VMEM, 1, 0);             \
    __builtin_amdgcn_sched_group_barrier(V8_SGB_VALU, 26, 0);                                    \
  }                                                                                              \
  __builtin_amdgcn_sched_group_barrier(V8_SGB_DSWR, 8, 0);                                       \
  __builtin_amdgcn_sched_group_barrier(V8_SGB_DSRD, 8, 0);                                       \
  __builtin_amdgcn_sched_group_barrier(V8_SGB_VMEM, 8, 0);                                       \
  __builtin_amdgcn_sched_barrier(0);

// one K-tile of the rotated loop.  F0..F3: filler statements of the four steps (the previous tile's epilogue).
#define V8_KTILE(ACC, ZC, F0, F1, F2, F3)                                                         \
  {                                                                                              \
    const char* xa = smem + a_c + x_off;                                                         \
    const char* ww = smem + w_c + w_off;                                                         \
    V8_STEP(x1, w1, xa, ww, 1, x0, w0, ACC, ZC, v8_piece_w(CW, 2, ldw, smem + w_1, wave), v8_piece_w(CW, 3, ldw, smem + w_1, wave), F0) \
    advance_w();                                                                                 \
    V8_STEP(x0, w0, xa, ww, 2, x1, w1, ACC, 0, v8_piece_a(CA, 0, smem + a_2, wave), (void)0, F1) \
    V8_STEP(x1, w1, xa, ww, 3, x0, w0, ACC, 0, v8_piece_a(CA, 1, smem + a_2, wave), (void)0, F2) \
    advance_a();                                                                                 \
    asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");                                  \
    __builtin_amdgcn_s_barrier();                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                           \
    V8_STEP(x0, w0, smem + a_1 + x_off, smem + w_1 + w_off, 0, x1, w1, ACC, 0, v8_piece_w(CW, 0, ldw, smem + w_c, wave), \
            v8_piece_w(CW, 1, ldw, smem + w_c, wave), F3)                                        \
    { const int t_ = a_c; a_c = a_1; a_1 = a_2; a_2 = t_; }                                      \
    { const int t_ = w_c; w_c = w_1; w_1 = t_; }                                                 \
  }

// epilogue pieces of the previous tile (accumulator set PRV), spread over the first four K-tiles of the current one:
//   K-tile 0: block (0,0) -> slab                K-tile 1: block (0,1) -> slab, read back, store rows 0..31
//   K-tile 2: block (1,0) -> slab                K-tile 3: block (1,1) -> slab, read back, store rows 32..63
// (fp32 form: every K-tile is one whole block: slab, read back, + residual, store)
template <int EPI>
__device__ __forceinline__ void v8_store_rows(const V8Drain& D, const GemmArgs& p, const u32x4 (&v)[4], int i, int jcol) {
  // v[tt] = 16 bytes of row i*32 + tt*8 + rrow at column group c8 (both already in D.row_off)
#pragma unroll
  for (int tt = 0; tt < 4; ++tt) {
    const int esz = (EPI == EPI_BF16) ? 2 : 4;
    const int off = D.row_off + ((i * 32 + tt * 8) * p.ldc + jcol) * esz;
    __builtin_amdgcn_raw_buffer_store_b128(v[tt], D.out_r, off, 0, 0);
  }
}

template <int EPI, int ACT>
__device__ __forceinline__ void v8_body(const GemmArgs& p, char* smem) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int lda = p.lda ? p.lda : p.K, ldw = p.ldw ? p.ldw : p.K;
  constexpr int ESZ = (EPI == EPI_BF16) ? 2 : 4;

  const int tiles_n = p.N / 256;
  const int tiles_m = (int)((p.M + 127) / 128);
  const int nwg = tiles_m * tiles_n;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
  const int q = nwg >> 3, r = nwg & 7;
  const int cnt = q + (xcd < r ? 1 : 0);
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  if (slot >= cnt) return;
  const int nk = p.K / VBK;

  const int frow = lane & 31, fhalf = lane >> 5;
  const int x_off = wm * 64 * VROWB, w_off = wn * 64 * VROWB;
  char* slab = smem + V_SLAB_OFF + wave * V_SLAB;
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  V8CurA CA;
  V8CurW CW;
  CA.li = CW.li = slot;
  CA.kt = CW.kt = 0;
  v8_setup_a(CA, p, base + slot, tiles_n, lda, wave, lane);
  v8_setup_w(CW, p, base + slot, tiles_n, ldw, wave, lane);
  auto advance_a = [&]() {
    if (++CA.kt == nk) {
      CA.kt = 0;
      if (CA.li + slots < cnt) { CA.li += slots; v8_setup_a(CA, p, base + CA.li, tiles_n, lda, wave, lane); }
    }
  };
  auto advance_w = [&]() {
    if (++CW.kt == nk) {
      CW.kt = 0;
      if (CW.li + slots < cnt) { CW.li += slots; v8_setup_w(CW, p, base + CW.li, tiles_n, ldw, wave, lane); }
    }
  };

  int a_c = V_AOFF, a_1 = V_AOFF + V_ASLOT, a_2 = V_AOFF + 2 * V_ASLOT;
  int w_c = V_WOFF, w_1 = V_WOFF + V_WSLOT;

  // prologue: A(0) W(0) | A(1) W(1)[0..1]
  v8_piece_a(CA, 0, smem + a_c, wave);
  v8_piece_a(CA, 1, smem + a_c, wave);
  advance_a();
#pragma unroll
  for (int it = 0; it < 4; ++it) v8_piece_w(CW, it, ldw, smem + w_c, wave);
  advance_w();
  v8_piece_a(CA, 0, smem + a_1, wave);
  v8_piece_a(CA, 1, smem + a_1, wave);
  advance_a();
  v8_piece_w(CW, 0, ldw, smem + w_1, wave);
  v8_piece_w(CW, 1, ldw, smem + w_1, wave);
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  bf16x8 x0[2], w0[2], x1[2], w1[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    x0[i] = v8_frag(smem + a_c + x_off, i * 32 + frow, fhalf);
    w0[i] = v8_frag(smem + w_c + w_off, i * 32 + frow, fhalf);
  }

  // drain descriptor of "the previous tile": none yet -> zero-size descriptors make every store a no-op
  V8Drain D;
  D.out_r = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, 0, 0x00020000);
  D.res_r = D.out_r;
  D.col0 = 0;
  D.row_off = ((lane >> 3) * p.ldc) * ESZ + (lane & 7) * 16;
  float* bias_lds = reinterpret_cast<float*>(smem + V_BIAS_OFF + wave * 256);
  float bias_reg = 0.f;                    // bias of the tile being ACCUMULATED, column lane of the wave's 64
  auto load_bias = [&](int t) {
    const int n0 = (t % tiles_n) * 256 + wn * 64;
    bias_reg = p.bias ? p.bias[n0 + lane] : 0.f;
  };
  auto set_drain = [&](int t) {            // the tile whose accumulators were just completed
    const int64_t m0 = (int64_t)(t / tiles_n) * 128 + wm * 64;     // this wave's first row
    const int n0 = (t % tiles_n) * 256 + wn * 64;
    const int64_t rows = min((int64_t)64, p.M - m0);               // valid rows of the wave's sub-tile (may be <= 0)
    // num_records bounds the ROWS: offsets of rows >= `rows` lie beyond it and are dropped by the hardware
    const int bytes = rows > 0 ? (int)(((rows - 1) * p.ldc + 64) * ESZ) : 0;
    char* ob = reinterpret_cast<char*>(p.out) + (m0 * p.ldc + n0) * ESZ;
    D.out_r = __builtin_amdgcn_make_buffer_rsrc(ob, 0, bytes, 0x00020000);
    if constexpr (EPI == EPI_F32) {
      const char* rb = reinterpret_cast<const char*>(p.resid ? p.resid : reinterpret_cast<const float*>(p.out)) + (m0 * p.ldc + n0) * ESZ;
      D.res_r = __builtin_amdgcn_make_buffer_rsrc((void*)rb, 0, p.resid ? bytes : 0, 0x00020000);
    }
    D.col0 = n0;
    bias_lds[lane] = bias_reg;             // the previous drain finished in K-tile 3 of this tile: its reads are done
  };

  f32x16 accA[2][2], accB[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) accB[i][j] = zero16;
  bias_lds[lane] = 0.f;

  u32x4 rows_v[4];      // slab read-back in flight between its ds_read and the stores in ks3
  u32x4 res_v[4];
#pragma unroll
  for (int tt = 0; tt < 4; ++tt) { rows_v[tt] = u32x4{0u, 0u, 0u, 0u}; res_v[tt] = u32x4{0u, 0u, 0u, 0u}; }

  // fillers (lambdas: no top-level commas inside the macro arguments)
  auto to_slab = [&](const f32x16& a, int j) { v8_block_to_slab<EPI, ACT>(a, bias_lds, j, lane, slab); };
  auto readback = [&]() { v8_slab_read(slab, lane, rows_v); };
  auto store_bf16 = [&](int i) { v8_store_rows<EPI>(D, p, rows_v, i, 0); };
  auto res_fetch = [&](int i, int j) {
#pragma unroll
    for (int tt = 0; tt < 4; ++tt)
      res_v[tt] = __builtin_amdgcn_raw_buffer_load_b128(D.res_r, D.row_off + ((i * 32 + tt * 8) * p.ldc + j * 32) * 4, 0, 0);
  };
  auto add_store_f32 = [&](int i, int j) {
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
      f32x4 a = __builtin_bit_cast(f32x4, rows_v[tt]);
      a += __builtin_bit_cast(f32x4, res_v[tt]);
      rows_v[tt] = __builtin_bit_cast(u32x4, a);
    }
    v8_store_rows<EPI>(D, p, rows_v, i, j * 32);
  };
// bf16: block -> slab in ks0 (one block per K-tile), read back in ks2 of the odd K-tiles, stores in ks3
#define V8_TO_SLAB(PRV, I, J) to_slab(PRV[I][J], J)
// fp32: one block per K-tile: residual fetch + block -> slab in ks0, read back in ks1, add + store in ks3

#define V8_TILE(ACC, PRV)                                                                         \
  if constexpr (EPI == EPI_BF16) {                                                               \
    V8_KTILE(ACC, 1, V8_TO_SLAB(PRV, 0, 0), (void)0, (void)0, (void)0)                           \
    V8_KTILE(ACC, 0, V8_TO_SLAB(PRV, 0, 1), (void)0, readback(), store_bf16(0))                  \
    V8_KTILE(ACC, 0, V8_TO_SLAB(PRV, 1, 0), (void)0, (void)0, (void)0)                           \
    V8_KTILE(ACC, 0, V8_TO_SLAB(PRV, 1, 1), (void)0, readback(), store_bf16(1))                  \
  } else {                                                                                       \
    V8_KTILE(ACC, 1, res_fetch(0, 0); V8_TO_SLAB(PRV, 0, 0), readback(), (void)0, add_store_f32(0, 0)) \
    V8_KTILE(ACC, 0, res_fetch(0, 1); V8_TO_SLAB(PRV, 0, 1), readback(), (void)0, add_store_f32(0, 1)) \
    V8_KTILE(ACC, 0, res_fetch(1, 0); V8_TO_SLAB(PRV, 1, 0), readback(), (void)0, add_store_f32(1, 0)) \
    V8_KTILE(ACC, 0, res_fetch(1, 1); V8_TO_SLAB(PRV, 1, 1), readback(), (void)0, add_store_f32(1, 1)) \
  }                                                                                              \
  for (int kt = 4; kt < nk; ++kt) { V8_KTILE(ACC, 0, (void)0, (void)0, (void)0, (void)0) }

  int c_li = slot;
  while (true) {
    load_bias(base + c_li);
    V8_TILE(accA, accB)
    set_drain(base + c_li);
    c_li += slots;
    if (c_li >= cnt) {        // accA holds the last tile
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) accB[i][j] = accA[i][j];
      break;
    }
    load_bias(base + c_li);
    V8_TILE(accB, accA)
    set_drain(base + c_li);
    c_li += slots;
    if (c_li >= cnt) break;   // accB holds the last tile
  }
  // final drain of accB, nothing left to overlap it with
  if constexpr (EPI == EPI_BF16) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      to_slab(accB[i][0], 0);
      to_slab(accB[i][1], 1);
      readback();
      store_bf16(i);
    }
  } else {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        res_fetch(i, j);
        to_slab(accB[i][j], j);
        readback();
        add_store_f32(i, j);
      }
  }
}

template <int EPI, int ACT>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_bf16_v8_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  v8_body<EPI, ACT>(p, smem);
}

template <int EPI, int ACT>
int launch_v8(const GemmArgs& p, hipStream_t st) {
  auto kern = gemm_bf16_v8_kernel<EPI, ACT>;
  static bool attr_done = false;
  if (!attr_done) {
    CACO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, V_SMEM));
    attr_done = true;
  }
  static int num_cu = 0;
  if (!num_cu) {
    int dev = 0;
    hipDeviceProp_t prop;
    CACO_HIP(hipGetDevice(&dev));
    CACO_HIP(hipGetDeviceProperties(&prop, dev));
    num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  const int tiles = (int)((p.M + 127) / 128) * (p.N / 256);
  const int grid = tiles < num_cu ? (tiles + 7) / 8 * 8 : num_cu / 8 * 8;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), V_SMEM, st, p);
  return check_hip(hipGetLastError(), "gemm_bf16_v8 launch");
}

}  // namespace

bool gemm_bf16_v8_ok(const GemmArgs& p, int epi) {
  const int lda = p.lda ? p.lda : p.K, ldw = p.ldw ? p.ldw : p.K;
  return p.N % 256 == 0 && p.K % VBK == 0 && p.K >= 4 * VBK && !p.fold_mr && !p.xb_out && !p.stats_part &&
         (int64_t)128 * lda * 2 < 0x7fffffff && (int64_t)256 * ldw * 2 < 0x7fffffff && (int64_t)64 * p.ldc * 4 < 0x7fffffff;
}

int gemm_bf16_v8(const GemmArgs& p, int epi, int act, hipStream_t st) {
  CACO_REQUIRE(gemm_bf16_v8_ok(p, epi), "gemm_bf16_v8: shape not supported");
  if (epi == EPI_BF16) {
    if (act == ACT_NONE) return launch_v8<EPI_BF16, ACT_NONE>(p, st);
    if (act == ACT_SILU) return launch_v8<EPI_BF16, ACT_SILU>(p, st);
    if (act == ACT_GELU) return launch_v8<EPI_BF16, ACT_GELU>(p, st);
  } else if (epi == EPI_F32 && act == ACT_NONE) {
    return launch_v8<EPI_F32, ACT_NONE>(p, st);
  }
  set_error("gemm_bf16_v8: unsupported epilogue %d / activation %d", epi, act);
  return CACO_ERR_INVALID;
}

}  // namespace caco
