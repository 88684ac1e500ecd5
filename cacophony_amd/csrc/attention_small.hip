// Attention for SHORT sequences (at most 64 queries and 64 keys, head_dim 64): the text tower's causal AND padding
// self-attention at T = 32 (text_models/roberta.py:67-104, mask :297-310) and the first steps of the caption decoder.
//
// Why a second kernel: attention.hip gives a workgroup of four waves 128 query rows of ONE (caption, head) and walks
// 64-key tiles through a double-buffered LDS ring.  At T = 32 three of its four waves have no rows, half of every key tile
// is padding, and each workgroup pays the ring's prologue and two barriers for 16 MFMAs of which 8 multiply zeros: 17.7 us
// per launch, 0.018 of the MFMA roofline (profiles/r2_v3/bench.json).  Here
//   * one WAVE owns one (caption, head, 32-query block); the four waves of a workgroup are four different units, so every
//     wave works and no workgroup barrier exists at all (nothing is shared between waves);
//   * keys come in 32-key tiles (NKT = 1 or 2): no padded half tile at T = 32;
//   * Q and K fragments are k-contiguous in the fused QKV rows, so each lane fetches its 16 bytes straight from global
//     memory (L2) into the MFMA operand registers: no LDS, no DMA wait for the score product;
//   * only V, the transposed operand of P.V, passes through LDS: 16-byte LDS-DMA into a wave-private image, read back with
//     ds_read_b64_tr_b16 exactly as attention.hip does;
//   * all <= 64 scores of a row are in registers at once: a plain (not online) softmax, no rescale pass.
// Math, fragment layouts and the key permutation that lets P stay in registers are those of attention.hip:
//   S^T[key, q] = K Q^T,  O^T[d, q] = V^T P^T  on v_mfma_f32_32x32x16_bf16, fp32 statistics, exp2 with the scale folded in.
// A query row whose keys are all masked yields 0 (attention.hip's convention).
//
// Opt-in until it has run on hardware (CACO_ATTN_SMALL=1, see attention.hip::attention_qkv); verified on the wavesim build
// against the same checker as the big kernel (tests/test_wavesim.py).
// a kernel that has not run on hardware yet: the intra-wave LDS hand-offs are also fenced for the compiler (common.h)
#define CACO_WAVE_SYNC_FENCE 1
#include "common.h"
#include "kernels.h"

namespace caco {
namespace {

typedef __attribute__((address_space(3))) void* lds_vptr;
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int key_perm_s(int i) { return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1); }

constexpr int SM_HD = 64;
constexpr int SM_RP = SM_HD * 2;              // row pitch of the V image and of a head's slice: 128 bytes
constexpr int SM_OPITCH = SM_RP + 16;         // output staging pitch (8-byte writes conflict-free)
constexpr int SM_NW = 4;                      // waves (= independent units) per workgroup

template <int NKT>
constexpr int sm_region() { return (NKT * 32 * SM_RP > 32 * SM_OPITCH) ? NKT * 32 * SM_RP : 32 * SM_OPITCH; }

// (the body lives in a __device__ function: the buffer-descriptor builtins are not visible to the host pass)
template <bool CAUSAL, int NKT>
__device__ __forceinline__ void attention_small_body(const bf16_t* __restrict__ qp_, int q_ld, int Sq, const bf16_t* __restrict__ kv,
                                                     int ld, int k_off, int v_off, const float* __restrict__ key_mask, int S,
                                                     int heads, int batch, bf16_t* __restrict__ out, float scale_log2, int kv_rows) {
  constexpr int REGION = sm_region<NKT>();
  __shared__ __attribute__((aligned(16))) char smem[SM_NW * REGION];
  const int tid = threadIdx.x, lane = tid & 63, hf = lane >> 5, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qblks = (Sq + 31) >> 5;
  const int unit = blockIdx.x * SM_NW + wave;                    // (clip, head, query block), query block fastest
  if (unit >= batch * heads * qblks) return;                     // wave-uniform; no barrier anywhere below
  const int qb = unit % qblks, h = (unit / qblks) % heads, b = unit / (qblks * heads);
  const int H = heads * SM_HD;
  char* region = smem + wave * REGION;

  const bf16_t* q_base = qp_ + (int64_t)b * Sq * q_ld + h * SM_HD;
  const bf16_t* kv_base = kv + (int64_t)b * kv_rows * ld + h * SM_HD;
  const int q0 = qb * 32, q_row = q0 + l31;
  const int ntiles = CAUSAL ? min((S + 31) >> 5, qb + 1) : (S + 31) >> 5;       // <= NKT

  // V tiles -> wave-private LDS image (key-major rows of 128 bytes), 8 rows per DMA instruction
  {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)kv_base, 0, 0x7fffffff, 0x00020000);
    const int r8 = lane >> 3, c8 = lane & 7;
#pragma unroll
    for (int pc = 0; pc < NKT * 4; ++pc) {
      if (pc < ntiles * 4) {
        const int row = min(pc * 8 + r8, S - 1);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_vptr)(region + pc * 1024), 16, row * ld * 2 + (v_off + c8 * 8) * 2, 0, 0, 0);
      }
    }
  }
  // Q fragments (B operand: column = query, 8 contiguous head-dim elements) and K fragments (A operand: row = key)
  bf16x8 qf[4], kf[NKT][4];
  {
    const bf16_t* qp = q_base + (int64_t)min(q_row, Sq - 1) * q_ld + hf * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + ks * 16);
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
      if (kt < ntiles) {
        const bf16_t* kp = kv_base + (int64_t)min(kt * 32 + key_perm_s(l31), S - 1) * ld + k_off + hf * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) kf[kt][ks] = *reinterpret_cast<const bf16x8*>(kp + ks * 16);
      }
  }
  // per-key keep flags: s[kt][g * 8 + e] is key kt * 32 + 16 g + 8 hf + e.  The mask row is read through a buffer
  // descriptor of exactly S floats: keys past S read as 0 = masked, whatever follows the row in memory.
  const __amdgpu_buffer_rsrc_t mrs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(key_mask ? key_mask + (int64_t)b * S : nullptr), 0, key_mask ? S * 4 : 0, 0x00020000);

  f32x16 s[NKT];
  float m_row = -INFINITY;
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) {
    if (kt < ntiles) {
      const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      f32x16 acc = zero16;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kt][ks], qf[ks], acc, 0, 0, 0);
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const int kl = kt * 32 + 16 * g + 8 * hf;
        f32x4 k0 = {1.f, 1.f, 1.f, 1.f}, k1 = {1.f, 1.f, 1.f, 1.f};
        if (key_mask) {
          k0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(mrs, kl * 4, 0, 0));
          k1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(mrs, kl * 4 + 16, 0, 0));
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int key = kl + e;
          const bool keep = key < S && (e < 4 ? k0[e] : k1[e - 4]) != 0.f && !(CAUSAL && key > q_row);
          const float v = keep ? acc[g * 8 + e] : -INFINITY;
          acc[g * 8 + e] = v;
          m_row = fmaxf(m_row, v);
        }
      }
      s[kt] = acc;
    }
  }
  {   // a row's keys sit in lanes l and l + 32
    const unsigned mu = __builtin_bit_cast(unsigned, m_row);
    const auto sw = __builtin_amdgcn_permlane32_swap(mu, mu, false, false);
    m_row = fmaxf(__builtin_bit_cast(float, sw[0]), __builtin_bit_cast(float, sw[1]));
  }
  const float neg = (m_row == -INFINITY) ? 0.f : -m_row * scale_log2;       // all keys masked: every p = exp2(-inf) = 0
  float l_row = 0.f;
  bf16x8 pf[NKT][2];
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt)
    if (kt < ntiles) {
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][g * 8 + e], scale_log2, neg));
          l_row += p;
          pf[kt][g][e] = (bf16_t)p;
        }
    }
  l_row += __shfl_xor(l_row, 32, 64);
  const float inv = l_row > 0.f ? 1.0f / l_row : 0.f;

  // O^T += V^T P^T.  Transposed V fragment: 16-lane group (lane >> 4) & 1 selects the 16-column half, lane >> 5 the 8-key half,
  // (lane & 15) >> 2 the key within a 4-key block, lane & 3 the 4-column piece (attention.hip)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's V image has landed (wave-private: no barrier)
  CACO_WAVE_LDS_SYNC();
  f32x16 o[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
  const int v_lane = (8 * hf + ((lane & 15) >> 2)) * SM_RP + ((((lane >> 4) & 1) * 16 + (lane & 3) * 4) * 2);
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt)
    if (kt < ntiles) {
#pragma unroll
      for (int sp = 0; sp < 2; ++sp)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const char* p = region + v_lane + dt * 64 + (kt * 32 + sp * 16) * SM_RP;
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p + 4 * SM_RP));
          typedef short s16x8 __attribute__((ext_vector_type(8)));
          const bf16x8 vf = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
          o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[kt][sp], o[dt], 0, 0, 0);
        }
    }

  // whole-row stores through the (now dead) V image: lane (l31, hf) holds query row l31, columns dt*32 + g*8 + 4*hf .. +3
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  CACO_WAVE_LDS_SYNC();
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      bf16x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (bf16_t)(o[dt][g * 4 + e] * inv);
      *reinterpret_cast<bf16x4*>(region + l31 * SM_OPITCH + (dt * 32 + g * 8 + 4 * hf) * 2) = v;
    }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  CACO_WAVE_LDS_SYNC();
  const int rows_valid = min(32, Sq - q0);
  const __amdgpu_buffer_rsrc_t out_r = __builtin_amdgcn_make_buffer_rsrc(
      out + ((int64_t)b * Sq + q0) * H + h * SM_HD, 0, (rows_valid - 1) * H * 2 + SM_RP, 0x00020000);      // rows past Sq fall outside
#pragma unroll
  for (int it = 0; it < 4; ++it) {                 // 32 rows x 8 chunks of 16 B = 4 wave instructions
    const int L = it * 64 + lane;
    const int r = L >> 3, c = L & 7;
    const u32x4 v = *reinterpret_cast<const u32x4*>(region + r * SM_OPITCH + c * 16);
    __builtin_amdgcn_raw_buffer_store_b128(v, out_r, r * H * 2 + c * 16, 0, 0);
  }
}

// (register budget held to 128 / 256: with the default 512, hipcc keeps the MFMA results in AGPRs and copies all 48 of them
// out for the softmax and the stores)
template <bool CAUSAL, int NKT>
__global__ __launch_bounds__(SM_NW * 64, NKT == 1 ? 4 : 2) void attention_small_kernel(const bf16_t* __restrict__ q, int q_ld, int Sq,
                                                                     const bf16_t* __restrict__ kv, int ld, int k_off, int v_off,
                                                                     const float* __restrict__ key_mask, int S, int heads, int batch,
                                                                     bf16_t* __restrict__ out, float scale_log2, int kv_rows) {
  attention_small_body<CAUSAL, NKT>(q, q_ld, Sq, kv, ld, k_off, v_off, key_mask, S, heads, batch, out, scale_log2, kv_rows);
}

}  // namespace

bool attention_small_ok(int seq_q, int seq, int head_dim) { return head_dim == SM_HD && seq_q >= 1 && seq_q <= 64 && seq >= 1 && seq <= 64; }

// same contract as attention_qkv (attention.hip) for seq_q, seq <= 64 and head_dim 64; arguments already validated there
int attention_small(const bf16_t* q, int q_ld, int seq_q, const bf16_t* kv, int ld, int k_off, int v_off, const float* key_mask,
                    int batch, int seq, int heads, int causal, bf16_t* out, hipStream_t st, int kv_batch_rows) {
  CACO_REQUIRE(attention_small_ok(seq_q, seq, SM_HD), "attention_small: Sq=%d S=%d outside 1..64", seq_q, seq);
  CACO_REQUIRE((int64_t)kv_batch_rows * ld * 2 < 0x7fffffff, "attention_small: key / value rows of one clip exceed 2 GiB");
  const float scale_log2 = 1.4426950408889634f / sqrtf((float)SM_HD);
  const int units = batch * heads * ((seq_q + 31) / 32);
  const dim3 grid((units + SM_NW - 1) / SM_NW);
#define CACO_ATTN_S(C_, N_) \
  hipLaunchKernelGGL((attention_small_kernel<C_, N_>), grid, dim3(SM_NW * 64), 0, st, q, q_ld, seq_q, kv, ld, k_off, v_off, key_mask, seq, heads, batch, out, scale_log2, kv_batch_rows)
  if (seq <= 32) { if (causal) CACO_ATTN_S(true, 1); else CACO_ATTN_S(false, 1); }
  else { if (causal) CACO_ATTN_S(true, 2); else CACO_ATTN_S(false, 2); }
#undef CACO_ATTN_S
  return check_hip(hipGetLastError(), "attention_small launch");
}

}  // namespace caco
