// Fused (flash-style) multi-head self-attention for the two encoder stacks: never materialises
// the S x S score matrix.
//
//   audio: 8 heads x 96, S = 500 (496 valid), key-padding mask        (audio_models/mae.py:89-92,
//          torch.nn.MultiheadAttention with key_padding_mask, need_weights=False)
//   text : 12 heads x 64, T = 32, causal AND key-padding mask          (text_models/roberta.py:86-104,297-310)
//
// Layout contract (produced by the QKV GEMMs): qk[B*S, 2H] bf16 with Q in columns [0,H) and K in
// [H,2H), head h owning the contiguous slice h*HD..; V arrives TRANSPOSED per clip,
// vt[B, H, S_pad] (S_pad = multiple of 64, written by the GEMM's transposed epilogue), so that both
// MFMA operands of P.V are k-contiguous and no transpose is needed on chip.
//
// Work split: one workgroup = 4 waves = 128 query rows of one (clip, head); each wave owns 32 query
// rows and the full head dimension.  K / V^T tiles of 64 keys are register-staged into padded
// (conflict-free) LDS, double-buffered, with the next tile's global loads issued before the
// current tile's MFMAs (issue-early / write-late).
//
// Math per 64-key tile, all on v_mfma_f32_32x32x16_bf16 with fp32 accumulation:
//   S^T[key, q] = K Q^T      (operands swapped so that every lane owns ONE query column: the
//                             row-wise softmax is lane-local plus a single lane^32 exchange)
//   O^T[d, q]  += V^T P^T
// The MFMA row <-> key assignment is permuted (bits 2 and 3 swapped) so that the S^T accumulator
// registers of a lane are, in order, exactly the 8-key groups the P operand of the second MFMA
// wants: P never leaves registers and needs no cross-lane shuffle.
// Softmax statistics (running max / sum) are fp32; exp is evaluated as exp2 with the 1/sqrt(HD)
// scale and log2(e) folded into one multiply.  A query row whose keys are all masked yields 0
// (the reference yields NaN there; it cannot happen with right-padded inputs).
#include "common.h"
#include "kernels.h"

namespace caco {
namespace {

constexpr int QB = 128;        // query rows per workgroup
constexpr int KT = 64;         // keys per tile
constexpr int VP = KT * 2 + 16;  // V^T row pitch in bytes (144: 9 x 16 B -> 16 consecutive rows hit 16 slots)

__device__ __forceinline__ int key_perm(int i) { return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1); }

template <int HD, bool CAUSAL>
__global__ __launch_bounds__(256) void attention_kernel(const bf16_t* __restrict__ qk, const bf16_t* __restrict__ vt,
                                                        const float* __restrict__ key_mask, int S, int S_pad, int heads,
                                                        bf16_t* __restrict__ out, float scale_log2) {
  constexpr int KP = HD * 2 + 16;              // K row pitch (208 / 144 bytes)
  constexpr int KCH = HD / 8;                  // 16-byte chunks per K row
  constexpr int NKC = KT * KCH / 256;          // K chunks per thread per tile
  constexpr int NVC = HD * 8 / 256;            // V^T chunks per thread per tile
  constexpr int KS = HD / 16;                  // MFMA k-steps over the head dim
  constexpr int DT = HD / 32;                  // 32-row output tiles over the head dim
  constexpr int K_BYTES = KT * KP, V_BYTES = HD * VP, BUF = K_BYTES + V_BYTES + KT * 4;
  __shared__ __attribute__((aligned(16))) char smem[2 * BUF];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hf = lane >> 5, l31 = lane & 31;
  const int qb = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int H = heads * HD;
  const int64_t row_base = (int64_t)b * S;
  const bf16_t* q_base = qk + row_base * (2 * H) + h * HD;
  const bf16_t* k_base = q_base + H;
  const bf16_t* v_base = vt + ((int64_t)b * H + h * HD) * S_pad;

  const int q_row = qb * QB + wave * 32 + l31;
  const bool wave_active = (qb * QB + wave * 32) < S;

  // Q fragments (B operand: column j = query, k = 8 contiguous head-dim elements)
  bf16x8 qf[KS];
  {
    const bf16_t* qp = q_base + (int64_t)(q_row < S ? q_row : S - 1) * (2 * H) + hf * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + ks * 16);
  }

  int ntiles = (S + KT - 1) / KT;
  if (CAUSAL) {
    const int last_q = min(qb * QB + QB - 1, S - 1);
    ntiles = min(ntiles, last_q / KT + 1);
  }

  bf16x8 kreg[NKC], vreg[NVC];
  float breg = 0.f;
  auto load_tile = [&](int t) {
    const int key0 = t * KT;
#pragma unroll
    for (int i = 0; i < NKC; ++i) {
      const int id = tid + i * 256, r = id / KCH, c = id % KCH;
      const int key = min(key0 + r, S - 1);
      kreg[i] = *reinterpret_cast<const bf16x8*>(k_base + (int64_t)key * (2 * H) + c * 8);
    }
#pragma unroll
    for (int i = 0; i < NVC; ++i) {
      const int id = tid + i * 256, d = id >> 3, c = id & 7;
      vreg[i] = *reinterpret_cast<const bf16x8*>(v_base + (int64_t)d * S_pad + key0 + c * 8);
    }
    if (tid < KT) {
      const int key = key0 + tid;
      const bool keep = key < S && (key_mask == nullptr || key_mask[row_base + key] != 0.f);
      breg = keep ? 0.f : -INFINITY;
    }
  };
  auto store_tile = [&](int buf) {
    char* kb = smem + buf * BUF;
    char* vb = kb + K_BYTES;
#pragma unroll
    for (int i = 0; i < NKC; ++i) {
      const int id = tid + i * 256, r = id / KCH, c = id % KCH;
      *reinterpret_cast<bf16x8*>(kb + r * KP + c * 16) = kreg[i];
    }
#pragma unroll
    for (int i = 0; i < NVC; ++i) {
      const int id = tid + i * 256, d = id >> 3, c = id & 7;
      *reinterpret_cast<bf16x8*>(vb + d * VP + c * 16) = vreg[i];
    }
    if (tid < KT) reinterpret_cast<float*>(vb + V_BYTES)[tid] = breg;
  };

  f32x16 o[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  load_tile(0);
  store_tile(0);
  __syncthreads();

  for (int t = 0; t < ntiles; ++t) {
    if (t + 1 < ntiles) load_tile(t + 1);
    if (wave_active) {
      const char* kb = smem + (t & 1) * BUF;
      const char* vb = kb + K_BYTES;
      const float* bias = reinterpret_cast<const float*>(vb + V_BYTES);
      f32x16 s[2];
#pragma unroll
      for (int st = 0; st < 2; ++st) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[st][r] = 0.f;
        const char* kr = kb + (st * 32 + key_perm(l31)) * KP + hf * 16;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kr + ks * 32);
          s[st] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[st], 0, 0, 0);
        }
      }
      // scores -> log2 domain, masks, tile max
      float m_tile = -INFINITY;
#pragma unroll
      for (int st = 0; st < 2; ++st) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          const int kl = st * 32 + 16 * g + 8 * hf;     // first of this lane's 8 consecutive keys
          const f32x4 b0 = *reinterpret_cast<const f32x4*>(bias + kl);
          const f32x4 b1 = *reinterpret_cast<const f32x4*>(bias + kl + 4);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int r = g * 8 + e;
            float v = s[st][r] * scale_log2 + (e < 4 ? b0[e] : b1[e - 4]);
            if (CAUSAL && (t * KT + kl + e) > q_row) v = -INFINITY;
            s[st][r] = v;
            m_tile = fmaxf(m_tile, v);
          }
        }
      }
      m_tile = fmaxf(m_tile, __shfl_xor(m_tile, 32, 64));
      const float m_new = fmaxf(m_run, m_tile);
      const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);
      float psum = 0.f;
      bf16x8 pf[4];
#pragma unroll
      for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float p = __builtin_amdgcn_exp2f(s[st][g * 8 + e] - m_use);
            psum += p;
            pf[st * 2 + g][e] = (bf16_t)p;
          }
      l_run = l_run * alpha + psum;
      m_run = m_new;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
      // O^T += V^T P^T
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const char* vr = vb + (dt * 32 + l31) * VP + hf * 16;
#pragma unroll
        for (int sp = 0; sp < 4; ++sp) {
          const bf16x8 vf = *reinterpret_cast<const bf16x8*>(vr + sp * 32);
          o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[sp], o[dt], 0, 0, 0);
        }
      }
    }
    if (t + 1 < ntiles) store_tile((t + 1) & 1);
    __syncthreads();
  }

  if (!wave_active) return;
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
  if (q_row < S) {
    bf16_t* op = out + (row_base + q_row) * H + h * HD + 4 * hf;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        bf16x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (bf16_t)(o[dt][g * 4 + e] * inv);
        *reinterpret_cast<bf16x4*>(op + dt * 32 + g * 8) = v;
      }
  }
}

}  // namespace

int attn_seq_pad(int seq) { return (seq + KT - 1) / KT * KT; }

int attention(const bf16_t* qk, const bf16_t* vt, const float* key_mask, int batch, int seq, int heads, int head_dim,
              int causal, bf16_t* out, hipStream_t st) {
  CACO_REQUIRE(batch > 0 && seq > 0 && heads > 0, "attention: bad shape B=%d S=%d heads=%d", batch, seq, heads);
  CACO_REQUIRE(head_dim == 64 || head_dim == 96, "attention: head_dim %d not in {64, 96}", head_dim);
  CACO_REQUIRE(heads <= 65535 && batch <= 65535, "attention: heads / batch exceed the grid limit");
  const int S_pad = attn_seq_pad(seq);
  const dim3 grid((seq + QB - 1) / QB, heads, batch);
  const float scale_log2 = 1.4426950408889634f / sqrtf((float)head_dim);
#define CACO_ATTN(HD_, C_) \
  hipLaunchKernelGGL((attention_kernel<HD_, C_>), grid, dim3(256), 0, st, qk, vt, key_mask, seq, S_pad, heads, out, scale_log2)
  if (head_dim == 96) {
    if (causal) CACO_ATTN(96, true); else CACO_ATTN(96, false);
  } else {
    if (causal) CACO_ATTN(64, true); else CACO_ATTN(64, false);
  }
#undef CACO_ATTN
  return check_hip(hipGetLastError(), "attention launch");
}

}  // namespace caco
