// Fused front end: 16 kHz waveform -> STFT magnitude -> HTK mel filterbank -> log -> 16x16 patches.
//
// Replaces compute_mel_spectrogram + spectrogram_to_patches + the four H2D copies of
// prepare_audio_batch (src/eval/eval_caco_torch.py:41-151,181-206) with one kernel whose only HBM
// traffic is the sample buffer in (coalesced float4) and the patch rows out (16-byte stores, already
// in the encoder's [B, S, 256] token layout, bf16 for the GEMM or fp32 for API parity).
//
// Geometry is the reference's fixed front end: hop 160, periodic Hann(400) centred in a 512-point
// frame (56 zeros each side, torch.stft semantics, eval_caco_torch.py:81-89), |rFFT| (power 1, :91),
// 128 HTK mel filters over 257 bins (:94-103), log(x + 1e-5) * scale + bias (:104).
//
// One workgroup = 256 threads = 16 consecutive frames (= one row of 8 patches) x 16 threads per
// frame; it is persistent over several 16-frame blocks.  The 512-point real FFT is a 256-point complex
// FFT of the even/odd packed frame, done as two register-resident radix-16 passes with one
// transposition through LDS, then the real-input split.  Everything that does not change from block to
// block lives in REGISTERS for the life of the workgroup: the thread's 28 Hann taps, its 30 W256
// twiddles, its 38 mel filter weights and 8 filter starts (since round 2; round 1 staged them in LDS);
// only the W512 twiddles of the real-input split are staged in LDS, once per workgroup.  fp32 throughout.
#include <math.h>
#include <vector>

#include "common.h"
#include "kernels.h"

namespace caco {
namespace {

constexpr int HOP = 160, WIN = 400, NFFT = 512, NMEL = 128, NBIN = 257, WOFF = (NFFT - WIN) / 2;
constexpr int FPB = 16;                               // frames per workgroup
constexpr int SOFF = 32;                              // first sample a frame's FFT ever reads: taps below 56 carry a zero window
constexpr int SPAN = (FPB - 1) * HOP + NFFT - 2 * SOFF;   // 2848 samples feed one workgroup
// HTK filter supports at 16 kHz / 257 bins, as the maximum over each group of 16 consecutive filters (thread t owns filters
// t, t+16, ...): 38 multiply-adds per thread, weights zero-padded to the group maximum.  ensure_tables() checks the
// filterbank it builds against these bounds.
constexpr int MEL_GROUP_TAPS[NMEL / 16] = {2, 2, 2, 3, 4, 6, 8, 11};
constexpr int MEL_TAPS = 2 + 2 + 2 + 3 + 4 + 6 + 8 + 11;

struct MelTables {            // device image; the first LDS_FLOATS floats are staged into LDS verbatim
  float tw512[512];           // e^{-2 pi i k / 512}, k = 0..255, interleaved re/im
  float tw256[512];           // e^{-2 pi i n1 k2 / 256}, index (n1*16 + k2) (symmetric), interleaved re/im
  float hann512[NFFT];        // periodic Hann(400) centred in the 512-point frame, zeros outside
  float melw[MEL_TAPS * 16];  // [tap slot][t]: weight of thread t's filter t + 16 j at bin mel_start + i
  int mel_start[NMEL];        // first bin of filter m
};
constexpr int LDS_FLOATS = 512;
static_assert(sizeof(MelTables) % 16 == 0, "MelTables must be float4-copyable");

typedef float c32 __attribute__((ext_vector_type(2)));
constexpr int XP = 272;       // complex pitch per frame: 16*17, and 2*XP = 32 (mod 64) banks

struct __attribute__((aligned(16))) MelSmem {
  float samp[SPAN];                 // samples, later the [16][128] log-mel tile
  float tw512[512];
  c32 buf[FPB * XP];                // FFT transposition / spectrum, later magnitudes
};
constexpr int TP = NMEL + 16;  // log-mel tile pitch
static_assert(SPAN % 4 == 0 && SPAN >= FPB * TP, "sample span is float4-copyable and holds the log-mel tile");

// Complex numbers are native 2-vectors so that every complex add / multiply maps onto the packed fp32 VALU ops
// (v_pk_add_f32, v_pk_mul_f32, v_pk_fma_f32: two floats per lane per issue slot); the swaps and sign flips of conj and
// multiply-by-i fold into the op_sel / neg modifiers of those instructions.
__device__ __forceinline__ c32 cmul(c32 a, c32 b) {   // (a.x b.x - a.y b.y, a.x b.y + a.y b.x)
  const c32 bs = {-b.y, b.x};
  return a.xx * b + a.yy * bs;
}
__device__ __forceinline__ c32 mul_mi(c32 a) { return (c32){a.y, -a.x}; }   // a * (-i)

// forward 4-point DFT (W4 = -i)
__device__ __forceinline__ void dft4(c32& x0, c32& x1, c32& x2, c32& x3) {
  const c32 s02 = x0 + x2, d02 = x0 - x2, s13 = x1 + x3, d13 = mul_mi(x1 - x3);
  x0 = s02 + s13;
  x2 = s02 - s13;
  x1 = d02 + d13;
  x3 = d02 - d13;
}

// forward 16-point DFT in registers, natural order in and out (two radix-4 passes).
__device__ __forceinline__ void dft16(c32 (&v)[16]) {
  constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, R2 = 0.70710678118654752f;
  // pass 1: for each b, DFT4 over a of v[b + 4a]
#pragma unroll
  for (int b = 0; b < 4; ++b) dft4(v[b], v[b + 4], v[b + 8], v[b + 12]);
  // now v[b + 4c] = y[b][c]; twiddle by W16^{b c}
  v[1 + 4] = cmul(v[1 + 4], (c32){C1, -S1});        // bc = 1
  v[1 + 8] = cmul(v[1 + 8], (c32){R2, -R2});        // 2
  v[1 + 12] = cmul(v[1 + 12], (c32){S1, -C1});      // 3
  v[2 + 4] = cmul(v[2 + 4], (c32){R2, -R2});        // 2
  v[2 + 8] = mul_mi(v[2 + 8]);                      // 4
  v[2 + 12] = cmul(v[2 + 12], (c32){-R2, -R2});     // 6
  v[3 + 4] = cmul(v[3 + 4], (c32){S1, -C1});        // 3
  v[3 + 8] = cmul(v[3 + 8], (c32){-R2, -R2});       // 6
  v[3 + 12] = cmul(v[3 + 12], (c32){-C1, S1});      // 9
  // pass 2: for each c, DFT4 over b of y[b][c] -> X[c + 4d]
#pragma unroll
  for (int c = 0; c < 4; ++c) dft4(v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]);
  // v[4c + d] holds X[c + 4d]: transpose the 4x4 index grid back to natural order
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int d = c + 1; d < 4; ++d) {
      const c32 tmp = v[4 * c + d];
      v[4 * c + d] = v[4 * d + c];
      v[4 * d + c] = tmp;
    }
}

// MODE: MEL_NATURAL_F32 -> out fp32 [B, frames_out, 128]; MEL_PATCH_F32 / MEL_PATCH_BF16 -> [B, S, 256]
//
// Instruction budget (the kernel is VALU/LDS-issue bound, not HBM bound: 9e7 wave instructions per batch of 256 clips in
// round 1).  Everything a thread needs that does not change from block to block lives in registers for the life of the
// workgroup (its 28 window taps, its 38 filter weights and 8 filter starts); the window is a zero-padded 512-tap table so
// pass A has no lane-dependent branch, taps n2 = 0 and 15 are compile-time zeros, the filterbank loop is fully unrolled
// with compile-time trip counts, and the magnitude uses the raw v_sqrt_f32 (1 ulp; no denormal rescue sequence).
static_assert(sizeof(MelSmem) * 2 <= 160 * 1024, "two workgroups per CU");
template <int MODE>
__global__ __launch_bounds__(256, 2) void mel_kernel(const float* __restrict__ wav, int64_t n_samples,
                                                  const MelTables* __restrict__ tables, void* __restrict__ out,
                                                  int frames_out, int rows_out, int S, float scale, float bias, int nblk,
                                                  float* __restrict__ tinds, float* __restrict__ finds,
                                                  float* __restrict__ mask, const int64_t* __restrict__ lengths) {
  __shared__ MelSmem sm;
  const int tid = threadIdx.x, fl = tid >> 4, t = tid & 15;
  const int b = blockIdx.y;
  const float* w = wav + (int64_t)b * n_samples;
  int64_t n_valid = n_samples;          // samples past this bound read as the STFT's zero padding (:78)
  if constexpr (MODE != MEL_NATURAL_F32) {
    // ---- patch bookkeeping of spectrogram_to_patches (eval_caco_torch.py:132-144), done by the same launch --------------
    // lengths != null: clip b holds lengths[b] real samples (the rest of its row is zero padding): its spectrogram has
    // ceil(len / 160) frames and only the patches of those frames are valid - what the reference gets by running
    // prepare_audio_batch (:181-206) clip by clip: the sample fetch is bounded by lengths[b], so the tail frames see the
    // STFT's zero padding (:78) whatever the row holds past the clip (a clip cut out of a longer buffer, say).
    // Rows [valid, S) of the patch tensor are zero, their indices 0, their mask 0.
    constexpr int nfreq = NMEL / 16;
    if (lengths) {
      int64_t len = lengths[b];
      len = len < 0 ? 0 : (len > n_samples ? n_samples : len);
      n_valid = len;                    // whatever the caller's buffer holds past lengths[b] is not part of the clip
      const int64_t full_b = ((len + HOP - 1) / HOP / FPB) * nfreq;
      if (full_b < rows_out) rows_out = (int)full_b;
      const int nblk_b = (rows_out + nfreq - 1) / nfreq;
      if (nblk_b < nblk) nblk = nblk_b;
    }
    for (int p = blockIdx.x * 256 + tid; p < S; p += gridDim.x * 256) {
      const bool keep = p < rows_out;
      const int q = keep ? p : 0;
      if (tinds) tinds[(int64_t)b * S + p] = (float)(q / nfreq);
      if (finds) finds[(int64_t)b * S + p] = (float)(q % nfreq);
      if (mask) mask[(int64_t)b * S + p] = keep ? 1.f : 0.f;
    }
    for (int p = rows_out + blockIdx.x; p < S; p += gridDim.x) {
      const int64_t o = ((int64_t)b * S + p) * 256 + tid;
      if constexpr (MODE == MEL_PATCH_BF16) reinterpret_cast<bf16_t*>(out)[o] = (bf16_t)0.f;
      else reinterpret_cast<float*>(out)[o] = 0.f;
    }
    if ((int)blockIdx.x >= nblk) return;      // nothing to transform (whole-workgroup exit: no barrier is skipped by a part)
  }

  // ---- constant tables: staged ONCE per workgroup; the workgroup then walks several 16-frame blocks ----------
  {
    const f32x4* src = reinterpret_cast<const f32x4*>(tables);
    f32x4* dst = reinterpret_cast<f32x4*>(sm.tw512);
    for (int i = tid; i < LDS_FLOATS / 4; i += 256) dst[i] = src[i];
  }
  c32 hreg[14];                          // window taps of complex samples n = t + 16 n2, n2 = 1..14
#pragma unroll
  for (int n2 = 1; n2 < 15; ++n2) hreg[n2 - 1] = *reinterpret_cast<const c32*>(&tables->hann512[2 * (t + 16 * n2)]);
  // W256^(n1 k2) is symmetric in (n1, k2): fetched as [k2][n1 = t], consecutive lanes read consecutive words
  c32 twreg[15];
#pragma unroll
  for (int k2 = 1; k2 < 16; ++k2) twreg[k2 - 1] = *reinterpret_cast<const c32*>(&tables->tw256[2 * (k2 * 16 + t)]);
  float wreg[MEL_TAPS];
#pragma unroll
  for (int i = 0; i < MEL_TAPS; ++i) wreg[i] = tables->melw[i * 16 + t];
  int mstart[NMEL / 16];
#pragma unroll
  for (int j = 0; j < NMEL / 16; ++j) mstart[j] = tables->mel_start[t + 16 * j];
  const bool vec_ok = (n_samples & 3) == 0 && (reinterpret_cast<uintptr_t>(wav) & 15) == 0;

  // the 2848 samples of a block as 712 float4, <= 3 per thread: fetched into registers one block ahead, so the HBM
  // latency of block i+1 hides under the transform of block i
  auto fetch = [&](int blk_, f32x4 (&pf)[3]) {
    const int64_t g0 = (int64_t)blk_ * (FPB * HOP) + SOFF;
#pragma unroll
    for (int r3 = 0; r3 < 3; ++r3) {
      const int i = tid + 256 * r3;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (i < SPAN / 4) {
        const int64_t g = g0 + 4 * i;
        if (vec_ok && g + 3 < n_valid) {
          v = *reinterpret_cast<const f32x4*>(w + g);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = (g + r < n_valid) ? w[g + r] : 0.f;     // zero pad, :78
        }
      }
      pf[r3] = v;
    }
  };
  f32x4 pf[3];
  if ((int)blockIdx.x < nblk) fetch(blockIdx.x, pf);

  for (int blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const int f0 = blk * FPB;
    __syncthreads();          // previous block's tile (aliases samp) fully stored; tables visible on the first pass
#pragma unroll
    for (int r3 = 0; r3 < 3; ++r3) {
      const int i = tid + 256 * r3;
      if (i < SPAN / 4) *reinterpret_cast<f32x4*>(&sm.samp[4 * i]) = pf[r3];
    }
    __syncthreads();
    if (blk + (int)gridDim.x < nblk) fetch(blk + gridDim.x, pf);

    // From here to the tile write every exchange stays inside one frame = 16 lanes of ONE wave: LDS operations of a
    // wave execute in order, so wave-level ordering (no workgroup barrier) is enough between the passes.
    // ---- pass A: thread n1 = t transforms z[n1 + 16 n2] over n2, twiddles by W256^{n1 k2} ----------
    c32 v[16];
    {
      // complex sample n of this frame is the c32 at index sbase + n - t of the block's samples; the skipped first 16 complex
      // samples (window zero) would lie before the array for the block's first frame, so the index - not a pointer - carries
      // the offset (a pointer before the array is undefined behaviour even when never dereferenced: UBSan on the wavesim build)
      const c32* sc = reinterpret_cast<const c32*>(sm.samp);
      const int sbase = (fl * HOP - SOFF) / 2 + t;
      v[0] = (c32){0.f, 0.f};                  // n < 16: real samples < 32, window zero
      v[15] = (c32){0.f, 0.f};                 // n >= 240: real samples >= 480, window zero
#pragma unroll
      for (int n2 = 1; n2 < 15; ++n2) v[n2] = sc[sbase + 16 * n2] * hreg[n2 - 1];
    }
    dft16(v);
    {
      c32* col = sm.buf + fl * XP + t * 17;
      col[0] = v[0];                           // W^0
#pragma unroll
      for (int k2 = 1; k2 < 16; ++k2) col[k2] = cmul(v[k2], twreg[k2 - 1]);
    }
    __builtin_amdgcn_wave_barrier();
    // ---- pass B: thread k2 = t transforms over n1 -> X[k2 + 16 k1] -----------------------------------
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) v[n1] = sm.buf[fl * XP + n1 * 17 + t];
    dft16(v);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) sm.buf[fl * XP + t + 16 * k1] = v[k1];
    __builtin_amdgcn_wave_barrier();

    // ---- real-input split + magnitude: R[k] = ((Zk + conj Z-k) - i w^k (Zk - conj Z-k)) / 2 ----------
    // (the factor 1/2 is folded into the filterbank weights: exact in binary floating point)
    float mag[16], mag256;
    {
      const c32* X = sm.buf + fl * XP;
      const c32* tw = reinterpret_cast<const c32*>(sm.tw512) + t;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int k = t + 16 * j;
        const c32 zk = X[k];
        const c32 zr = X[(256 - k) & 255];
        const c32 zc = {zr.x, -zr.y};
        const c32 e = zk + zc, d = zk - zc;
        const c32 r = e + mul_mi(cmul(tw[16 * j], d));       // 2 R[k] = e - i w^k d
        mag[j] = __builtin_amdgcn_sqrtf(r.x * r.x + r.y * r.y);
      }
      const c32 z0 = X[0];
      mag256 = 2.f * fabsf(z0.x - z0.y);       // 2 R[256] = 2 (Re Z0 - Im Z0) (real)
    }
    __builtin_amdgcn_wave_barrier();
    // inside this frame's own spectrum region (544 floats for 257 magnitudes); odd frames are shifted by 16 floats so that
    // the two frames of a 32-lane ds_read_b32 group do not gather from identical banks (544 = 0 mod 32)
    float* magbuf = reinterpret_cast<float*>(sm.buf) + fl * (2 * XP) + (fl & 1) * 16;
#pragma unroll
    for (int j = 0; j < 16; ++j) magbuf[t + 16 * j] = mag[j];
    if (t == 0) magbuf[256] = mag256;
    __builtin_amdgcn_wave_barrier();

    // ---- mel filterbank + log; thread owns mels t, t+16, ...: compile-time trip counts, weights in registers -------------
    float melv[NMEL / 16];
    {
      int slot = 0;
#pragma unroll
      for (int j = 0; j < NMEL / 16; ++j) {
        const float* mp = magbuf + mstart[j];
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < MEL_GROUP_TAPS[j]; ++i) acc += mp[i] * wreg[slot + i];
        slot += MEL_GROUP_TAPS[j];
        melv[j] = __logf(acc + 1e-5f) * scale + bias;
      }
    }
    __syncthreads();                            // every frame is done reading the samples: samp becomes the tile
    float* tile = sm.samp;                      // [16 frames][TP]: pitch 144 = 16 (mod 64) banks keeps the two frames of a
                                                // 32-lane write group and the 4 frames of a b128 read group on distinct banks
#pragma unroll
    for (int j = 0; j < NMEL / 16; ++j) tile[fl * TP + t + 16 * j] = melv[j];
    __syncthreads();

    // ---- coalesced 16/32-byte stores -----------------------------------------------------------------
    if constexpr (MODE == MEL_NATURAL_F32) {
      const int frame = f0 + (tid >> 4);
      if (frame < frames_out) {
        const int m0 = (tid & 15) * 8;
        float* op = reinterpret_cast<float*>(out) + ((int64_t)b * frames_out + frame) * NMEL + m0;
        const float* tp = tile + (tid >> 4) * TP + m0;
        *reinterpret_cast<f32x4*>(op) = *reinterpret_cast<const f32x4*>(tp);
        *reinterpret_cast<f32x4*>(op + 4) = *reinterpret_cast<const f32x4*>(tp + 4);
      }
    } else {
      // patch row p = blk*8 + f holds mel[f0 + tt][f*16 + m], tt-major (eval_caco_torch.py:124-129)
      const int f = tid >> 5, rem = tid & 31, tt = rem >> 1, m0 = (rem & 1) * 8;
      const int p = blk * 8 + f;
      if (p < rows_out) {
        const float* tp = tile + tt * TP + f * 16 + m0;
        const int64_t o = ((int64_t)b * S + p) * 256 + tt * 16 + m0;
        if constexpr (MODE == MEL_PATCH_BF16) {
          bf16x8 pk;
#pragma unroll
          for (int r = 0; r < 8; ++r) pk[r] = (bf16_t)tp[r];
          *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16_t*>(out) + o) = pk;
        } else {
          float* op = reinterpret_cast<float*>(out) + o;
          *reinterpret_cast<f32x4*>(op) = *reinterpret_cast<const f32x4*>(tp);
          *reinterpret_cast<f32x4*>(op + 4) = *reinterpret_cast<const f32x4*>(tp + 4);
        }
      }
    }
  }
}

MelTables* g_tables_dev[CACO_MAX_DEVICES] = {};   // device copies, one per device, created at first use

int ensure_tables(MelTables** out) {
  int dev = 0;
  CACO_HIP(hipGetDevice(&dev));
  CACO_REQUIRE(dev >= 0 && dev < CACO_MAX_DEVICES, "mel: device index %d out of range", dev);
  if (g_tables_dev[dev]) { *out = g_tables_dev[dev]; return CACO_OK; }
  std::vector<char> hostbuf(sizeof(MelTables), 0);
  MelTables* h = reinterpret_cast<MelTables*>(hostbuf.data());
  const double PI = 3.14159265358979323846;
  for (int k = 0; k < WIN; ++k) h->hann512[WOFF + k] = (float)(0.5 - 0.5 * cos(2.0 * PI * k / WIN));   // periodic; zeros outside
  for (int n1 = 0; n1 < 16; ++n1)
    for (int k2 = 0; k2 < 16; ++k2) {
      const double a = -2.0 * PI * (double)(n1 * k2) / 256.0;
      h->tw256[2 * (n1 * 16 + k2)] = (float)cos(a);
      h->tw256[2 * (n1 * 16 + k2) + 1] = (float)sin(a);
    }
  for (int k = 0; k < 256; ++k) {
    const double a = -2.0 * PI * (double)k / 512.0;
    h->tw512[2 * k] = (float)cos(a);
    h->tw512[2 * k + 1] = (float)sin(a);
  }
  // torchaudio.functional.melscale_fbanks(257, 0, 8000, 128, 16000, norm=None, mel_scale="htk")
  const double f_max = 8000.0, m_max = 2595.0 * log10(1.0 + f_max / 700.0);
  std::vector<double> f_pts(NMEL + 2);
  for (int i = 0; i < NMEL + 2; ++i) f_pts[i] = 700.0 * (pow(10.0, (m_max * i / (NMEL + 1)) / 2595.0) - 1.0);
  int slot0[NMEL / 16];
  for (int j = 0, s0 = 0; j < NMEL / 16; ++j) { slot0[j] = s0; s0 += MEL_GROUP_TAPS[j]; }
  for (int m = 0; m < NMEL; ++m) {
    const int t = m % 16, j = m / 16;
    int start = -1, cnt = 0;
    for (int k = 0; k < NBIN; ++k) {
      const double f = 8000.0 * k / (NBIN - 1);
      const double down = (f - f_pts[m]) / (f_pts[m + 1] - f_pts[m]);
      const double up = (f_pts[m + 2] - f) / (f_pts[m + 2] - f_pts[m + 1]);
      const double wgt = fmax(0.0, fmin(down, up));
      if (wgt > 0.0) {
        if (start < 0) start = k;
        if (k != start + cnt || cnt >= MEL_GROUP_TAPS[j]) {
          set_error("mel filterbank: filter %d has a non-contiguous support or more than %d bins", m, MEL_GROUP_TAPS[j]);
          return CACO_ERR_INVALID;
        }
        h->melw[(slot0[j] + cnt) * 16 + t] = 0.5f * (float)wgt;     // the kernel hands over 2 |R[k]|
        ++cnt;
      }
    }
    // zero-weight padding taps read bins start .. start + taps - 1: keep them inside the 257 magnitudes
    if (start < 0) start = 0;                   // empty filter (HTK filter 0 at this resolution): all-zero weights
    if (start + MEL_GROUP_TAPS[j] > NBIN) {
      set_error("mel filterbank: padded taps of filter %d run past bin %d", m, NBIN - 1);
      return CACO_ERR_INVALID;
    }
    h->mel_start[m] = start;
  }
  MelTables* d = nullptr;
  CACO_HIP(hipMalloc(reinterpret_cast<void**>(&d), sizeof(MelTables)));
  CACO_HIP(hipMemcpy(d, h, sizeof(MelTables), hipMemcpyHostToDevice));
  g_tables_dev[dev] = d;
  *out = d;
  return CACO_OK;
}

// workgroups per clip: enough to fill the chip a few times over, few enough that each one amortises its table
// staging over several 16-frame blocks
int mel_grid_x(int nblk, int batch) {
  const int want = (2048 + batch - 1) / batch;      // ~8 workgroups per CU chip-wide
  return nblk < want ? nblk : (want < 1 ? 1 : want);
}

}  // namespace

int mel_frontend(const float* wav, int batch, int64_t n_samples, int max_patches, float scale, float bias, void* out,
                 int mode, float* tinds, float* finds, float* mask, hipStream_t st, const int64_t* lengths) {
  CACO_REQUIRE(wav && out && batch > 0 && n_samples > 0, "mel: bad arguments (batch %d, n_samples %lld)", batch, (long long)n_samples);
  CACO_REQUIRE(batch <= 65535, "mel: batch %d exceeds the grid limit", batch);
  MelTables* g_tables = nullptr;
  int rc = ensure_tables(&g_tables);
  if (rc) return rc;
  const int frames = (int)((n_samples + HOP - 1) / HOP);
  if (mode == MEL_NATURAL_F32) {
    const int nblk = (frames + FPB - 1) / FPB;
    const dim3 grid(mel_grid_x(nblk, batch), batch);
    hipLaunchKernelGGL(mel_kernel<MEL_NATURAL_F32>, grid, dim3(256), 0, st, wav, n_samples, g_tables, out, frames, 0, 0,
                       scale, bias, nblk, nullptr, nullptr, nullptr, nullptr);
    return check_hip(hipGetLastError(), "mel launch");
  }
  CACO_REQUIRE(max_patches > 0, "mel: max_patches must be positive");
  const int n_tp = frames / FPB, nfreq = NMEL / 16;
  const int full = n_tp * nfreq;
  const int valid = full < max_patches ? full : max_patches;   // truncation branch keeps the first max_patches
  const int blocks = (valid + nfreq - 1) / nfreq;
  // one launch: the transform blocks, the zero tail rows [valid, max_patches) and the index / mask arrays
  const dim3 grid(mel_grid_x(blocks < 1 ? 1 : blocks, batch), batch);
  if (mode == MEL_PATCH_BF16)
    hipLaunchKernelGGL(mel_kernel<MEL_PATCH_BF16>, grid, dim3(256), 0, st, wav, n_samples, g_tables, out, frames, valid,
                       max_patches, scale, bias, blocks, tinds, finds, mask, lengths);
  else
    hipLaunchKernelGGL(mel_kernel<MEL_PATCH_F32>, grid, dim3(256), 0, st, wav, n_samples, g_tables, out, frames, valid,
                       max_patches, scale, bias, blocks, tinds, finds, mask, lengths);
  return check_hip(hipGetLastError(), "mel patch launch");
}

}  // namespace caco
