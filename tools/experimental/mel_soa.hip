// NOT BUILT.  Two-frames-per-thread (structure-of-arrays) variant of the mel front end, kept for the record; it is a drop-in
// replacement of cacophony_amd/csrc/mel.hip (same entry points, passes the same tests).
//
// Idea: every value is a pair (frame A, frame B = A + 8) in one 64-bit register pair, real and imaginary parts in separate
// pairs, all LDS buffers pair-interleaved: multiply-by-(-i), conj and the twiddle products need no swizzle / sign-flip /
// s_nop instructions (250 of the 1100 loop instructions of the shipped kernel), and the magnitude, filterbank and log
// stages are packed too.  Result on MI355X, 256 clips x 10 s (tools/mel_bench.py):
//   shipped kernel (interleaved re/im, 256 threads, 2 workgroups = 8 waves per CU)        171 us
//   this kernel (128 threads, 53 KB LDS -> 3 workgroups = 6 waves per CU)                  202-208 us
//   (an earlier form with planar instead of pair-interleaved LDS: the compiler merged neighbouring loads of ONE frame into
//    ds_read2_b32 and re-paired with 320 v_mov: 234 us)
// Why: the interleaved layout already packs the complex adds two floats per instruction; what this form removes is only
// the overhead share, 240 -> 175 hot instructions per frame (-27 %), and it pays for that with 25 % fewer resident waves
// (LDS: the pair layout needs both frames' transposition buffers in one workgroup).  The kernel is bound by instructions
// issued per wave (one per ~12 cycles), so 6 waves x 175 is no better than 8 waves x 240.  A version that reads the
// samples straight from global memory (no 12.5 KB sample stage -> 4 workgroups per CU) is the untested next step.

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
// frame.  The 512-point real FFT is a 256-point complex FFT of the even/odd packed frame, done as
// two register-resident radix-16 passes with one transposition through LDS, then the real-input
// split.  The Hann window, the FFT twiddles and the mel filterbank (CSR: 506 non-zero weights) are
// staged in LDS once per workgroup.  fp32 throughout.
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
  float melw[MEL_TAPS * 16];  // [tap slot][t]: weight of thread t's filter t + 16 j at bin mel_start + i
  int mel_start[NMEL];        // first bin of filter m
  float tw256[512];           // e^{-2 pi i n1 k2 / 256}, index (n1*16 + k2) (symmetric), interleaved re/im
  float hann512[NFFT];        // periodic Hann(400) centred in the 512-point frame, zeros outside
};
constexpr int LDS_FLOATS = 512 + MEL_TAPS * 16 + NMEL;
static_assert(sizeof(MelTables) % 16 == 0, "MelTables must be float4-copyable");

constexpr int XP = 272;       // complex pitch per frame: 16*17, and 2*XP = 32 (mod 64) banks

// LDS.  A thread transforms TWO frames (A = G, B = G + 8) and every value it moves is a pair (A, B): all hot buffers are
// laid out pair-interleaved, so one 8-byte LDS access moves the pair into / out of an aligned register pair.
//   samp: pairs (x[r], x[r + 8 hops]) of even samples, then of odd samples (frame B's sample is frame A's + 1280);
//   buf:  per group G a plane of XP real pairs and a plane of XP imaginary pairs; the group pitch is 32 banks (mod 64) so
//         that the two groups of a 32-lane b64 pass never share a bank.
constexpr int FB = FPB / 2;                                // frame B = frame A + FB
constexpr int NPAIR = (FB - 1) * (HOP / 2) + (NFFT - 2 * SOFF) / 2;   // 784 sample pairs per parity
constexpr int GP = 4 * XP + 32;                            // floats per group in buf
constexpr int TP = NMEL + 16;                              // log-mel tile pitch
struct __attribute__((aligned(16))) MelSmem {
  float samp[4 * NPAIR];            // even pairs | odd pairs, later the [16][TP] log-mel tile
  float tw512[512];                 // staged tables: tw512 | melw | mel_start (contiguous, LDS_FLOATS)
  float melw[MEL_TAPS * 16];
  int mel_start[NMEL];
  float buf[FB * GP];               // FFT transposition / spectrum, later magnitudes
};
static_assert(SPAN % 4 == 0 && 4 * NPAIR >= FPB * TP && GP % 64 == 32, "layout");

// Two frames per thread, structure-of-arrays: every value of the transform is a PAIR (frame A, frame B) in one 64-bit
// register pair, real and imaginary parts in separate pairs.  All arithmetic is then packed fp32 (v_pk_add / v_pk_mul /
// v_pk_fma: two frames per instruction) with scalar twiddles broadcast by op_sel; multiply-by-(-i) and conj are register
// renaming, so none of the swizzle / sign-flip / s_nop traffic of an interleaved (re, im) layout remains (round 2
// counted 250 of 1100 loop instructions as such overhead).  The kernel is bound by instructions issued per wave
// (one per ~12 cycles), so this is what pays: LDS operations move pairs too (ds_read2st64 / ds_write2st64: frame B sits a
// multiple of 256 bytes behind frame A in every buffer).
typedef float p2 __attribute__((ext_vector_type(2)));
struct cx { p2 re, im; };
__device__ __forceinline__ cx cadd(cx a, cx b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ cx csub(cx a, cx b) { return {a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ cx cmulc(cx a, float c, float s) {   // a * (c + i s), scalar twiddle
  return {a.re * c - a.im * s, a.re * s + a.im * c};
}

// forward 4-point DFT (W4 = -i)
__device__ __forceinline__ void dft4(cx& x0, cx& x1, cx& x2, cx& x3) {
  const cx s02 = cadd(x0, x2), d02 = csub(x0, x2), s13 = cadd(x1, x3), d13 = csub(x1, x3);
  x0 = cadd(s02, s13);
  x2 = csub(s02, s13);
  x1 = {d02.re + d13.im, d02.im - d13.re};     // d02 + (-i) d13
  x3 = {d02.re - d13.im, d02.im + d13.re};     // d02 - (-i) d13
}

// forward 16-point DFT in registers, natural order in and out (two radix-4 passes).
__device__ __forceinline__ void dft16(cx (&v)[16]) {
  constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, R2 = 0.70710678118654752f;
  // pass 1: for each b, DFT4 over a of v[b + 4a]
#pragma unroll
  for (int b = 0; b < 4; ++b) dft4(v[b], v[b + 4], v[b + 8], v[b + 12]);
  // now v[b + 4c] = y[b][c]; twiddle by W16^{b c}
  v[1 + 4] = cmulc(v[1 + 4], C1, -S1);                                              // bc = 1
  v[1 + 8] = {(v[1 + 8].re + v[1 + 8].im) * R2, (v[1 + 8].im - v[1 + 8].re) * R2};   // 2: (R2, -R2)
  v[1 + 12] = cmulc(v[1 + 12], S1, -C1);                                            // 3
  v[2 + 4] = {(v[2 + 4].re + v[2 + 4].im) * R2, (v[2 + 4].im - v[2 + 4].re) * R2};   // 2
  v[2 + 8] = {v[2 + 8].im, -v[2 + 8].re};                                           // 4: -i
  v[2 + 12] = {(v[2 + 12].im - v[2 + 12].re) * R2, (v[2 + 12].re + v[2 + 12].im) * -R2};   // 6: (-R2, -R2)
  v[3 + 4] = cmulc(v[3 + 4], S1, -C1);                                              // 3
  v[3 + 8] = {(v[3 + 8].im - v[3 + 8].re) * R2, (v[3 + 8].re + v[3 + 8].im) * -R2};  // 6
  v[3 + 12] = cmulc(v[3 + 12], -C1, S1);                                            // 9
  // pass 2: for each c, DFT4 over b of y[b][c] -> X[c + 4d]
#pragma unroll
  for (int c = 0; c < 4; ++c) dft4(v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]);
  // v[4c + d] holds X[c + 4d]: transpose the 4x4 index grid back to natural order
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int d = c + 1; d < 4; ++d) {
      const cx tmp = v[4 * c + d];
      v[4 * c + d] = v[4 * d + c];
      v[4 * d + c] = tmp;
    }
}

// MODE: MEL_NATURAL_F32 -> out fp32 [B, frames_out, 128]; MEL_PATCH_F32 / MEL_PATCH_BF16 -> [B, S, 256]
//
// One workgroup = 128 threads = 8 groups of 16 lanes; group G transforms frames G and G + 8 of a 16-frame block (the 8
// frames between them keep the two groups of a 32-lane LDS pass on different banks: 160 samples = 32 banks, 544 = 32).
// Instruction budget: everything a thread needs that does not change from block to block lives in registers for the
// life of the workgroup (28 window taps, 30 W256 twiddles, 38 filter weights, 8 filter starts); the window is a
// zero-padded 512-tap table so pass A has no lane-dependent branch, taps n2 = 0 and 15 are compile-time zeros, the
// filterbank loop is fully unrolled with compile-time trip counts, the magnitude uses the raw v_sqrt_f32.
constexpr int NT = 128;                        // threads per workgroup
static_assert(sizeof(MelSmem) * 3 <= 160 * 1024, "three workgroups per CU");
template <int MODE>
__global__ __launch_bounds__(NT, 2) void mel_kernel(const float* __restrict__ wav, int64_t n_samples,
                                                  const MelTables* __restrict__ tables, void* __restrict__ out,
                                                  int frames_out, int rows_out, int S, float scale, float bias, int nblk,
                                                  float* __restrict__ tinds, float* __restrict__ finds,
                                                  float* __restrict__ mask, const int64_t* __restrict__ lengths) {
  __shared__ MelSmem sm;
  const int tid = threadIdx.x, G = tid >> 4, t = tid & 15;
  const int b = blockIdx.y;
  const float* w = wav + (int64_t)b * n_samples;
  if constexpr (MODE != MEL_NATURAL_F32) {
    // ---- patch bookkeeping of spectrogram_to_patches (eval_caco_torch.py:132-144), done by the same launch --------------
    // lengths != null: clip b holds lengths[b] real samples (the rest of its row is zero padding): its spectrogram has
    // ceil(len / 160) frames and only the patches of those frames are valid - what the reference gets by running
    // prepare_audio_batch (:181-206) clip by clip.  The mel values of the frames it does have are the same either way: the
    // STFT pads with zeros (:78).  Rows [valid, S) of the patch tensor are zero, their indices 0, their mask 0.
    constexpr int nfreq = NMEL / 16;
    if (lengths) {
      int64_t len = lengths[b];
      len = len < 0 ? 0 : (len > n_samples ? n_samples : len);
      const int64_t full_b = ((len + HOP - 1) / HOP / FPB) * nfreq;
      if (full_b < rows_out) rows_out = (int)full_b;
      const int nblk_b = (rows_out + nfreq - 1) / nfreq;
      if (nblk_b < nblk) nblk = nblk_b;
    }
    for (int p = blockIdx.x * NT + tid; p < S; p += gridDim.x * NT) {
      const bool keep = p < rows_out;
      const int q = keep ? p : 0;
      if (tinds) tinds[(int64_t)b * S + p] = (float)(q / nfreq);
      if (finds) finds[(int64_t)b * S + p] = (float)(q % nfreq);
      if (mask) mask[(int64_t)b * S + p] = keep ? 1.f : 0.f;
    }
    for (int p = rows_out + blockIdx.x; p < S; p += gridDim.x) {
      const int64_t o = ((int64_t)b * S + p) * 256 + 2 * tid;
      if constexpr (MODE == MEL_PATCH_BF16) *reinterpret_cast<unsigned int*>(reinterpret_cast<bf16_t*>(out) + o) = 0u;
      else *reinterpret_cast<float2*>(reinterpret_cast<float*>(out) + o) = make_float2(0.f, 0.f);
    }
    if ((int)blockIdx.x >= nblk) return;      // nothing to transform (whole-workgroup exit: no barrier is skipped by a part)
  }

  // ---- constant tables: staged ONCE per workgroup; the workgroup then walks several 16-frame blocks ----------
  {
    const f32x4* src = reinterpret_cast<const f32x4*>(tables);
    f32x4* dst = reinterpret_cast<f32x4*>(sm.tw512);
    for (int i = tid; i < LDS_FLOATS / 4; i += NT) dst[i] = src[i];
  }
  float he[14], ho[14];                  // window taps of complex samples n = t + 16 n2, n2 = 1..14 (even / odd real sample)
#pragma unroll
  for (int n2 = 1; n2 < 15; ++n2) {
    const float2 h = *reinterpret_cast<const float2*>(&tables->hann512[2 * (t + 16 * n2)]);
    he[n2 - 1] = h.x;
    ho[n2 - 1] = h.y;
  }
  // W256^(n1 k2) is symmetric in (n1, k2): fetched as [k2][n1 = t], consecutive lanes read consecutive words
  float twc[15], tws[15];
#pragma unroll
  for (int k2 = 1; k2 < 16; ++k2) {
    const float2 tw = *reinterpret_cast<const float2*>(&tables->tw256[2 * (k2 * 16 + t)]);
    twc[k2 - 1] = tw.x;
    tws[k2 - 1] = tw.y;
  }
  const bool vec_ok = (n_samples & 3) == 0 && (reinterpret_cast<uintptr_t>(wav) & 15) == 0;

  // the 2848 samples of a block as 712 float4, <= 6 per thread: fetched into registers one block ahead, so the HBM
  // latency of block i+1 hides under the transform of block i
  constexpr int NPF = (SPAN / 4 + NT - 1) / NT;
  auto fetch = [&](int blk_, f32x4 (&pf)[NPF]) {
    const int64_t g0 = (int64_t)blk_ * (FPB * HOP) + SOFF;
#pragma unroll
    for (int r3 = 0; r3 < NPF; ++r3) {
      const int i = tid + NT * r3;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (i < SPAN / 4) {
        const int64_t g = g0 + 4 * i;
        if (vec_ok && g + 3 < n_samples) {
          v = *reinterpret_cast<const f32x4*>(w + g);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = (g + r < n_samples) ? w[g + r] : 0.f;   // zero pad, :78
        }
      }
      pf[r3] = v;
    }
  };
  f32x4 pf[NPF];
  if ((int)blockIdx.x < nblk) fetch(blockIdx.x, pf);

  // per-thread LDS bases (floats)
  constexpr int IM = 2 * XP;                                              // imaginary plane behind the real plane (floats)
  const float* evA = &sm.samp[2 * (G * (HOP / 2) - SOFF / 2 + t)];        // pair of real sample 2n at evA[2 (n - t)]
  const float* odA = evA + 2 * NPAIR;                                     // real sample 2n + 1
  float* colA = sm.buf + G * GP + 2 * (t * 17);                           // pass A output column: re at [2 k2], im at [IM + 2 k2]
  float* rowA = sm.buf + G * GP + 2 * t;                                  // element n of the frame pair at rowA[2 (n - t)]
  const float* revA = sm.buf + G * GP + 2 * (256 - t);                    // X[256 - t - 16 j] at revA[-32 j]
  const float* rev0 = sm.buf + G * GP + 2 * ((256 - t) & 255);            // j = 0: X[(256 - t) & 255]
  float* magA = sm.buf + G * GP;                                          // magnitudes overwrite the real plane (257 <= XP)
  auto ld = [](const float* p) { return *reinterpret_cast<const p2*>(p); };
  auto st = [](float* p, p2 v) { *reinterpret_cast<p2*>(p) = v; };
  float* tile = sm.samp;                      // [16 frames][TP], aliases the samples

  for (int blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const int f0 = blk * FPB;
    __syncthreads();          // previous block's tile (aliases samp) fully stored; tables visible on the first pass
#pragma unroll
    for (int r3 = 0; r3 < NPF; ++r3) {       // sample r is the A half of pair r / 2 and the B half of pair (r - 1280) / 2
      const int i = tid + NT * r3;
      if (i < SPAN / 4) {
        float* ev = sm.samp + 4 * i;         // pair 2i of the even plane; the odd plane follows 2 * NPAIR floats later
        if (2 * i < NPAIR) {
          ev[0] = pf[r3][0]; ev[2] = pf[r3][2];
          ev[2 * NPAIR] = pf[r3][1]; ev[2 * NPAIR + 2] = pf[r3][3];
        }
        if (4 * i >= FB * HOP) {
          float* evb = ev - 2 * (FB * HOP / 2) + 1;
          evb[0] = pf[r3][0]; evb[2] = pf[r3][2];
          evb[2 * NPAIR] = pf[r3][1]; evb[2 * NPAIR + 2] = pf[r3][3];
        }
      }
    }
    __syncthreads();
    if (blk + (int)gridDim.x < nblk) fetch(blk + gridDim.x, pf);

    // From here to the tile write every exchange stays inside one frame pair = 16 lanes of ONE wave: LDS operations of a
    // wave execute in order, so wave-level ordering (no workgroup barrier) is enough between the passes.
    // ---- pass A: thread n1 = t transforms z[n1 + 16 n2] over n2, twiddles by W256^{n1 k2} ----------
    cx v[16];
    v[0] = {p2{0.f, 0.f}, p2{0.f, 0.f}};       // n < 16: real samples < 32, window zero
    v[15] = v[0];                              // n >= 240: real samples >= 480, window zero
#pragma unroll
    for (int n2 = 1; n2 < 15; ++n2) v[n2] = {ld(evA + 32 * n2) * he[n2 - 1], ld(odA + 32 * n2) * ho[n2 - 1]};
    dft16(v);
    st(colA, v[0].re); st(colA + IM, v[0].im);                           // W^0
#pragma unroll
    for (int k2 = 1; k2 < 16; ++k2) {
      const cx z = cmulc(v[k2], twc[k2 - 1], tws[k2 - 1]);
      st(colA + 2 * k2, z.re); st(colA + IM + 2 * k2, z.im);
    }
    __builtin_amdgcn_wave_barrier();
    // ---- pass B: thread k2 = t transforms over n1 -> X[k2 + 16 k1] -----------------------------------
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) v[n1] = {ld(rowA + 34 * n1), ld(rowA + IM + 34 * n1)};
    dft16(v);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) { st(rowA + 32 * k1, v[k1].re); st(rowA + IM + 32 * k1, v[k1].im); }
    __builtin_amdgcn_wave_barrier();

    // ---- real-input split + magnitude: 2 R[k] = (Zk + conj Z-k) - i w^k (Zk - conj Z-k) ----------
    // (the factor 1/2 is folded into the filterbank weights: exact in binary floating point)
    p2 mag[16], mag256;
    {
      const float2* tw = reinterpret_cast<const float2*>(sm.tw512) + t;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float* rp = j == 0 ? rev0 : revA - 32 * j;
        const cx zk = {ld(rowA + 32 * j), ld(rowA + IM + 32 * j)};
        const cx zr = {ld(rp), ld(rp + IM)};
        const float2 wk = tw[16 * j];                                   // (c, s) of w^k, k = t + 16 j
        // zc = conj zr; e = zk + zc; d = zk - zc; r = e - i w d:  r.re = e.re + (w d).im, r.im = e.im - (w d).re
        const p2 ere = zk.re + zr.re, eim = zk.im - zr.im, dre = zk.re - zr.re, dim = zk.im + zr.im;
        const p2 rre = ere + dre * wk.y + dim * wk.x;
        const p2 rim = eim - dre * wk.x + dim * wk.y;
        const p2 m2 = rre * rre + rim * rim;
        mag[j] = p2{__builtin_amdgcn_sqrtf(m2[0]), __builtin_amdgcn_sqrtf(m2[1])};
      }
      const p2 z0re = ld(rowA - 2 * t), z0im = ld(rowA + IM - 2 * t);
      const p2 dz = z0re - z0im;
      mag256 = p2{2.f * fabsf(dz[0]), 2.f * fabsf(dz[1])};              // 2 R[256] = 2 (Re Z0 - Im Z0) (real)
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < 16; ++j) st(magA + 2 * (t + 16 * j), mag[j]);
    if (t == 0) st(magA + 512, mag256);
    __builtin_amdgcn_wave_barrier();

    // ---- mel filterbank + log; thread owns mels t, t+16, ...: compile-time trip counts, weights in registers -------------
    p2 melv[NMEL / 16];
    {
      int slot = 0;
#pragma unroll
      for (int j = 0; j < NMEL / 16; ++j) {
        const float* mp = magA + 2 * sm.mel_start[t + 16 * j];
        p2 acc = {0.f, 0.f};
#pragma unroll
        for (int i = 0; i < MEL_GROUP_TAPS[j]; ++i) acc += ld(mp + 2 * i) * sm.melw[(slot + i) * 16 + t];
        slot += MEL_GROUP_TAPS[j];
        acc += 1e-5f;
        melv[j] = p2{__logf(acc[0]), __logf(acc[1])} * scale + bias;
      }
    }
    __syncthreads();                            // every frame is done reading the samples: samp becomes the tile
                                                // (pitch 144 = 16 (mod 64) banks between the frames of a write pass)
#pragma unroll
    for (int j = 0; j < NMEL / 16; ++j) { tile[G * TP + t + 16 * j] = melv[j][0]; tile[(G + FB) * TP + t + 16 * j] = melv[j][1]; }
    __syncthreads();

    // ---- coalesced stores: 16 consecutive mels of one frame per thread ----------------------------------------------
    if constexpr (MODE == MEL_NATURAL_F32) {
      const int fr = tid >> 3, c = tid & 7;
      if (f0 + fr < frames_out) {
        float* op = reinterpret_cast<float*>(out) + ((int64_t)b * frames_out + f0 + fr) * NMEL + c * 16;
        const float* tp = tile + fr * TP + c * 16;
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(op + 4 * q) = *reinterpret_cast<const f32x4*>(tp + 4 * q);
      }
    } else {
      // patch row p = blk*8 + f holds mel[f0 + tt][f*16 + m], tt-major (eval_caco_torch.py:124-129)
      const int f = tid >> 4, tt = tid & 15;
      const int p = blk * 8 + f;
      if (p < rows_out) {
        const float* tp = tile + tt * TP + f * 16;
        const int64_t o = ((int64_t)b * S + p) * 256 + tt * 16;
        if constexpr (MODE == MEL_PATCH_BF16) {
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            bf16x8 pk;
#pragma unroll
            for (int r = 0; r < 8; ++r) pk[r] = (bf16_t)tp[8 * q + r];
            *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16_t*>(out) + o + 8 * q) = pk;
          }
        } else {
          float* op = reinterpret_cast<float*>(out) + o;
#pragma unroll
          for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(op + 4 * q) = *reinterpret_cast<const f32x4*>(tp + 4 * q);
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
  const int want = (3072 + batch - 1) / batch;      // ~12 workgroups (of 128 threads, 3 resident) per CU chip-wide
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
    hipLaunchKernelGGL(mel_kernel<MEL_NATURAL_F32>, grid, dim3(NT), 0, st, wav, n_samples, g_tables, out, frames, 0, 0,
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
    hipLaunchKernelGGL(mel_kernel<MEL_PATCH_BF16>, grid, dim3(NT), 0, st, wav, n_samples, g_tables, out, frames, valid,
                       max_patches, scale, bias, blocks, tinds, finds, mask, lengths);
  else
    hipLaunchKernelGGL(mel_kernel<MEL_PATCH_F32>, grid, dim3(NT), 0, st, wav, n_samples, g_tables, out, frames, valid,
                       max_patches, scale, bias, blocks, tinds, finds, mask, lengths);
  return check_hip(hipGetLastError(), "mel patch launch");
}

}  // namespace caco
