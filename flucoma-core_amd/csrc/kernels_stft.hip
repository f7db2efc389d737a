// kernels_stft.hip -- gfx950 kernel K1: batched frame gather + window + real FFT + magnitude.
//
// Restates, for a whole batch of frames at once, what the reference does one frame at a time:
//   algorithm::STFT::process    include/flucoma/algorithms/public/STFT.hpp:90-108
//   algorithm::FFT::process     include/flucoma/algorithms/util/FFT.hpp:92-108   (plain DFT, bins 0..fft/2,
//                                                                   DC and Nyquist purely real)
//   algorithm::STFT::magnitude  include/flucoma/algorithms/public/STFT.hpp:61-66
// Frame t of a buffer covers samples [t*hop - win/2, t*hop - win/2 + win), zero outside [0, n);
// T = (n + hop) / hop frames (integer division) -- bit-exact index arithmetic.
//
// One workgroup transforms one frame at a time and strides over frames:
//   1. coalesced load of the frame's f32 (or f64) samples, multiply by the f64 window (LDS copy),
//      packed as n = fft/2 complex points z[m] = x[2m] + i x[2m+1] into LDS
//   2. Stockham autosort radix-4 (one radix-2 pass when log2(n) is odd) between two LDS
//      buffers; twiddles e^{-2 pi i j / fft} come from an LDS-resident table computed on the
//      host in f64
//   3. real-FFT split  X[k] = (Z[k] + conj Z[n-k])/2 - i/2 e^{-2 pi i k / fft} (Z[k] - conj Z[n-k])
//      fused with |X| and the coalesced store of the magnitude row (and optionally X itself).
// HBM traffic per frame: hop*4 B of unique samples in (overlapping reads hit L2), F*8 B out.
#include "fluhip_kernels.h"

#include <cstdlib>

namespace fluhip {

typedef double d2 __attribute__((ext_vector_type(2)));

struct StftKArgs
{
  const float* audio;
  const double* audio64;
  int64_t n, audioStride;
  int win, fft, hop, T, F, B;
  int nc;          // complex points = fft/2
  const double* window;
  const double* twiddle;
  double* mag;
  int64_t magStride, ldMag;
  double* spec;
  int64_t specStride;
  int winInLds, twInLds;
  int frameOffset;
  int64_t totalFrames;
};

__device__ __forceinline__ d2 cmul(d2 a, d2 b)
{
  return d2{a[0] * b[0] - a[1] * b[1], a[0] * b[1] + a[1] * b[0]};
}

// table holds the half circle; m in [0, fft): e^{-2 pi i m / fft}
__device__ __forceinline__ d2 twid(const d2* tw, int m, int half)
{
  if (m >= half)
  {
    d2 t = tw[m - half];
    return d2{-t[0], -t[1]};
  }
  return tw[m];
}

__global__ __launch_bounds__(256) void stft_r2c_mag_kernel(StftKArgs a)
{
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int nc = a.nc;
  d2* bufA = reinterpret_cast<d2*>(lds);
  d2* bufB = bufA + nc;
  d2* twl = bufB + nc;                        // [nc] = fft/2 twiddles when twInLds
  double* wlds = reinterpret_cast<double*>(twl + (a.twInLds ? nc : 0)); // [win] when winInLds
  const int tid = threadIdx.x, nt = blockDim.x;

  if (a.twInLds)
    for (int j = tid; j < nc; j += nt) twl[j] = reinterpret_cast<const d2*>(a.twiddle)[j];
  const d2* tw = a.twInLds ? twl : reinterpret_cast<const d2*>(a.twiddle);
  if (a.winInLds)
    for (int j = tid; j < a.win; j += nt) wlds[j] = a.window[j];
  __syncthreads();
  const double* wptr = a.winInLds ? wlds : a.window;
  const int halfWin = a.win / 2; // alg/STFT.hpp:92

  for (int64_t frame = blockIdx.x; frame < a.totalFrames; frame += gridDim.x)
  {
    const int b = (int) (frame / a.T), t = (int) (frame % a.T);
    const int64_t s0 = (int64_t) t * a.hop - halfWin + a.frameOffset; // first sample of the frame
    // ---- 1. gather + window (alg/STFT.hpp:94-97,104-105; clients/nrt/NMFClient.hpp:240) ----
    for (int m = tid; m < nc; m += nt)
    {
      double x0 = 0.0, x1 = 0.0;
      const int i0 = 2 * m, i1 = 2 * m + 1;
      const int64_t p0 = s0 + i0, p1 = s0 + i1;
      if (a.audio)
      {
        const float* src = a.audio + (int64_t) b * a.audioStride;
        if (i0 < a.win && p0 >= 0 && p0 < a.n) x0 = (double) src[p0];
        if (i1 < a.win && p1 >= 0 && p1 < a.n) x1 = (double) src[p1];
      }
      else
      {
        const double* src = a.audio64 + (int64_t) b * a.audioStride;
        if (i0 < a.win && p0 >= 0 && p0 < a.n) x0 = src[p0];
        if (i1 < a.win && p1 >= 0 && p1 < a.n) x1 = src[p1];
      }
      if (i0 < a.win) x0 *= wptr[i0];
      if (i1 < a.win) x1 *= wptr[i1];
      bufA[m] = d2{x0, x1};
    }
    __syncthreads();
    // ---- 2. Stockham autosort complex FFT of nc points -------------------------------------
    d2* src = bufA;
    d2* dst = bufB;
    for (int Ns = 1; Ns < nc;)
    {
      if (nc / Ns >= 4)
      {
        const int q = nc >> 2;
        const int tstep = a.fft / (Ns * 4); // table index step per (j mod Ns)
        for (int j = tid; j < q; j += nt)
        {
          const int k = j & (Ns - 1);
          d2 v0 = src[j], v1 = src[j + q], v2 = src[j + 2 * q], v3 = src[j + 3 * q];
          if (k)
          {
            const int m1 = k * tstep;
            v1 = cmul(v1, twid(tw, m1, nc));
            v2 = cmul(v2, twid(tw, 2 * m1, nc));
            v3 = cmul(v3, twid(tw, 3 * m1, nc));
          }
          const d2 t0 = v0 + v2, t1 = v0 - v2, t2 = v1 + v3;
          const d2 d13 = v1 - v3;
          const d2 t3 = d2{d13[1], -d13[0]}; // (v1 - v3) * (-i)
          const int o = ((j - k) << 2) + k;  // (j / Ns) * Ns * 4 + k
          dst[o] = t0 + t2;
          dst[o + Ns] = t1 + t3;
          dst[o + 2 * Ns] = t0 - t2;
          dst[o + 3 * Ns] = t1 - t3;
        }
        Ns <<= 2;
      }
      else
      {
        const int h = nc >> 1;
        const int tstep = a.fft / (Ns * 2);
        for (int j = tid; j < h; j += nt)
        {
          const int k = j & (Ns - 1);
          d2 v0 = src[j], v1 = src[j + h];
          if (k) v1 = cmul(v1, twid(tw, k * tstep, nc));
          const int o = ((j - k) << 1) + k;
          dst[o] = v0 + v1;
          dst[o + Ns] = v0 - v1;
        }
        Ns <<= 1;
      }
      __syncthreads();
      d2* tmp = src; src = dst; dst = tmp;
    }
    // ---- 3. real split + magnitude (util/FFT.hpp:99-106, alg/STFT.hpp:61-66) -----------------
    double* magRow = a.mag ? a.mag + (int64_t) b * a.magStride + (int64_t) t * a.ldMag : nullptr;
    double* specRow = a.spec ? a.spec + (int64_t) b * a.specStride + (int64_t) t * a.F * 2 : nullptr;
    for (int k = tid; k <= nc; k += nt)
    {
      double xr, xi;
      if (k == 0) { const d2 z = src[0]; xr = z[0] + z[1]; xi = 0.0; }
      else if (k == nc) { const d2 z = src[0]; xr = z[0] - z[1]; xi = 0.0; }
      else
      {
        const d2 A = src[k], Bc = src[nc - k];
        const double er = 0.5 * (A[0] + Bc[0]), ei = 0.5 * (A[1] - Bc[1]);
        const double dr = 0.5 * (A[0] - Bc[0]), di = 0.5 * (A[1] + Bc[1]);
        const d2 w = tw[k];
        xr = er + (w[0] * di + w[1] * dr);
        xi = ei - (w[0] * dr - w[1] * di);
      }
      if (magRow) magRow[k] = sqrt(xr * xr + xi * xi);
      if (specRow) reinterpret_cast<d2*>(specRow)[k] = d2{xr, xi};
    }
    __syncthreads(); // bufA/bufB are rewritten by the next frame
  }
}

// ---------------------------------------------------------------------------------------
// Wave-per-frame STFT for the common sizes (fft 1024 / 2048 / 4096).
//
// One wavefront transforms one frame: the n = fft/2 complex points live in registers (n/64 per
// lane) and the FFT is three Stockham passes of radix R1 x R2 x R3 = n whose butterflies are done
// entirely in registers (radix 8 / 16 from radix-4 kernels with compile-time twiddles); between
// passes the points are exchanged through a private, padded LDS buffer.  A wavefront never
// synchronises with another one (its LDS instructions execute in program order), so there is no
// s_barrier in the frame loop at all; 8 wavefronts per workgroup share only the read-only twiddle
// table.  Same arithmetic order of magnitude and same DFT convention as the generic kernel above.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ d2 mul_mi(d2 a) { return d2{a[1], -a[0]}; } // a * (-i)

// |X| = sqrt(x) for x = re^2 + im^2 >= 0: v_rsq_f64 seed (~2^-23) + two Goldschmidt steps (~2^-92,
// i.e. rounding error only) in 9 FP64 ops instead of the ~25 of the correctly-rounded library sqrt.
// x is clamped to the smallest normal so that rsq stays finite; sqrt(2.2e-308) = 1.5e-154 stands in for 0.
__device__ __forceinline__ double mag_sqrt(double x)
{
  x = fmax(x, 2.2250738585072014e-308);
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  const double d = __builtin_fma(-g, g, x);
  return __builtin_fma(d, h, g);
}

__device__ __forceinline__ void dft4(d2& a0, d2& a1, d2& a2, d2& a3)
{
  const d2 t0 = a0 + a2, t1 = a0 - a2, t2 = a1 + a3, t3 = mul_mi(a1 - a3);
  a0 = t0 + t2; a1 = t1 + t3; a2 = t0 - t2; a3 = t1 - t3;
}

constexpr double kC1 = 0.92387953251128673848; // cos(pi/8)
constexpr double kS1 = 0.38268343236508977173; // sin(pi/8)
constexpr double kC2 = 0.70710678118654752440; // sqrt(2)/2

template <int R>
__device__ __forceinline__ void dft_inplace(d2 (&v)[R]);

template <>
__device__ __forceinline__ void dft_inplace<8>(d2 (&v)[8])
{
  // r = a + 2c: S_a = DFT4 over c; y[b + 4d] = S_0[b] + (-1)^d W8^b S_1[b]
  d2 e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6];
  d2 o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
  dft4(e0, e1, e2, e3);
  dft4(o0, o1, o2, o3);
  o1 = d2{kC2 * (o1[0] + o1[1]), kC2 * (o1[1] - o1[0])};   // * (c2, -c2)
  o2 = mul_mi(o2);
  o3 = d2{kC2 * (o3[1] - o3[0]), -kC2 * (o3[0] + o3[1])};  // * (-c2, -c2)
  v[0] = e0 + o0; v[4] = e0 - o0;
  v[1] = e1 + o1; v[5] = e1 - o1;
  v[2] = e2 + o2; v[6] = e2 - o2;
  v[3] = e3 + o3; v[7] = e3 - o3;
}

template <>
__device__ __forceinline__ void dft_inplace<16>(d2 (&v)[16])
{
  // r = a + 4c: S_a[b] = DFT4 over c; T_a[b] = W16^{ab} S_a[b]; y[b + 4d] = DFT4 over a of T_a[b]
  d2 s[4][4];
#pragma unroll
  for (int a = 0; a < 4; a++)
  {
    s[a][0] = v[a]; s[a][1] = v[a + 4]; s[a][2] = v[a + 8]; s[a][3] = v[a + 12];
    dft4(s[a][0], s[a][1], s[a][2], s[a][3]);
  }
  // W16^1 = (c1,-s1)  W16^2 = (c2,-c2)  W16^3 = (s1,-c1)  W16^4 = -i  W16^6 = (-c2,-c2)  W16^9 = (-c1, s1)
  s[1][1] = cmul(s[1][1], d2{kC1, -kS1});
  s[1][2] = d2{kC2 * (s[1][2][0] + s[1][2][1]), kC2 * (s[1][2][1] - s[1][2][0])};
  s[1][3] = cmul(s[1][3], d2{kS1, -kC1});
  s[2][1] = d2{kC2 * (s[2][1][0] + s[2][1][1]), kC2 * (s[2][1][1] - s[2][1][0])};
  s[2][2] = mul_mi(s[2][2]);
  s[2][3] = d2{kC2 * (s[2][3][1] - s[2][3][0]), -kC2 * (s[2][3][0] + s[2][3][1])};
  s[3][1] = cmul(s[3][1], d2{kS1, -kC1});
  s[3][2] = d2{kC2 * (s[3][2][1] - s[3][2][0]), -kC2 * (s[3][2][0] + s[3][2][1])};
  s[3][3] = cmul(s[3][3], d2{-kC1, kS1});
#pragma unroll
  for (int b = 0; b < 4; b++)
  {
    dft4(s[0][b], s[1][b], s[2][b], s[3][b]);
    v[b] = s[0][b]; v[b + 4] = s[1][b]; v[b + 8] = s[2][b]; v[b + 12] = s[3][b];
  }
}

__device__ __forceinline__ int lds_pad(int i) { return i + (i >> 4); } // one 16-byte slot per 16

// one Stockham pass: inputs in `pts` (NB butterflies of radix R per lane, lane-major), twiddle
// by e^{-2 pi i k r / (Ns R)}, DFT_R, scatter to LDS at the autosort position.  `ptw` is the
// pass's own twiddle table laid out [r-1][k] (k < Ns): lanes read consecutive 16-byte slots, so
// the reads are bank-conflict free (a single e^{-2 pi i m / fft} table is read at stride 16 r
// slots in the middle pass: a 16-way conflict).
// KEEP: the last pass (NS_ == points / R, so k == j and the outputs land at j + r NS_): the lane's own outputs are
// exactly the bins k = lane + 64 i it handles in the real-FFT split, i = b + r NB -- they are handed back in `pts`
// in that order and the split does not read them from LDS again.
template <int R, int NB, int NS_, bool KEEP = false>
__device__ __forceinline__ void stockham_pass(d2 (&pts)[NB * R], d2* buf, const d2* ptw, int lane)
{
  d2 keep[KEEP ? NB * R : 1];
#pragma unroll
  for (int b = 0; b < NB; b++)
  {
    const int j = lane + 64 * b;
    const int k = j & (NS_ - 1);
    d2 v[R];
#pragma unroll
    for (int r = 0; r < R; r++) v[r] = pts[b * R + r];
    if constexpr (NS_ > 1)
    {
#pragma unroll
      for (int r = 1; r < R; r++) v[r] = cmul(v[r], ptw[(r - 1) * NS_ + k]);
    }
    dft_inplace<R>(v);
    const int o = (j - k) * R + k;
#pragma unroll
    for (int r = 0; r < R; r++) buf[lds_pad(o + r * NS_)] = v[r];
    if constexpr (KEEP)
    {
#pragma unroll
      for (int r = 0; r < R; r++) keep[b + r * NB] = v[r];
    }
  }
  if constexpr (KEEP)
  {
#pragma unroll
    for (int i = 0; i < NB * R; i++) pts[i] = keep[i];
  }
}

#ifdef FLUHIP_AB_SWITCHES // round 1's wave-per-frame kernel: production reaches it for no shape any more (the block form of
                         // kernels_stft2.hip covers fft 1024 / 2048 / 4096 with an even window); kept for the A/B build (FLUHIP_STFT_BLOCK=0)
template <int R1, int R2, int R3, int MAXW>
__global__ __launch_bounds__(64 * MAXW) void stft_wave_kernel(StftKArgs a)
{
  constexpr int N = R1 * R2 * R3;     // complex points per frame = fft/2
  constexpr int PPL = N / 64;         // points per lane
  constexpr int NB1 = N / (64 * R1), NB2 = N / (64 * R2), NB3 = N / (64 * R3);
  constexpr int BUF = N + N / 16;     // padded slots per wavefront
  constexpr int T2 = (R2 - 1) * R1, T3 = (R3 - 1) * R1 * R2; // per-pass twiddle tables
  extern __shared__ __attribute__((aligned(16))) double lds[];
  d2* tw2 = reinterpret_cast<d2*>(lds);                 // [R2-1][R1]
  d2* tw3 = tw2 + T2;                                   // [R3-1][R1*R2]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wpb = blockDim.x >> 6;
  d2* wlds = tw3 + T3;                                  // [N] window pairs (w[2m], w[2m+1]), zero past win
  d2* buf = wlds + N + wave * BUF;

  // pass tables from the natural table e^{-2 pi i m / fft} (m < fft/2, second half by symmetry)
  const d2* twg = reinterpret_cast<const d2*>(a.twiddle);
  for (int i = threadIdx.x; i < T2 + T3; i += blockDim.x)
  {
    int m;
    if (i < T2) { const int r = i / R1 + 1, k = i % R1; m = k * r * (2 * N / (R1 * R2)); }
    else { const int ii = i - T2; const int r = ii / (R1 * R2) + 1, k = ii % (R1 * R2); m = k * r * (2 * N / (R1 * R2 * R3)); }
    d2 w = twg[m >= N ? m - N : m];
    if (m >= N) w = d2{-w[0], -w[1]};
    tw2[i] = w;
  }
  // LDS copy of the window (measured: 623 us vs 912 us with the table read from L1/L2 per frame)
  for (int m = threadIdx.x; m < N; m += blockDim.x) wlds[m] = reinterpret_cast<const d2*>(a.window)[m];
  __syncthreads();
  const int halfWin = a.win / 2;

  // raw samples of one frame, two per point; out-of-range samples read as zero
  auto gather = [&](int64_t frame, float2 (&raw)[PPL]) {
    const int b = (int) (frame / a.T), t = (int) (frame % a.T);
    const int64_t s0 = (int64_t) t * a.hop - halfWin + a.frameOffset;
    const float* src = a.audio + (int64_t) b * a.audioStride;
    const int64_t last = a.n - 1;
#pragma unroll
    for (int bb = 0; bb < NB1; bb++)
#pragma unroll
      for (int r = 0; r < R1; r++)
      {
        const int m = lane + 64 * bb + r * (N / R1);
        const int64_t p0 = s0 + 2 * m, p1 = p0 + 1;
        const float v0 = src[p0 < 0 ? 0 : (p0 > last ? last : p0)];
        const float v1 = src[p1 < 0 ? 0 : (p1 > last ? last : p1)];
        raw[bb * R1 + r] = make_float2((p0 >= 0 && p0 <= last) ? v0 : 0.f, (p1 >= 0 && p1 <= last) ? v1 : 0.f);
      }
  };
  auto gather64 = [&](int64_t frame, d2 (&pts)[PPL]) {
    const int b = (int) (frame / a.T), t = (int) (frame % a.T);
    const int64_t s0 = (int64_t) t * a.hop - halfWin + a.frameOffset;
    const double* src = a.audio64 + (int64_t) b * a.audioStride;
    const int64_t last = a.n - 1;
#pragma unroll
    for (int bb = 0; bb < NB1; bb++)
#pragma unroll
      for (int r = 0; r < R1; r++)
      {
        const int m = lane + 64 * bb + r * (N / R1);
        const int64_t p0 = s0 + 2 * m, p1 = p0 + 1;
        const double v0 = src[p0 < 0 ? 0 : (p0 > last ? last : p0)];
        const double v1 = src[p1 < 0 ? 0 : (p1 > last ? last : p1)];
        const d2 w = wlds[m];
        pts[bb * R1 + r] = d2{((p0 >= 0 && p0 <= last) ? v0 : 0.0) * w[0], ((p1 >= 0 && p1 <= last) ? v1 : 0.0) * w[1]};
      }
  };

  const int64_t stride = (int64_t) gridDim.x * wpb;
  int64_t frame = (int64_t) blockIdx.x * wpb + wave;
  constexpr bool kPrefetch = PPL <= 8; // a second set of raw samples only where registers allow
  float2 raw[PPL];
  if (kPrefetch && a.audio && frame < a.totalFrames) gather(frame, raw);

  for (; frame < a.totalFrames; frame += stride)
  {
    const int b = (int) (frame / a.T), t = (int) (frame % a.T);
    d2 pts[PPL];
    if (a.audio && !kPrefetch)
    {
      // straight into the pass-1 registers.  All loads are unconditional (clamped address, value
      // selected afterwards) so the whole frame's samples are in flight together.
      const int64_t s0 = (int64_t) t * a.hop - halfWin + a.frameOffset;
      const float* src = a.audio + (int64_t) b * a.audioStride;
      const bool inside = s0 >= 0 && s0 + 2 * N <= a.n;                      // wave-uniform
      const bool aligned = inside && ((reinterpret_cast<uintptr_t>(src + s0) & 7) == 0);
      if (aligned)
      {
#pragma unroll
        for (int bb = 0; bb < NB1; bb++)
#pragma unroll
          for (int r = 0; r < R1; r++)
          {
            const int m = lane + 64 * bb + r * (N / R1);
            const float2 x = *reinterpret_cast<const float2*>(src + s0 + 2 * m);
            const d2 w = wlds[m];
            pts[bb * R1 + r] = d2{(double) x.x * w[0], (double) x.y * w[1]};
          }
      }
      else
      {
        const int64_t last = a.n - 1;
#pragma unroll
        for (int bb = 0; bb < NB1; bb++)
#pragma unroll
          for (int r = 0; r < R1; r++)
          {
            const int m = lane + 64 * bb + r * (N / R1);
            const int64_t p0 = s0 + 2 * m, p1 = p0 + 1;
            const float v0 = src[p0 < 0 ? 0 : (p0 > last ? last : p0)];
            const float v1 = src[p1 < 0 ? 0 : (p1 > last ? last : p1)];
            const double x0 = (p0 >= 0 && p0 <= last) ? (double) v0 : 0.0;
            const double x1 = (p1 >= 0 && p1 <= last) ? (double) v1 : 0.0;
            const d2 w = wlds[m]; // zero past the window length
            pts[bb * R1 + r] = d2{x0 * w[0], x1 * w[1]};
          }
      }
    }
    else if (a.audio)
    {
#pragma unroll
      for (int bb = 0; bb < NB1; bb++)
#pragma unroll
        for (int r = 0; r < R1; r++)
        {
          const d2 w = wlds[lane + 64 * bb + r * (N / R1)];
          const int i = bb * R1 + r;
          pts[i] = d2{(double) raw[i].x * w[0], (double) raw[i].y * w[1]};
        }
      // software pipeline: the next frame's samples are in flight during this frame's FFT
      if (kPrefetch && frame + stride < a.totalFrames) gather(frame + stride, raw);
    }
    else
      gather64(frame, pts);
    // ---- pass 1 (Ns = 1) ------------------------------------------------------------------
    stockham_pass<R1, NB1, 1>(pts, buf, tw2, lane);
    // ---- pass 2 (Ns = R1): re-read in the new distribution ---------------------------------
    d2 p2[PPL];
#pragma unroll
    for (int bb = 0; bb < NB2; bb++)
#pragma unroll
      for (int r = 0; r < R2; r++) p2[bb * R2 + r] = buf[lds_pad(lane + 64 * bb + r * (N / R2))];
    stockham_pass<R2, NB2, R1>(p2, buf, tw2, lane);
    // ---- pass 3 (Ns = R1 R2) ----------------------------------------------------------------
    d2 p3[PPL];
#pragma unroll
    for (int bb = 0; bb < NB3; bb++)
#pragma unroll
      for (int r = 0; r < R3; r++) p3[bb * R3 + r] = buf[lds_pad(lane + 64 * bb + r * (N / R3))];
    stockham_pass<R3, NB3, R1 * R2, true>(p3, buf, tw3, lane);
    // ---- real split + magnitude: X[k], k = lane + 64 i (and k = N on lane 0) ------------------
    double* magRow = a.mag ? a.mag + (int64_t) b * a.magStride + (int64_t) t * a.ldMag : nullptr;
    double* specRow = a.spec ? a.spec + (int64_t) b * a.specStride + (int64_t) t * a.F * 2 : nullptr;
    // k = 0 falls out of the general formula with Z[N] := Z[0]; k = N is the one extra bin (lane 0)
#pragma unroll
    for (int i = 0; i < PPL; i++)
    {
      const int k = lane + 64 * i;
      const d2 A = p3[i], Bc = buf[lds_pad((N - k) & (N - 1))]; // A: this lane's own output of the last pass
      const double er = 0.5 * (A[0] + Bc[0]), ei = 0.5 * (A[1] - Bc[1]);
      const double dr = 0.5 * (A[0] - Bc[0]), di = 0.5 * (A[1] + Bc[1]);
      const d2 w = twg[k]; // natural-order table from global memory (L1/L2 resident, coalesced)
      const double xr = er + (w[0] * di + w[1] * dr);
      const double xi = (k == 0) ? 0.0 : ei - (w[0] * dr - w[1] * di);
      if (magRow) magRow[k] = mag_sqrt(xr * xr + xi * xi);
      if (specRow) reinterpret_cast<d2*>(specRow)[k] = d2{xr, xi};
    }
    if (lane == 0)
    {
      const d2 z = buf[0];
      const double xr = z[0] - z[1];
      if (magRow) magRow[N] = fabs(xr);
      if (specRow) reinterpret_cast<d2*>(specRow)[N] = d2{xr, 0.0};
    }
  }
}

template <int R1, int R2, int R3, int MAXW>
static void launch_stft_wave(const StftKArgs& k, hipStream_t s)
{
  constexpr int N = R1 * R2 * R3;
  constexpr int BUF = N + N / 16;
  // as many wavefronts per workgroup as the LDS allows (twiddle table + one padded buffer each)
  constexpr int TW = (R2 - 1) * R1 + (R3 - 1) * R1 * R2;
  int waves = (int) ((160 * 1024 - (size_t) (TW + N) * 16) / ((size_t) BUF * 16));
  if (waves > MAXW) waves = MAXW;
  const size_t shmem = ((size_t) (TW + N) + (size_t) waves * BUF) * 16;
  auto kern = stft_wave_kernel<R1, R2, R3, MAXW>;
  request_dynamic_lds(kern, (size_t) (shmem));
  int64_t wgs = (k.totalFrames + waves - 1) / waves;
  if (wgs > 256 * 4) wgs = 256 * 4;
  if (wgs < 1) return;
  hipLaunchKernelGGL(kern, dim3((unsigned) wgs), dim3((unsigned) (64 * waves)), shmem, s, k);
}
#endif // FLUHIP_AB_SWITCHES

// ---------------------------------------------------------------------------------------
// fft sizes whose frame does not fit the LDS (above 8192): the same Stockham passes through global memory, a
// thread per butterfly over a chunk of frames, ping-ponging between two scratch buffers [frames][fft/2] complex.
// Correctness-first (the reference's static FFT setup goes to 65536, util/FFT.hpp:113-122; such windows are rare).
// ---------------------------------------------------------------------------------------
__global__ void big_pack_kernel(StftKArgs a, int64_t f0, int nf, d2* buf)
{
  const int nc = a.nc;
  const int64_t idx = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t) nf * nc) return;
  const int m = (int) (idx % nc);
  const int64_t frame = f0 + idx / nc;
  const int b = (int) (frame / a.T), t = (int) (frame % a.T);
  const int64_t s0 = (int64_t) t * a.hop - a.win / 2 + a.frameOffset;
  const int i0 = 2 * m, i1 = 2 * m + 1;
  const int64_t p0 = s0 + i0, p1 = s0 + i1;
  double x0 = 0.0, x1 = 0.0;
  if (a.audio)
  {
    const float* src = a.audio + (int64_t) b * a.audioStride;
    if (i0 < a.win && p0 >= 0 && p0 < a.n) x0 = (double) src[p0];
    if (i1 < a.win && p1 >= 0 && p1 < a.n) x1 = (double) src[p1];
  }
  else
  {
    const double* src = a.audio64 + (int64_t) b * a.audioStride;
    if (i0 < a.win && p0 >= 0 && p0 < a.n) x0 = src[p0];
    if (i1 < a.win && p1 >= 0 && p1 < a.n) x1 = src[p1];
  }
  if (i0 < a.win) x0 *= a.window[i0];
  if (i1 < a.win) x1 *= a.window[i1];
  buf[idx] = d2{x0, x1};
}

// one Stockham pass (radix 4, or 2 for the last pass of an odd log2) over nf frames of nc points
__global__ void big_pass_kernel(const d2* src, d2* dst, int nc, int Ns, int radix, int fft, const d2* tw, int nf)
{
  const int per = nc / radix;
  const int64_t idx = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t) nf * per) return;
  const int j = (int) (idx % per);
  const d2* in = src + (idx / per) * nc;
  d2* out = dst + (idx / per) * nc;
  const int k = j & (Ns - 1);
  if (radix == 4)
  {
    const int tstep = fft / (Ns * 4);
    d2 v0 = in[j], v1 = in[j + per], v2 = in[j + 2 * per], v3 = in[j + 3 * per];
    if (k)
    {
      const int m1 = k * tstep;
      v1 = cmul(v1, twid(tw, m1, nc));
      v2 = cmul(v2, twid(tw, 2 * m1, nc));
      v3 = cmul(v3, twid(tw, 3 * m1, nc));
    }
    const d2 t0 = v0 + v2, t1 = v0 - v2, t2 = v1 + v3;
    const d2 d13 = v1 - v3;
    const d2 t3 = d2{d13[1], -d13[0]};
    const int o = ((j - k) << 2) + k;
    out[o] = t0 + t2;
    out[o + Ns] = t1 + t3;
    out[o + 2 * Ns] = t0 - t2;
    out[o + 3 * Ns] = t1 - t3;
  }
  else
  {
    const int tstep = fft / (Ns * 2);
    d2 v0 = in[j], v1 = in[j + per];
    if (k) v1 = cmul(v1, twid(tw, k * tstep, nc));
    const int o = ((j - k) << 1) + k;
    out[o] = v0 + v1;
    out[o + Ns] = v0 - v1;
  }
}

__global__ void big_split_kernel(const d2* z, StftKArgs a, int64_t f0, int nf)
{
  const int nc = a.nc;
  const int64_t idx = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t) nf * (nc + 1)) return;
  const int k = (int) (idx % (nc + 1));
  const int64_t fr = idx / (nc + 1), frame = f0 + fr;
  const int b = (int) (frame / a.T), t = (int) (frame % a.T);
  const d2* src = z + fr * nc;
  const d2* tw = reinterpret_cast<const d2*>(a.twiddle);
  double xr, xi;
  if (k == 0) { const d2 v = src[0]; xr = v[0] + v[1]; xi = 0.0; }
  else if (k == nc) { const d2 v = src[0]; xr = v[0] - v[1]; xi = 0.0; }
  else
  {
    const d2 A = src[k], Bc = src[nc - k];
    const double er = 0.5 * (A[0] + Bc[0]), ei = 0.5 * (A[1] - Bc[1]);
    const double dr = 0.5 * (A[0] - Bc[0]), di = 0.5 * (A[1] + Bc[1]);
    const d2 w = tw[k];
    xr = er + (w[0] * di + w[1] * dr);
    xi = ei - (w[0] * dr - w[1] * di);
  }
  if (a.mag) a.mag[(int64_t) b * a.magStride + (int64_t) t * a.ldMag + k] = sqrt(xr * xr + xi * xi);
  if (a.spec) reinterpret_cast<d2*>(a.spec + (int64_t) b * a.specStride + (int64_t) t * a.F * 2)[k] = d2{xr, xi};
}

// the passes on their own: nf frames of nc complex points in bufA -> result in the returned buffer (A or B)
double* launch_big_fft_passes(double* bufA, double* bufB, int nc, int fft, const double* twiddle, int nf, hipStream_t s)
{
  d2* src = reinterpret_cast<d2*>(bufA);
  d2* dst = reinterpret_cast<d2*>(bufB);
  for (int Ns = 1; Ns < nc;)
  {
    const int radix = (nc / Ns >= 4) ? 4 : 2;
    const int64_t total = (int64_t) nf * (nc / radix);
    hipLaunchKernelGGL(big_pass_kernel, dim3((unsigned) ((total + 255) / 256)), dim3(256), 0, s, src, dst, nc, Ns, radix, fft,
                       reinterpret_cast<const d2*>(twiddle), nf);
    Ns *= radix;
    d2* t = src; src = dst; dst = t;
  }
  return reinterpret_cast<double*>(src);
}

int64_t big_fft_scratch_bytes(int64_t fft, int64_t frames, int64_t* chunkFrames)
{
  const int64_t perFrame = fft / 2 * 16; // one buffer
  int64_t cf = std::max<int64_t>(1, ((int64_t) 512 << 20) / perFrame);
  cf = std::min(cf, std::max<int64_t>(frames, 1));
  if (chunkFrames) *chunkFrames = cf;
  return 2 * cf * perFrame;
}

static void launch_stft_big(const StftKArgs& k, double* scratch, hipStream_t s)
{
  int64_t cf = 1;
  (void) big_fft_scratch_bytes(k.fft, k.totalFrames, &cf);
  double* bufA = scratch;
  double* bufB = scratch + cf * k.nc * 2;
  for (int64_t f0 = 0; f0 < k.totalFrames; f0 += cf)
  {
    const int nf = (int) std::min<int64_t>(cf, k.totalFrames - f0);
    const int64_t np = (int64_t) nf * k.nc;
    hipLaunchKernelGGL(big_pack_kernel, dim3((unsigned) ((np + 255) / 256)), dim3(256), 0, s, k, f0, nf,
                       reinterpret_cast<d2*>(bufA));
    const double* z = launch_big_fft_passes(bufA, bufB, k.nc, k.fft, k.twiddle, nf, s);
    const int64_t ns = (int64_t) nf * (k.nc + 1);
    hipLaunchKernelGGL(big_split_kernel, dim3((unsigned) ((ns + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const d2*>(z),
                       k, f0, nf);
  }
}


static size_t stft_lds_bytes(int64_t win, int64_t fft, bool twInLds, bool winInLds)
{
  const size_t nc = (size_t) fft / 2;
  return ((twInLds ? 3 : 2) * nc * 2 + (winInLds ? (size_t) win : 0)) * sizeof(double);
}

bool stft_supported(int64_t win, int64_t fft)
{
  if (fft < 4 || (fft & (fft - 1)) || win > fft || win < 1) return false;
  return fft <= 65536; // util/FFT.hpp:113-122: the reference's shared setup ends there too
}
bool stft_needs_scratch(int64_t win, int64_t fft) { return stft_lds_bytes(win, fft, false, false) > 160 * 1024; }

void launch_stft(const StftArgs& a, hipStream_t s)
{
  StftKArgs k;
  k.audio = a.audio; k.audio64 = a.audio64; k.n = a.n; k.audioStride = a.audioStride;
  k.win = a.win; k.fft = a.fft; k.hop = a.hop; k.T = a.T; k.F = a.F; k.B = a.B;
  k.nc = a.fft / 2;
  k.window = a.window; k.twiddle = a.twiddle;
  k.mag = a.mag; k.magStride = a.magStride; k.ldMag = a.ldMag;
  k.spec = a.spec; k.specStride = a.specStride;
  k.frameOffset = a.frameOffset;
  k.totalFrames = (int64_t) a.B * a.T;
  k.twInLds = 1; k.winInLds = 0;
  if (stft_needs_scratch(a.win, a.fft))
  {
    if (a.bigScratch && k.totalFrames > 0) launch_stft_big(k, a.bigScratch, s);
    return;
  }
  {
    static const bool generic = fluhip::ab_getenv("FLUHIP_STFT_GENERIC") != nullptr;
    // wave-per-frame kernels: power-of-two fft with an even window (pairs of window values)
    if (!generic && (a.win % 2) == 0)
    {
      // the block form (kernels_stft2.hip) without its bin-major output is the faster wave-per-frame kernel
      if (launch_stft_block(a, nullptr, 0, 0, s)) return;
#ifdef FLUHIP_AB_SWITCHES // (only with FLUHIP_STFT_BLOCK=0: the block form takes these sizes)
      if (a.fft == 2048) { launch_stft_wave<16, 8, 8, 8>(k, s); return; }
      if (a.fft == 1024) { launch_stft_wave<8, 8, 8, 8>(k, s); return; }
#endif
    }
  }
  // window table in LDS when it keeps >= 2 workgroups per CU
  k.twInLds = stft_lds_bytes(a.win, a.fft, true, false) <= 160 * 1024 ? 1 : 0;
  k.winInLds = (k.twInLds && stft_lds_bytes(a.win, a.fft, true, true) <= 80 * 1024) ? 1 : 0;
  const size_t shmem = stft_lds_bytes(a.win, a.fft, k.twInLds != 0, k.winInLds != 0);
  int threads = k.nc / 4;
  if (threads < 64) threads = 64;
  if (threads > 256) threads = 256;
  request_dynamic_lds(stft_r2c_mag_kernel, (size_t) (160 * 1024));
  int64_t grid = k.totalFrames;
  const int64_t cap = 256 * 8; // persistent-ish: each workgroup strides over frames
  if (grid > cap) grid = cap;
  if (grid < 1) return;
  hipLaunchKernelGGL(stft_r2c_mag_kernel, dim3((unsigned) grid), dim3((unsigned) threads), shmem,
                     s, k);
}

} // namespace fluhip
