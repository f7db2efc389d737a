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
    const int64_t s0 = (int64_t) t * a.hop - halfWin; // first sample of the frame
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

static size_t stft_lds_bytes(int64_t win, int64_t fft, bool twInLds, bool winInLds)
{
  const size_t nc = (size_t) fft / 2;
  return ((twInLds ? 3 : 2) * nc * 2 + (winInLds ? (size_t) win : 0)) * sizeof(double);
}

bool stft_supported(int64_t win, int64_t fft)
{
  if (fft < 4 || (fft & (fft - 1)) || win > fft || win < 1) return false;
  return stft_lds_bytes(win, fft, false, false) <= 160 * 1024;
}

void launch_stft(const StftArgs& a, hipStream_t s)
{
  StftKArgs k;
  k.audio = a.audio; k.audio64 = a.audio64; k.n = a.n; k.audioStride = a.audioStride;
  k.win = a.win; k.fft = a.fft; k.hop = a.hop; k.T = a.T; k.F = a.F; k.B = a.B;
  k.nc = a.fft / 2;
  k.window = a.window; k.twiddle = a.twiddle;
  k.mag = a.mag; k.magStride = a.magStride; k.ldMag = a.ldMag;
  k.spec = a.spec; k.specStride = a.specStride;
  k.totalFrames = (int64_t) a.B * a.T;
  // window table in LDS when it keeps >= 2 workgroups per CU
  k.twInLds = stft_lds_bytes(a.win, a.fft, true, false) <= 160 * 1024 ? 1 : 0;
  k.winInLds = (k.twInLds && stft_lds_bytes(a.win, a.fft, true, true) <= 80 * 1024) ? 1 : 0;
  const size_t shmem = stft_lds_bytes(a.win, a.fft, k.twInLds != 0, k.winInLds != 0);
  int threads = k.nc / 4;
  if (threads < 64) threads = 64;
  if (threads > 256) threads = 256;
  (void) hipFuncSetAttribute(reinterpret_cast<const void*>(stft_r2c_mag_kernel),
                             hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  int64_t grid = k.totalFrames;
  const int64_t cap = 256 * 8; // persistent-ish: each workgroup strides over frames
  if (grid > cap) grid = cap;
  if (grid < 1) return;
  hipLaunchKernelGGL(stft_r2c_mag_kernel, dim3((unsigned) grid), dim3((unsigned) threads), shmem,
                     s, k);
}

} // namespace fluhip
