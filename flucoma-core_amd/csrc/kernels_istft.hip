// kernels_istft.hip -- resynthesis of one NMF component:
//   algorithm::NMF::estimate      include/flucoma/algorithms/public/NMF.hpp:33-42      est[t][f] = H1[t][k] W1[k][f]
//   algorithm::RatioMask::process include/flucoma/algorithms/public/RatioMask.hpp:33-57  Y = X * min(est * (1/max(Vhat,eps)), 1)
//   algorithm::ISTFT::process     include/flucoma/algorithms/public/STFT.hpp:178-199     inverse real FFT, * 1/fft, * window,
//                                                                           overlap-add, / max(sum w^2, eps), trim win/2
//   driver                        include/flucoma/clients/nrt/NMFClient.hpp:302-334
//
// Kernel R1 (one workgroup per frame): masked spectrum -> C2R through an n = fft/2 point complex
// FFT in LDS (Z[k] = E[k] + i O[k] with E, O the spectra of the even / odd samples; inverse taken
// as conj(FFT(conj Z)) / n with the forward Stockham passes of the STFT kernel) -> first `win`
// samples * window -> frames[t][win].
// Kernel R2 (one thread per output sample): gathers the <= ceil(win/hop) overlapping frames in
// increasing frame order -- the same summation order as the reference's sequential overlap-add,
// so the result does not depend on scheduling -- and divides by the window-power normaliser.
#include "fluhip_kernels.h"

#include <algorithm>

namespace fluhip {

typedef double d2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ d2 cmul_i(d2 a, d2 b)
{
  return d2{a[0] * b[0] - a[1] * b[1], a[0] * b[1] + a[1] * b[0]};
}

__device__ __forceinline__ d2 twid_i(const d2* tw, int m, int half)
{
  if (m >= half)
  {
    d2 t = tw[m - half];
    return d2{-t[0], -t[1]};
  }
  return tw[m];
}

__global__ __launch_bounds__(256) void resynth_frames_kernel(ResynthArgs a)
{
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int nc = a.fft / 2;
  d2* bufA = reinterpret_cast<d2*>(lds);
  d2* bufB = bufA + nc;
  const d2* tw = reinterpret_cast<const d2*>(a.twiddle); // e^{-2 pi i m / fft}, m < fft/2 (global, L1/L2 resident)
  const int tid = threadIdx.x, nt = blockDim.x;
  const int t = blockIdx.x;
  const int comp = a.k + blockIdx.y; // a launch covers nComp consecutive components (frames / out strided by them)
  const double* spec = a.spec + (int64_t) t * a.F * 2;
  const double* vhat = a.Vhat + (int64_t) t * a.ldV;
  const bool useMask = a.Wf != nullptr;
  const double hk = useMask ? a.H1[(int64_t) t * a.Kp + comp] : 0.0;

  auto masked = [&](int f) -> d2 {
    d2 x = reinterpret_cast<const d2*>(spec)[f];
    if (useMask)
    {
      const double est = hk * a.Wf[(int64_t) f * a.Kp + comp];  // NMF.hpp:41
      const double mult = 1.0 / fmax(vhat[f], kEpsilon);          // RatioMask.hpp:39-41
      const double m = fmin(est * mult, 1.0);                     // RatioMask.hpp:52-56 (exponent 1)
      x = d2{x[0] * m, x[1] * m};
    }
    if (f == 0 || f == nc) x[1] = 0.0;                          // util/FFT.hpp:155-160 packed DC / Nyquist
    return x;
  };

  // ---- Z[k] = E[k] + i O[k], stored conjugated for the forward transform ----------------------
  for (int k = tid; k < nc; k += nt)
  {
    const d2 X = masked(k), Xn = masked(nc - k);
    const d2 Xc = d2{Xn[0], -Xn[1]};
    const d2 E = d2{0.5 * (X[0] + Xc[0]), 0.5 * (X[1] + Xc[1])};
    d2 D = d2{0.5 * (X[0] - Xc[0]), 0.5 * (X[1] - Xc[1])};
    const d2 w = tw[k];                       // e^{-2 pi i k / fft}; we need its conjugate e^{+...}
    const d2 O = cmul_i(D, d2{w[0], -w[1]});
    const d2 Z = d2{E[0] - O[1], E[1] + O[0]}; // E + i O
    bufA[k] = d2{Z[0], -Z[1]};                // conj
  }
  __syncthreads();
  d2* src = bufA;
  d2* dst = bufB;
  for (int Ns = 1; Ns < nc;)
  {
    if (nc / Ns >= 4)
    {
      const int q = nc >> 2;
      const int tstep = a.fft / (Ns * 4);
      for (int j = tid; j < q; j += nt)
      {
        const int k = j & (Ns - 1);
        d2 v0 = src[j], v1 = src[j + q], v2 = src[j + 2 * q], v3 = src[j + 3 * q];
        if (k)
        {
          const int m1 = k * tstep;
          v1 = cmul_i(v1, twid_i(tw, m1, nc));
          v2 = cmul_i(v2, twid_i(tw, 2 * m1, nc));
          v3 = cmul_i(v3, twid_i(tw, 3 * m1, nc));
        }
        const d2 t0 = v0 + v2, t1 = v0 - v2, t2 = v1 + v3;
        const d2 d13 = v1 - v3;
        const d2 t3 = d2{d13[1], -d13[0]};
        const int o = ((j - k) << 2) + k;
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
        if (k) v1 = cmul_i(v1, twid_i(tw, k * tstep, nc));
        const int o = ((j - k) << 1) + k;
        dst[o] = v0 + v1;
        dst[o + Ns] = v0 - v1;
      }
      Ns <<= 1;
    }
    __syncthreads();
    d2* tmp = src; src = dst; dst = tmp;
  }
  // z[m] = conj(src[m]) / nc ; x[2m] = Re z, x[2m+1] = Im z ; keep the first `win` samples
  // (alg/STFT.hpp:191), scale by 1/fft is already contained in the normalised inverse
  // (unnormalised C2R = fft * x, times mScale = 1/fft), then * window (:193)
  double* fr = a.frames + ((int64_t) blockIdx.y * a.T + t) * a.win;
  const double inv = 1.0 / (double) nc;
  for (int m = tid; m < nc; m += nt)
  {
    const d2 z = src[m];
    const int i0 = 2 * m, i1 = 2 * m + 1;
    if (i0 < a.win) fr[i0] = (z[0] * inv) * a.window[i0];
    if (i1 < a.win) fr[i1] = (-z[1] * inv) * a.window[i1];
  }
}

__global__ void resynth_ola_kernel(ResynthArgs a)
{
  const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  const double* frames = a.frames + (int64_t) blockIdx.y * a.T * a.win;
  const int64_t p = i + a.trim; // position in the padded output (alg/STFT.hpp:197; BufSTFTClient.hpp:272)
  // frames t with t*hop <= p < t*hop + win
  int64_t tlo = (p - a.win + a.hop) / a.hop; // ceil((p - win + 1) / hop) for p - win + 1 > 0
  if (p - a.win + 1 <= 0) tlo = 0;
  int64_t thi = p / a.hop;
  if (thi > a.T - 1) thi = a.T - 1;
  double acc = 0.0, nrm = 0.0;
  for (int64_t t = tlo; t <= thi; t++)
  {
    const int64_t off = p - t * a.hop;
    const double w = a.window[off];
    acc += frames[t * a.win + off];
    nrm += w * w;
  }
  const double y = acc / fmax(nrm, kEpsilon); // :196
  const int64_t os = a.outStride > 0 ? a.outStride : a.n; // components outStride apart (ragged corpora: the longest buffer's samples)
  if (a.out) a.out[(int64_t) blockIdx.y * os + i] = y;
  if (a.out32) a.out32[(int64_t) blockIdx.y * os + i] = (float) y;
}

// ---- fft sizes whose frame does not fit the LDS: the inverse through the global-memory passes ---------------------
// pack: Z[k] = conj(E[k] + i O[k]) of the masked spectrum (as in resynth_frames_kernel) for a chunk of frames of one
// component; unpack: frame[i] = (conj(z) / nc) interleaved, times the window.
__global__ void big_ipack_kernel(ResynthArgs a, int comp, int t0, int nf, d2* buf)
{
  const int nc = a.fft / 2;
  const int64_t idx = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t) nf * nc) return;
  const int k = (int) (idx % nc), t = t0 + (int) (idx / nc);
  const d2* tw = reinterpret_cast<const d2*>(a.twiddle);
  const double* spec = a.spec + (int64_t) t * a.F * 2;
  const double* vhat = a.Vhat + (int64_t) t * a.ldV;
  const bool useMask = a.Wf != nullptr;
  const double hk = useMask ? a.H1[(int64_t) t * a.Kp + comp] : 0.0;
  auto masked = [&](int f) -> d2 {
    d2 x = reinterpret_cast<const d2*>(spec)[f];
    if (useMask)
    {
      const double est = hk * a.Wf[(int64_t) f * a.Kp + comp];
      const double m = fmin(est * (1.0 / fmax(vhat[f], kEpsilon)), 1.0);
      x = d2{x[0] * m, x[1] * m};
    }
    if (f == 0 || f == nc) x[1] = 0.0;
    return x;
  };
  const d2 X = masked(k), Xn = masked(nc - k);
  const d2 Xc = d2{Xn[0], -Xn[1]};
  const d2 E = d2{0.5 * (X[0] + Xc[0]), 0.5 * (X[1] + Xc[1])};
  const d2 D = d2{0.5 * (X[0] - Xc[0]), 0.5 * (X[1] - Xc[1])};
  const d2 w = tw[k];
  const d2 O = cmul_i(D, d2{w[0], -w[1]});
  const d2 Z = d2{E[0] - O[1], E[1] + O[0]};
  buf[idx] = d2{Z[0], -Z[1]};
}

__global__ void big_iunpack_kernel(const d2* z, ResynthArgs a, int t0, int nf, double* frames)
{
  const int nc = a.fft / 2;
  const int64_t idx = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t) nf * nc) return;
  const int m = (int) (idx % nc), t = t0 + (int) (idx / nc);
  const d2 v = z[idx];
  const double inv = 1.0 / (double) nc;
  double* fr = frames + (int64_t) t * a.win;
  const int i0 = 2 * m, i1 = 2 * m + 1;
  if (i0 < a.win) fr[i0] = (v[0] * inv) * a.window[i0];
  if (i1 < a.win) fr[i1] = (-v[1] * inv) * a.window[i1];
}

static void launch_resynth_big(const ResynthArgs& a, hipStream_t s)
{
  const int nc = a.fft / 2;
  int64_t cf = 1;
  (void) big_fft_scratch_bytes(a.fft, a.T, &cf);
  double* bufA = a.bigScratch;
  double* bufB = a.bigScratch + cf * nc * 2;
  const int ncomp = a.nComp < 1 ? 1 : a.nComp;
  for (int c = 0; c < ncomp; c++)
  {
    double* frames = a.frames + (int64_t) c * a.T * a.win;
    for (int t0 = 0; t0 < a.T; t0 += (int) cf)
    {
      const int nf = (int) std::min<int64_t>(cf, a.T - t0);
      const int64_t np = (int64_t) nf * nc;
      hipLaunchKernelGGL(big_ipack_kernel, dim3((unsigned) ((np + 255) / 256)), dim3(256), 0, s, a, a.k + c, t0, nf,
                         reinterpret_cast<d2*>(bufA));
      const double* z = launch_big_fft_passes(bufA, bufB, nc, a.fft, a.twiddle, nf, s);
      hipLaunchKernelGGL(big_iunpack_kernel, dim3((unsigned) ((np + 255) / 256)), dim3(256), 0, s,
                         reinterpret_cast<const d2*>(z), a, t0, nf, frames);
    }
  }
  const unsigned nc2 = (unsigned) ncomp;
  hipLaunchKernelGGL(resynth_ola_kernel, dim3((unsigned) ((a.n + 255) / 256), nc2), dim3(256), 0, s, a);
}

void launch_resynth(const ResynthArgs& a, hipStream_t s)
{
  if (a.bigScratch) { launch_resynth_big(a, s); return; }
  const size_t shmem = (size_t) a.fft * 2 * sizeof(double); // two complex buffers of fft/2 points
  request_dynamic_lds(resynth_frames_kernel, (size_t) (160 * 1024));
  int threads = a.fft / 8;
  if (threads < 64) threads = 64;
  if (threads > 256) threads = 256;
  const unsigned nc = (unsigned) (a.nComp < 1 ? 1 : a.nComp);
  hipLaunchKernelGGL(resynth_frames_kernel, dim3((unsigned) a.T, nc), dim3((unsigned) threads), shmem, s, a);
  hipLaunchKernelGGL(resynth_ola_kernel, dim3((unsigned) ((a.n + 255) / 256), nc), dim3(256), 0, s, a);
}

// ---- BufSTFT plumbing --------------------------------------------------------------------
__global__ void spec_to_magphase_kernel(const double* spec, int T, int F, float* mag, float* phase)
{
  // tile transpose [T][F] -> [F][T] so both sides are coalesced
  __shared__ float tm[32][33], tp[32][33];
  const int f0 = blockIdx.x * 32, t0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int j = ty; j < 32; j += 8)
  {
    const int t = t0 + j, f = f0 + tx;
    float m = 0.f, p = 0.f;
    if (t < T && f < F)
    {
      const d2 x = reinterpret_cast<const d2*>(spec)[(int64_t) t * F + f];
      m = (float) sqrt(x[0] * x[0] + x[1] * x[1]); // alg/STFT.hpp:61-66
      p = (float) atan2(x[1], x[0]);                // alg/STFT.hpp:75-79 (arg)
    }
    tm[j][tx] = m;
    tp[j][tx] = p;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8)
  {
    const int f = f0 + j, t = t0 + tx;
    if (f < F && t < T)
    {
      if (mag) mag[(int64_t) f * T + t] = tm[tx][j];
      if (phase) phase[(int64_t) f * T + t] = tp[tx][j];
    }
  }
}

void launch_spec_to_magphase(const double* spec, int T, int F, float* mag, float* phase, hipStream_t s)
{
  dim3 g((unsigned) ((F + 31) / 32), (unsigned) ((T + 31) / 32));
  hipLaunchKernelGGL(spec_to_magphase_kernel, g, dim3(256), 0, s, spec, T, F, mag, phase);
}

__global__ void polar_to_spec_kernel(const float* mag, const float* phase, int T, int F, double* spec)
{
  __shared__ double tr[32][33], ti[32][33];
  const int t0 = blockIdx.x * 32, f0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int j = ty; j < 32; j += 8)
  {
    const int f = f0 + j, t = t0 + tx;
    double re = 0.0, im = 0.0;
    if (f < F && t < T)
    {
      // std::polar(m, p) on the buffers' float samples (nrt/BufSTFTClient.hpp:248-250): std::polar<float>, i.e.
      // single-precision m cos p and m sin p, widened into the complex<double> frame afterwards
      const float m = mag[(int64_t) f * T + t], p = phase[(int64_t) f * T + t];
      re = (double) (m * cosf(p));
      im = (double) (m * sinf(p));
    }
    tr[j][tx] = re;
    ti[j][tx] = im;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8)
  {
    const int t = t0 + j, f = f0 + tx;
    if (t < T && f < F) reinterpret_cast<d2*>(spec)[(int64_t) t * F + f] = d2{tr[tx][j], ti[tx][j]};
  }
}

void launch_polar_to_spec(const float* mag, const float* phase, int T, int F, double* spec, hipStream_t s)
{
  dim3 g((unsigned) ((T + 31) / 32), (unsigned) ((F + 31) / 32));
  hipLaunchKernelGGL(polar_to_spec_kernel, g, dim3(256), 0, s, mag, phase, T, F, spec);
}

} // namespace fluhip
