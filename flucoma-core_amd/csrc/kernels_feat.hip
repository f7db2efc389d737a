// kernels_feat.hip -- MelBands / MFCC on top of the batched STFT magnitudes.
//
//   algorithm::MelBands::processFrame   include/flucoma/algorithms/public/MelBands.hpp:79-97
//   algorithm::DCT::processFrame        include/flucoma/algorithms/public/DCT.hpp:65-75
//   client glue                         include/flucoma/clients/rt/MFCCClient.hpp:122-130,
//                                       include/flucoma/clients/rt/MelBandsClient.hpp:104-113
//
// One wavefront owns FT frames; lane = mel band (bands beyond 64 are walked in chunks).  A mel filter is a triangle:
// of the F bins a band touches a few dozen, so a lane only walks its band's support -- the magnitude rows of the
// wavefront's frames are staged in LDS (coalesced loads) and gathered from there, the weights come from a table
// packed over the support (i-major, band contiguous: coalesced).  Skipping the zero weights leaves the sums
// bit-identical to the dense ascending-bin dot product (adding w = 0 terms changes nothing).
// Sums run over bins / bands in ascending order, like a sequential dot product.
#include "fluhip_kernels.h"

namespace fluhip {

constexpr int kFT = 4; // frames per wavefront (4 x F doubles of LDS each: 16 KB at fft 1024, 64 KB per workgroup)

__global__ __launch_bounds__(256) void mel_kernel(FeatArgs a)
{
  extern __shared__ double lds[]; // [4 waves][kFT][bandsPad] log-band energies for the DCT, then [4 waves][kFT][F] magnitude rows
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.y;
  const int nw = blockDim.x >> 6; // wavefronts per workgroup: 4, fewer when the magnitude rows are long
  const int t0 = (blockIdx.x * nw + wave) * kFT;
  if (t0 >= a.T) return;
  const double* mag = a.mag + (int64_t) b * a.magStride;
  double* wl = lds + (size_t) wave * kFT * a.bandsPad;
  const double scale1 = 1.0 / ((double) a.win / 4.0);                     // alg/MelBands.hpp:49
  const double scale2 = 1.0 / (2.0 * (double) (2 * (a.F - 1)) / (double) a.win); // :52

  // energy of each frame (only used when magNorm): sum_f mag * scale1, times scale2 (:86-87)
  double energy[kFT];
#pragma unroll
  for (int i = 0; i < kFT; i++) energy[i] = 0.0;
  if (a.magNorm)
  {
    for (int f = 0; f < a.F; f++)
#pragma unroll
      for (int i = 0; i < kFT; i++)
      {
        const int t = min(t0 + i, a.T - 1);
        energy[i] += mag[(int64_t) t * a.ldMag + f] * scale1;
      }
#pragma unroll
    for (int i = 0; i < kFT; i++) energy[i] *= scale2;
  }

  double bandSum[kFT];
#pragma unroll
  for (int i = 0; i < kFT; i++) bandSum[i] = 0.0;

  // magnitude rows of this wavefront's frames -> LDS (behind the band scratch)
  double* rows = lds + (size_t) nw * kFT * a.bandsPad + (size_t) wave * kFT * a.F;
#pragma unroll
  for (int i = 0; i < kFT; i++)
  {
    const int t = min(t0 + i, a.T - 1);
    for (int f = lane; f < a.F; f += 64) rows[i * a.F + f] = mag[(int64_t) t * a.ldMag + f];
  }
  for (int c0 = 0; c0 < a.nBands; c0 += 64)
  {
    const int band = c0 + lane;
    double acc[kFT];
#pragma unroll
    for (int i = 0; i < kFT; i++) acc[i] = 0.0;
    const int lo = a.bandLo[band];
    for (int j = 0; j < a.maxLen; j++)
    {
      const double w = a.wpack[(int64_t) j * a.bandsPad + band]; // zero past the band's support / beyond nBands
      const int f = min(lo + j, a.F - 1);
#pragma unroll
      for (int i = 0; i < kFT; i++)
      {
        double m = rows[i * a.F + f];
        if (a.magNorm) m = m * scale1;
        if (a.usePower) m = m * m;
        acc[i] += w * m;                                          // :90-91
      }
    }
#pragma unroll
    for (int i = 0; i < kFT; i++) wl[i * a.bandsPad + band] = acc[i];
    // sum over bands for the magNorm renormalisation (:93), ascending band order per chunk
    if (a.magNorm)
    {
#pragma unroll
      for (int i = 0; i < kFT; i++)
      {
        double s = (band < a.nBands) ? acc[i] : 0.0;
        for (int off = 1; off < 64; off <<= 1) s += __shfl_xor(s, off);
        bandSum[i] += s;
      }
    }
  }
  // finish the band energies in place (normalise, dB)
  for (int c0 = 0; c0 < a.nBands; c0 += 64)
  {
    const int band = c0 + lane;
#pragma unroll
    for (int i = 0; i < kFT; i++)
    {
      double v = wl[i * a.bandsPad + band];
      if (a.magNorm) v = v * energy[i] / fmax(kEpsilon, bandSum[i]);
      if (a.logOutput) v = 20.0 * log10(fmax(v, kEpsilon));        // :95
      wl[i * a.bandsPad + band] = v;
    }
  }
  // LDS writes above and reads below are by the same wavefront: program order suffices
  if (!a.dct)
  {
    // BufMelBands output: out[b][band][t]
    for (int c0 = 0; c0 < a.nBands; c0 += 64)
    {
      const int band = c0 + lane;
      if (band < a.nBands)
#pragma unroll
        for (int i = 0; i < kFT; i++)
          if (t0 + i < a.T)
            a.out[((int64_t) b * a.nOut + band) * a.T + t0 + i] = (float) wl[i * a.bandsPad + band];
    }
    return;
  }
  // MFCC: coef[j] = sum_band dct[j][band] * logband[band], j = startCoeff .. startCoeff + nOut - 1
  for (int o = lane; o < kFT * a.nOut; o += 64)
  {
    const int i = o / a.nOut, j = o % a.nOut;
    double s = 0.0;
    if (a.startCoeff + j < a.nDct)
    {
      const double* drow = a.dct + (int64_t) (a.startCoeff + j) * a.nBands;
      for (int band = 0; band < a.nBands; band++) s += drow[band] * wl[i * a.bandsPad + band]; // alg/DCT.hpp:73-75
    }
    if (t0 + i < a.T) a.out[((int64_t) b * a.nOut + j) * a.T + t0 + i] = (float) s;
  }
}

void launch_features(const FeatArgs& a, hipStream_t s)
{
  int nw = 4;
  while (nw > 1 && (size_t) nw * kFT * (a.bandsPad + a.F) * sizeof(double) > 144 * 1024) nw >>= 1;
  const size_t shmem = (size_t) nw * kFT * (a.bandsPad + a.F) * sizeof(double);
  request_dynamic_lds(mel_kernel, (size_t) (160 * 1024));
  dim3 grid((unsigned) ((a.T + nw * kFT - 1) / (nw * kFT)), (unsigned) a.B);
  hipLaunchKernelGGL(mel_kernel, grid, dim3((unsigned) (64 * nw)), shmem, s, a);
}

} // namespace fluhip
