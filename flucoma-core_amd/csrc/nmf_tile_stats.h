// Tile records of the bin-tiled W update (kernels_nmf_bintile.hip): column statistics of W' per tile of four bins,
// stat[kind][k][rec] with kind = sum x^2, sum x, max x.  Included by the kernels that consume them (the bin-tiled W update
// itself and the frame-strip kernel's prologue, kernels_nmf_strip.hip): every consumer adds a column's records in the same
// order, so every workgroup of every launch sees the same norms bit for bit (deferred normalisation, alg/NMF.hpp:162).
#pragma once
#include "fluhip_kernels.h"

namespace fluhip {
namespace bintile {

constexpr int kRecU = 17; // records per thread of the column statistics: 16 threads per column, up to 272 records

// Column statistics of W' from the tile records, stat[kind][k][rec] (kind = sum x^2, sum x, max x).  Thread (k = tid >> 4,
// j = tid & 15) takes the records j, j + 16, ...; the requests go out first (load), the sums later (reduce: two barriers).
// Every consumer adds in this order, so every workgroup of every launch sees the same norms bit for bit.
struct TileStatRaw
{
  double v2[kRecU], v1[kRecU], vm[kRecU];
};
__device__ __forceinline__ TileStatRaw tile_column_stats_load(const double* stat, int nRec, int tid)
{
  const int k = (tid >> 4) & 15, j = tid & 15;
  TileStatRaw w;
#pragma unroll
  for (int u = 0; u < kRecU; u++)
  {
    const int i = min(j + 16 * u, nRec - 1);
    w.v2[u] = stat[(int64_t) k * nRec + i];
    w.v1[u] = stat[(int64_t) (16 + k) * nRec + i];
    w.vm[u] = stat[(int64_t) (32 + k) * nRec + i];
  }
  return w;
}
// sc: 48 doubles of LDS scratch; out nrmL[16] (1 when W is normalised), csL[16] column sums; the first 256 threads work
__device__ __forceinline__ void tile_column_stats(const TileStatRaw& w, int nRec, int K, int wPend, double* sc, double* nrmL,
                                                  double* csL, int tid)
{
  if (tid < 256)
  {
    const int k = tid >> 4, j = tid & 15;
    double s2 = 0.0, s1 = 0.0, mx = 0.0;
#pragma unroll
    for (int u = 0; u < kRecU; u++)
      if (j + 16 * u < nRec)
      {
        s2 += w.v2[u];
        s1 += w.v1[u];
        mx = fmax(mx, w.vm[u]);
      }
#pragma unroll
    for (int sh = 1; sh < 16; sh <<= 1)
    {
      s2 += __shfl_xor(s2, sh);
      s1 += __shfl_xor(s1, sh);
      mx = fmax(mx, __shfl_xor(mx, sh));
    }
    if (j == 0)
    {
      sc[k] = s2;
      sc[16 + k] = s1;
      sc[32 + k] = mx;
    }
  }
  __syncthreads();
  if (tid < 16)
  {
    double gmax = 0.0;
#pragma unroll
    for (int k = 0; k < 16; k++) gmax = fmax(gmax, sc[32 + k]);
    // alg/NMF.hpp:162 "if (W.maxCoeff() > epsilon) W.colwise().normalize()"; padded columns keep a divisor of one
    nrmL[tid] = (wPend && tid < K && gmax > kEpsilon) ? sqrt(sc[tid]) : 1.0;
    csL[tid] = sc[16 + tid];
  }
  __syncthreads();
}

} // namespace bintile
} // namespace fluhip
