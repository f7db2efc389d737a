// Tile records of the bin-tiled W update (kernels_nmf_bintile.hip): column statistics of W' per tile of four bins,
// stat[kind][k][rec] with kind = sum x^2, sum x, max x.  Included by the kernels that consume them (the bin-tiled W update
// itself and the frame-strip kernel's prologue, kernels_nmf_strip.hip).  ONE summation order for every consumer, so that
// every workgroup of every launch sees the same norms bit for bit (deferred normalisation, alg/NMF.hpp:162):
//   lane j of the 16 lanes of a column takes the records j + 16 u, u = 0 .. kRecU - 1, as FOUR partial sums over u = q (mod 4)
//   (u ascending), adds them ((p0 + p1) + p2) + p3, and the 16 lanes close with an xor butterfly.
// A 256-thread consumer (the strip kernel) forms the four partials in one thread (QT = 1); the 1024-thread tile kernel gives
// column k to wavefront k and partial q to its lanes 16 q .. 16 q + 15 (QT = 4: a fourth of the requests and registers per
// thread), joined by lane reads in the same order.
#pragma once
#include "fluhip_kernels.h"

namespace fluhip {
namespace bintile {

constexpr int kRecU = 20; // records per lane of a column: up to 320 records (F <= 1281)

template <int QT>
struct TileStatRaw
{
  static constexpr int N = kRecU / QT;
  double v2[N], v1[N], vm[N];
};
// the requests (no sum yet: a caller with other loads to issue puts them behind these and consumes nothing in between).
// QT = 1: thread tid < 256 is lane j = tid & 15 of column k = tid >> 4.  QT = 4: wavefront k = tid >> 6 (16 wavefronts),
// lane j = tid & 15, partial q = (tid >> 4) & 3.
template <int QT>
__device__ __forceinline__ TileStatRaw<QT> tile_column_stats_load(const double* stat, int nRec, int tid)
{
  static_assert(QT == 1 || QT == 4, "partials per thread");
  const int k = QT == 4 ? (tid >> 6) & 15 : (tid >> 4) & 15, j = tid & 15, q0 = QT == 4 ? (tid >> 4) & 3 : 0;
  TileStatRaw<QT> w;
#pragma unroll
  for (int n = 0; n < TileStatRaw<QT>::N; n++)
  {
    const int u = QT == 4 ? q0 + 4 * n : n;
    const int i = min(j + 16 * u, nRec - 1);
    w.v2[n] = stat[(int64_t) k * nRec + i];
    w.v1[n] = stat[(int64_t) (16 + k) * nRec + i];
    w.vm[n] = stat[(int64_t) (32 + k) * nRec + i];
  }
  return w;
}
// sc: 48 doubles of LDS scratch; out nrmL[16] (1 when W is normalised), csL[16] column sums.  Two barriers; every thread of
// the workgroup calls it (256 threads with QT = 1, 1024 with QT = 4).
template <int QT>
__device__ __forceinline__ void tile_column_stats(const TileStatRaw<QT>& w, int nRec, int K, int wPend, double* sc, double* nrmL,
                                                  double* csL, int tid)
{
  const int k = QT == 4 ? (tid >> 6) & 15 : (tid >> 4) & 15, j = tid & 15;
  double s2 = 0.0, s1 = 0.0, mx = 0.0;
  if constexpr (QT == 1)
  {
    double p2[4] = {0.0, 0.0, 0.0, 0.0}, p1[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int u = 0; u < kRecU; u++)
      if (j + 16 * u < nRec)
      {
        p2[u & 3] += w.v2[u];
        p1[u & 3] += w.v1[u];
        mx = fmax(mx, w.vm[u]);
      }
    s2 = ((p2[0] + p2[1]) + p2[2]) + p2[3];
    s1 = ((p1[0] + p1[1]) + p1[2]) + p1[3];
  }
  else
  {
    const int q0 = (tid >> 4) & 3;
    double p2 = 0.0, p1 = 0.0, pm = 0.0;
#pragma unroll
    for (int n = 0; n < TileStatRaw<QT>::N; n++)
      if (j + 16 * (q0 + 4 * n) < nRec)
      {
        p2 += w.v2[n];
        p1 += w.v1[n];
        pm = fmax(pm, w.vm[n]);
      }
    // the four partials of lane j sit 16 lanes apart: every lane reads them in the order q = 0 .. 3
    s2 = ((__shfl(p2, j) + __shfl(p2, j + 16)) + __shfl(p2, j + 32)) + __shfl(p2, j + 48);
    s1 = ((__shfl(p1, j) + __shfl(p1, j + 16)) + __shfl(p1, j + 32)) + __shfl(p1, j + 48);
    mx = fmax(fmax(__shfl(pm, j), __shfl(pm, j + 16)), fmax(__shfl(pm, j + 32), __shfl(pm, j + 48)));
  }
#pragma unroll
  for (int sh = 1; sh < 16; sh <<= 1)
  {
    s2 += __shfl_xor(s2, sh);
    s1 += __shfl_xor(s1, sh);
    mx = fmax(mx, __shfl_xor(mx, sh));
  }
  if ((QT == 4 ? (tid & 63) : j) == 0)
  {
    sc[k] = s2;
    sc[16 + k] = s1;
    sc[32 + k] = mx;
  }
  __syncthreads();
  if (tid < 16)
  {
    double gmax = 0.0;
#pragma unroll
    for (int kk = 0; kk < 16; kk++) gmax = fmax(gmax, sc[32 + kk]);
    // alg/NMF.hpp:162 "if (W.maxCoeff() > epsilon) W.colwise().normalize()"; padded columns keep a divisor of one
    nrmL[tid] = (wPend && tid < K && gmax > kEpsilon) ? sqrt(sc[tid]) : 1.0;
    csL[tid] = sc[16 + tid];
  }
  __syncthreads();
}

} // namespace bintile
} // namespace fluhip
