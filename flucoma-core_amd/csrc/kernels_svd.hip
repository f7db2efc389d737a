// kernels_svd.hip -- one-sided Jacobi (Hestenes) SVD for NNDSVD seeding (alg/NNDSVD.hpp:42-46 uses
// Eigen's BDCSVD on the host).
//
// The magnitude spectrogram's transposed copy G [n][ldg] holds one bin per row (n = bins, rows of length T =
// frames, contiguous).  Rotating pairs of rows until all rows are mutually orthogonal, G <- R G with R orthogonal,
// gives  X^T = R^T (R G) = U S V^T  with  U = R^T (columns = rows of the accumulated rotations, kept in Jt [n][n],
// which starts as the identity and receives the same row rotations), S = the row norms of the final G, V^T = the
// normalised rows.  Pairs of one round of a round-robin tournament are disjoint, so a round is one launch with a
// workgroup per pair: a coalesced pass for (|x|^2, |y|^2, x.y), the rotation that zeroes x.y, a second pass that
// applies it to the two rows of G and of Jt.  Sweeps repeat until no pair of a whole sweep was further from
// orthogonal than kTol.  Everything is plain FP64 VALU work at HBM / L2 speed -- the rows of a pair are read twice
// and written once per round -- which for the shapes of this path (n ~ 1e3, T ~ 1e3..1e4) is a few hundred
// milliseconds, with singular vectors accurate to rounding (Jacobi's relative accuracy), no library behind it.
#include "fluhip_kernels.h"

#include <vector>

namespace fluhip {

constexpr double kJacobiTol = 1e-15;

// round-robin tournament on m (even) players: round r in [0, m-1), pair i in [0, m/2)
__device__ __forceinline__ void tournament_pair(int m, int r, int i, int& p, int& q)
{
  const int mm = m - 1;
  if (i == 0) { p = mm; q = r % mm; }
  else { p = (r + i) % mm; q = (r - i + mm) % mm; }
  if (p > q) { const int t = p; p = q; q = t; }
}

__global__ __launch_bounds__(256) void jacobi_round_kernel(double* G, int64_t ldg, int T, double* Jt, int n, int m,
                                                           int round, unsigned* maxOff, double zero2)
{
  __shared__ double red[3][256 / 64];
  __shared__ double cs[2];
  int p, q;
  tournament_pair(m, round, blockIdx.x, p, q);
  if (q >= n) return; // the padding player of an odd n
  double* x = G + (int64_t) p * ldg;
  double* y = G + (int64_t) q * ldg;
  double a = 0.0, b = 0.0, d = 0.0;
  for (int t = threadIdx.x; t < T; t += blockDim.x)
  {
    const double xv = x[t], yv = y[t];
    a = fma(xv, xv, a);
    b = fma(yv, yv, b);
    d = fma(xv, yv, d);
  }
  for (int off = 32; off > 0; off >>= 1)
  {
    a += __shfl_down(a, off);
    b += __shfl_down(b, off);
    d += __shfl_down(d, off);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red[0][wave] = a; red[1][wave] = b; red[2][wave] = d; }
  __syncthreads();
  if (threadIdx.x == 0)
  {
    a = b = d = 0.0;
    for (int w = 0; w < (int) (blockDim.x >> 6); w++) { a += red[0][w]; b += red[1][w]; d += red[2][w]; }
    double c = 1.0, s = 0.0;
    const double lim = sqrt(a) * sqrt(b);
    // rows that have shrunk to rounding level of the whole matrix are zero (rank-deficient input: more bins than
    // frames); relative orthogonality among them is noise and would rotate for ever
    if (a > zero2 && b > zero2 && fabs(d) > kJacobiTol * lim)
    {
      atomicMax(maxOff, __float_as_uint((float) (fabs(d) / lim)));
      const double zeta = (b - a) / (2.0 * d);
      const double t = copysign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
      c = 1.0 / sqrt(1.0 + t * t);
      s = c * t;
    }
    cs[0] = c;
    cs[1] = s;
  }
  __syncthreads();
  const double c = cs[0], s = cs[1];
  if (s == 0.0) return;
  for (int t = threadIdx.x; t < T; t += blockDim.x)
  {
    const double xv = x[t], yv = y[t];
    x[t] = c * xv - s * yv;
    y[t] = s * xv + c * yv;
  }
  double* jx = Jt + (int64_t) p * n;
  double* jy = Jt + (int64_t) q * n;
  for (int t = threadIdx.x; t < n; t += blockDim.x)
  {
    const double xv = jx[t], yv = jy[t];
    jx[t] = c * xv - s * yv;
    jy[t] = s * xv + c * yv;
  }
}

__global__ void identity_kernel(double* Jt, int n)
{
  const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (int64_t) n * n) Jt[i] = (i / n == i % n) ? 1.0 : 0.0;
}

__global__ __launch_bounds__(256) void rownorm_kernel(const double* G, int64_t ldg, int T, double* out)
{
  __shared__ double red[256 / 64];
  const double* x = G + (int64_t) blockIdx.x * ldg;
  double a = 0.0;
  for (int t = threadIdx.x; t < T; t += blockDim.x) a = fma(x[t], x[t], a);
  for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0)
  {
    a = 0.0;
    for (int w = 0; w < (int) (blockDim.x >> 6); w++) a += red[w];
    out[blockIdx.x] = sqrt(a);
  }
}

// G [n][ldg] (rows of length T) is overwritten by S V^T (row i = s_i v_i^T, unsorted); Jt [n][n] receives U^T
// (row i = u_i^T); norms [n] = the singular values in row order.  flag: one device word of scratch.
// Returns the number of sweeps, or -1 when maxSweeps did not reach orthogonality.
int launch_jacobi_svd(double* G, int64_t ldg, int n, int T, double* Jt, double* norms, unsigned* flag, int maxSweeps,
                      hipStream_t s)
{
  const int m = (n + 1) & ~1;
  // squared Frobenius norm (invariant under the rotations) for the "this row is zero" threshold
  hipLaunchKernelGGL(rownorm_kernel, dim3((unsigned) n), dim3(256), 0, s, G, ldg, T, norms);
  std::vector<double> h0((size_t) n);
  if (hipMemcpyAsync(h0.data(), norms, (size_t) n * sizeof(double), hipMemcpyDeviceToHost, s) != hipSuccess) return -1;
  if (hipStreamSynchronize(s) != hipSuccess) return -1;
  double fro2 = 0.0;
  for (double v : h0) fro2 += v * v;
  const double zero2 = fro2 * 2.0e-29; // (64 eps)^2 / ~10: a row below ~1e-14 of the matrix norm
  hipLaunchKernelGGL(identity_kernel, dim3((unsigned) (((int64_t) n * n + 255) / 256)), dim3(256), 0, s, Jt, n);
  int sweeps = -1;
  for (int sw = 0; sw < maxSweeps; sw++)
  {
    (void) hipMemsetAsync(flag, 0, sizeof(unsigned), s);
    for (int r = 0; r < m - 1; r++)
      hipLaunchKernelGGL(jacobi_round_kernel, dim3((unsigned) (m / 2)), dim3(256), 0, s, G, ldg, T, Jt, n, m, r, flag,
                         zero2);
    unsigned h = 1;
    if (hipMemcpyAsync(&h, flag, sizeof(h), hipMemcpyDeviceToHost, s) != hipSuccess) return -1;
    if (hipStreamSynchronize(s) != hipSuccess) return -1;
    if (h == 0) { sweeps = sw + 1; break; } // a whole sweep without a rotation
  }
  hipLaunchKernelGGL(rownorm_kernel, dim3((unsigned) n), dim3(256), 0, s, G, ldg, T, norms);
  return sweeps;
}

} // namespace fluhip
