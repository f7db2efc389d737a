// kernels_nmf_wide.hip -- factor update for ranks above 128: the un-fused form.
//
// The fused kernels (kernels_nmf5.hip) keep a wavefront's stationary rows in registers, which ends at Kp = 128.
// Beyond that the update is done the way the reference writes it (alg/NMF.hpp:158-170), in three steps over a
// materialised ratio matrix, with one LDS-tiled FP64 GEMM kernel on the matrix cores (v_mfma_f64_16x16x4):
//   Q[r][c]  = sum_k Mv[r][k] S[c][k]            (GEMM, "NT")          R x C
//   Q[r][c]  = V[r][c] / max(Q[r][c], eps)       (in place)
//   N[c][k]  = sum_r Q[r][c] Mv[r][k]            (GEMM, "TN")          C x Kp
//   S[c][k] <- S[c][k] N[c][k] / max(sum_r Mv[r][k], eps)
// Rare in practice (FluCoMa ranks are a handful), so this path is correct and on the matrix cores, not tuned; it also
// serves the tests as an independent second implementation of the same update.
#include "fluhip_kernels.h"

namespace fluhip {

constexpr int TM = 64, TN = 64, TK = 16; // workgroup tile; four wavefronts x (2 x 2) MFMA tiles of 16 x 16

// C[m][n] = sum_k A(m,k) B(n,k)   with A(m,k) = A[m*lda + k] (TRANSA = 0) or A[k*lda + m] (TRANSA = 1),
//                                       B(n,k) = B[n*ldb + k] (TRANSB = 0) or B[k*ldb + n] (TRANSB = 1)
template <int TRANSA, int TRANSB>
__global__ __launch_bounds__(256) void dgemm_tile_kernel(const double* A, int64_t lda, int64_t strideA, const double* B,
                                                         int64_t ldb, int64_t strideB, double* C, int64_t ldc,
                                                         int64_t strideC, int M, int N, int K)
{
  __shared__ double As[TK][TM + 1], Bs[TK][TN + 1];
  const int b = blockIdx.z;
  A += (int64_t) b * strideA; B += (int64_t) b * strideB; C += (int64_t) b * strideC;
  const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN;
  // four wavefronts, a 32 x 32 quadrant each = 2 x 2 tiles of v_mfma_f64_16x16x4: A-operand lane l = (row l % 16, k l / 16),
  // B-operand lane l = (k l / 16, column l % 16), result register e of lane l = (row l / 16 + 4 e, column l % 16)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
  const int lr = lane & 15, lk = lane >> 4;
  typedef double d4 __attribute__((ext_vector_type(4)));
  d4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};
  for (int k0 = 0; k0 < K; k0 += TK)
  {
    for (int e = threadIdx.x; e < TM * TK; e += 256)
    {
      int m, k;
      if (TRANSA) { m = e % TM; k = e / TM; } else { k = e % TK; m = e / TK; }
      const int gm = m0 + m, gk = k0 + k;
      As[k][m] = (gm < M && gk < K) ? (TRANSA ? A[(int64_t) gk * lda + gm] : A[(int64_t) gm * lda + gk]) : 0.0;
    }
    for (int e = threadIdx.x; e < TN * TK; e += 256)
    {
      int n, k;
      if (TRANSB) { n = e % TN; k = e / TN; } else { k = e % TK; n = e / TK; }
      const int gn = n0 + n, gk = k0 + k;
      Bs[k][n] = (gn < N && gk < K) ? (TRANSB ? B[(int64_t) gk * ldb + gn] : B[(int64_t) gn * ldb + gk]) : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < TK; kk += 4)
    {
      double a[2], bb[2];
#pragma unroll
      for (int i = 0; i < 2; i++) a[i] = As[kk + lk][wm + 16 * i + lr];
#pragma unroll
      for (int j = 0; j < 2; j++) bb[j] = Bs[kk + lk][wn + 16 * j + lr];
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], bb[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int e = 0; e < 4; e++)
      {
        const int gm = m0 + wm + 16 * i + lk + 4 * e, gn = n0 + wn + 16 * j + lr;
        if (gm < M && gn < N) C[(int64_t) gm * ldc + gn] = acc[i][j][e];
      }
}

__global__ void ratio_inplace_kernel(double* Q, int64_t ldq, int64_t strideQ, const double* V, int64_t ldv,
                                     int64_t strideV, int R, int C)
{
  const int b = blockIdx.z;
  const int c = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
  if (c >= C || r >= R) return;
  double* q = Q + (int64_t) b * strideQ + (int64_t) r * ldq + c;
  *q = V[(int64_t) b * strideV + (int64_t) r * ldv + c] / fmax(*q, kEpsilon);
}

// S[c][k] <- S[c][k] N[c][k] / max(den[k], eps), den = column sums of Mv (launch_colsum)
__global__ void wide_apply_kernel(double* S, int64_t strideS, const double* Nm, int64_t strideN, const double* den,
                                  int C, int Kp)
{
  const int b = blockIdx.y;
  const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t) C * Kp) return;
  const double d = fmax(den[(int64_t) b * Kp + (i % Kp)], kEpsilon);
  double* s = S + (int64_t) b * strideS + i;
  *s = (*s * Nm[(int64_t) b * strideN + i]) / d;
}

int64_t nmf_update_wide_scratch_doubles(int R, int C, int Kp, int B)
{
  // ratio matrix [B][R][C] + numerators [B][C][Kp] + denominators [B][Kp] + column-sum partials
  return (int64_t) B * ((int64_t) R * C + (int64_t) C * Kp + Kp) + colsum_scratch_doubles(R, Kp, B);
}

void launch_nmf_update_wide(const UpdateArgs& a, double* scratch, hipStream_t s)
{
  const int R = a.R, C = a.C, Kp = a.Kp, B = a.B;
  double* Q = scratch;
  double* Nm = Q + (int64_t) B * R * C;
  double* den = Nm + (int64_t) B * C * Kp;
  double* csum = den + (int64_t) B * Kp;
  // Q = Mv S^T
  hipLaunchKernelGGL((dgemm_tile_kernel<0, 0>), dim3((unsigned) ((C + TN - 1) / TN), (unsigned) ((R + TM - 1) / TM), (unsigned) B),
                     dim3(256), 0, s, a.Mv, (int64_t) Kp, a.strideM, a.S, (int64_t) Kp, a.strideS, Q, (int64_t) C,
                     (int64_t) R * C, R, C, Kp);
  hipLaunchKernelGGL(ratio_inplace_kernel, dim3((unsigned) ((C + 255) / 256), (unsigned) R, (unsigned) B), dim3(256), 0, s, Q,
                     (int64_t) C, (int64_t) R * C, a.V, a.ldv, a.strideV, R, C);
  // N = ratio^T Mv
  hipLaunchKernelGGL((dgemm_tile_kernel<1, 1>), dim3((unsigned) ((Kp + TN - 1) / TN), (unsigned) ((C + TM - 1) / TM), (unsigned) B),
                     dim3(256), 0, s, Q, (int64_t) C, (int64_t) R * C, a.Mv, (int64_t) Kp, a.strideM, Nm, (int64_t) Kp,
                     (int64_t) C * Kp, C, Kp, R);
  launch_colsum(a.Mv, a.strideM, R, Kp, B, den, (int64_t) Kp, csum, s);
  const int64_t total = (int64_t) C * Kp;
  hipLaunchKernelGGL(wide_apply_kernel, dim3((unsigned) ((total + 255) / 256), (unsigned) B), dim3(256), 0, s, a.S, a.strideS,
                     Nm, (int64_t) C * Kp, den, C, Kp);
}

} // namespace fluhip
