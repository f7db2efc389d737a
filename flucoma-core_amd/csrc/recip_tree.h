// recip_tree.h -- reciprocals of N positive doubles with ONE v_rcp_f64 per group of up to G (round 6).
//
// The factor updates divide by max(Q, eps) element by element (alg/NMF.hpp:160, 168); on gfx950 v_rcp_f64 is a quarter-rate
// instruction on the FP64 datapath the MFMAs share, so a reciprocal costs four operation slots before its Newton step.  A
// group shares one: a binary tree of products up (a[l][j] = the product of node j of level l), v_rcp_f64 + one Newton step
// at the root (2^-46, as the single reciprocal of rounds 1 - 5), and the inverses handed down -- the inverse of a child is
// the inverse of its parent times its sibling.  A group of n costs (n - 1) + 6 + 2 (n - 1) slots: 9 for two (rounds 2 - 5's
// pairs), 12 for three, 15 for four, 21 for six, against 6 n alone.  Every product adds one rounding (1.1e-16) under the
// Newton step's 1.4e-14.  Range: the root is a product of up to G operands in [eps, max Q]; with Q <= 3.4e38 * window
// (any float input) a group of four reaches 2.4e167 and a group of six 1e251 -- inside the double range -- and eps^6 = 1e-94.
// Written level by level across the groups: independent chains side by side for the in-order VALU.
#pragma once

namespace fluhip {

template <int N, int G>
__device__ __forceinline__ void recip_tree(const double (&d)[N], double (&y)[N])
{
  static_assert(G >= 1 && G <= 8, "operands per reciprocal");
  constexpr int NGR = (N + G - 1) / G;
  constexpr int L = G > 4 ? 3 : (G > 2 ? 2 : (G > 1 ? 1 : 0));      // levels above the leaves
  // (the levels of chunk c live at a[l][c G ..]; nodes(cnt, l) = its nodes at level l)
  auto nodes = [](int cnt, int l) { for (int i = 0; i < l; i++) cnt = (cnt + 1) / 2; return cnt; };
  double a[L > 0 ? L : 1][N], inv[N];
  // up
#pragma unroll
  for (int l = 0; l < L; l++)
#pragma unroll
    for (int c = 0; c < NGR; c++)
    {
      const int cnt = (c == NGR - 1) ? N - c * G : G;
      const int nl = nodes(cnt, l), nu = nodes(cnt, l + 1);
#pragma unroll
      for (int j = 0; j < (G + 1) / 2; j++)
      {
        if (j >= nu) break;
        const double x0 = l == 0 ? d[c * G + 2 * j] : a[l > 0 ? l - 1 : 0][c * G + 2 * j];
        if (2 * j + 1 < nl)
        {
          const double x1 = l == 0 ? d[c * G + 2 * j + 1] : a[l > 0 ? l - 1 : 0][c * G + 2 * j + 1];
          a[l][c * G + j] = x0 * x1;
        }
        else a[l][c * G + j] = x0;
      }
    }
  // root: reciprocal + one Newton step
  double rr[NGR], ee[NGR];
#pragma unroll
  for (int c = 0; c < NGR; c++) rr[c] = __builtin_amdgcn_rcp(L > 0 ? a[L > 0 ? L - 1 : 0][c * G] : d[c * G]);
#pragma unroll
  for (int c = 0; c < NGR; c++) ee[c] = __builtin_fma(-(L > 0 ? a[L > 0 ? L - 1 : 0][c * G] : d[c * G]), rr[c], 1.0);
#pragma unroll
  for (int c = 0; c < NGR; c++) inv[c * G] = __builtin_fma(rr[c], ee[c], rr[c]);
  // down: inv[c G + j] = the inverse of node j of the level at hand
#pragma unroll
  for (int l = L - 1; l >= 0; l--)
#pragma unroll
    for (int c = 0; c < NGR; c++)
    {
      const int cnt = (c == NGR - 1) ? N - c * G : G;
      const int nl = nodes(cnt, l), nu = nodes(cnt, l + 1);
      double nx[G];
#pragma unroll
      for (int j = 0; j < (G + 1) / 2; j++)
      {
        if (j >= nu) break;
        const double up = inv[c * G + j];
        if (2 * j + 1 < nl)
        {
          const double x0 = l == 0 ? d[c * G + 2 * j] : a[l > 0 ? l - 1 : 0][c * G + 2 * j];
          const double x1 = l == 0 ? d[c * G + 2 * j + 1] : a[l > 0 ? l - 1 : 0][c * G + 2 * j + 1];
          nx[2 * j] = up * x1;
          nx[2 * j + 1] = up * x0;
        }
        else nx[2 * j] = up;
      }
#pragma unroll
      for (int j = 0; j < G; j++)
      {
        if (j >= nl) break;
        inv[c * G + j] = nx[j];
      }
    }
#pragma unroll
  for (int g = 0; g < N; g++) y[g] = inv[g];
}

} // namespace fluhip
