// Host build of csrc/recip_tree.h (tests/test_recip_tree.py): v_rcp_f64 modelled as a reciprocal good to ~23 bits (what the
// instruction delivers), every (operands, group size) the kernels instantiate and the ones around them, operands spread over the
// range the factor updates see (eps .. 1e6) -- the worst relative error of y[i] d[i] - 1 per combination on stdout.
#include <cmath>
#include <cstdio>
#include <random>
#define __device__
#define __forceinline__ inline
static inline double __builtin_amdgcn_rcp(double x)
{
  const double r = 1.0 / x;
  int e;
  const double m = std::frexp(r, &e);
  return std::ldexp(std::floor(m * 8388608.0) / 8388608.0, e);      // 23 bits kept
}
#include "../../flucoma-core_amd/csrc/recip_tree.h"

template <int N, int G>
static double worst()
{
  std::mt19937_64 rng(1000 * N + G);
  std::uniform_real_distribution<double> ex(-15.6, 6.0);
  double w = 0.0;
  for (int rep = 0; rep < 2000; rep++)
  {
    double d[N], y[N];
    for (int i = 0; i < N; i++) d[i] = std::pow(10.0, ex(rng));
    if (rep == 0) for (int i = 0; i < N; i++) d[i] = 2.220446049250313e-16;   // every operand at the clamp
    if (rep == 1) for (int i = 0; i < N; i++) d[i] = 7e41;                    // the largest Q a float input can produce (3.4e38 x 2048)
    fluhip::recip_tree<N, G>(d, y);
    for (int i = 0; i < N; i++) w = std::fmax(w, std::fabs(y[i] * d[i] - 1.0));
  }
  return w;
}
template <int G>
static void row()
{
  std::printf("%d %.3e %.3e %.3e %.3e %.3e %.3e %.3e %.3e %.3e\n", G, worst<1, G>(), worst<2, G>(), worst<3, G>(), worst<4, G>(), worst<5, G>(),
              worst<6, G>(), worst<7, G>(), worst<8, G>(), worst<9, G>());
}
int main()
{
  row<1>(); row<2>(); row<3>(); row<4>(); row<6>();
  return 0;
}
