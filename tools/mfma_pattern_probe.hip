// mfma_pattern_probe -- does v_mfma_f64_4x4x4_4b_f64 keep its 16-cycle issue rate with the operand
// pattern of the NMF update (72 distinct B registers in the Q phase, 72 accumulators in the out
// phase, results of one phase feeding the next)?  No memory traffic, no division.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(256, 1) void k(double* out, int iters, double seed)
{
  constexpr int NG = 9, M = 8;
  double sb[NG][M], acc[NG][M], ma[M], mb[M], q[NG];
  const double t = seed + threadIdx.x * 1e-7;
  for (int g = 0; g < NG; g++)
    for (int m = 0; m < M; m++) { sb[g][m] = t * (g + 1) + m; acc[g][m] = 0.0; }
  for (int m = 0; m < M; m++) { ma[m] = t + m; mb[m] = t - m; }
  for (int it = 0; it < iters; it++)
  {
    if (MODE & 1)
    {
#pragma unroll
      for (int g = 0; g < NG; g++) q[g] = 0.0;
#pragma unroll
      for (int m = 0; m < M; m++)
#pragma unroll
        for (int g = 0; g < NG; g++) q[g] = __builtin_amdgcn_mfma_f64_4x4x4f64(ma[m], sb[g][m], q[g], 0, 0, 0);
    }
    else
    {
#pragma unroll
      for (int g = 0; g < NG; g++) q[g] = ma[g % M] * 1.0000001;
    }
    if (MODE & 2)
    {
#pragma unroll
      for (int g = 0; g < NG; g++)
#pragma unroll
        for (int m = 0; m < M; m++) acc[g][m] = __builtin_amdgcn_mfma_f64_4x4x4f64(q[g], mb[m], acc[g][m], 0, 0, 0);
    }
    else
    {
#pragma unroll
      for (int g = 0; g < NG; g++) acc[g][0] += q[g];
    }
#pragma unroll
    for (int m = 0; m < M; m++) ma[m] += 1e-9; // keep the loop from being hoisted
  }
  double s = 0;
  for (int g = 0; g < NG; g++)
    for (int m = 0; m < M; m++) s += acc[g][m];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename F>
static double time_ms(F f)
{
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  hipEventRecord(e0); f(); f(); f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  return ms / 3;
}
int main()
{
  double* out; hipMalloc(&out, 256 * 256 * 8);
  const int iters = 2000;
  auto rep = [&](const char* name, double ms, int mfmaPerIter) {
    double tf = 512.0 * mfmaPerIter * iters * 1024 / (ms * 1e-3) / 1e12;
    printf("%-28s %.3f ms  %.1f TF  (%.1f cycles/MFMA @2.4GHz)\n", name, ms, tf, ms * 1e-3 * 2.4e9 / ((double) mfmaPerIter * iters));
  };
  rep("Q phase only (72 MFMA)", time_ms([&] { hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, out, iters, 1.0); }), 72);
  rep("out phase only (72 MFMA)", time_ms([&] { hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, out, iters, 1.0); }), 72);
  rep("Q -> out chained (144 MFMA)", time_ms([&] { hipLaunchKernelGGL(k<3>, dim3(256), dim3(256), 0, 0, out, iters, 1.0); }), 144);
  return 0;
}
