// valu_rate_probe.hip -- issue cost of single FP64/FP32 VALU instructions on gfx950, one wavefront per SIMD:
// 32 independent instructions of one kind per loop body, shader cycles per instruction.
//   hipcc -O3 --offload-arch=gfx950 tools/valu_rate_probe.hip -o tools/bin/valu_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>

enum { FMA64, RCP64, RCP32, CVT_F32_F64, CVT_F64_F32, MUL64, MAX64, RSQ64, FMA32, SQRT64 };

template <int KIND>
__global__ __launch_bounds__(256, 1) void probe(double* out, long long* cyc, int iters)
{
  double r[32];
  float f[32];
  for (int i = 0; i < 32; i++) { r[i] = 1.0 + threadIdx.x * 1e-3 + i; f[i] = 1.0f + i; }
  const double c = 1.0000001;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++)
  {
#pragma unroll
    for (int i = 0; i < 32; i++)
    {
      if (KIND == FMA64) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(r[i]) : "v"(c));
      if (KIND == RCP64) asm volatile("v_rcp_f64 %0, %0" : "+v"(r[i]));
      if (KIND == RSQ64) asm volatile("v_rsq_f64 %0, %0" : "+v"(r[i]));
      if (KIND == SQRT64) asm volatile("v_sqrt_f64 %0, %0" : "+v"(r[i]));
      if (KIND == RCP32) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[i]));
      if (KIND == FMA32) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f[i]));
      if (KIND == CVT_F32_F64) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(r[i]));
      if (KIND == CVT_F64_F32) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(r[i]) : "v"(f[i]));
      if (KIND == MUL64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(r[i]) : "v"(c));
      if (KIND == MAX64) asm volatile("v_max_f64 %0, %0, %1" : "+v"(r[i]) : "v"(c));
    }
  }
  long long t1 = __builtin_readcyclecounter();
  double s = 0; for (int i = 0; i < 32; i++) s += r[i] + f[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 3) cyc[0] = t1 - t0;
}

template <int KIND>
static void run(const char* name, double* out, long long* cyc)
{
  const int iters = 2000;
  for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL((probe<KIND>), dim3(256), dim3(256), 0, 0, out, cyc, iters); hipDeviceSynchronize(); }
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-16s %6.2f cycles per wavefront instruction\n", name, (double) c / iters / 32.0);
}

int main()
{
  double* out; long long* cyc;
  hipMalloc(&out, 256 * 256 * 8); hipMalloc(&cyc, 64);
  run<FMA64>("v_fma_f64", out, cyc); run<MUL64>("v_mul_f64", out, cyc); run<MAX64>("v_max_f64", out, cyc);
  run<RCP64>("v_rcp_f64", out, cyc); run<RSQ64>("v_rsq_f64", out, cyc); run<SQRT64>("v_sqrt_f64", out, cyc);
  run<RCP32>("v_rcp_f32", out, cyc); run<FMA32>("v_fma_f32", out, cyc);
  run<CVT_F32_F64>("v_cvt_f32_f64", out, cyc); run<CVT_F64_F32>("v_cvt_f64_f32", out, cyc);
  return 0;
}
