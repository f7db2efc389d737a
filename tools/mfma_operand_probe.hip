// mfma_operand_probe.hip -- does the register file an operand of v_mfma_f64_4x4x4_4b_f64 comes from change its issue
// rate on gfx950?  24 independent accumulators per loop body (the second product of the frame-strip kernel), one
// wavefront per SIMD, operands forced into VGPRs ("v") or AGPRs ("a") by inline-asm constraints; prints shader cycles per
// MFMA.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_operand_probe.hip -o tools/bin/mfma_operand_probe
#include <hip/hip_runtime.h>
#include <cstdio>

#define MF(ACC, A, B, CA, CB, CC) asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+" CC(ACC) : CA(A), CB(B))

template <int VAR>
__global__ __launch_bounds__(256, 1) void probe(double* out, long long* cyc, int iters)
{
  const int lane = threadIdx.x & 63;
  double acc[24];
  for (int i = 0; i < 24; i++) acc[i] = 0.0;
  double a[6], b[4];
  for (int i = 0; i < 6; i++) a[i] = lane * 0.25 + i;
  for (int i = 0; i < 4; i++) b[i] = 1.0 + lane + i;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++)
  {
#pragma unroll
    for (int q = 0; q < 6; q++)
#pragma unroll
      for (int m = 0; m < 4; m++)
      {
        double& c = acc[q * 4 + m];
        if (VAR == 0) MF(c, a[q], b[m], "v", "v", "v");
        if (VAR == 1) MF(c, a[q], b[m], "v", "v", "a");
        if (VAR == 2) MF(c, a[q], b[m], "a", "v", "a");
        if (VAR == 3) MF(c, a[q], b[m], "v", "a", "a");
        if (VAR == 4) MF(c, a[q], b[m], "a", "a", "a");
        if (VAR == 5) MF(c, a[q], b[m], "a", "a", "v");
        if (VAR == 6) MF(c, a[q], b[m], "v", "a", "v");
        if (VAR == 7) MF(c, a[q], b[m], "a", "v", "v");
      }
  }
  long long t1 = __builtin_readcyclecounter();
  double s = 0;
  for (int i = 0; i < 24; i++) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 5) cyc[0] = t1 - t0;
}

template <int VAR>
static void run(const char* name, double* out, long long* cyc)
{
  const int iters = 2000;
  for (int rep = 0; rep < 2; rep++)
  {
    hipLaunchKernelGGL(probe<VAR>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
  }
  long long c;
  hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-34s : %6.2f cycles per MFMA\n", name, (double) c / iters / 24.0);
}

int main()
{
  double* out;
  long long* cyc;
  hipMalloc(&out, 256 * 256 * 8);
  hipMalloc(&cyc, 64);
  run<0>("A vgpr, B vgpr, C/D vgpr", out, cyc);
  run<1>("A vgpr, B vgpr, C/D agpr", out, cyc);
  run<2>("A agpr, B vgpr, C/D agpr", out, cyc);
  run<3>("A vgpr, B agpr, C/D agpr", out, cyc);
  run<4>("A agpr, B agpr, C/D agpr", out, cyc);
  run<5>("A agpr, B agpr, C/D vgpr", out, cyc);
  run<6>("A vgpr, B agpr, C/D vgpr", out, cyc);
  run<7>("A agpr, B vgpr, C/D vgpr", out, cyc);
  return 0;
}
