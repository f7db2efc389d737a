// dma_placement_probe.hip -- where in a step should the ring refill of the factor-update kernel be issued?  (round 5)
// One wavefront per SIMD (the update kernel's occupancy).  Loop body = a block of 32 independent v_mfma_f64_4x4x4_4b_f64
// (512 cycles at the pipe rate) + a block of NV independent v_fma_f64 (the quotient block's stand-in) + ND LDS-DMA
// instructions (global_load_lds_dwordx4, scalar base + lane offset, the kernel's form) placed
//   0 nowhere   1 spread through the MFMA block   2 spread through the VALU block   3 in front of the VALU block
//   4 behind the VALU block (in front of the MFMA block)
// with a counted wait that leaves three bodies' worth of DMA in flight.  Prints shader cycles per body.
//   hipcc -O3 --offload-arch=gfx950 tools/dma_placement_probe.hip -o tools/bin/dma_placement_probe
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ void glds16(const void* sbase, unsigned voff, unsigned ldsAddr)
{
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(ldsAddr), "v"(voff), "s"(sbase) : "memory", "m0");
}

template <int PLACE, int ND, int NV>
__global__ __launch_bounds__(256, 1) void probe(double* out, const char* src, long long* cyc, int iters)
{
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* my = lds + wave * 32768;
  const unsigned ldsA = (unsigned) (size_t) (__attribute__((address_space(3))) const void*) my;
  double acc[32];
  for (int i = 0; i < 32; i++) acc[i] = 0.0;
  double a = lane * 0.25, b = 1.0 + lane;
  double f[8];
  for (int i = 0; i < 8; i++) f[i] = 1.0 + i;
  const double g = 0.999;
  const unsigned voff = lane * 16;
  const char* base = src + (size_t) (blockIdx.x * 4 + wave) * (1 << 16);
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++)
  {
    const char* sb = base + (it & 7) * 4096;
    if (ND > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * ND) : "memory");
    if (PLACE == 4)
#pragma unroll
      for (int d = 0; d < ND; d++) glds16(sb + d * 1024, voff, ldsA + ((it & 3) * ND + d) * 1024);
#pragma unroll
    for (int i = 0; i < 32; i++)
    {
      acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
      if (PLACE == 1 && ND > 0 && (i % (32 / ND)) == (32 / ND) / 2)
        glds16(sb + (i / (32 / ND)) * 1024, voff, ldsA + ((it & 3) * ND + i / (32 / ND)) * 1024);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (PLACE == 3)
#pragma unroll
      for (int d = 0; d < ND; d++) glds16(sb + d * 1024, voff, ldsA + ((it & 3) * ND + d) * 1024);
#pragma unroll
    for (int i = 0; i < NV; i++)
    {
      asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(f[i & 7]) : "v"(g));
      if (PLACE == 2 && ND > 0 && (i % (NV / ND)) == (NV / ND) / 2)
        glds16(sb + (i / (NV / ND)) * 1024, voff, ldsA + ((it & 3) * ND + i / (NV / ND)) * 1024);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  long long t1 = __builtin_readcyclecounter();
  double s = 0.0;
  for (int i = 0; i < 32; i++) s += acc[i];
  for (int i = 0; i < 8; i++) s += f[i];
  out[blockIdx.x * 256 + threadIdx.x] = s + reinterpret_cast<double*>(my)[lane];
  if (threadIdx.x == 0 && blockIdx.x == 5) cyc[0] = t1 - t0;
}

template <int PLACE, int ND, int NV>
static double run(const char* name, double* out, char* src, long long* cyc)
{
  const int iters = 2000;
  hipFuncSetAttribute(reinterpret_cast<const void*>(probe<PLACE, ND, NV>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  for (int rep = 0; rep < 2; rep++)
  {
    hipLaunchKernelGGL((probe<PLACE, ND, NV>), dim3(256), dim3(256), 128 * 1024, 0, out, src, cyc, iters);
    hipDeviceSynchronize();
  }
  long long c;
  hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double per = (double) c / iters;
  printf("%-34s ND=%d NV=%-2d : %7.1f cycles per body\n", name, ND, NV, per);
  return per;
}

template <int ND, int NV>
static void family(double* out, char* src, long long* cyc)
{
  const double b0 = run<0, 0, NV>("no DMA", out, src, cyc);
  const double p1 = run<1, ND, NV>("DMA spread through the MFMA block", out, src, cyc);
  const double p2 = run<2, ND, NV>("DMA spread through the VALU block", out, src, cyc);
  const double p3 = run<3, ND, NV>("DMA in front of the VALU block", out, src, cyc);
  const double p4 = run<4, ND, NV>("DMA in front of the MFMA block", out, src, cyc);
  printf("   per DMA instruction: in the MFMA block %.1f, in the VALU block %.1f, in front of it %.1f, in front of the MFMAs %.1f\n",
         (p1 - b0) / ND, (p2 - b0) / ND, (p3 - b0) / ND, (p4 - b0) / ND);
}

int main()
{
  double* out;
  char* src;
  long long* cyc;
  hipMalloc(&out, 256 * 256 * 8);
  hipMalloc(&src, (size_t) 1024 << 16);
  hipMemset(src, 0, (size_t) 1024 << 16);
  hipMalloc(&cyc, 64);
  family<4, 32>(out, src, cyc);
  family<8, 32>(out, src, cyc);
  family<4, 64>(out, src, cyc);
  return 0;
}
