// mfma_coissue_probe.hip -- what does one extra instruction cost inside a stream of independent
// v_mfma_f64_4x4x4_4b_f64 on gfx950 when the SIMD holds a single wavefront (the update kernel's
// occupancy)?  Loop body = 16 independent MFMAs (256 cycles at the pipe rate) with NPER instructions
// of one kind spread evenly between them; prints shader cycles per loop body.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_coissue_probe.hip -o tools/bin/mfma_coissue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double d2 __attribute__((ext_vector_type(2)));

enum { NONE, DSR64, DSR128, GLDS, VFMA, VMOV, GLD128, SMOV, WAVES2 };

template <int KIND, int NPER>
__global__ __launch_bounds__(256, 1) void probe(double* out, const double* src, long long* cyc, int iters)
{
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  char* my = lds + wave * 8192;
  for (int i = lane; i < 1024; i += 64) reinterpret_cast<double*>(my)[i] = i;
  __syncthreads();
  double acc[16];
  for (int i = 0; i < 16; i++) acc[i] = 0.0;
  double a = lane * 0.25, b = 1.0 + lane;
  double sink = 0.0, f0 = 1.0, f1 = 0.5;
  d2 sink2 = {0.0, 0.0};
  const unsigned laddr = (unsigned) (size_t) (my) + lane * 16;   // LDS byte address (low 32 bits of the flat shared pointer)
  const double* g = src + (size_t) (blockIdx.x * 256 + threadIdx.x) * 2;
  unsigned goff = lane * 16;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++)
  {
#pragma unroll
    for (int i = 0; i < 16; i++)
    {
      acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
      if (NPER > 0 && (i % (16 / (NPER > 16 ? 16 : NPER))) == 0)
      {
        constexpr int REP = NPER > 16 ? NPER / 16 : 1;
#pragma unroll
        for (int r = 0; r < REP; r++)
        {
          if (KIND == DSR64) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(sink) : "v"(laddr), "n"(64 * (i + 1)) : "memory");
          if (KIND == DSR128) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(sink2) : "v"(laddr), "n"(64 * (i + 1)) : "memory");
          if (KIND == GLDS)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*) (reinterpret_cast<const char*>(src) + goff),
                                             (__attribute__((address_space(3))) void*) (my + 1024 * (i & 3)), 16, 0, 0);
          if (KIND == VFMA) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(f0) : "v"(f1), "v"(f1));
          if (KIND == VMOV) asm volatile("v_mov_b32 %0, %1" : "=v"(goff) : "v"(goff));
          if (KIND == GLD128) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(sink2) : "v"(g) : "memory");
          if (KIND == SMOV) asm volatile("s_mov_b32 s20, 0" ::: "s20");
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  }
  long long t1 = __builtin_readcyclecounter();
  double s = sink + sink2[0] + sink2[1] + f0;
  for (int i = 0; i < 16; i++) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 5) cyc[0] = t1 - t0;
}

template <int KIND, int NPER>
static void run(const char* name, double* out, double* src, long long* cyc)
{
  const int iters = 2000;
  hipFuncSetAttribute(reinterpret_cast<const void*>(probe<KIND, NPER>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  for (int rep = 0; rep < 2; rep++)
  {
    hipLaunchKernelGGL((probe<KIND, NPER>), dim3(256), dim3(256), 64 * 1024, 0, out, src, cyc, iters);
    hipDeviceSynchronize();
  }
  long long c;
  hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double per = (double) c / iters;
  printf("%-28s x%-2d : %7.1f cycles / 16 MFMAs  (+%6.1f, %5.1f per extra instr)\n", name, NPER, per, per - 256.0,
         NPER ? (per - 256.0) / NPER : 0.0);
}

int main()
{
  double *out, *src;
  long long* cyc;
  hipMalloc(&out, 256 * 256 * 8);
  hipMalloc(&src, 1 << 24);
  hipMemset(src, 0, 1 << 24);
  hipMalloc(&cyc, 64);
  run<NONE, 0>("mfma only", out, src, cyc);
  run<DSR64, 4>("ds_read_b64", out, src, cyc);
  run<DSR64, 16>("ds_read_b64", out, src, cyc);
  run<DSR128, 4>("ds_read_b128", out, src, cyc);
  run<DSR128, 16>("ds_read_b128", out, src, cyc);
  run<GLDS, 2>("global_load_lds_dwordx4", out, src, cyc);
  run<GLDS, 4>("global_load_lds_dwordx4", out, src, cyc);
  run<GLDS, 8>("global_load_lds_dwordx4", out, src, cyc);
  run<GLD128, 4>("global_load_dwordx4", out, src, cyc);
  run<GLD128, 8>("global_load_dwordx4", out, src, cyc);
  run<VFMA, 4>("v_fma_f64", out, src, cyc);
  run<VFMA, 16>("v_fma_f64", out, src, cyc);
  run<VMOV, 16>("v_mov_b32", out, src, cyc);
  run<VMOV, 32>("v_mov_b32", out, src, cyc);
  run<SMOV, 16>("s_mov_b32", out, src, cyc);
  run<SMOV, 32>("s_mov_b32", out, src, cyc);
  return 0;
}
