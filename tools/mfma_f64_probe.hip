// mfma_f64_probe -- confirms the MI355X roofs the bench normalises against:
//   * v_mfma_f64_16x16x4_f64 issue rate (FP64 matrix peak), 1/2/4 independent accumulators
//   * v_fma_f64 vector rate
//   * streaming HBM read bandwidth (f64 sum over a 4 GiB buffer)
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_probe.hip -o gpurun_out/mfma_f64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_kernel(double* out, int iters, double a0, double b0, long long* cyc)
{
  d4 acc[NACC];
  for (int i = 0; i < NACC; i++) acc[i] = d4{0, 0, 0, 0};
  // full-entropy mantissas (rule 25: bench on random-like data, zeros clock higher)
  unsigned long long h = (threadIdx.x + 1) * 0x9E3779B97F4A7C15ull + blockIdx.x * 0xD1B54A32D192ED03ull;
  h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
  double a = a0 * (0.5 + (double) (h >> 11) * (1.0 / 9007199254740992.0));
  h *= 0x94D049BB133111EBull; h ^= h >> 31;
  double b = b0 * (0.5 + (double) (h >> 11) * (1.0 / 9007199254740992.0));
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++)
  {
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  long long t1 = __builtin_readcyclecounter();
  double s = 0;
  for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NACC>
__global__ __launch_bounds__(256) void mfma4_kernel(double* out, int iters, double a0, double b0, long long* cyc)
{
  double acc[NACC];
  for (int i = 0; i < NACC; i++) acc[i] = 0.0;
  unsigned long long h = (threadIdx.x + 1) * 0x9E3779B97F4A7C15ull + blockIdx.x * 0xD1B54A32D192ED03ull;
  h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
  double a = a0 * (0.5 + (double) (h >> 11) * (1.0 / 9007199254740992.0));
  h *= 0x94D049BB133111EBull; h ^= h >> 31;
  double b = b0 * (0.5 + (double) (h >> 11) * (1.0 / 9007199254740992.0));
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++)
  {
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
  }
  long long t1 = __builtin_readcyclecounter();
  double s = 0;
  for (int i = 0; i < NACC; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { cyc[0] = t1 - t0; }
}

// VALU FMA with one scalar (SGPR) operand: the shape of a lane-per-column contraction
__global__ __launch_bounds__(256) void fma_sgpr_kernel(double* out, int iters, const double* __restrict__ sc)
{
  double acc[16];
  unsigned long long h = (threadIdx.x + 1) * 0x9E3779B97F4A7C15ull + blockIdx.x * 0xD1B54A32D192ED03ull;
  h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
  double x = 0.5 + (double) (h >> 11) * (1.0 / 9007199254740992.0);
  for (int i = 0; i < 16; i++) acc[i] = x + i;
  for (int it = 0; it < iters; it++)
  {
    const double s0 = sc[it & 15], s1 = sc[(it + 1) & 15];   // uniform -> s_load
#pragma unroll
    for (int i = 0; i < 16; i += 2) { acc[i] = __builtin_fma(x, s0, acc[i]); acc[i + 1] = __builtin_fma(x, s1, acc[i + 1]); }
  }
  double s = 0;
  for (int i = 0; i < 16; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// half the waves of a block run MFMA, the other half VALU FMA: do the pipes add up?
__global__ __launch_bounds__(512) void mixed_kernel(double* out, int iters, double a0, double b0)
{
  unsigned long long h = (threadIdx.x + 1) * 0x9E3779B97F4A7C15ull + blockIdx.x * 0xD1B54A32D192ED03ull;
  h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
  double a = a0 * (0.5 + (double) (h >> 11) * (1.0 / 9007199254740992.0));
  double s = 0;
  if (__builtin_amdgcn_readfirstlane(threadIdx.x) < 256)
  {
    d4 acc[4];
    for (int i = 0; i < 4; i++) acc[i] = d4{0, 0, 0, 0};
    for (int it = 0; it < iters; it++)
#pragma unroll
      for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b0, acc[i], 0, 0, 0);
    for (int i = 0; i < 4; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  }
  else
  {
    double acc[8];
    for (int i = 0; i < 8; i++) acc[i] = a + i;
    for (int it = 0; it < iters * 8; it++)   // 8x: ~ the same wall time as the MFMA half
#pragma unroll
      for (int i = 0; i < 8; i++) acc[i] = __builtin_fma(acc[i], a0, b0);
    for (int i = 0; i < 8; i++) s += acc[i];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void fma_kernel(double* out, int iters, double a0, double b0)
{
  double acc[8];
  for (int i = 0; i < 8; i++) acc[i] = threadIdx.x * 1e-9 + i;
  for (int it = 0; it < iters; it++)
  {
#pragma unroll
    for (int i = 0; i < 8; i++) acc[i] = __builtin_fma(acc[i], a0, b0);
  }
  double s = 0;
  for (int i = 0; i < 8; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void read_kernel(const d2* in, size_t n2, double* out)
{
  double s = 0;
  for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (size_t) gridDim.x * blockDim.x)
  {
    d2 v = in[i];
    s += v[0] + v[1];
  }
  if (s == 12345.678) out[0] = s;
}

template <typename F>
static double time_ms(F f, int reps)
{
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  f();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < reps; i++) f();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

int main()
{
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  printf("device: %s (%s), %d CUs, clock %d kHz\n", p.name, p.gcnArchName, p.multiProcessorCount, p.clockRate);
  double* out;
  hipMalloc(&out, (size_t) 256 * 4096 * sizeof(double));
  const int iters = 20000;
  long long* dcyc;
  hipMalloc(&dcyc, 8);
  for (int wavesPerSimd = 1; wavesPerSimd <= 8; wavesPerSimd *= 2)
  {
    const int blocks = 256 * wavesPerSimd; // 4 waves per block -> one per SIMD
    auto run = [&](auto kern, int nacc) {
      double ms = time_ms([&] { hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 1e-3, dcyc); }, 3);
      long long c = 0;
      hipMemcpy(&c, dcyc, 8, hipMemcpyDeviceToHost);
      double tf = 2048.0 * nacc * iters * blocks * 4 / (ms * 1e-3) / 1e12;
      // s_memtime-style counter ticks per MFMA per SIMD, and the implied shader clock
      double ticksPerInst = (double) c / ((double) nacc * iters * wavesPerSimd);
      printf("  mfma_f64_16x16x4 %d wave/SIMD %d acc: %.3f ms  %.1f TF  counter ticks/inst/SIMD %.1f  (counter rate %.0f MHz)\n",
             wavesPerSimd, nacc, ms, tf, ticksPerInst, (double) c / (ms * 1e-3) / 1e6);
    };
    run(mfma_kernel<1>, 1);
    run(mfma_kernel<2>, 2);
    run(mfma_kernel<4>, 4);
    run(mfma_kernel<8>, 8);
  }
  for (int wavesPerSimd = 1; wavesPerSimd <= 8; wavesPerSimd *= 2)
  {
    const int blocks = 256 * wavesPerSimd;
    auto run4 = [&](auto kern, int nacc) {
      double ms = time_ms([&] { hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 1e-3, dcyc); }, 3);
      printf("  mfma_f64_4x4x4_4b %d wave/SIMD %d acc: %.3f ms  %.1f TF\n", wavesPerSimd, nacc, ms,
             512.0 * nacc * iters * blocks * 4 / (ms * 1e-3) / 1e12);
    };
    run4(mfma4_kernel<4>, 4);
    run4(mfma4_kernel<8>, 8);
  }
  for (int wavesPerSimd = 1; wavesPerSimd <= 8; wavesPerSimd *= 2)
  {
    const int blocks = 256 * wavesPerSimd;
    double ms = time_ms([&] { hipLaunchKernelGGL(fma_kernel, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0000001, 1e-9); }, 3);
    printf("  v_fma_f64 (vgpr operands) %d wave/SIMD: %.1f TF\n", wavesPerSimd, 2.0 * 8 * iters * blocks * 256 / (ms * 1e-3) / 1e12);
    double* sc;
    hipMalloc(&sc, 16 * 8);
    double hs[16];
    for (int i = 0; i < 16; i++) hs[i] = 1e-9 * (i + 1);
    hipMemcpy(sc, hs, sizeof(hs), hipMemcpyHostToDevice);
    ms = time_ms([&] { hipLaunchKernelGGL(fma_sgpr_kernel, dim3(blocks), dim3(256), 0, 0, out, iters, sc); }, 3);
    printf("  v_fma_f64 (one sgpr operand) %d wave/SIMD: %.1f TF\n", wavesPerSimd, 2.0 * 16 * iters * blocks * 256 / (ms * 1e-3) / 1e12);
    hipFree(sc);
  }
  for (int bpc = 1; bpc <= 2; bpc++)
  {
    const int blocks = 256 * bpc;
    double ms = time_ms([&] { hipLaunchKernelGGL(mixed_kernel, dim3(blocks), dim3(512), 0, 0, out, iters, 1.0000001, 1e-3); }, 3);
    double mf = 2048.0 * 4 * iters * blocks * 4, vf = 2.0 * 8 * (iters * 8.0) * blocks * 256;
    printf("  mixed (4 MFMA waves + 4 VALU waves per block, %d block/CU): %.3f ms  MFMA %.1f TF + VALU %.1f TF = %.1f TF\n", bpc, ms,
           mf / (ms * 1e-3) / 1e12, vf / (ms * 1e-3) / 1e12, (mf + vf) / (ms * 1e-3) / 1e12);
  }
  {
    size_t bytes = (size_t) 4 << 30;
    d2* buf;
    hipMalloc(&buf, bytes);
    hipMemset(buf, 0, bytes);
    double ms = time_ms([&] { hipLaunchKernelGGL(read_kernel, dim3(256 * 16), dim3(256), 0, 0, buf, bytes / 16, out); }, 5);
    printf("HBM streaming read: %.0f GB/s\n", bytes / (ms * 1e-3) / 1e9);
    hipFree(buf);
  }
  return 0;
}
