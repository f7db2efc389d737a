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
__global__ __launch_bounds__(256) void mfma_kernel(double* out, int iters, double a0, double b0)
{
  d4 acc[NACC];
  for (int i = 0; i < NACC; i++) acc[i] = d4{0, 0, 0, 0};
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  for (int it = 0; it < iters; it++)
  {
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
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
  for (int wavesPerSimd = 1; wavesPerSimd <= 2; wavesPerSimd++)
  {
    const int blocks = 256 * wavesPerSimd; // 4 waves per block -> one per SIMD
    double ms1 = time_ms([&] { hipLaunchKernelGGL(mfma_kernel<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 1e-3); }, 3);
    double ms2 = time_ms([&] { hipLaunchKernelGGL(mfma_kernel<2>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 1e-3); }, 3);
    double ms4 = time_ms([&] { hipLaunchKernelGGL(mfma_kernel<4>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 1e-3); }, 3);
    auto tf = [&](double ms, int nacc) { return 2048.0 * nacc * iters * blocks * 4 / (ms * 1e-3) / 1e12; };
    auto cyc = [&](double ms, int nacc) { return ms * 1e-3 * 2.4e9 / ((double) nacc * iters * wavesPerSimd); };
    printf("mfma_f64_16x16x4 %d wave(s)/SIMD: 1 acc %.1f TF (%.1f cyc/inst @2.4GHz), 2 acc %.1f TF (%.1f), 4 acc %.1f TF (%.1f)\n",
           wavesPerSimd, tf(ms1, 1), cyc(ms1, 1), tf(ms2, 2), cyc(ms2, 2), tf(ms4, 4), cyc(ms4, 4));
  }
  {
    const int blocks = 256 * 8;
    double ms = time_ms([&] { hipLaunchKernelGGL(fma_kernel, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0000001, 1e-9); }, 3);
    printf("v_fma_f64: %.1f TF\n", 2.0 * 8 * iters * blocks * 256 / (ms * 1e-3) / 1e12);
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
